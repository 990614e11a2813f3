// bvh.h -- per-mesh SAH BVH builder restating src/core/bvh.rs operation for
// operation in f32 (the comparison `cost < best_cost` and the in-place
// partition decide triangle order, hence traversal order, hence the floats
// the shader produces).  Host-side; results-identical, not accelerated.
#ifndef RT_BVH_H
#define RT_BVH_H

#include <cstdint>
#include <vector>

#include "../../../include/rt_abi.h"
#include "glam_math.h"

namespace rt2 {

struct Vertex {  // src/scene/components/geometry/vertex.rs:3-8
    Vec3 pos, normal;
    float uv[2] = {0, 0};
};

enum class Quality { Low = 0, High = 1, Disabled = 2 };  // bvh.rs:126-131

struct BvhResult {
    std::vector<rt_packed_triangle> triangles;
    std::vector<rt_node> nodes;
};

// ≙ BVH::build (bvh.rs:208-290)
BvhResult bvh_build(const std::vector<Vertex>& vertices, const std::vector<uint32_t>& indices,
                    Quality quality);

}  // namespace rt2

#endif
