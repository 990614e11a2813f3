// bvh.h -- per-mesh SAH BVH builder restating src/core/bvh.rs operation for
// operation in f32 (the comparison `cost < best_cost` and the in-place
// partition decide triangle order, hence traversal order, hence the floats
// the shader produces).  Host-side; results-identical, not accelerated.
#ifndef RT_BVH_H
#define RT_BVH_H

#include <cstdint>
#include <functional>
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

// Level-wise variant for large meshes: the same nodes, node numbering and triangle order as
// bvh_build, built breadth-first so that the SAH searches of a whole level (find_best_split,
// bvh.rs:299-351 -- the O(150 n)-per-level part) can be delegated in one batch, e.g. to the GPU
// (csrc/rt_bvh_search.hip).  The partitions and everything else run here, as in bvh_build.
struct SplitQuery {  // a node of the level: its range in the current triangle order and its bounds
    uint32_t start, count;
    float aabb_min[3], aabb_max[3];
};
struct SplitResult {  // find_best_split's (cost, axis, split_pos)
    int axis;
    float pos, cost;
};
// order[p] = original triangle index at position p of the current order (n entries)
using LevelSearch = std::function<void(const uint32_t* order, size_t n, const std::vector<SplitQuery>&, std::vector<SplitResult>&)>;
// per original triangle: centroid, min, max (9 floats) -- what a LevelSearch needs to price planes
std::vector<float> bvh_search_data(const std::vector<Vertex>& vertices, const std::vector<uint32_t>& indices);
BvhResult bvh_build_levels(const std::vector<Vertex>& vertices, const std::vector<uint32_t>& indices, Quality quality,
                           const LevelSearch& search);
// the searches on the GPU (throws std::runtime_error on HIP errors)
LevelSearch make_device_level_search(int device, const float* tri9, size_t n_tris);
// the searches on the host (reference implementation of a LevelSearch; tests)
LevelSearch make_host_level_search(const float* tri9, size_t n_tris);

}  // namespace rt2

#endif
