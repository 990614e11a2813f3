// rt_device.h -- device-side data layout shared by rt_api.hip (upload) and
// rt_kernel.hip (render).  The reference's arrays (include/rt_abi.h) are
// re-laid-out once at upload time for 16-byte vector loads; every derived
// value is computed with the same IEEE operations the shader would execute
// per ray, so the re-layout is results-preserving.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_abi.h"

namespace rtd {

// Per-mesh record.  Matrices keep rows 0..2 of each column (the shader only
// takes .xyz of mat4 * vec4): m[col*4 + row], row 3 unused.
struct alignas(16) DMesh {
    float w2m[16];
    float m2w[16];
    uint32_t node_offset;
    uint32_t tri_offset;
    uint32_t flags;  // DMESH_*
    uint32_t root_count;  // nodes[node_offset].count (0 => internal root)
};
enum : uint32_t {
    DMESH_SAME_XFORM = 1u,  // world_to_model bit-identical to the previous mesh's
    DMESH_GLASS = 2u,       // material.flag == GLASS  => no backface culling (wgsl:375)
};

// BVH node, 48 B = 3 x float4, unchanged from rt_node:
//   q0 = (left, right, first, count) as bits, q1 = (min.xyz, -), q2 = (max.xyz, -)
//
// Triangle, split in two 48-B records:
//   isect: q0 = (v1.xyz, n.x) q1 = (edge_ab.xyz, n.y) q2 = (edge_ac.xyz, n.z)
//          with edge_ab = v2 - v1, edge_ac = v3 - v1, n = cross(edge_ab,
//          edge_ac) exactly as wgsl:261-263 computes them per test
//   shade: q0 = (n1.xyz, uv10) q1 = (n2.xyz, uv11)... see pack in rt_api.hip:
//          q0 = (n1.xyz, u10) q1 = (n2.xyz, u11) q2 = (n3.xyz, u20) and
//          q3 = (u21, u30, u31, 0)  -> 64 B
struct alignas(16) DSphere {
    float cx, cy, cz, radius;
};

struct DTexture {
    const uint8_t* rgba8;
    uint32_t width, height;
};

struct Counters {
    unsigned long long segments;
    unsigned long long paths;
    unsigned long long node_tests;
    unsigned long long triangle_tests;
};

// Kernel argument block (lives in kernarg SGPRs).
struct RenderArgs {
    rt_params params;
    rt_camera_uniform camera;
    const DMesh* meshes;
    const rt_material* mesh_materials;
    const float4* nodes;      // 3 float4 per node
    const float4* tri_isect;  // 3 float4 per triangle
    const float4* tri_shade;  // 4 float4 per triangle
    const DSphere* spheres;
    const rt_material* sphere_materials;
    const DTexture* textures;
    const float* srgb_lut;
    float4* image;  // full frame, or compact strips when strip_world > 1
    Counters* counters;
    uint32_t n_meshes, n_spheres, n_textures;
    uint32_t stack_entries;  // per-lane BVH stack depth (LDS dwords per lane)
    uint32_t strip_rank, strip_world;
    uint32_t tiles_x, tiles_y;  // 8x8 tiles of the (local) image
    uint32_t count_tests;       // 1 => accumulate node/triangle test counters
};

}  // namespace rtd

#endif
