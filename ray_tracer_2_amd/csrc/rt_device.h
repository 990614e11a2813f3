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

// Features that were built, parity-tested and measured SLOWER than what ships -- the LDS-staged top of a big mesh's BVH
// (option "lds_top"), the LDS-staged top-level tree ("lds_tlas"), the hybrid launches of a deferred-walk sequence
// ("hybrid") and the wavefront sequence ("wavefront"); measurements in DESIGN.md sections 5.4 / 5.5 -- are compiled only
// with -DRT_EXPERIMENTS=1 (tools/build_variant.sh exp -DRT_EXPERIMENTS=1 -> librt2_mi355x_exp.so; their parity tests run
// against that build: RT2_LIB=... pytest).  The product library contains none of their kernels and rejects the options.
#ifndef RT_EXPERIMENTS
#define RT_EXPERIMENTS 0
#endif

// Experiment of round 5 (tools/experiments/README.md): two-level records for rt_walk_kernel --
// the record of an internal node carries its children's boxes AND a copy of both children's own records (192 B), so that
// the near-first descent fetches once per two levels.  -DRT_WALK2=1 (tools/build_variant.sh walk2 -DRT_WALK2=1).
#ifndef RT_WALK2
#define RT_WALK2 0
#endif

namespace rtd {

// The whole scene is one blob of 16-byte words, either read in place (global
// memory: uniform reads become scalar loads) or, when it fits the LDS budget,
// staged into LDS once per workgroup with coalesced 16-byte loads.  All
// offsets below are byte offsets into that blob.
struct SceneLayout {
    uint32_t mesh_off;    // MESH_REC_BYTES per mesh
    uint32_t wide_off;    // WIDE_REC_BYTES per internal BVH node
    uint32_t tri_off;     // TRI_ISECT_BYTES per triangle
    uint32_t shade_off;   // TRI_SHADE_BYTES per triangle
    uint32_t mat_off;     // 96 B rt_material per mesh, then per sphere
    uint32_t sphere_off;  // 16 B (centre, radius) per sphere
    uint32_t item_off;    // ITEM_BYTES per item of the mesh loop
    uint32_t tlas_off;    // WIDE_REC_BYTES per node of the top-level trees over mesh root boxes
    uint32_t forest_off;  // FOREST_ENTRY_BYTES per member of the forest items
    uint32_t bytes;       // total, multiple of 16
    uint32_t _pad[2];
};

// The mesh loop (wgsl:369) runs over items.  An item is one mesh, or a top-level tree
// (TLAS) over the root boxes of a run of meshes that share one world_to_model matrix and
// have internal roots; the order in which meshes are visited is free because ties between
// equal world distances are broken by mesh index, exactly as the shader's in-order loop with
// its strict `<` does.  Two 16-byte words per item:
//   q0 = (kind, a, b, c): kind = ITEM_* flags; b = mesh whose matrices give the local ray when
//        ITEM_NEW_XFORM is set; single mesh: a = mesh index, c = its wide_base;
//        TLAS: a = root node index, c = number of meshes below it
//   q1 = single mesh: a copy of the mesh record's q8 (flags, root_idx, root_count, tri_base), so
//        that a mesh visit needs no dependent load.
//   forest (ITEM_FOREST): a = first member entry, c = number of members (<= 32): meshes with an
//        internal, non-deep root that share both matrices; each lane walks the members whose root
//        box it hits one after the other, independently of the other lanes (traverse_forest).
constexpr uint32_t ITEM_BYTES = 32;
//   flat (ITEM_FLAT2, an attribute of a single-mesh item): the mesh's BVH is a root with two LEAF children
//        (a quad split into its two triangles, ...).  Its whole traversal is two box tests and the leaves' triangles,
//        near leaf first: the few-mesh kernels run it as straight-line code with every lane of the wave in step,
//        no stack and no loop (traverse_flat2), instead of as a forest member or a mesh walk.
enum : uint32_t { ITEM_TLAS = 1u, ITEM_NEW_XFORM = 2u, ITEM_FOREST = 4u, ITEM_FLAT2 = 8u,
                  ITEM_DEFER = 16u,       // the big mesh whose walk a launch with RenderArgs::park != 0 defers (last item)
                  ITEM_DEFER_CULL = 32u,  // ... and its root box provably contains its children's (missing it = missing the mesh)
                  ITEM_PRUNE = 64u        // cross-mesh pruning may cut this item's meshes (RenderArgs::cross_prune): every mesh of the
                                          // item has the model_to_world of the mesh that gives the local ray, bit for bit, and a BVH
                                          // that is a proper bounding hierarchy (checked at upload, rt_api.hip)
};
// A top-level tree's reference to a mesh (the child index of a tree record whose child count is non-zero; with bit
// 31 set, an entry of the tree stack): everything a lane needs to enter the mesh -- the mesh's index (the caps allow
// 400), the absolute index of its root's wide record (<= 1.3 M internal nodes) and whether it is glass (wgsl:376: no
// backface culling).  Meshes that do not fit these fields stay single items.
enum : uint32_t {
    TLAS_REF_ROOT_MASK = 0x001fffffu,  // bits 0-20
    TLAS_REF_MESH_SHIFT = 21u,
    TLAS_REF_MESH_MASK = 0x1ffu,       // bits 21-29
    TLAS_REF_GLASS = 0x40000000u,      // bit 30
};
// Forest member entry, 3 x 16 B: q0 = (root wide index, mesh index, flags, 0), q1/q2 = the root's
// packed box (as in a wide record).  flags: DMESH_GLASS, FOREST_CULLABLE = the root box provably
// contains the boxes of the root's children (so missing it means missing the mesh).
constexpr uint32_t FOREST_ENTRY_BYTES = 48;
constexpr uint32_t FOREST_MAX_MEMBERS = 32;
constexpr uint32_t FOREST_CULLABLE = 0x100u;
constexpr uint32_t TLAS_MIN_MESHES = 8;

// Mesh record, 12 x 16 B:
//   q0..q3  world_to_model columns   q4..q7  model_to_world columns
//   q8 = (flags, root_idx, root_count, tri_base)
//   q9 = (wide_base, S, C, 0): S >= the largest absolute row sum of model_to_world's 3 x 3 part, C >= the largest
//        absolute component of its translation (rounded up by the host; cross-mesh pruning's error terms)
//   q10 = root (min.x, max.x, min.y, max.y)   q11 = root (min.z, max.z, 0, 0)
// root_count > 0: the root is a leaf with triangles [root_idx, root_idx+count);
// root_count == 0: root_idx is the mesh-local index of its wide record.
constexpr uint32_t MESH_REC_BYTES = 192;
enum : uint32_t {
    DMESH_GLASS = 2u,       // material.flag == GLASS  => no backface culling (wgsl:375)
    DMESH_DEEP = 4u,        // BVH height >= 32: the shader's 32-entry stack can overflow; traverse it
                            // with the shader's literal push/pop and index clamping (naga Restrict)
};
// Wide BVH record of one internal node (both children's boxes inline, so a
// visit is one round trip instead of three dependent ones), 4 x 16 B:
//   q0 = (a.min.x, a.max.x, a.min.y, a.max.y) q1 = (a.min.z, a.max.z, a_idx, a_count)
//   q2, q3 = the same for child b   (min/max of an axis adjacent: one packed-f32 pair per axis)
// child leaf: idx = first triangle (mesh-local), count > 0;
// child internal: idx = mesh-local wide index, count = 0.
constexpr uint32_t WIDE_REC_BYTES = 64;
// Triangle intersection record, 3 x 16 B:
//   q0 = (v1.xyz, n.x) q1 = (edge_ab.xyz, n.y) q2 = (edge_ac.xyz, n.z)
//   with edge_ab = v2 - v1, edge_ac = v3 - v1, n = cross(edge_ab, edge_ac)
//   exactly as wgsl:261-263 computes them per test.
constexpr uint32_t TRI_ISECT_BYTES = 48;
// Triangle shading record, 4 x 16 B:
//   q0 = (n1.xyz, u10) q1 = (n2.xyz, u11) q2 = (n3.xyz, u20) q3 = (u21, u30, u31, 0)
constexpr uint32_t TRI_SHADE_BYTES = 64;
constexpr uint32_t MATERIAL_BYTES = 96;
constexpr uint32_t SPHERE_BYTES = 16;

// Occupancy target: workgroups of 256 threads (one wave per SIMD) resident per CU = waves per
// SIMD.  The render kernels are compiled for that register budget (512 / RT_MIN_WAVES VGPRs) and
// the LDS budget per workgroup is the CU's 160 KiB divided by it.
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 5
#endif
constexpr uint32_t BLOCKS_PER_CU = RT_MIN_WAVES;
constexpr uint32_t LDS_BUDGET_BYTES = 160 * 1024 / BLOCKS_PER_CU;
constexpr uint32_t BLOCK_THREADS = 256;
constexpr uint32_t WAVES_PER_BLOCK = BLOCK_THREADS / 64;
// Per-lane state kept in LDS instead of registers (see path_step): the pixel's running sum
// `total`, 4 floats.  It is touched once per path but would otherwise occupy 4 VGPRs across the
// whole traversal.
#ifndef RT_TOTAL_IN_LDS
#define RT_TOTAL_IN_LDS 1   // 0 (experiment): the pixel sum stays in registers in every kernel
#endif
constexpr uint32_t LANE_STATE_DWORDS = RT_TOTAL_IN_LDS ? 4 : 0;
// The same in the kernels that read the scene from global memory: held in registers the sum was spilled
// around the traversal in every iteration (6 scratch stores + 6 loads per wave-iteration = 1.2 GB of writes
// per frame on the config 3 stand-in, profiles/r02_dragon9_summary.txt); 4 KiB of LDS per workgroup.
#ifndef RT_TOTAL_LDS_GLOBAL
#define RT_TOTAL_LDS_GLOBAL 1
#endif
constexpr bool total_in_lds(bool lds_scene) { return RT_TOTAL_IN_LDS != 0 && (lds_scene || RT_TOTAL_LDS_GLOBAL != 0); }
// Per-lane primary-ray memo (see path_step): rd, hit record (dst, point, normal, u, v), and
// one word = mat_off | hit | backface << 1 | ray valid << 2 | hit valid << 3.
constexpr uint32_t PIXEL_MEMO_DWORDS = 13;

struct DTexture {
    const uint8_t* rgba8;
    uint32_t width, height;
};

struct Counters {
    unsigned long long segments;
    unsigned long long paths;
    unsigned long long node_tests;
    unsigned long long triangle_tests;
    unsigned long long reused;  // segments whose hit came from the primary-ray memo (no traversal)
};

// Kernel argument block (lives in kernarg SGPRs).
struct RenderArgs {
    rt_params params;
    rt_camera_uniform camera;
    const float4* blob;  // the scene, see SceneLayout
    SceneLayout lay;
    const DTexture* textures;
    const float* srgb_lut;
    float4* image;  // full frame, or compact strips when strip_world > 1
    Counters* counters;
    uint32_t* work_counter;  // persistent kernel: next 8x8 tile to hand out (zeroed per launch)
    const uint32_t* tile_order;  // optional: tiles sorted by last frame's cost, heaviest first
    uint32_t* tile_cost;         // optional: rays per tile of this frame (feeds the next frame's order)
    uint32_t n_meshes, n_spheres, n_textures, n_items;
    uint32_t stack_entries;  // per-lane BVH stack depth
    uint32_t stack_wide;     // 1 => two dwords per entry (a leaf reference does not fit one dword)
    uint32_t tlas_entries;   // per-lane TLAS stack depth (1 dword per entry), >= 1
    uint32_t many_mesh;      // 1 => use the kernels with top-level trees / root-box culling compiled in
    uint32_t simple;         // 1 => no spheres, no glass, no textured material, camera without jitter: the few-mesh product
                             // kernels have an instantiation with those branches compiled out (option "specialise")
    uint32_t pixel_cache;    // per-lane primary-ray memo (PIXEL_MEMO_DWORDS per lane): 0 off, 1 in LDS,
                             // 2 in `pixel_cache_mem` (persistent kernel only, when LDS has no room)
    uint32_t* pixel_cache_mem;
    const void* primary;     // primary table of the frame (rt_primary_kernel: one 64-byte memo entry per pixel -- the constant
                             // primary ray and, with option "primary_hits", its hit), or null: compute per pixel
    uint32_t primary_complete;  // != 0: the table holds the hits too, so a lane's memo EQUALS its table entry for the whole pixel
                             // (memo_hit_store rewrites the same bits): park records leave the memo out and a resumed pixel
                             // reloads it from the table
    float spp_reciprocal;     // 1 / rays_per_pixel when that is a power of two (exact), else 0
    float blend_weight, blend_rest;  // 1 / f32(frames + 1) and 1 - that (wgsl:157-158)
    float memo_ro[3];        // (origin + right * 0) + up * 0: the memoised primary rays' common origin
    uint32_t vote_eighths;   // intersection vote: run when wanting lanes * 8 >= lanes * vote_eighths
    uint32_t vote_patience;  // ... or when somebody has waited this many iterations
    uint32_t strip_rank, strip_world;
    uint32_t tiles_x, tiles_y;  // 8x8 tiles of the (local) image
    uint32_t count_tests;       // 1 => accumulate node/triangle test counters
    uint32_t kernel_variant;    // 0 = persistent waves + lane refill, 1 = one wave per tile
    uint32_t persistent_blocks; // grid size of the persistent kernel
    uint32_t lds_scene;         // 1 => the blob is staged into LDS
    uint32_t forest_cull;       // 1 => forest members are skipped when the ray misses their root box
    uint32_t cull_roots;        // 1 => skip a mesh with an internal root when the ray misses the root box
    // Cross-mesh pruning (option "cross_prune"; many-mesh product kernels): a tree box, a mesh's
    // root box or a BVH box whose entry distance lies beyond what the closest hit so far allows -- t_cut, a local-space
    // bound with a written error budget and 12.5 % of slack on top -- is not entered.  The counter (STATS) and debug
    // kernels never prune.
    uint32_t cross_prune;
    // Frame batch (rt_render_frames): one persistent launch renders frames params.frames ..
    // params.frames + batch_frames - 1; work items are (frame, tile) pairs, frame k's samples go
    // unblended to image + k * batch_stride and rt_blend_frames_kernel applies wgsl:154-161 in frame
    // order afterwards.  0 = a plain one-frame launch that blends in place.
    uint32_t batch_frames;
    uint32_t batch_tile_major;  // 1 => work items in (tile, frame) order instead of (frame, tile)
    unsigned long long batch_stride;  // texels between the scratch frames
    // LDS-staged top of a big mesh's BVH (scenes read from global memory): wide records top_base ..
    // top_base + top_count - 1 -- the first levels of the biggest mesh, numbered breadth-first at upload
    // -- are copied into LDS by every workgroup (coalesced 16-byte loads) and read from there.
    uint32_t top_base, top_count;
    // LDS-staged top-level tree (many-mesh scenes read from global memory; option "lds_tlas"): all tlas_lds wide
    // records of the scene's top-level tree(s) are copied into LDS by every workgroup, behind the staged BVH top;
    // 0 = read in place.
    uint32_t tlas_lds;
    // Deferred walks (option "sort_rounds"; scenes with one big mesh, few-mesh kernels).  The walk through a big
    // mesh is entered by a few lanes of a wave at a time and costs the whole wave its full length.  With park != 0
    // a render launch does not walk that mesh (its item, ITEM_DEFER, is the last of the mesh loop -- the order of the
    // loop is free): a lane whose ray can enter it PARKS its pixel -- the pixel's state and the closest-hit record
    // of the segment so far go to a queue in global memory (park records, below) -- and takes other work.
    // rt_walk_kernel then walks the big mesh for all parked rays, every lane fetching its next ray as soon as its
    // walk ends, and the next render launch resumes the parked pixels behind the walk: offer the mesh's hit, finish
    // the segment, go on until the next entry into the big mesh.  A pixel's operations and their order are those of
    // the undeferred loop, so the image is the same bit for bit; the last launch of a sequence has park = 0 and
    // walks what is left inline.
    uint32_t fast_miss;           // != 0: a memoised primary ray that misses ends its pixel in one step (path_end, option "fast_miss")
    uint32_t park;
    uint32_t park_levels;         // != 0: a ray parks only if the walk's first two levels reach a grandchild box (option "park_levels")
    float4* q_in;                 // park records to resume instead of tiles (null: the work items are tiles)
    const uint32_t* q_in_count;   // their number (written by the launch before)
    float4* q_out;                // where this launch parks
    uint32_t* q_out_count;
    uint32_t defer_mesh, defer_xform;  // the deferred mesh and the mesh whose matrices give its local ray
#if RT_WALK2
    const float4* walk2;               // 12 float4 per internal node of the deferred mesh: its wide record, then copies of
    uint32_t walk2_base;               // its two children's wide records (zeros for a leaf child); index = wide index - base
#endif
    // Hybrid launch of a deferred-walk sequence (option "hybrid"): `blob` / `lay` are the SMALL blob -- the scene without
    // the deferred mesh's BVH and triangles, staged into LDS -- and only a winner on the deferred mesh reads its shading
    // record from the full blob in global memory.
    uint32_t hybrid;
    uint32_t big_shade_off;
    const float4* big_blob;
    // Wavefront sequence (option "wavefront"; many-mesh scenes).  Every ray of a many-mesh scene walks the top-level
    // tree and a few meshes, the walks are of very unequal length, and inside the render kernel a wave-iteration lasts
    // as long as its longest walk (lane utilisation 0.18 on the sponza-sized stand-in).  Here the path state of every
    // pixel of the launch lives in a SLOT in global memory and two kernels alternate: rt_wf_shade_kernel finishes a
    // segment (winner's normal / uv, shading, russian roulette, end of path / pixel) and begins the next one -- the
    // operations of path_step around the traversal, unchanged -- and lists the slots whose new segment needs a
    // traversal; rt_wf_walk_kernel intersects the scene for the listed rays with per-lane refill (a lane takes its
    // next ray the moment its walk ends) and phase-scheduled passes (box steps / leaf triangles / bookkeeping, the
    // most populated phase next).  A pixel's operations and their order are those of the undeferred loop.
    //   wf_state: 6 float4 planes per slot, blocks of 64 slots: (x | out_row << 16, rng, j | fresh << 31, seg)
    //             (ro, rd.x) (rd.yz, T.xy) (T.zw, light.xy) (light.zw, total.xy) (total.zw, meta, -)
    //   wf_hit:   2 planes: (closest, object code | any << 24 | sphere inside << 25, win_u, win_v) (win_tri, win_point)
    //             object code: mesh index, or 0x800000 | sphere index
    //   the primary-ray memo of a slot: pixel_cache_mem, 13 dwords per slot (pixel_cache == 3)
    float4* wf_state;
    float4* wf_hit;
    const uint32_t* wf_list_in;    // slots to shade (their hits are in wf_hit) / rays to walk
    const uint32_t* wf_count_in;
    uint32_t* wf_list_out;         // slots whose next segment needs a traversal
    uint32_t* wf_count_out;
    uint32_t wf_round0;            // 1 => the shade launch that takes the pixels: slots 0 .. wf_slots - 1, no hits yet
    uint32_t wf_slots;             // slots of the launch = wf_frame_slots x frames
    uint32_t wf_frame_slots;       // slots per frame = tiles x 64 (whole 8x8 tiles; pixels outside the image stay empty)
};
constexpr uint32_t WF_STATE_PLANES = 6, WF_HIT_PLANES = 2;
// Park record of a pixel: 14 x 16 B, stored in blocks of 64 records, plane by plane (plane p of record i of block b at
// float4 index (b * PARK_PLANES + p) * 64 + i), so that the lanes of a wave, which hold consecutive records, store and
// load contiguous kilobytes.  Planes: 0 (x, out_row, rng, j)  1 (seg, fresh, meta, memo[12])  2 (ro, rd.x)
// 3 (rd.yz, T.xy)  4 (T.zw, light.xy)  5 (light.zw, win_point.yz)  6 total  7-9 memo[0..11] (left untouched when the
// primary table holds the whole memo: RenderArgs::primary_complete)  10 (closest, object, any | inside << 1, sphere
// dst)  11 (win_u, win_v, win_tri, win_point.x)  12 unused  13 the walk's result (t, u, v, tri | det sign; tri = ~0:
// no hit), written by rt_walk_kernel.  A park writes 9 planes (144 B) and a resume reads 11 (176 B) in the usual case.
constexpr uint32_t PARK_PLANES = 14;

constexpr uint32_t RT_MAX_BATCH_FRAMES = 64;  // (config 3 stand-in: 5.24 ms per frame at 32 per launch, 5.09 at 64; config 2: 1.126 / 1.119)
// rt_blend_frames_kernel: image = the accumulation image, scratch = batch frame 0
struct BlendArgs {
    float4* image;
    const float4* scratch;
    unsigned long long texels, stride;
    uint32_t n;
    int32_t frames0;  // Params.frames of batch frame 0
    float weight[RT_MAX_BATCH_FRAMES], rest[RT_MAX_BATCH_FRAMES];  // wgsl:157-158 per frame, from the host
};

}  // namespace rtd

#endif
