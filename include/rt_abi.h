/*
 * rt_abi.h -- C ABI of librt2_mi355x.so, the MI355X-native replacement for the
 * per-pixel path-trace loop of addiswebb/ray_tracer_2.
 *
 * Everything in this header is plain C: fixed-layout POD structs, pointers and
 * sizes.  No torch, HIP or C++ types cross the boundary.  A Rust (or any other
 * FFI-capable) host binds exactly these symbols; INTEGRATION.md shows the
 * replacement `src/rendering/ray_tracer.rs` a maintainer would add.
 *
 * Section 1 reproduces, byte for byte, the `#[repr(C)]`/bytemuck structs the
 * reference uploads to its WGSL shader (citations are relative to the
 * reference checkout).  Section 2 is the device-side API that replaces
 * `RayTracer::{new, create_gpu_resources, load_scene_gpu_resources,
 * update_buffers, render}` (src/rendering/ray_tracer.rs:49,316,237,397,420).
 * Section 3 is the host-side scene pipeline (OBJ/MTL loader, SAH BVH builder,
 * built-in scene library) that feeds it, replacing `AssetManager::load_model`
 * (src/core/asset.rs:102), `BVH::build_per_mesh` (src/core/bvh.rs:152) and
 * `Scene::instantiate_scene` (src/scene/scene.rs:179).
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* 1. Data contract (byte-exact with the reference)                          */
/* ------------------------------------------------------------------------ */

/* src/core/app.rs:27-40  <->  shaders/ray_tracer.wgsl:2-12 */
typedef struct rt_params {
    uint32_t width;
    uint32_t height;
    int32_t number_of_bounces; /* a path has number_of_bounces + 1 segments */
    int32_t rays_per_pixel;
    int32_t skybox;
    int32_t frames; /* the seed: -1 after a reset, 0 first frame, ... */
    int32_t accumulate;
    int32_t debug_flag; /* 0 = path trace, 1..7 = debug views */
    int32_t debug_scale;
    float _p1[3];
} rt_params;

/* src/scene/components/material.rs:3-18  <->  wgsl:14-27 */
typedef struct rt_material {
    float color[4];
    float emission_color[4];
    float specular_color[4];
    float absorption[4];
    float absorption_strength;
    float emission_strength;
    float smoothness;
    float specular;
    float ior;
    int32_t flag; /* RT_MATERIAL_* */
    int32_t diffuse_index;
    int32_t normal_index;
} rt_material;

enum { RT_MATERIAL_DEFAULT = 0, RT_MATERIAL_GLASS = 1, RT_MATERIAL_TEXTURE = 2 };

/* src/scene/components/geometry/sphere.rs:4-10  <->  wgsl:29-33 */
typedef struct rt_sphere {
    float pos[3];
    float radius;
    rt_material material;
} rt_sphere;

/* src/scene/components/geometry/mesh.rs:52-62  <->  wgsl:35-42.
 * Matrices are column-major ([col][row]), as glam's to_cols_array_2d. */
typedef struct rt_mesh_uniform {
    float world_to_model[4][4];
    float model_to_world[4][4];
    uint32_t node_offset;
    uint32_t triangles;
    uint32_t triangle_offset;
    float _p1;
    rt_material material;
} rt_mesh_uniform;

/* src/core/bvh.rs:55-66  <->  wgsl:60-67.  left/right are mesh-local node
 * indices, first is a mesh-local triangle index, count > 0 marks a leaf. */
typedef struct rt_node {
    uint32_t left;
    uint32_t right;
    uint32_t first;
    uint32_t count;
    float aabb_min[3];
    float _p1;
    float aabb_max[3];
    float _p2;
} rt_node;

/* src/core/bvh.rs:19-34  <->  wgsl:69-82 */
typedef struct rt_packed_triangle {
    float v1[3];
    float uv10;
    float v2[3];
    float uv11;
    float v3[3];
    float uv20;
    float n1[3];
    float uv21;
    float n2[3];
    float uv30;
    float n3[3];
    float uv31;
} rt_packed_triangle;

/* src/scene/camera.rs:15-22  <->  wgsl:44-49 (84 bytes on the host) */
typedef struct rt_camera_uniform {
    float cam_to_world[4][4];
    float view_params[3]; /* plane_width, plane_height, focus_dist */
    float defocus_strength;
    float diverge_strength;
} rt_camera_uniform;

/* src/scene/scene.rs:1016-1026  <->  wgsl:51-58.  The host layout is kept
 * as is, including `nodes` at offset 100 (the shader never reads it). */
typedef struct rt_scene_uniform {
    uint32_t spheres;
    uint32_t n_vertices;
    uint32_t n_indices;
    uint32_t meshes;
    rt_camera_uniform camera;
    uint32_t nodes;
    float padding[6];
} rt_scene_uniform;

/* One RGBA8 sRGB texture as the reference uploads it
 * (src/rendering/ray_tracer.rs:237-275): tightly packed rows, already
 * flipped horizontally by the loader (src/core/asset.rs:77). */
typedef struct rt_texture_desc {
    const uint8_t* rgba8;
    uint32_t width;
    uint32_t height;
} rt_texture_desc;

/* Capacity limits the reference's fixed-size buffers impose
 * (src/rendering/ray_tracer.rs:15-19, src/core/bvh.rs:140); kept as input
 * validation. */
#define RT_MAX_MESHES 400u
#define RT_MAX_SPHERES 500u
#define RT_MAX_TRIANGLES 1375000u
#define RT_MAX_NODES 2600000u
#define RT_MAX_TEXTURES 64u
#define RT_BVH_STACK 32u /* wgsl:297 */

/* status codes: 0 = ok, negative = error (text via rt_last_error) */
enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_CAPACITY = -2,
    RT_ERR_DEVICE = -3,
    RT_ERR_NO_SCENE = -4,
    RT_ERR_BVH_DEPTH = -5,
    RT_ERR_IO = -6,
    RT_ERR_PARSE = -7,
    RT_ERR_OUT_OF_MEMORY = -8,
    RT_ERR_INDEX_RANGE = -9
};

/* ------------------------------------------------------------------------ */
/* 2. Device side: replaces RayTracer (src/rendering/ray_tracer.rs)          */
/* ------------------------------------------------------------------------ */

typedef struct rt_handle rt_handle;

/* Counters accumulated over the rt_render* calls since the last
 * rt_reset_timing (device-side atomics, one add per wave).  `segments` counts
 * calculate_ray_collions calls (wgsl:353) = rays. */
typedef struct rt_stats {
    uint64_t segments;
    uint64_t paths;
    uint64_t node_tests;     /* AABB tests, as wgsl:322 counts them */
    uint64_t triangle_tests; /* as wgsl:307 counts them */
    float kernel_ms;         /* sum of the hipEvent times of `launches` render launches (pipelined single frames --
                              * option "pipeline" -- overlap: their sum then exceeds the wall time) */
    uint32_t launches;       /* render launches since the last rt_reset_timing */
    uint32_t frames;         /* frames the host asked for since the last rt_reset_timing (rt_render_frames: several per launch) */
    uint64_t segments_reused; /* of `segments`: primary segments whose hit was taken from the per-pixel memo
                               * (same ray as the pixel's first sample: no traversal ran); segments -
                               * segments_reused rays were traversed */
    uint32_t frames_speculative; /* frames rendered AHEAD of the host's calls (option "frame_ahead") that no call has asked for
                               * (yet): the launches, rays and times above include them; a sequence that ends mid-batch
                               * leaves them here */
    uint32_t _reserved;
} rt_stats;

/* ≙ RayTracer::new + create_gpu_resources (ray_tracer.rs:49,316): picks the
 * device, creates the stream and the RGBA32F accumulation image of up to
 * max_width x max_height texels (engine.rs:142-158). */
int rt_create(int device_ordinal, uint32_t max_width, uint32_t max_height, rt_handle** out);

/* ≙ RayTracer::update_buffers (ray_tracer.rs:397-419), to be called when the
 * scene changes instead of every frame.  All arrays are copied before return.
 * Validates capacities and BVH indices (ranges, cycles).  A BVH of height >= 32 --
 * the shader's 32-entry stack (wgsl:297) can overflow on it -- is accepted and traversed
 * with the shader's literal push/pop and index clamping, so it renders exactly as the
 * shader would. */
int rt_upload_scene(rt_handle* h, const rt_scene_uniform* scene, const rt_sphere* spheres,
                    uint32_t n_spheres, const rt_mesh_uniform* meshes, uint32_t n_meshes,
                    const rt_packed_triangle* triangles, uint32_t n_triangles,
                    const rt_node* nodes, uint32_t n_nodes);

/* ≙ RayTracer::load_scene_gpu_resources (ray_tracer.rs:237-315). n <= 64. */
int rt_upload_textures(rt_handle* h, const rt_texture_desc* descs, uint32_t n);

/* The cheap per-frame part of update_buffers: only the camera changed. */
int rt_set_camera(rt_handle* h, const rt_camera_uniform* camera);

/* ≙ RayTracer::render (ray_tracer.rs:420-434): one frame of wgsl `main` over
 * params->width x params->height; accumulates into the device image when
 * params->frames >= 1 (wgsl:154-161).  Asynchronous on the handle's stream. */
int rt_render(rt_handle* h, const rt_params* params);

/* n_frames consecutive frames of RayTracer::render with Params.frames advancing by one per frame, as
 * App::update does while accumulating (app.rs:44-53, 160-162): the image afterwards is bit-identical to
 * n_frames calls of rt_render with frames, frames + 1, ...  The only dependency between frames is the
 * per-texel blend (wgsl:154-161), so the frames of a batch (option "batch_frames", default 32, at most
 * 64) are sampled by ONE persistent launch over (frame, tile) work items -- the waves never drain
 * between frames -- into scratch images, and a dense second kernel blends them in frame order with
 * the shader's two operations.  Intermediate frames of a batch are not observable. */
int rt_render_frames(rt_handle* h, const rt_params* params, uint32_t n_frames);

/* Multi-GPU tile split: renders only the 8-row strips s with
 * s % world == rank into the compact local image (strip-major).  Seeds use
 * the full-frame pixel index, so the stitched image equals rt_render's. */
int rt_render_strips(rt_handle* h, const rt_params* params, uint32_t rank, uint32_t world);

/* rt_render_frames for one rank's strips. */
int rt_render_strips_frames(rt_handle* h, const rt_params* params, uint32_t n_frames, uint32_t rank,
                            uint32_t world);

/* Number of texels rt_render_strips(rank, world) writes. */
uint64_t rt_strip_texels(uint32_t width, uint32_t height, uint32_t rank, uint32_t world);

/* Scatter a gathered [world][max_local_texels][4] float device buffer
 * (rank-major, each rank padded to rt_strip_texels(.., 0, world)) into the
 * handle's full-frame image. */
int rt_assemble_strips(rt_handle* h, const void* gathered_device, uint32_t width, uint32_t height,
                       uint32_t world);

/* Single-process multi-GPU frame: per_gpu[r] (one handle per device, same scene
 * uploaded to each) renders the strips of rank r, per_gpu[0]'s device gathers them
 * device-to-device over xGMI and assembles the frame; if rgba32f_out is not NULL the
 * full frame (width*height RGBA32F) is copied there (blocking).  Every handle keeps
 * its own strip accumulation across frames.  The frame is bit-identical to rt_render's. */
int rt_render_multi(rt_handle** per_gpu, int n_gpus, const rt_params* params, float* rgba32f_out);
/* The same for n_frames consecutive frames (rt_render_strips_frames on every GPU), ONE gather at the
 * end.  Transport of the gather: RCCL (grouped ncclSend/ncclRecv, every peer on its own xGMI link to the
 * root; librccl is loaded on first use) between distinct devices, device-to-device copies (peer access
 * enabled where the devices allow it) between handles that share a device or when option "multi_rccl"
 * of per_gpu[0] is 0.  Non-blocking when rgba32f_out is NULL: a later call may follow at once (each
 * rank's next render waits for the root's copy of its previous strips). */
int rt_render_multi_frames(rt_handle** per_gpu, int n_gpus, const rt_params* params, uint32_t n_frames,
                           float* rgba32f_out);
/* Blocking read of the frame the last rt_render_multi* call assembled on the root. */
int rt_read_multi_frame(rt_handle* root, float* rgba32f_out, size_t bytes);

/* ≙ copy_texture_to_buffer in save_render_to_file (app.rs:341-407): blocking
 * copy of width*height RGBA32F texels (row 0 = bottom of the view). */
int rt_read_image(rt_handle* h, float* rgba32f_out, size_t bytes);
/* The two halves of rt_read_image for a host that shows every frame (the reference blits the storage texture every
 * redraw, src/rendering/renderer.rs): rt_snapshot_image keeps the first `bytes` of the image as it is after the calls
 * made so far -- a device-to-device copy in stream order, it does not block --, rt_read_snapshot brings that copy to the
 * host on a stream of its own and blocks until it has arrived, NOT until later frames are rendered:
 *     rt_render(k); rt_snapshot_image(h, n); rt_render(k + 1); rt_read_snapshot(h, out, n);   // frame k, while k + 1 renders
 * costs max(render, read) per frame instead of their sum (config 2: 1.13 + 0.59 ms). */
int rt_snapshot_image(rt_handle* h, size_t bytes);
int rt_read_snapshot(rt_handle* h, float* rgba32f_out, size_t bytes);
/* Restore a previously read accumulation image (checkpoint/resume). */
int rt_write_image(rt_handle* h, const float* rgba32f_in, size_t bytes);

int rt_synchronize(rt_handle* h);
int rt_get_stats(rt_handle* h, rt_stats* out);
/* Shape of the last render launch (what a profile summary needs next to the code object's static resources):
 * out[0] = dynamic LDS bytes per workgroup, out[1] = workgroups of the render kernel, out[2] = 1 when the scene blob
 * was staged into LDS, out[3] = bit 0: many-mesh kernels, bit 1: specialised instantiation, bit 2: one-wave-per-tile
 * variant, bit 3: a deferred-walk sequence ran, bit 4: a wavefront sequence ran (then out[0] / out[1] are the walk kernel's);
 * out[4] = MiB of device memory the handle holds on its own initiative right now (batch and pipeline scratch images, primary
 * tables, global-memory memos, park queues, snapshot: what option "max_device_mb" bounds), out[5] = that cap (0 = none). */
int rt_last_launch(rt_handle* h, uint32_t out[6]);
/* Zero the counters and forget the recorded launch times. */
int rt_reset_timing(rt_handle* h);
/* Run this handle's work on a caller-owned HIP stream (e.g. the stream a
 * collective library orders its transfers on) so render -> gather -> assemble
 * need no host synchronisation; NULL restores the internal stream. */
int rt_set_stream(rt_handle* h, void* hip_stream);
/* Render into caller-owned device memory of `texels` RGBA32F texels (e.g. a
 * buffer a collective library will send); NULL restores the internal image. */
int rt_bind_image(rt_handle* h, void* device_ptr, uint64_t texels);
/* Tuning knobs.  The image, the ray count and the test counters NEVER depend on them (every option is swept by the
 * parity tests); they choose between schedules and data placements.  Unknown names and out-of-range values return
 * RT_ERR_INVALID_ARGUMENT.  "(upload)" = takes effect at the next rt_upload_scene.
 *
 *   name                  values (default)        meaning
 *   --------------------  ----------------------  ---------------------------------------------------------------
 *   kernel_variant        -1 / 0 / 1 (-1)         -1 automatic; 0 persistent waves with active-lane refill; 1 one wave
 *                                                 per 8x8 tile (chosen automatically at <= 1.25 tiles per resident wave)
 *   persistent_blocks     >= 1 (CUs x 5)          grid of variant 0 (256-thread workgroups)
 *   specialise            0 / 1 (1)               1: scenes without spheres, glass and textured materials, rendered by a
 *                                                 camera without jitter, run kernels with those branches compiled out
 *   lds_scene             0 / 1 (1)               0: never stage the scene blob into LDS
 *   pixel_cache           0 / 1 / 2 (1)           per-pixel memo of the primary ray and its hit: off / in LDS when it
 *                                                 fits (else global memory) / always in global memory
 *   primary_table         0 / 1 (1)               0: compute the memoised primary ray per pixel in the render kernel
 *                                                 instead of once per (camera, frame size)
 *   primary_hits          0 / 1 (1)               the primary table also holds every pixel's primary HIT (once per camera, frame
 *                                                 size, strip layout and scene): while the camera stands still no primary ray is
 *                                                 traversed at all; 0: a pixel's first sample of every frame traverses it again
 *   max_device_mb         >= 0 (0)                upper bound (MiB) on the device memory the library takes on its OWN initiative: batch
 *                                                 and pipeline scratch images (64 x 33 MB for a 1080p batch), primary tables (133 MB),
 *                                                 global-memory memos, the park queues of deferred walks (up to half of the free
 *                                                 memory), the snapshot; not the image, the scene and the textures.  What does not
 *                                                 fit runs the plainer path -- smaller batches, no pipeline, no table, no deferred
 *                                                 walks: same image, other frame time; rt_last_launch reports what is held.  0 = no cap
 *   memo_in_table         0 / 1 (1)               kernels whose memo has no room in LDS read it in place from a complete primary
 *                                                 table instead of copying it into a buffer in global memory (round 5: that copy
 *                                                 was 96 % of what those kernels wrote to memory); 0: the copy
 *   vote_eighths          -1 / 0..8 (-1)          intersection vote: traverse when wanting lanes x 8 >= lanes x this
 *   vote_patience         -1 / >= 0 (-1)          ... or when some lane has waited this many iterations; -1: by the kind of
 *                                                 launch (6 and 3 for a scene in LDS on the few-mesh kernels, 7 and 16 for
 *                                                 the others, 8 and 16 inside a deferred-walk sequence)
 *   tile_feedback         0 / 1 (1)               order the tiles by an earlier frame's rays per tile, heaviest first
 *   tile_feedback_period  >= 1 (8)                frames an order is kept before it is refreshed
 *   pipeline              -1 / 0 / 2 .. 8 (-1)    frames in flight: consecutive rt_render calls sample into that many scratch images,
 *                                                 each on an internal stream of its own, and are blended in frame order on the
 *                                                 handle's stream: frame k + 1's launch takes the CUs frame k's draining waves free;
 *                                                 every frame stays observable (config 2: 1.51 / 1.27 / 1.23 / 1.20 ms per frame at
 *                                                 0 / 2 / 3 / 4, config 3 stand-in 7.26 / 6.83 at 3 / 4).  Needs a hardware queue per
 *                                                 stream: -1 (and 1) = four frames when the host has set GPU_MAX_HW_QUEUES >= 5, three
 *                                                 otherwise; seven for a rank of a strip split of 4 ranks and more when there are
 *                                                 12 queues (the library never touches the environment)
 *   pipeline_when_idle    0 / 1 (0)               1 = also pipeline a frame that finds the handle's stream idle; by default such a
 *                                                 frame -- a host that renders, reads, renders: nothing to overlap with -- takes the
 *                                                 plain in-place launch (no scratch image, no blend kernel)
 *   primary_per_slot      0 / 1 (1)               a pipelined frame whose camera is not the shared primary table's builds a table
 *                                                 of its own pipeline slot on its own stream instead of rewriting the shared one
 *                                                 behind a barrier (the frames right after a camera has stopped; while it MOVES --
 *                                                 the camera differs from the previous frame's -- frames render without a table)
 *   frame_ahead           -1 / 0 / 2..64 (-1)     one-frame calls (rt_render, rt_render_strips, rt_render_multi) that continue an
 *                                                 accumulation (same parameters, camera, scene and options, frames = f, f + 1, ...):
 *                                                 the call for frame f renders frames f .. f + d - 1 in one batched launch and blends
 *                                                 frame f; the next d - 1 calls only blend theirs (the image after every call is
 *                                                 bit-identical; a call that does not continue the sequence drops the rest).  -1 =
 *                                                 automatic: ONLY for a host that runs ahead of the device (the call finds the
 *                                                 handle's stream busy) -- a host that renders, waits and renders again never has a
 *                                                 frame held back behind frames it has not asked for (round 5; before, its longest
 *                                                 wait grew from 1.5 to 8 ms) and opts in with an explicit depth --, and then for an
 *                                                 LDS-resident scene and a share too small for a launch of its own to keep the lanes
 *                                                 full (batches of about 4 ms: 28 / 14 / 7 frames for a strip share of 8 / 4 / 2 ranks
 *                                                 of config 2, none for the whole frame), reached by doubling from 2 so that a
 *                                                 sequence of n frames renders at most n in vain; never for a scene read from global
 *                                                 memory (rays of unknown cost: a host asks for batches there itself -- frame_ahead = 8:
 *                                                 config 5's geometry 4.43 -> 3.40 ms per call); 0 = off; 2 .. 64 = that many frames per batch whatever
 *                                                 the host does (the first call of a batch returns its frame after the whole batch).
 *                                                 rt_get_stats: `frames` counts the frames asked for, `frames_speculative` the ones
 *                                                 rendered ahead that no call has asked for (yet); rays and times include both
 *   cross_prune           0 / 1 (0)               many-mesh kernels: boxes and meshes whose entry distance lies beyond a bound derived
 *                                                 from the closest hit so far (error budget + 12.5 % slack) are not
 *                                                 entered; never in the counter / debug kernels.  OFF by default since round 5: exact
 *                                                 on every random ray ever compared (102 G), but a ray that grazes a far triangle at
 *                                                 about 1e-6 rad from an origin within about 1e-4 of that triangle's plane, in a
 *                                                 generic orientation, gets a t from the shader's own arithmetic that the 12.5 % do
 *                                                 not cover (tests/test_gpu_prune_directed.py: 1 texel in 2 M such rays).  1 = take
 *                                                 the 35-45 % shorter many-mesh frames and that risk
 *   batch_frames          1..64 (32)              frames per launch of rt_render_frames
 *   batch_tile_major      0 / 1 (1)               a batch's work items in (tile, frame) order instead of (frame, tile)
 *   forest                0 / 1 (1)  (upload)     BVH meshes of one local space walked per lane back to back
 *   flat2                 0 / 1 (1)  (upload)     meshes whose BVH is a root with two leaves run as straight-line code
 *   stack_wide            -1 / 0 / 1 (-1)         BVH stack entries: automatic / one dword when legal / two dwords
 *   tlas                  0 / 1 (1)  (upload)     top-level trees over the root boxes of many meshes in one local space
 *   tlas_min              >= 2 (8)   (upload)     smallest run of meshes that gets a top-level tree
 *   cull_roots            -1 / 0 / 1 (-1)         results-preserving root-box culling in the mesh loop (-1: from 16 meshes)
 *   sort_rounds           -1 / 0 / 1..64 (-1)     deferred walks (one big mesh among a few): park pixels in front of the
 *                                                 mesh, walk it for all parked rays in a kernel of its own, resume --
 *                                                 for that many rounds; -1: by the work of the launch and the mesh's size,
 *                                                 and only while both park queues fit half of the free memory
 *   defer_min_nodes       >= 1 (1024) (upload)    smallest BVH (internal nodes) whose mesh may be the deferred one
 *   fast_miss             0 / 1 (1)               a memoised primary ray that leaves the scene ends its pixel in one step: the
 *                                                 remaining samples are that segment again and their light is added in order
 *   park_levels           0 / 1 (1)               a parking launch runs the deferred walk's first two levels inline and parks
 *                                                 only the rays that reach a grandchild box (0: every ray that can hit the root box)
 *   multi_rccl            0 / 1 / 2 (1)           gather of rt_render_multi: device-to-device copies / RCCL between
 *                                                 distinct devices, copies otherwise (falls back to copies when librccl
 *                                                 cannot be loaded) / RCCL or an error
 *
 * Only in a library built with -DRT_EXPERIMENTS=1 (tools/build_variant.sh exp -DRT_EXPERIMENTS=1; rt_version() then ends in
 * "+experiments") -- features that were built, parity-tested and measured slower than what ships; the product library
 * accepts 0 for them and rejects anything else with RT_ERR_INVALID_ARGUMENT:
 *
 *   lds_top               -1 / 0 / N (0)          wide BVH records of the biggest mesh staged into LDS when the scene is read
 *                                                 from global memory: what fits at full occupancy / none / N (<= 2048)
 *   lds_tlas              0 / 1 / 2 (0)           top-level tree records staged into LDS when the scene is read from global
 *                                                 memory: none / the top levels that fit at full occupancy / the whole tree
 *                                                 (measured slower both ways)
 *   hybrid                0 / 1 (0)               the parking launches of a deferred-walk sequence stage everything but the big
 *                                                 mesh into LDS (measured no faster: DESIGN.md section 5.4)
 *   wavefront             0 / 1 (0)               wavefront sequences (many-mesh scenes): path state in memory slots, a shading
 *                                                 kernel and a ray-walk kernel with per-lane refill alternate (measured slower
 *                                                 than the inline kernels: DESIGN.md section 5.5)
 */
int rt_set_option(rt_handle* h, const char* name, int value);
/* Enable/disable the optional per-ray counters (node/triangle tests). */
int rt_set_counters(rt_handle* h, int enabled);
/* Raw device pointer of the image / stream, for zero-copy interop
 * (torch.distributed gathers over RCCL operate on this memory). */
void* rt_device_image(rt_handle* h);
void* rt_stream(rt_handle* h);

/* The test-only entry points (rt_test_*: the kernels' arithmetic building blocks evaluated element-wise on the device, raw
 * copies of sequence buffers, the RCCL gather against a stub, the frame_ahead policy) are NOT exported by the product
 * library: include/rt_test_abi.h declares them and ray_tracer_2_amd/librt2_mi355x_test.so -- the same sources compiled
 * with -DRT_TEST_ENTRIES=1 -- exports them beside everything declared here (round 5). */

const char* rt_last_error(rt_handle* h);
void rt_destroy(rt_handle* h);

/* Library-level queries (no device needed). */
const char* rt_version(void);
int rt_device_count(void);
/* sizeof() of every struct above, for FFI self-checks:
 * params, material, sphere, mesh, node, triangle, camera, scene. */
void rt_abi_sizes(uint32_t out[8]);

/* ------------------------------------------------------------------------ */
/* 3. Host side: scene pipeline (no device needed)                           */
/* ------------------------------------------------------------------------ */

typedef struct rt_scene rt_scene;

/* ≙ Scene::from_name + Scene::instantiate_scene (scene.rs:1003,179).
 * name: "cornell_box", "room", "room_2", "metal", "balls", "sponza",
 * "texture_test", "obj_test", "random_balls" / "random_balls:<seed>" (the reference draws this
 * scene from an OS-seeded generator, scene.rs:403, so its own runs never agree; this build makes
 * the same draws from a seeded one: a documented divergence).  assets_dir ≙ CARGO_MANIFEST_DIR/assets (asset.rs:50,108). */
int rt_scene_load_builtin(const char* name, const char* assets_dir, rt_scene** out);

/* Scene-definition API ≙ SceneDefinition::{set_camera, add_mesh, add_sphere}
 * (scene.rs:75-99). */
typedef struct rt_transform {
    float pos[3];
    float rot[4]; /* quaternion x, y, z, w */
    float scale[3];
} rt_transform;

typedef struct rt_camera_desc { /* ≙ CameraDescriptor (camera.rs:38-66) */
    rt_transform transform;
    float fov;
    float aspect;
    float near_plane;
    float far_plane;
    float focus_dist;
    float defocus_strength;
    float diverge_strength;
} rt_camera_desc;

int rt_scene_create(rt_scene** out);
int rt_scene_set_camera(rt_scene* s, const rt_camera_desc* cam);
/* ≙ Transform::cam (transform.rs:13-19): look-at camera transform. */
void rt_transform_cam(const float origin[3], const float look_at[3], rt_transform* out);
int rt_scene_add_sphere(rt_scene* s, const float centre[3], float radius, const rt_material* m);
/* MeshDefinition::FromFile (mesh.rs:33-36) */
int rt_scene_add_obj(rt_scene* s, const char* assets_dir, const char* path,
                     const rt_transform* t, int use_mtl, const rt_material* m);
/* MeshDefinition::FromData: n_vertices x (pos[3], normal[3], uv[2]) floats */
int rt_scene_add_mesh_data(rt_scene* s, const float* vertices8, uint32_t n_vertices,
                           const uint32_t* indices, uint32_t n_indices,
                           const rt_transform* t, const rt_material* m);
/* Texture registration for material.diffuse_index; returns the index or <0. */
int rt_scene_add_texture_rgba8(rt_scene* s, const uint8_t* rgba8, uint32_t w, uint32_t h);
/* ≙ BVH::build_per_mesh(meshes, Quality::High) (bvh.rs:152-207);
 * quality: 0 = Low, 1 = High, 2 = Disabled (bvh.rs:126-131). */
int rt_scene_build(rt_scene* s, int quality);
/* The same build with the SAH plane searches (find_best_split, src/core/bvh.rs:299-351: the
 * O(150 n)-per-level part) of meshes with at least `min_triangles` triangles (0 = default 16384)
 * done on GPU `device`; nodes, node order and triangle order are bit-identical to rt_scene_build's
 * (evaluate_sah is order-independent: minima, maxima and integer counts).  device = -1 runs the
 * same level-wise build with the searches on the host (validation).  SURVEY 8(f)-4. */
int rt_scene_build_device(rt_scene* s, int quality, int device, uint32_t min_triangles);

/* Accessors: pointers stay valid until the scene is modified or destroyed. */
int rt_scene_get_uniform(const rt_scene* s, rt_scene_uniform* out); /* ≙ Scene::to_uniform */
uint32_t rt_scene_num_spheres(const rt_scene* s);
uint32_t rt_scene_num_meshes(const rt_scene* s);
uint32_t rt_scene_num_triangles(const rt_scene* s);
uint32_t rt_scene_num_nodes(const rt_scene* s);
uint32_t rt_scene_num_textures(const rt_scene* s);
const rt_sphere* rt_scene_spheres(const rt_scene* s);
const rt_mesh_uniform* rt_scene_meshes(const rt_scene* s);
const rt_packed_triangle* rt_scene_triangles(const rt_scene* s);
const rt_node* rt_scene_nodes(const rt_scene* s);
int rt_scene_get_texture(const rt_scene* s, uint32_t i, rt_texture_desc* out);
const char* rt_scene_mesh_label(const rt_scene* s, uint32_t i);
/* Raw (pre-BVH) geometry of mesh instance i, in file order: n_vertices x
 * (pos[3], normal[3], uv[2]) floats and the index list (≙ MeshData,
 * geometry/mesh.rs:8-12).  Pass NULL outputs to query the counts. */
int rt_scene_mesh_data(const rt_scene* s, uint32_t i, float* vertices8, uint32_t* n_vertices,
                       uint32_t* indices, uint32_t* n_indices, rt_transform* transform,
                       rt_material* material);
uint32_t rt_scene_num_mesh_instances(const rt_scene* s);
const char* rt_scene_last_error(const rt_scene* s);
void rt_scene_destroy(rt_scene* s);

/* Convenience: upload everything a built scene holds to a device handle
 * (≙ load_scene_gpu_resources + update_buffers). */
int rt_upload_built_scene(rt_handle* h, const rt_scene* s);

/* Uniform n x n barycentric split of every triangle of every mesh added so
 * far (SURVEY 8d stand-in for the missing Dragon_80K / dragon_large assets). */
int rt_scene_subdivide_meshes(rt_scene* s, uint32_t n);

/* ≙ the pixel loop of save_render_to_file (app.rs:408-460): gamma 1/2.2,
 * clamp, truncating u8, net vertical flip.  out: width*height*4 bytes. */
int rt_export_rgba8(const float* rgba32f, uint32_t width, uint32_t height, uint8_t* out);

#ifdef __cplusplus
}
#endif

#endif /* RT_ABI_H */
