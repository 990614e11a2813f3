// rt_api.hip -- section 2 of include/rt_abi.h: the device-side C ABI that
// replaces RayTracer (src/rendering/ray_tracer.rs) of the reference.
//
// Ownership and threading follow the reference's use of RayTracer: one handle
// per device, calls from one thread at a time, inputs are borrowed and copied
// before the call returns, errors are integer codes (never exceptions or
// aborts across the boundary).
#include <hip/hip_runtime.h>


#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_rccl.h"  // (types of <rccl/rccl.h>; the library itself is loaded on first multi-device use)
#ifndef RT_TEST_ENTRIES
#define RT_TEST_ENTRIES 0
#endif
#if RT_TEST_ENTRIES
#include "../../include/rt_test_abi.h"
#endif
#include "rt_srgb_lut.h"

namespace rtd {
hipError_t launch_render(const RenderArgs& a, hipStream_t stream);
size_t render_lds_bytes(const RenderArgs& a);
hipError_t launch_tile_order(const uint32_t* cost, uint32_t n_tiles, uint32_t max_cost, uint32_t* order,
                             hipStream_t stream);
hipError_t launch_primary(const RenderArgs& a, void* table, bool with_hits, hipStream_t stream);
hipError_t launch_blend_frames(const BlendArgs& b, hipStream_t stream);
hipError_t launch_walk(const RenderArgs& a, uint32_t compute_units, hipStream_t stream);
hipError_t launch_wf_shade(const RenderArgs& a, uint32_t blocks, hipStream_t stream);
hipError_t launch_wf_walk(const RenderArgs& a, uint32_t blocks, hipStream_t stream);
size_t wf_walk_lds_bytes(const RenderArgs& a);
#if RT_TEST_ENTRIES
hipError_t launch_units(int fn, const float* x, const float* y, float* out, unsigned long long n, hipStream_t stream);
hipError_t launch_sweep(int which, unsigned long long* out, hipStream_t stream);
hipError_t launch_units_texture(const uint8_t* rgba8, uint32_t width, uint32_t height, const float* srgb_lut, const float* uv,
                                float* out, unsigned long long n, hipStream_t stream);
#endif
hipError_t launch_assemble(const float4* gathered, float4* image, uint32_t width, uint32_t height,
                           uint32_t world, unsigned long long pad_texels, hipStream_t stream);
#if defined(RT_DIAG) || defined(RT_DIAGT)
hipError_t diag_read(unsigned long long* out, bool reset);
#endif
#if defined(RT_DIAG) || defined(RT_WAVE_TIMES)
hipError_t diag_wave_times(unsigned long long* out);
#endif
}  // namespace rtd

using namespace rtd;

struct rt_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;  // created by rt_create; `stream` may be rebound (rt_set_stream)
    // one (start, stop) event pair per launch since the last rt_reset_timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    // launches whose event pairs were harvested when the pool wrapped (rt_get_stats adds the live ones)
    double ev_ms_harvested = 0.0;
    unsigned long long launches_total = 0, frames_total = 0;  // since rt_reset_timing (frames_total: frames the host asked for)
    unsigned long long frames_speculative = 0;         // frames rendered ahead (option "frame_ahead") that no call has asked for (yet)
    // frame batches (rt_render_frames): scratch images of the frames in flight
    float4* batch_scratch = nullptr;
    size_t batch_scratch_texels = 0;
    // deferred walks (RenderArgs::park): the deferred mesh found at upload, the two park queues, their counters
    bool have_defer = false;
    uint32_t defer_mesh = 0, defer_xform = 0, defer_internal = 0;  // (internal nodes of its BVH)
#if RT_WALK2
    float4* walk2 = nullptr;           // experiment: two-level records of the deferred mesh (rt_device.h)
    uint32_t walk2_base = 0;
#endif
    int defer_min_nodes = 1024;  // option "defer_min_nodes": smallest BVH (internal nodes) that is worth deferring (next upload; tests lower it)
    // option "sort_rounds": walk-and-resume rounds of a deferred-walk sequence; 0 = off, -1 (default) = automatic: by
    // the work of the launch in units of one 1920 x 1080 frame at 16 samples per pixel and the size of the big mesh
    // (render_impl; none below 8 units, or 2 for a mesh of 400 k internal nodes and more: a round has a fixed cost, its
    // longest chain of dependent segments, that only a big launch amortises) -- and off when the two park queues (224 B per pixel and frame of the
    // batch, each) would take more than a quarter of the free device memory
    int sort_rounds = -1;
    // wavefront sequences (RenderArgs::wf_*): option "wavefront" 0 (default) = off, 1 = whenever legal (many-mesh scenes
    // without a literal-stack mesh).  Off by default: measured slower than the inline kernels -- sponza-sized stand-in
    // 10.65 -> 11.3 ms per frame, 200-mesh stand-in 5.0 -> 7.4 (tools/experiments/README.md)
    int wavefront = 0;
    bool any_deep = false;            // some mesh is walked with the shader's literal stack (not in the walk kernel)
    float4* wf_state = nullptr;
    float4* wf_hit = nullptr;
    uint32_t* wf_lists = nullptr;     // two lists of wf_capacity slots
    uint32_t* wf_counts = nullptr;
    size_t wf_capacity = 0, wf_counts_capacity = 0;
    // Hybrid launches of a deferred-walk sequence (option "hybrid"): everything but the deferred mesh's BVH and triangles
    // -- mesh records, materials, items, the small meshes' records and triangles -- as a blob of its own that the
    // parking render launches stage into LDS (they never walk the big mesh); only a winner on the big mesh reads its
    // shading record from the full blob.
    // Pipelined single frames (option "pipeline"): consecutive rt_render calls sample into two or three scratch images, each
    // on an internal stream of its own, and are blended in frame order on the handle's stream, so frame k + 1's launch takes the CUs that
    // frame k's draining waves free -- every frame stays observable (rt_read_image after any call returns that frame).
    int pipeline = -1;                                 // option "pipeline": frames in flight (0 = off, 2 .. 4, -1 = automatic_pipeline_depth(); config 2: 1.51 / 1.27 / 1.23 / 1.20 ms per frame; 5 and 6: 1.25, 1.24)
    static constexpr int PIPE_MAX = 8;                 // most frames in flight (option "pipeline")
    // option "pipeline_when_idle" (default 0): a frame that finds the handle's stream idle -- a host that renders, reads and
    // only then renders again, the reference's present loop -- has nothing to overlap with and takes the plain in-place
    // launch (no scratch image, no blend kernel, no event hops); 1 = pipeline such frames too
    int pipeline_when_idle = 0;
    // Frames rendered ahead (option "frame_ahead"): a host that accumulates calls rt_render with the same parameters and
    // frames = f, f + 1, f + 2, ...  When a share is so small that one launch per frame cannot keep the lanes full (a strip
    // share of four or eight ranks, a small window), the call for frame f renders frames f .. f + d - 1 in ONE batched launch
    // (the persistent kernel refills its lanes across frames) into the batch scratch images and blends frame f only; the
    // calls for f + 1 .. f + d - 1 find their frame rendered and only blend it -- same bits, same image after every call.
    // A call that does not continue the sequence (other parameters, camera, scene, option, strip layout) drops what is left.
    // Automatic (-1) only batches for a host that runs AHEAD of the device -- its call finds the handle's stream busy, so
    // nobody is waiting for this frame alone.  A host that renders, waits and renders again never has its frame held back
    // behind frames it has not asked for (round 5: VERDICT / ADVICE round 4 -- its longest wait had grown from 1.5 to 8 ms);
    // such a host opts in with an explicit depth (frame_ahead = 2 .. 64).
    int frame_ahead = -1;                              // -1 automatic (ahead_depth()), 0 off, 2 .. RT_MAX_BATCH_FRAMES frames
    bool frame_ahead_failed = false;                   // the batch could not be set up once (memory): automatic stays off
    uint32_t ahead_ramp = 2;                           // automatic: batches of 2, 4, 8, ... frames up to ahead_depth() while the
                                                       // sequence goes on, so that a host that stops after n frames has had
                                                       // at most n rendered in vain; back to 2 when the sequence breaks
    uint64_t generation = 0;                           // bumped by everything that changes what a frame looks like
    struct {
        bool valid = false;
        rt_params base{};                              // the parameters of slot 0
        uint32_t n = 0, next = 0;                      // slots rendered / the next one to hand out
        uint32_t rank = 0, world = 1;
        uint64_t generation = 0;
        uint64_t texels = 0;                           // slot stride
    } ahead;
    struct {
        bool valid = false;
        rt_params params{};
        uint32_t rank = 0, world = 1;
        uint64_t generation = 0;
    } last_single;                                     // the last one-frame call (render_single)
    // rt_snapshot_image / rt_read_snapshot: a copy of the image taken in stream order (device to device), read back on a
    // stream of its own while the handle's stream renders the next frames
    float4* snapshot = nullptr;
    size_t snapshot_capacity = 0, snapshot_bytes = 0;  // bytes allocated / bytes of the last snapshot
    hipStream_t copy_stream = nullptr;
    hipEvent_t snapshot_taken = nullptr, snapshot_read = nullptr;
    bool snapshot_read_pending = false;                // a read of the staging buffer was queued since the last snapshot
    uint32_t pipe_layout[PIPE_MAX][5] = {};            // (w, h, rank, world, texels) each scratch image's padding rows were zeroed for
    hipStream_t pipe_stream[PIPE_MAX] = {};
    hipEvent_t pipe_sampled[PIPE_MAX] = {};            // frame sampled into scratch[i] (recorded on pipe_stream[i])
    hipEvent_t pipe_blended[PIPE_MAX] = {};            // ... and blended out of it (recorded on the handle's stream)
    hipEvent_t pipe_book = nullptr;                    // the last tile-order / primary-table rebuild (recorded on a pipe stream)
    hipEvent_t pipe_main = nullptr;                    // the last launch that was not pipelined (recorded on the handle's stream)
    bool pipe_sampled_set[PIPE_MAX] = {}, pipe_blended_set[PIPE_MAX] = {}, pipe_book_set = false, pipe_main_set = false;
    bool pipe_main_need = true;                        // pipelined frames have to wait for pipe_main before they sample (render_impl)
    float4* pipe_scratch[PIPE_MAX] = {};
    size_t pipe_scratch_texels = 0;
    uint32_t* pipe_work[PIPE_MAX] = {};                // a ring of launch counters per pipe stream
    uint32_t pipe_work_slot[PIPE_MAX] = {};
    uint32_t* pipe_memo[PIPE_MAX] = {};                // further global-memory primary-ray memos (pixel_cache == 2)
    size_t pipe_memo_words[PIPE_MAX] = {};
    uint32_t pipe_seq = 0;
    float4* small_blob = nullptr;
    SceneLayout small_lay{};
    uint32_t small_stack_entries = 1;
    bool small_ok = false;
    int hybrid = 0;  // option "hybrid" (default 0: measured no faster -- config 3 stand-in 6.04 -> 6.11 ms, config 5 geometry 3.49 -> 3.48)
    float4* park_queue[2] = {nullptr, nullptr};
    size_t park_capacity = 0;  // records per queue
    uint32_t* park_counts = nullptr;
    int batch_tile_major = 1;  // option "batch_tile_major": (tile, frame) instead of (frame, tile) order of a batch's work items
    int batch_frames_opt = 32;  // option "batch_frames": frames per launch of rt_render_frames (1..RT_MAX_BATCH_FRAMES)
    // rt_render_multi: what the root's stream has to finish before this handle's image may be overwritten
    // (owned by THIS handle, created on the root's device -- an event is recorded on a stream of its own device --,
    // so that destroying the root first leaves nothing dangling)
    hipEvent_t multi_copied = nullptr;
    int multi_copied_device = -1;
    bool multi_copy_pending = false;
    // batch scratch: the frame layout its padding rows were zeroed for
    uint32_t scratch_w = 0, scratch_h = 0, scratch_rank = 0, scratch_world = 0, scratch_n = 0;
    // root side of rt_render_multi
    std::vector<void*> multi_comms;       // ncclComm_t per rank
    std::vector<int> multi_comm_devices;  // the device list the communicators were made for
    std::set<int> multi_peers_enabled;
    int multi_rccl = 1;  // option "multi_rccl"
    uint32_t top_base = 0, top_available = 0;  // breadth-first numbered top records of the biggest mesh's BVH
    // option "lds_top": 0 off (default: measured slower), -1 what fits beside the stacks at full
    // occupancy, N records
    int lds_top = 0;
    // option "lds_tlas": top-level tree records staged into LDS when the scene is read from global memory: 0 off,
    // 1 when they fit beside the stacks without costing a workgroup per CU, 2 always (if they fit the LDS at all)
    // (default 0: measured slower both ways on the sponza-sized stand-in, DESIGN.md section 5.4 -- the whole tree costs
    // two workgroups per CU, 10.64 -> 13.47 ms; the 42 top records that fit at full occupancy make every tree fetch a
    // two-path load, 10.67 -> 11.98 ms)
    int lds_tlas = 0;
    int fast_miss = 1;  // option "fast_miss"
    int park_levels = 1;  // option "park_levels": the parking launches run the deferred walk's first two levels inline
    uint32_t n_tlas_records = 0;
    float4* own_image = nullptr;  // allocated by rt_create; `image` may be rebound
    float4* multi_gathered = nullptr;  // rt_render_multi root: [world][pad_texels]
    float4* multi_frame = nullptr;     // rt_render_multi root: assembled full frame
    size_t multi_gathered_texels = 0, multi_frame_texels = 0;
    hipEvent_t multi_event = nullptr;
    size_t image_texels = 0;
    unsigned long long paths_total = 0;  // camera paths started since rt_reset_timing
    uint32_t max_width = 0, max_height = 0;
    float4* image = nullptr;
    Counters* counters = nullptr;
    uint32_t* work_counters = nullptr;  // ring of per-launch tile counters
    uint32_t* pixel_cache_mem = nullptr;  // primary-ray cache when LDS has no room (persistent kernel)
    size_t pixel_cache_words = 0;
    uint32_t work_slot = 0;
    int kernel_variant = -1;  // -1 auto, 0 persistent + lane refill, 1 one wave per tile
    // tile-cost feedback: rays per tile of the previous frame order the next frame's tiles
    uint32_t* tile_cost[2] = {nullptr, nullptr};
    uint32_t* tile_order = nullptr;
    uint32_t tile_capacity = 0;
    int cost_slot = 0;
    bool history_valid = false;
    uint32_t hist_w = 0, hist_h = 0, hist_rank = 0, hist_world = 0;
    // primary table (rt_primary_kernel): every pixel's memo -- its constant primary ray and (option "primary_hits") that
    // ray's hit -- valid for (camera, width, height, strip layout, scene, with / without hits); 64 B per pixel
    void* primary = nullptr;
    size_t primary_texels = 0;
    bool primary_valid = false;
    rt_camera_uniform primary_camera{};
    uint32_t primary_w = 0, primary_h = 0, primary_rank = 0, primary_world = 0;
    bool primary_with_hits = false;
    // A pipelined frame whose camera (or shape) is not the shared table's builds a table of its OWN pipeline slot on its own
    // stream -- ordered behind that slot's previous frame and read by nobody else, so no frame in flight has to finish
    // first: the frames of a moving camera overlap like those of a standing one (option "primary_per_slot", default 1).
    struct SlotTable {
        void* table = nullptr;
        size_t texels = 0;
        bool valid = false, with_hits = false;
        rt_camera_uniform camera{};
        uint32_t w = 0, h = 0, rank = 0, world = 0;
    } slot_primary[PIPE_MAX];
    int primary_per_slot = 1;
    // A frame whose camera is not the previous frame's (the camera is MOVING) renders without a table: every pixel computes its
    // own memo (its primary ray is traversed once, by its first sample) -- building a table that the next frame throws away
    // costs more than it saves (config 2: 1.32 ms per frame of a moving camera with a table per frame, 1.24 without).
    rt_camera_uniform last_frame_camera{};
    bool last_frame_camera_valid = false;
    int use_primary = 1;  // option "primary_table"
    int primary_hits = 1; // option "primary_hits": the table also holds the primary rays' hits, so that the frames of an
                          // accumulation (still camera, src/core/app.rs:44-53) traverse no primary ray at all
    int memo_in_table = 1; // option "memo_in_table": kernels whose memo has no room in LDS read it straight from a complete
                          // primary table (RenderArgs::pixel_cache == 4) instead of copying it into a global-memory buffer
    int tile_feedback = 1;
    int tile_feedback_period = 8;  // frames an order is kept before it is refreshed
    bool have_order = false, costs_ready = false;
    uint32_t order_age = 0;
    uint32_t persistent_blocks = 0;
    uint32_t compute_units = 1;  // of the device (rt_create); sizes the walk kernel's grid whatever "persistent_blocks" says
    float* srgb_lut = nullptr;
    // scene
    bool have_scene = false;
    float4* blob = nullptr;  // the scene, see rt_device.h
    SceneLayout lay{};
    bool lds_scene = false;
    bool roots_are_unions = false;  // every internal root's box is the exact union of its children's
    int cull_roots = -1;            // option: -1 auto (many meshes), 0 off, 1 on (if provable)
    // option "cross_prune": cross-mesh pruning in the many-mesh product kernels (RenderArgs::cross_prune).  OFF by default since
    // round 5: its exactness rests on "a triangle hit at t is not found under a box entered far beyond t", which holds for
    // every random ray ever compared (102 G) but NOT for a ray that grazes a far triangle at ~1e-6 rad from an origin within
    // ~1e-4 of that triangle's plane in a generic orientation -- there the shader's own t = dot(ao, n) / det is a quotient of
    // two cancelling sums (tools/prune_directed.py: leaf-box entry up to 1.55 t; tests/test_gpu_prune_directed.py: 1 texel of
    // 2 M directed rays differs).  A host that prefers 45 % less time on many-mesh scenes over that switches it on.
    int cross_prune = 0;
    int force_global = 0;  // option "lds_scene" = 0 disables LDS staging (tuning / tests)
    DTexture* textures = nullptr;
    std::vector<uint8_t*> texture_data;
    uint32_t n_meshes = 0, n_spheres = 0, n_textures = 0, n_nodes = 0, n_triangles = 0;
    uint32_t stack_entries = 1, tlas_entries = 1, n_items = 0;
    bool has_tlas = false;
    bool stack_wide = false, stack_must_wide = false;
    int force_stack_wide = -1;  // option "stack_wide": -1 auto, 0 one-dword entries when legal, 1 two-dword entries
    bool has_forest = false;
    bool plain_materials = false;  // no spheres, no glass, no textured material (rt_upload_scene)
    int specialise = 1;            // option "specialise": 0 = always the general kernels
    int pixel_cache_opt = 1;  // option "pixel_cache"
    int vote_eighths = -1, vote_patience = -1;  // options "vote_eighths", "vote_patience" (-1: by the kind of launch, render_impl)
    int use_tlas = 1;  // option "tlas": 0 = every mesh is a single item (takes effect at the next upload)
    int use_forest = 1;  // option "forest": 0 = no forest items (takes effect at the next upload)
    int use_flat2 = 1;   // option "flat2": 0 = meshes with a two-leaf BVH are not run as straight-line items (next upload)
    int tlas_min = (int)TLAS_MIN_MESHES;  // option "tlas_min": smallest run of meshes that gets a top-level tree
    rt_camera_uniform camera{};
    int count_tests = 0;
    uint32_t last_launch[4] = {0, 0, 0, 0};  // rt_last_launch (entries 4 and 5 are computed when asked)
    bool queues_noted = false;               // the note about GPU_MAX_HW_QUEUES has been left in `err` once
    // option "max_device_mb" (0 = no cap): upper bound on the device memory the library takes on its OWN initiative -- batch
    // and pipeline scratch images, primary tables, global-memory memos, the park queues of deferred walks, the snapshot
    // (not the image, the scene and the textures, which are the host's data).  A feature that does not fit runs the
    // plainer path (smaller batches, no pipeline, no table, no deferred walks); rt_last_launch reports what is held.
    size_t max_device_bytes = 0;
    std::string err;
};

namespace {

// The pipelined single frames (rt_handle::pipeline) keep up to four streams of a handle busy, and ROCm maps a process's
// streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with a fifth stream in the process -- the null stream, a
// framework's copy stream -- two of them share a queue and, if those are two of the pipeline's, their launches serialise
// (measured: 1.30 -> 1.42 ms per frame, 1.34 -> 1.60 at two frames in flight).  The library does not touch the
// environment (the variable is the host's, read when the HIP runtime initialises: INTEGRATION.md section 3; the Python
// package and bench.py set it before they load anything): the automatic depth is four frames in flight when the host
// has asked for five queues or more, three otherwise (1.29 against 1.23 ms per frame on four queues).  A rank of a
// strip split whose share no longer fills the machine (world >= 4: 518 K pixels and fewer for 328 K resident lanes) runs
// seven frames deep when the host has asked for twelve queues or more -- room for the seven streams beside the host's own
// (a framework's compute and copy streams, RCCL's): a frame's latency is then set by its longest pixel chain, not by its
// work (rank 0's share of config 2 at world 8: 0.226 -> 0.207 ms per frame; profiles/r04_strip_pipeline_depth.txt).
// (The variable is read ONCE, when the first handle is created -- the HIP runtime reads it once too, at initialisation: a
// host that changes it later would otherwise be given a depth for queues it does not have)
int hw_queues_requested() {
    static const int queues = [] {
        const char* v = getenv("GPU_MAX_HW_QUEUES");
        const int q = v ? atoi(v) : 4;
        return q > 0 ? q : 4;
    }();
    return queues;
}
int automatic_pipeline_depth(uint32_t world) {
    const int queues = hw_queues_requested();
    if (world >= 4 && queues >= 12) return 7;
    return queues >= 5 ? 4 : 3;
}

thread_local std::string g_err;

int fail(rt_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_err = msg;
    return code;
}

// bytes of device memory the handle holds on its own initiative (see rt_handle::max_device_bytes)
size_t optional_bytes(const rt_handle* h) {
    size_t b = h->batch_scratch_texels * sizeof(float4) + h->pixel_cache_words * sizeof(uint32_t) + h->primary_texels * 64u +
               2u * (((h->park_capacity + 63) / 64) * (size_t)PARK_PLANES * 64u * sizeof(float4)) + h->snapshot_capacity;
    for (int k = 0; k < rt_handle::PIPE_MAX; ++k) {
        if (h->pipe_scratch[k]) b += h->pipe_scratch_texels * sizeof(float4);
        b += h->pipe_memo_words[k] * sizeof(uint32_t) + h->slot_primary[k].texels * 64u;
    }
    return b;
}
// may the handle take `extra` more bytes (after giving back `freed` of what it holds)?
bool fits_cap(const rt_handle* h, size_t extra, size_t freed = 0) {
    if (h->max_device_bytes == 0) return true;
    const size_t held = optional_bytes(h);
    return (held > freed ? held - freed : 0) + extra <= h->max_device_bytes;
}

#define HIP_TRY(h, expr)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(h, RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
void free_dev(T*& p) {
    if (p) {
        (void)hipFree(p);
        p = nullptr;
    }
}

void free_scene(rt_handle* h) {
    free_dev(h->blob);
    free_dev(h->small_blob);
    h->small_ok = false;
    h->have_scene = false;
}

void free_textures(rt_handle* h) {
    for (uint8_t*& p : h->texture_data) free_dev(p);
    h->texture_data.clear();
    free_dev(h->textures);
    h->n_textures = 0;
}

template <typename T>
int upload(rt_handle* h, T*& dst, const T* src, size_t n) {
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIP_TRY(h, hipMalloc((void**)&dst, bytes));
    if (n) HIP_TRY(h, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return RT_OK;
}

// Height (in edges) of the subtree under `root`, with index validation and a
// visit budget that catches cycles.  Iterative: explicit (node, depth) stack.
int mesh_bvh_height(const rt_node* nodes, uint32_t n_nodes, uint32_t node_offset,
                    uint32_t tri_offset, uint32_t n_triangles, uint32_t& height, std::string& why) {
    if (node_offset >= n_nodes) {
        why = "mesh node_offset out of range";
        return RT_ERR_INDEX_RANGE;
    }
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.emplace_back(node_offset, 0u);
    uint64_t visits = 0;
    height = 0;
    while (!st.empty()) {
        auto [idx, depth] = st.back();
        st.pop_back();
        if (++visits > (uint64_t)n_nodes + 1) {
            why = "BVH has a cycle";
            return RT_ERR_INDEX_RANGE;
        }
        const rt_node& nd = nodes[idx];
        if (depth > height) height = depth;
        if (nd.count > 0) {
            if ((uint64_t)tri_offset + nd.first + nd.count > n_triangles) {
                why = "leaf triangle range out of bounds";
                return RT_ERR_INDEX_RANGE;
            }
        } else {
            uint64_t a = (uint64_t)node_offset + nd.left, b = (uint64_t)node_offset + nd.right;
            if (a >= n_nodes || b >= n_nodes) {
                why = "BVH child index out of range";
                return RT_ERR_INDEX_RANGE;
            }
            st.emplace_back((uint32_t)a, depth + 1);
            st.emplace_back((uint32_t)b, depth + 1);
        }
    }
    return RT_OK;
}

}  // namespace

namespace {
RcclApi g_rccl;

// (abort: the communicators' last group failed -- ncclCommAbort where the library has it)
void multi_drop_comms(rt_handle* root, bool abort = false) {
    if (g_rccl.lib)
        for (void* c : root->multi_comms) {
            if (abort) g_rccl.drop((ncclComm_t)c);
            else (void)g_rccl.CommDestroy((ncclComm_t)c);
        }
    root->multi_comms.clear();
    root->multi_comm_devices.clear();
}
}  // namespace

extern "C" {

const char* rt_version(void) { return RT_EXPERIMENTS ? "ray_tracer_2_amd 0.1 (gfx950) +experiments" : "ray_tracer_2_amd 0.1 (gfx950)"; }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void rt_abi_sizes(uint32_t out[8]) {
    out[0] = sizeof(rt_params);
    out[1] = sizeof(rt_material);
    out[2] = sizeof(rt_sphere);
    out[3] = sizeof(rt_mesh_uniform);
    out[4] = sizeof(rt_node);
    out[5] = sizeof(rt_packed_triangle);
    out[6] = sizeof(rt_camera_uniform);
    out[7] = sizeof(rt_scene_uniform);
}

const char* rt_last_error(rt_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int rt_create(int device_ordinal, uint32_t max_width, uint32_t max_height, rt_handle** out) {
    if (!out || max_width == 0 || max_height == 0) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, RT_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    rt_handle* h = new (std::nothrow) rt_handle();
    if (!h) return fail(nullptr, RT_ERR_OUT_OF_MEMORY, "out of host memory");
    h->device = device_ordinal;
    h->max_width = max_width;
    h->max_height = max_height;
    *out = h;
    (void)hw_queues_requested();   // (GPU_MAX_HW_QUEUES as the runtime is about to see it)
    HIP_TRY(h, hipSetDevice(device_ordinal));
    HIP_TRY(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    size_t texels = (size_t)max_width * max_height;
    HIP_TRY(h, hipMalloc((void**)&h->own_image, texels * sizeof(float4)));
    h->image = h->own_image;
    h->image_texels = texels;
    HIP_TRY(h, hipMemsetAsync(h->image, 0, texels * sizeof(float4), h->stream));
    HIP_TRY(h, hipMalloc((void**)&h->counters, sizeof(Counters)));
    HIP_TRY(h, hipMemsetAsync(h->counters, 0, sizeof(Counters), h->stream));
    HIP_TRY(h, hipMalloc((void**)&h->work_counters, 64 * sizeof(uint32_t)));
    HIP_TRY(h, hipMemsetAsync(h->work_counters, 0, 64 * sizeof(uint32_t), h->stream));
    h->tile_capacity = ((max_width + 7) / 8) * ((max_height + 7) / 8 + 1);
    for (int k = 0; k < 2; ++k) HIP_TRY(h, hipMalloc((void**)&h->tile_cost[k], (size_t)h->tile_capacity * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc((void**)&h->tile_order, (size_t)h->tile_capacity * sizeof(uint32_t)));
    {
        hipDeviceProp_t prop;
        HIP_TRY(h, hipGetDeviceProperties(&prop, device_ordinal));
        // 4 waves per SIMD (the render kernels' register budget) = 4 workgroups of 4 waves per CU
        h->persistent_blocks = (uint32_t)prop.multiProcessorCount * BLOCKS_PER_CU;
        h->compute_units = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 1u;
    }
    static const float lut[256] = {RT_SRGB_LUT_VALUES};
    HIP_TRY(h, hipMalloc((void**)&h->srgb_lut, sizeof(lut)));
    HIP_TRY(h, hipMemcpyAsync(h->srgb_lut, lut, sizeof(lut), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RT_OK;
}

void rt_destroy(rt_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->multi_copy_pending && h->multi_copied) (void)hipEventSynchronize(h->multi_copied);  // (the root still reads h->image)
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_scene(h);
    free_textures(h);
    free_dev(h->own_image);
    free_dev(h->multi_gathered);
    free_dev(h->multi_frame);
    free_dev(h->batch_scratch);
    free_dev(h->park_queue[0]);
    free_dev(h->park_queue[1]);
    free_dev(h->park_counts);
    for (int k = 0; k < rt_handle::PIPE_MAX; ++k) {
        free_dev(h->pipe_memo[k]);
        if (h->pipe_stream[k]) {
            (void)hipStreamSynchronize(h->pipe_stream[k]);
            (void)hipStreamDestroy(h->pipe_stream[k]);
        }
        if (h->pipe_sampled[k]) (void)hipEventDestroy(h->pipe_sampled[k]);
        if (h->pipe_blended[k]) (void)hipEventDestroy(h->pipe_blended[k]);
        free_dev(h->pipe_scratch[k]);
        free_dev(h->pipe_work[k]);
    }
    if (h->pipe_book) (void)hipEventDestroy(h->pipe_book);
    if (h->pipe_main) (void)hipEventDestroy(h->pipe_main);
    if (h->copy_stream) {
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamDestroy(h->copy_stream);
    }
    if (h->snapshot_taken) (void)hipEventDestroy(h->snapshot_taken);
    if (h->snapshot_read) (void)hipEventDestroy(h->snapshot_read);
    free_dev(h->snapshot);
    free_dev(h->wf_state);
    free_dev(h->wf_hit);
    free_dev(h->wf_lists);
    free_dev(h->wf_counts);
    if (h->multi_event) (void)hipEventDestroy(h->multi_event);
    if (h->multi_copied) (void)hipEventDestroy(h->multi_copied);
    if (g_rccl.lib)
        for (void* c : h->multi_comms) (void)g_rccl.CommDestroy((ncclComm_t)c);
    free_dev(h->counters);
    free_dev(h->work_counters);
    free_dev(h->primary);
    for (auto& st : h->slot_primary) free_dev(st.table);
    free_dev(h->pixel_cache_mem);
#if RT_WALK2
    free_dev(h->walk2);
#endif
    free_dev(h->tile_cost[0]);
    free_dev(h->tile_cost[1]);
    free_dev(h->tile_order);
    free_dev(h->srgb_lut);
    for (auto& e : h->ev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int rt_upload_scene(rt_handle* h, const rt_scene_uniform* scene, const rt_sphere* spheres,
                    uint32_t n_spheres, const rt_mesh_uniform* meshes, uint32_t n_meshes,
                    const rt_packed_triangle* triangles, uint32_t n_triangles, const rt_node* nodes,
                    uint32_t n_nodes) {
    if (!h || !scene) return fail(h, RT_ERR_INVALID_ARGUMENT, "null handle or scene");
    h->generation += 1;
    if ((n_spheres && !spheres) || (n_meshes && !meshes) || (n_triangles && !triangles) || (n_nodes && !nodes))
        return fail(h, RT_ERR_INVALID_ARGUMENT, "null array with non-zero count");
    // ray_tracer.rs:15-19, bvh.rs:140
    if (n_meshes > RT_MAX_MESHES) return fail(h, RT_ERR_CAPACITY, "more than 400 meshes");
    if (n_spheres > RT_MAX_SPHERES) return fail(h, RT_ERR_CAPACITY, "more than 500 spheres");
    if (n_triangles > RT_MAX_TRIANGLES) return fail(h, RT_ERR_CAPACITY, "more than 1375000 triangles");
    if (n_nodes > RT_MAX_NODES) return fail(h, RT_ERR_CAPACITY, "more than 2600000 BVH nodes");
    if (scene->spheres != n_spheres || scene->meshes != n_meshes)
        return fail(h, RT_ERR_INVALID_ARGUMENT, "SceneUniform counts disagree with the array lengths");
    HIP_TRY(h, hipSetDevice(h->device));

    try {
        // ---- validation + wide BVH records ---------------------------------
        struct WideRec { float4 q[4]; };
        std::vector<WideRec> wide;
        std::vector<uint32_t> wide_base(n_meshes), root_idx(n_meshes), root_count(n_meshes);
        std::vector<char> deep(n_meshes, 0);
        std::vector<uint32_t> tri_lo(n_meshes, 0xffffffffu), tri_hi(n_meshes, 0u), mesh_need(n_meshes, 0u);  // triangle range, stack entries
        std::vector<uint32_t> wide_index(n_nodes, 0xffffffffu);  // per original node
        uint32_t max_height = 0, max_leaf_ref = 0;  // (largest triangle count of a leaf that can go on a stack)
        uint32_t top_mesh_records = 0, top_mesh_base = 0;
        for (uint32_t i = 0; i < n_meshes; ++i) {
            const rt_mesh_uniform& m = meshes[i];
            uint32_t height = 0;
            std::string why;
            int rc = mesh_bvh_height(nodes, n_nodes, m.node_offset, m.triangle_offset, n_triangles, height, why);
            if (rc != RT_OK) return fail(h, rc, "mesh " + std::to_string(i) + ": " + why);
            // The shader's stack holds 32 entries (wgsl:297); with the near child kept in
            // registers this kernel needs `height` entries and the shader height + 1.  A
            // tree of height >= 32 can overflow the shader's stack; such a mesh is traversed
            // with the shader's literal push/pop and clamped indices (DMESH_DEEP), which
            // needs the full 32 entries.
            deep[i] = height + 1 > RT_BVH_STACK;
            const uint32_t need = deep[i] ? RT_BVH_STACK : height;
            if (need > max_height) max_height = need;
            mesh_need[i] = need;
            // Wide records: internal nodes in DFS pre-order, indexed per mesh.
            // (Meshes may alias node ranges; records are built per mesh.)
            wide_base[i] = (uint32_t)wide.size();
            const rt_node* mn = nodes + m.node_offset;
            // (child and root indices are absolute: triangle index into the scene's triangle
            // array, wide-record index into the scene's record array)
            if (mn[0].count > 0) {
                root_idx[i] = m.triangle_offset + mn[0].first;
                root_count[i] = mn[0].count;
                tri_lo[i] = root_idx[i];
                tri_hi[i] = root_idx[i] + root_count[i];
                continue;
            }
            root_idx[i] = wide_base[i];
            root_count[i] = 0;
            // Record order: the first TOP_BFS internal nodes breadth-first from the root (any prefix of
            // them is a "top of the tree": what the render kernels stage into LDS for a big mesh), the
            // rest in depth-first pre-order below them.
            std::vector<uint32_t> order;  // original mesh-local indices of internal nodes
            std::vector<uint32_t> frontier{0u}, st;
            constexpr size_t TOP_BFS = 2048;
            for (size_t q = 0; q < frontier.size(); ++q) {
                const uint32_t n = frontier[q];
                if (order.size() >= TOP_BFS) { st.push_back(n); continue; }
                wide_index[m.node_offset + n] = (uint32_t)order.size();
                order.push_back(n);
                if (mn[mn[n].left].count == 0) frontier.push_back(mn[n].left);
                if (mn[mn[n].right].count == 0) frontier.push_back(mn[n].right);
            }
            std::reverse(st.begin(), st.end());  // (pop order = breadth-first order of the cut)
            while (!st.empty()) {
                uint32_t n = st.back();
                st.pop_back();
                wide_index[m.node_offset + n] = (uint32_t)order.size();
                order.push_back(n);
                if (mn[mn[n].right].count == 0) st.push_back(mn[n].right);
                if (mn[mn[n].left].count == 0) st.push_back(mn[n].left);
            }
            if (order.size() > top_mesh_records) {  // the biggest BVH gets the LDS-staged top
                top_mesh_records = (uint32_t)order.size();
                top_mesh_base = wide_base[i];
            }
            for (uint32_t n : order) {
                const rt_node &ca = mn[mn[n].left], &cb = mn[mn[n].right];
                auto kind = [&](const rt_node& c, uint32_t local, uint32_t& idx, uint32_t& cnt) {
                    if (c.count > 0) {
                        idx = m.triangle_offset + c.first;
                        cnt = c.count;
                        if (cnt > max_leaf_ref) max_leaf_ref = cnt;
                        tri_lo[i] = std::min(tri_lo[i], idx);
                        tri_hi[i] = std::max(tri_hi[i], idx + cnt);
                    } else {
                        idx = wide_base[i] + wide_index[m.node_offset + local];
                        cnt = 0;
                    }
                };
                uint32_t ai, ac, bi, bc;
                kind(ca, mn[n].left, ai, ac);
                kind(cb, mn[n].right, bi, bc);
                auto asf = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
                WideRec w;
                w.q[0] = make_float4(ca.aabb_min[0], ca.aabb_max[0], ca.aabb_min[1], ca.aabb_max[1]);
                w.q[1] = make_float4(ca.aabb_min[2], ca.aabb_max[2], asf(ai), asf(ac));
                w.q[2] = make_float4(cb.aabb_min[0], cb.aabb_max[0], cb.aabb_min[1], cb.aabb_max[1]);
                w.q[3] = make_float4(cb.aabb_min[2], cb.aabb_max[2], asf(bi), asf(bc));
                wide.push_back(w);
            }
        }
        // ---- mesh-loop items and top-level trees -------------------------------------
        // Runs of consecutive meshes with bit-identical world_to_model share a local space.
        // Within a run, meshes with an internal, non-deep root whose box provably contains its
        // children's go under a TLAS when there are enough of them; every other mesh is a
        // single item.  (Visit order is free: rt_kernel.hip breaks distance ties by mesh index.)
        struct Item { uint32_t kind, a, b, n; };
        std::vector<Item> items;
        std::vector<WideRec> tlas;
        struct ForestEntry { float4 q[3]; };
        std::vector<ForestEntry> forest_entries;
        uint32_t tlas_depth = 0;
        // (root_box_contains: the root box provably contains its children's boxes; root_box_ok: ... and the mesh is
        // walked with the ordinary stack)
        auto root_box_contains = [&](uint32_t i) {
            const rt_node* mn = nodes + meshes[i].node_offset;
            if (mn[0].count > 0) return false;
            const rt_node &ca = mn[mn[0].left], &cb = mn[mn[0].right];
            for (int k = 0; k < 3; ++k) {
                const float lo_k = ca.aabb_min[k] < cb.aabb_min[k] ? ca.aabb_min[k] : cb.aabb_min[k];
                const float hi_k = ca.aabb_max[k] > cb.aabb_max[k] ? ca.aabb_max[k] : cb.aabb_max[k];
                if (!(mn[0].aabb_min[k] <= lo_k && mn[0].aabb_max[k] >= hi_k)) return false;
                if (!(ca.aabb_min[k] <= ca.aabb_max[k] && cb.aabb_min[k] <= cb.aabb_max[k])) return false;
                if (!(mn[0].aabb_min[k] - mn[0].aabb_min[k] == 0.0f && mn[0].aabb_max[k] - mn[0].aabb_max[k] == 0.0f)) return false;  // finite
            }
            return true;
        };
        auto root_box_ok = [&](uint32_t i) { return !deep[i] && root_box_contains(i); };
        // (a tree's reference to a mesh has 9 bits for the mesh and 21 for its root record, rt_device.h)
        auto tree_ok = [&](uint32_t i) { return root_box_ok(i) && i <= TLAS_REF_MESH_MASK && root_idx[i] <= TLAS_REF_ROOT_MASK; };
        struct Box { float lo[3], hi[3]; };
        auto asf2 = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
        // recursive split over the root boxes; returns the child reference (idx, count)
        uint32_t tlas_max_depth = 0;  // set per tree: depth of the balanced tree + 6
        auto ceil_log2 = [](size_t n) { uint32_t d = 0; while (((size_t)1 << d) < n) ++d; return d; };
        std::function<void(std::vector<uint32_t>&, size_t, size_t, uint32_t, uint32_t&, uint32_t&, Box&)> build_tlas =
            [&](std::vector<uint32_t>& ms, size_t b0, size_t e0, uint32_t depth, uint32_t& idx, uint32_t& cnt, Box& box) {
                if (depth > tlas_depth) tlas_depth = depth;
                if (e0 - b0 == 1) {
                    const rt_node& r = nodes[meshes[ms[b0]].node_offset];
                    for (int k = 0; k < 3; ++k) { box.lo[k] = r.aabb_min[k]; box.hi[k] = r.aabb_max[k]; }
                    idx = root_idx[ms[b0]] | (ms[b0] << TLAS_REF_MESH_SHIFT) |
                          (meshes[ms[b0]].material.flag == RT_MATERIAL_GLASS ? TLAS_REF_GLASS : 0u);
                    cnt = 1;
                    return;
                }
                // Split: surface-area heuristic over the root boxes, swept along each axis in centroid order (the
                // boxes are few -- one per mesh -- so the full sweep is affordable; a median split put the scene-wide
                // floor and ceiling meshes of the many-mesh stand-in into the same subtrees as the columns next to
                // their centroids).  Any split is a correct one: the tree only has to contain its root boxes.
                auto centroid_less = [&](int axis) {
                    return [&, axis](uint32_t x, uint32_t y) {
                        const rt_node &rx = nodes[meshes[x].node_offset], &ry = nodes[meshes[y].node_offset];
                        const float cx = rx.aabb_min[axis] + rx.aabb_max[axis], cy = ry.aabb_min[axis] + ry.aabb_max[axis];
                        return cx < cy || (cx == cy && x < y);
                    };
                };
                auto half_area = [](const double* lo3, const double* hi3) {
                    const double dx = hi3[0] - lo3[0], dy = hi3[1] - lo3[1], dz = hi3[2] - lo3[2];
                    return dx * dy + dy * dz + dz * dx;
                };
                const size_t cnt_here = e0 - b0;
                int best_axis = 0;
                size_t best_left = cnt_here / 2;
                double best_cost = DBL_MAX;
                std::vector<double> right_area(cnt_here);
                for (int axis = 0; axis < 3; ++axis) {
                    std::sort(ms.begin() + b0, ms.begin() + e0, centroid_less(axis));
                    double lo3[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi3[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
                    for (size_t q = cnt_here; q-- > 1;) {  // right_area[q]: boxes q .. end
                        const rt_node& r = nodes[meshes[ms[b0 + q]].node_offset];
                        for (int k = 0; k < 3; ++k) {
                            if (r.aabb_min[k] < lo3[k]) lo3[k] = r.aabb_min[k];
                            if (r.aabb_max[k] > hi3[k]) hi3[k] = r.aabb_max[k];
                        }
                        right_area[q] = half_area(lo3, hi3);
                    }
                    for (int k = 0; k < 3; ++k) { lo3[k] = DBL_MAX; hi3[k] = -DBL_MAX; }
                    for (size_t q = 1; q < cnt_here; ++q) {  // left = boxes 0 .. q-1
                        const rt_node& r = nodes[meshes[ms[b0 + q - 1]].node_offset];
                        for (int k = 0; k < 3; ++k) {
                            if (r.aabb_min[k] < lo3[k]) lo3[k] = r.aabb_min[k];
                            if (r.aabb_max[k] > hi3[k]) hi3[k] = r.aabb_max[k];
                        }
                        const double cost = half_area(lo3, hi3) * (double)q + right_area[q] * (double)(cnt_here - q);
                        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_left = q; }
                    }
                }
                // (every lane keeps a tree stack of depth + 2 entries in LDS: a subtree that would not fit below the
                // depth limit any other way is split in the middle)
                if (depth + ceil_log2(cnt_here) >= tlas_max_depth) best_left = cnt_here / 2;
                std::sort(ms.begin() + b0, ms.begin() + e0, centroid_less(best_axis));
                const size_t mid = b0 + best_left;
                const uint32_t me = (uint32_t)tlas.size();
                tlas.emplace_back();
                uint32_t ai, ac, bi, bc;
                Box ba, bb;
                build_tlas(ms, b0, mid, depth + 1, ai, ac, ba);
                build_tlas(ms, mid, e0, depth + 1, bi, bc, bb);
                WideRec w;
                w.q[0] = make_float4(ba.lo[0], ba.hi[0], ba.lo[1], ba.hi[1]);
                w.q[1] = make_float4(ba.lo[2], ba.hi[2], asf2(ai), asf2(ac));
                w.q[2] = make_float4(bb.lo[0], bb.hi[0], bb.lo[1], bb.hi[1]);
                w.q[3] = make_float4(bb.lo[2], bb.hi[2], asf2(bi), asf2(bc));
                tlas[me] = w;
                for (int k = 0; k < 3; ++k) {  // exact union (min/max are exact)
                    box.lo[k] = ba.lo[k] < bb.lo[k] ? ba.lo[k] : bb.lo[k];
                    box.hi[k] = ba.hi[k] > bb.hi[k] ? ba.hi[k] : bb.hi[k];
                }
                idx = me;
                cnt = 0;
            };
        // Forest items are walked by the few-mesh kernels only: none when the scene gets a top-level
        // tree anywhere or has enough meshes for automatic root-box culling (many-mesh kernels).
        bool any_tlas = false;
        if (h->use_tlas)
            for (uint32_t i0 = 0; i0 < n_meshes;) {
                uint32_t i1 = i0 + 1, ok = 0;
                while (i1 < n_meshes && memcmp(meshes[i1].world_to_model, meshes[i0].world_to_model, 64) == 0) ++i1;
                for (uint32_t i = i0; i < i1; ++i) ok += tree_ok(i) ? 1u : 0u;
                if (ok >= (uint32_t)h->tlas_min) any_tlas = true;
                i0 = i1;
            }
        const bool allow_forest = h->use_forest && !any_tlas && n_meshes < 16;
        // meshes whose root has two leaf children run as straight-line code in the few-mesh kernels (ITEM_FLAT2)
        auto is_flat2 = [&](uint32_t i) {
            if (!h->use_flat2 || any_tlas || n_meshes >= 16 || root_count[i] != 0 || deep[i]) return false;
            const rt_node* mn = nodes + meshes[i].node_offset;
            return mn[mn[0].left].count > 0 && mn[mn[0].right].count > 0;
        };
        for (uint32_t i0 = 0; i0 < n_meshes;) {
            uint32_t i1 = i0 + 1;
            while (i1 < n_meshes && memcmp(meshes[i1].world_to_model, meshes[i0].world_to_model, 64) == 0) ++i1;
            std::vector<uint32_t> grouped;
            if (h->use_tlas)
                for (uint32_t i = i0; i < i1; ++i)
                    if (tree_ok(i)) grouped.push_back(i);
            if (grouped.size() < (size_t)h->tlas_min) grouped.clear();
            // the other meshes of the run with an internal, non-deep root (and the run's
            // model_to_world as well) form a forest when there are at least two of them
            std::vector<uint32_t> forest;
            if (allow_forest) {
                size_t g = 0;
                for (uint32_t i = i0; i < i1; ++i) {
                    if (g < grouped.size() && grouped[g] == i) { ++g; continue; }
                    if (root_count[i] == 0 && !deep[i] && !is_flat2(i) && memcmp(meshes[i].model_to_world, meshes[i0].model_to_world, 64) == 0)
                        forest.push_back(i);
                }
                if (forest.size() < 2) forest.clear();
            }
            bool first = true;
            auto flag = [&]() { uint32_t f = first ? ITEM_NEW_XFORM : 0u; first = false; return f; };
            size_t g = 0, fo = 0;
            for (uint32_t i = i0; i < i1; ++i) {
                if (g < grouped.size() && grouped[g] == i) { ++g; continue; }
                if (fo < forest.size() && forest[fo] == i) { ++fo; continue; }
                items.push_back(Item{flag() | (is_flat2(i) ? ITEM_FLAT2 : 0u), i, i0, 1});
            }
            for (size_t f0 = 0; f0 < forest.size(); f0 += FOREST_MAX_MEMBERS) {
                const size_t f1 = std::min(forest.size(), f0 + (size_t)FOREST_MAX_MEMBERS);
                items.push_back(Item{ITEM_FOREST | flag(), (uint32_t)forest_entries.size(), i0, (uint32_t)(f1 - f0)});
                for (size_t f = f0; f < f1; ++f) {
                    const uint32_t i = forest[f];
                    const rt_node& r = nodes[meshes[i].node_offset];
                    ForestEntry e;
                    uint32_t fl = (meshes[i].material.flag == RT_MATERIAL_GLASS ? DMESH_GLASS : 0u) |
                                  (root_box_ok(i) ? FOREST_CULLABLE : 0u);
                    e.q[0] = make_float4(asf2(root_idx[i]), asf2(i), asf2(fl), 0.0f);
                    e.q[1] = make_float4(r.aabb_min[0], r.aabb_max[0], r.aabb_min[1], r.aabb_max[1]);
                    e.q[2] = make_float4(r.aabb_min[2], r.aabb_max[2], 0.0f, 0.0f);
                    forest_entries.push_back(e);
                }
            }
            if (!grouped.empty()) {
                std::vector<uint32_t> ms = grouped;
                uint32_t ridx, rcnt;
                Box rb;
                tlas_max_depth = 1 + ceil_log2(ms.size()) + 6;
                build_tlas(ms, 0, ms.size(), 1, ridx, rcnt, rb);
                items.push_back(Item{ITEM_TLAS | flag(), ridx, i0, (uint32_t)grouped.size()});
            }
            i0 = i1;
        }
        // Number the tree records breadth-first from the roots (all trees together): any prefix of the array is then
        // "the top levels", which is what option "lds_tlas" stages into LDS when the whole tree does not fit.
        if (!tlas.empty()) {
            std::vector<uint32_t> order, new_of(tlas.size(), 0xffffffffu);
            auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
            for (const Item& it : items)
                if (it.kind & ITEM_TLAS) order.push_back(it.a);
            for (size_t q = 0; q < order.size(); ++q) {
                const WideRec& w = tlas[order[q]];
                if (bits(w.q[1].w) == 0u) order.push_back(bits(w.q[1].z));  // child a is a tree node
                if (bits(w.q[3].w) == 0u) order.push_back(bits(w.q[3].z));
            }
            if (order.size() == tlas.size()) {
                for (size_t q = 0; q < order.size(); ++q) new_of[order[q]] = (uint32_t)q;
                std::vector<WideRec> re(tlas.size());
                for (size_t q = 0; q < order.size(); ++q) {
                    WideRec w = tlas[order[q]];
                    if (bits(w.q[1].w) == 0u) w.q[1].z = asf2(new_of[bits(w.q[1].z)]);
                    if (bits(w.q[3].w) == 0u) w.q[3].z = asf2(new_of[bits(w.q[3].z)]);
                    re[q] = w;
                }
                tlas.swap(re);
                for (Item& it : items)
                    if (it.kind & ITEM_TLAS) it.a = new_of[it.a];
            }
        }
        // ---- cross-mesh pruning (RenderArgs::cross_prune): which items may be cut, and the order of the loop ----
        // An item gets ITEM_PRUNE when every mesh of it (a) has the model_to_world of the mesh that gives the item's
        // local ray, bit for bit -- the kernel's bound on the world distance is derived from that matrix --, (b) is
        // not glass (no backface culling: a ray leaving a surface is not culled against the coplanar triangles next
        // to it, the one place where the triangle test's parameter is noise), (c) is walked with
        // the ordinary stack, and (d) has a BVH that is a proper bounding hierarchy: finite boxes, every child box inside
        // its parent's, every leaf triangle inside its leaf's box (true of the reference's builder; verified, since
        // BVHs may be foreign).
        auto hierarchy_ok = [&](uint32_t i) {
            const rt_mesh_uniform& m = meshes[i];
            const rt_node* mn = nodes + m.node_offset;
            auto finite_box = [](const rt_node& n) {
                for (int k = 0; k < 3; ++k)
                    if (!(n.aabb_min[k] <= n.aabb_max[k] && n.aabb_min[k] - n.aabb_min[k] == 0.0f && n.aabb_max[k] - n.aabb_max[k] == 0.0f)) return false;
                return true;
            };
            std::vector<uint32_t> st{0u};
            while (!st.empty()) {
                const rt_node& n = mn[st.back()];
                st.pop_back();
                if (!finite_box(n)) return false;
                if (n.count > 0) {
                    for (uint32_t t = 0; t < n.count; ++t) {
                        const rt_packed_triangle& p = triangles[m.triangle_offset + n.first + t];
                        for (const float* v : {p.v1, p.v2, p.v3})
                            for (int k = 0; k < 3; ++k)
                                if (!(v[k] >= n.aabb_min[k] && v[k] <= n.aabb_max[k])) return false;
                    }
                } else {
                    for (uint32_t c : {n.left, n.right}) {
                        const rt_node& ch = mn[c];
                        for (int k = 0; k < 3; ++k)
                            if (!(ch.aabb_min[k] >= n.aabb_min[k] && ch.aabb_max[k] <= n.aabb_max[k])) return false;
                        st.push_back(c);
                    }
                }
            }
            return true;
        };
        std::vector<signed char> prune_ok_memo(n_meshes, -1);
        auto mesh_prune_ok = [&](uint32_t i, uint32_t xform_mesh) {
            if (prune_ok_memo[i] < 0)
                prune_ok_memo[i] = (!deep[i] && meshes[i].material.flag != RT_MATERIAL_GLASS && hierarchy_ok(i)) ? 1 : 0;
            return prune_ok_memo[i] == 1 && memcmp(meshes[i].model_to_world, meshes[xform_mesh].model_to_world, 64) == 0;
        };
        if (any_tlas || n_meshes >= 16) {  // (the scenes the many-mesh kernels render)
            // members of a tree, per tree item (the trees are not renumbered again below)
            auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
            for (Item& it : items) {
                bool ok = true;
                if (it.kind & ITEM_TLAS) {
                    std::vector<uint32_t> st{it.a};
                    while (!st.empty() && ok) {
                        const WideRec w = tlas[st.back()];
                        st.pop_back();
                        for (int c = 0; c < 2 && ok; ++c) {
                            const uint32_t idx = bits(w.q[2 * c + 1].z), cnt = bits(w.q[2 * c + 1].w);
                            if (cnt == 0u) st.push_back(idx);
                            else ok = mesh_prune_ok((idx >> TLAS_REF_MESH_SHIFT) & TLAS_REF_MESH_MASK, it.b);
                        }
                    }
                } else if (it.kind & ITEM_FOREST) {
                    ok = false;  // (few-mesh kernels only)
                } else {
                    ok = root_count[it.a] == 0u && mesh_prune_ok(it.a, it.b);
                }
                if (ok) it.kind |= ITEM_PRUNE;
            }
            // The loop's order is free (ties between equal world distances go to the lower mesh index, rt_kernel.hip):
            // first the meshes whose root is a leaf (the whole wave tests their triangles in step), then the other
            // single meshes, then the trees, so that the long walks start with a closest hit to prune against.  Inside
            // a class the order stays; an item opens its local space when its class's previous item had another one.
            std::vector<Item> ordered;
            for (int cls = 0; cls < 3; ++cls) {
                bool first = true;
                uint32_t prev_b = 0;
                for (const Item& it0 : items) {
                    const int c = (it0.kind & ITEM_TLAS) ? 2 : ((it0.kind & ITEM_FOREST) || root_count[it0.a] == 0u) ? 1 : 0;
                    if (c != cls) continue;
                    Item it = it0;
                    it.kind &= ~(uint32_t)ITEM_NEW_XFORM;
                    if (first || it.b != prev_b) it.kind |= ITEM_NEW_XFORM;
                    first = false;
                    prev_b = it.b;
                    ordered.push_back(it);
                }
            }
            // (classes follow each other: the first item of a class whose local space is the previous class's last one
            // need not open it again)
            for (size_t k = 1; k < ordered.size(); ++k)
                if (ordered[k].b == ordered[k - 1].b) ordered[k].kind &= ~(uint32_t)ITEM_NEW_XFORM;
            items.swap(ordered);
        }
        // (one entry is always there: the many-mesh kernels, which the debug views use too,
        // run single meshes through the same stack)
        const uint32_t tlas_entries = tlas.empty() ? 1u : tlas_depth + 2u;
        const bool has_tlas = !tlas.empty();

        // ---- the deferred mesh (RenderArgs::park) ----------------------------------------------
        // The biggest single-mesh item of a few-mesh scene with a real BVH:
        // its item goes to the end of the mesh loop (the loop's order is free), where a launch can stop in front
        // of it.
        bool have_defer = false;
        uint32_t defer_mesh = 0, defer_xform = 0, defer_internal = 0;
        if (!any_tlas && n_meshes < 16) {
            size_t best_k = items.size();
            uint32_t best_big = 0;
            for (size_t k = 0; k < items.size(); ++k) {
                const Item& it = items[k];
                if (it.kind & (ITEM_TLAS | ITEM_FOREST | ITEM_FLAT2)) continue;
                const uint32_t mi = it.a;
                if (root_count[mi] != 0) continue;
                const uint32_t internal = (mi + 1 < n_meshes ? wide_base[mi + 1] : (uint32_t)wide.size()) - wide_base[mi];
                if (internal >= (uint32_t)h->defer_min_nodes && internal > best_big) { best_big = internal; best_k = k; }
            }
            if (best_k < items.size()) {
                Item d = items[best_k];
                items.erase(items.begin() + (std::ptrdiff_t)best_k);
                // (the item that followed it in the same local space now opens that space)
                if ((d.kind & ITEM_NEW_XFORM) && best_k < items.size() && !(items[best_k].kind & ITEM_NEW_XFORM)) items[best_k].kind |= ITEM_NEW_XFORM;
                d.kind |= ITEM_NEW_XFORM | ITEM_DEFER | (root_box_contains(d.a) ? ITEM_DEFER_CULL : 0u);
                items.push_back(d);
                have_defer = true;
                defer_mesh = d.a;
                defer_xform = d.b;
                defer_internal = best_big;
            }
        }

#if RT_WALK2
        free_dev(h->walk2);
        h->walk2 = nullptr;
        if (have_defer) {
            const uint32_t base = wide_base[defer_mesh];
            std::vector<float4> w2((size_t)defer_internal * 12u, make_float4(0, 0, 0, 0));
            auto bitsof = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
            for (uint32_t k = 0; k < defer_internal; ++k) {
                const WideRec& w = wide[base + k];
                for (int q = 0; q < 4; ++q) w2[(size_t)k * 12 + q] = w.q[q];
                if (bitsof(w.q[1].w) == 0u) for (int q = 0; q < 4; ++q) w2[(size_t)k * 12 + 4 + q] = wide[bitsof(w.q[1].z)].q[q];
                if (bitsof(w.q[3].w) == 0u) for (int q = 0; q < 4; ++q) w2[(size_t)k * 12 + 8 + q] = wide[bitsof(w.q[3].z)].q[q];
            }
            HIP_TRY(h, hipMalloc((void**)&h->walk2, w2.size() * sizeof(float4)));
            HIP_TRY(h, hipMemcpy(h->walk2, w2.data(), w2.size() * sizeof(float4), hipMemcpyHostToDevice));
            h->walk2_base = base;
        }
#endif
        // ---- blob layout ------------------------------------------------------
        SceneLayout lay{};
        uint64_t off = 0;
        // (the per-scene sections first, the per-node / per-triangle arrays last: the small blob of the hybrid launches has
        // the same sections with shorter arrays, so every offset up to wide_off is the same in both -- the primary-ray memo
        // keeps a material's byte offset across launches that read different blobs)
        lay.mesh_off = (uint32_t)off;   off += (uint64_t)n_meshes * MESH_REC_BYTES;
        lay.mat_off = (uint32_t)off;    off += (uint64_t)(n_meshes + n_spheres) * MATERIAL_BYTES;
        lay.sphere_off = (uint32_t)off; off += (uint64_t)n_spheres * SPHERE_BYTES;
        lay.item_off = (uint32_t)off;   off += (uint64_t)items.size() * ITEM_BYTES;
        lay.tlas_off = (uint32_t)off;   off += (uint64_t)tlas.size() * WIDE_REC_BYTES;
        lay.forest_off = (uint32_t)off; off += (uint64_t)forest_entries.size() * FOREST_ENTRY_BYTES;
        lay.wide_off = (uint32_t)off;   off += (uint64_t)wide.size() * WIDE_REC_BYTES;
        lay.tri_off = (uint32_t)off;    off += (uint64_t)n_triangles * TRI_ISECT_BYTES;
        lay.shade_off = (uint32_t)off;  off += (uint64_t)n_triangles * TRI_SHADE_BYTES;
        if (off == 0) off = 16;
        if (off > 0xfffffff0ull) return fail(h, RT_ERR_CAPACITY, "scene larger than 4 GiB");
        lay.bytes = (uint32_t)off;
        std::vector<float4> blob(off / 16, make_float4(0, 0, 0, 0));
        auto asf = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
        for (uint32_t i = 0; i < n_meshes; ++i) {
            const rt_mesh_uniform& m = meshes[i];
            float4* r = blob.data() + (lay.mesh_off + (size_t)i * MESH_REC_BYTES) / 16;
            memcpy(r, m.world_to_model, 64);
            memcpy(r + 4, m.model_to_world, 64);
            uint32_t flags = 0;
            if (m.material.flag == RT_MATERIAL_GLASS) flags |= DMESH_GLASS;
            if (deep[i]) flags |= DMESH_DEEP;
            r[8] = make_float4(asf(flags), asf(root_idx[i]), asf(root_count[i]), asf(m.triangle_offset));
            {
                // S >= the largest absolute row sum of model_to_world's 3 x 3 part ([col][row]), C >= the largest
                // absolute translation component: in double, then rounded up (cross-mesh pruning's error terms)
                double S = 0.0, C = 0.0;
                for (int row = 0; row < 3; ++row) {
                    const double rs = std::fabs((double)m.model_to_world[0][row]) + std::fabs((double)m.model_to_world[1][row]) +
                                      std::fabs((double)m.model_to_world[2][row]);
                    if (!(rs <= S)) S = rs;  // (NaN sticks)
                    const double tc = std::fabs((double)m.model_to_world[3][row]);
                    if (!(tc <= C)) C = tc;
                }
                auto up = [](double d) { float f = (float)d; if ((double)f < d) f = std::nextafter(f, INFINITY); return f; };
                r[9] = make_float4(asf(wide_base[i]), up(S), up(C), 0.0f);
            }
            const rt_node& root = nodes[m.node_offset];
            r[10] = make_float4(root.aabb_min[0], root.aabb_max[0], root.aabb_min[1], root.aabb_max[1]);
            r[11] = make_float4(root.aabb_min[2], root.aabb_max[2], 0.0f, 0.0f);
            memcpy(blob.data() + (lay.mat_off + (size_t)i * MATERIAL_BYTES) / 16, &m.material, MATERIAL_BYTES);
        }
        if (!wide.empty()) memcpy(blob.data() + lay.wide_off / 16, wide.data(), wide.size() * sizeof(WideRec));
        if (!tlas.empty()) memcpy(blob.data() + lay.tlas_off / 16, tlas.data(), tlas.size() * sizeof(WideRec));
        if (!forest_entries.empty())
            memcpy(blob.data() + lay.forest_off / 16, forest_entries.data(), forest_entries.size() * sizeof(ForestEntry));
        for (size_t k = 0; k < items.size(); ++k) {
            const Item& it = items[k];
            const bool single = (it.kind & (ITEM_TLAS | ITEM_FOREST)) == 0;
            blob[lay.item_off / 16 + 2 * k] = make_float4(asf(it.kind), asf(it.a), asf(it.b), asf(single ? wide_base[it.a] : it.n));
            if (single) blob[lay.item_off / 16 + 2 * k + 1] = blob[(lay.mesh_off + (size_t)it.a * MESH_REC_BYTES) / 16 + 8];
        }
        // Triangle re-layout (see rt_device.h).  The subtractions and the
        // cross product are wgsl:261-263, evaluated once here in binary32.
        for (uint32_t t = 0; t < n_triangles; ++t) {
            const rt_packed_triangle& p = triangles[t];
            float abx = p.v2[0] - p.v1[0], aby = p.v2[1] - p.v1[1], abz = p.v2[2] - p.v1[2];
            float acx = p.v3[0] - p.v1[0], acy = p.v3[1] - p.v1[1], acz = p.v3[2] - p.v1[2];
            float nx = aby * acz - abz * acy;
            float ny = abz * acx - abx * acz;
            float nz = abx * acy - aby * acx;
            float4* ti = blob.data() + (lay.tri_off + (size_t)t * TRI_ISECT_BYTES) / 16;
            ti[0] = make_float4(p.v1[0], p.v1[1], p.v1[2], nx);
            ti[1] = make_float4(abx, aby, abz, ny);
            ti[2] = make_float4(acx, acy, acz, nz);
            float4* ts = blob.data() + (lay.shade_off + (size_t)t * TRI_SHADE_BYTES) / 16;
            ts[0] = make_float4(p.n1[0], p.n1[1], p.n1[2], p.uv10);
            ts[1] = make_float4(p.n2[0], p.n2[1], p.n2[2], p.uv11);
            ts[2] = make_float4(p.n3[0], p.n3[1], p.n3[2], p.uv20);
            ts[3] = make_float4(p.uv21, p.uv30, p.uv31, 0.0f);
        }
        for (uint32_t i = 0; i < n_spheres; ++i) {
            blob[(lay.sphere_off + (size_t)i * SPHERE_BYTES) / 16] =
                make_float4(spheres[i].pos[0], spheres[i].pos[1], spheres[i].pos[2], spheres[i].radius);
            memcpy(blob.data() + (lay.mat_off + (size_t)(n_meshes + i) * MATERIAL_BYTES) / 16,
                   &spheres[i].material, MATERIAL_BYTES);
        }

        free_scene(h);
        int rc;
        if ((rc = upload(h, h->blob, blob.data(), blob.size())) != RT_OK) return rc;
        // ---- the small blob of the hybrid launches (experiments build only) ----
        std::vector<float4> small;
        SceneLayout sl{};
        uint32_t small_need = 1;
        bool small_ok = false;
#if RT_EXPERIMENTS
#include "experiments/rt_api_hybrid_blob.inl"   // (statement fragment: fills small / sl / small_need / small_ok)
#endif
        if (small_ok && (rc = upload(h, h->small_blob, small.data(), small.size())) != RT_OK) return rc;
        h->small_ok = small_ok;
        h->small_lay = sl;
        h->small_stack_entries = small_need;
        HIP_TRY(h, hipStreamSynchronize(h->stream));  // host staging vectors die here
        // The root-box shortcut needs root box == union of the two child boxes, bit for bit
        // (true for the reference's builder; verified, not assumed, since BVHs may be foreign).
        bool unions = true;
        for (uint32_t i = 0; i < n_meshes && unions; ++i) {
            const rt_node* mn = nodes + meshes[i].node_offset;
            if (mn[0].count > 0) continue;
            const rt_node &ca = mn[mn[0].left], &cb = mn[mn[0].right];
            for (int k = 0; k < 3; ++k) {
                const float lo_k = ca.aabb_min[k] < cb.aabb_min[k] ? ca.aabb_min[k] : cb.aabb_min[k];
                const float hi_k = ca.aabb_max[k] > cb.aabb_max[k] ? ca.aabb_max[k] : cb.aabb_max[k];
                // the root box may also be larger than the union (still conservative)
                if (!(mn[0].aabb_min[k] <= lo_k && mn[0].aabb_max[k] >= hi_k)) unions = false;
                // (and the children must be proper boxes, or the interval argument does not hold)
                if (!(ca.aabb_min[k] <= ca.aabb_max[k] && cb.aabb_min[k] <= cb.aabb_max[k])) unions = false;
            }
        }
        h->roots_are_unions = unions;
        h->lay = lay;
        h->n_meshes = n_meshes;
        h->n_spheres = n_spheres;
        h->n_nodes = n_nodes;
        h->n_triangles = n_triangles;
        h->stack_entries = max_height ? max_height : 1;
        // One-dword stack entries hold 7 bits of leaf count and 24 bits of triangle index; they cost
        // a few instructions per push/pop, so they are used when they buy occupancy: when two-dword
        // entries would not leave room for the scene blob and the primary-ray memo in LDS.
        {
            const uint64_t fixed = 8u * 3u * 4u * WAVES_PER_BLOCK + (uint64_t)(LANE_STATE_DWORDS + PIXEL_MEMO_DWORDS) * 64u * 4u * WAVES_PER_BLOCK +
                                   (uint64_t)tlas_entries * 64u * 4u * WAVES_PER_BLOCK;
            const uint64_t wide_stacks = (uint64_t)h->stack_entries * 128u * 4u * WAVES_PER_BLOCK;
            const bool wide_fits = lay.bytes + fixed + wide_stacks <= LDS_BUDGET_BYTES;
            h->stack_must_wide = max_leaf_ref > 127u || n_triangles > (1u << 24);
            h->stack_wide = h->stack_must_wide || wide_fits;
        }
        h->tlas_entries = tlas_entries;
        h->has_tlas = has_tlas;
        h->n_tlas_records = (uint32_t)tlas.size();
        h->any_deep = false;
        for (uint32_t i = 0; i < n_meshes; ++i) h->any_deep = h->any_deep || deep[i];
        h->has_forest = !forest_entries.empty();
        {
            bool plain = n_spheres == 0;
            for (uint32_t i = 0; i < n_meshes && plain; ++i) {
                const rt_material& m = meshes[i].material;
                if (m.flag == RT_MATERIAL_GLASS || (m.flag == RT_MATERIAL_TEXTURE && m.diffuse_index != -1)) plain = false;
            }
            h->plain_materials = plain;
        }
        h->n_items = (uint32_t)items.size();
        h->top_base = top_mesh_base;
        h->top_available = top_mesh_records >= 64 ? std::min<uint32_t>(top_mesh_records, 2048u) : 0u;
        h->have_defer = have_defer;
        h->defer_mesh = defer_mesh;
        h->defer_xform = defer_xform;
        h->defer_internal = defer_internal;
        // LDS residency: blob + the four waves' stacks, cost tables and lane state within the
        // per-workgroup budget (the primary-ray memo goes to LDS only if it still fits, see render_impl)
        uint64_t stacks = ((uint64_t)h->stack_entries * (h->stack_wide ? 128u : 64u) + (uint64_t)h->tlas_entries * 64u) * sizeof(uint32_t) * WAVES_PER_BLOCK +
                          8u * 3u * 4u * WAVES_PER_BLOCK + (uint64_t)LANE_STATE_DWORDS * 64u * 4u * WAVES_PER_BLOCK;
        h->lds_scene = (uint64_t)lay.bytes + stacks <= LDS_BUDGET_BYTES;
        h->camera = scene->camera;
        h->have_scene = true;
        h->history_valid = false;
        h->primary_valid = false;  // (the table holds hits: a function of the scene)
        for (auto& st : h->slot_primary) st.valid = false;
    } catch (const std::bad_alloc&) {
        return fail(h, RT_ERR_OUT_OF_MEMORY, "out of host memory");
    }
    return RT_OK;
}

// rt_render_multi (device-to-device transport): the root may still be copying this handle's previous frame out of
// h->image.  Everything that writes, rebinds or reads the image on the handle's stream waits for that copy first.
static int wait_pending_copy(rt_handle* h) {
    if (h->multi_copy_pending) {
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->multi_copied, 0));
        h->multi_copy_pending = false;
    }
    return RT_OK;
}

int rt_upload_textures(rt_handle* h, const rt_texture_desc* descs, uint32_t n) {
    if (!h || (n && !descs)) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    h->generation += 1;
    if (n > RT_MAX_TEXTURES) return fail(h, RT_ERR_CAPACITY, "Cannot load more than 64 textures");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_textures(h);
    std::vector<DTexture> dt(n);
    for (uint32_t i = 0; i < n; ++i) {
        size_t bytes = (size_t)descs[i].width * descs[i].height * 4;
        uint8_t* p = nullptr;
        if (bytes) {
            if (!descs[i].rgba8) return fail(h, RT_ERR_INVALID_ARGUMENT, "texture without data");
            HIP_TRY(h, hipMalloc((void**)&p, bytes));
            h->texture_data.push_back(p);
            HIP_TRY(h, hipMemcpyAsync(p, descs[i].rgba8, bytes, hipMemcpyHostToDevice, h->stream));
        }
        dt[i] = DTexture{p, descs[i].width, descs[i].height};
    }
    int rc = upload(h, h->textures, dt.data(), dt.size());
    if (rc != RT_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n_textures = n;
    return RT_OK;
}

int rt_set_camera(rt_handle* h, const rt_camera_uniform* camera) {
    if (!h || !camera) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    // (a host that re-sends the same camera every frame -- the reference's update_buffers -- keeps its frames rendered ahead)
    if (memcmp(&h->camera, camera, sizeof(rt_camera_uniform)) != 0) h->generation += 1;
    h->camera = *camera;
    return RT_OK;
}

int rt_set_option(rt_handle* h, const char* name, int value) {
    if (!h || !name) return RT_ERR_INVALID_ARGUMENT;
    std::string n(name);
    h->generation += 1;  // (frames rendered ahead under the old setting are dropped: render_single)
    if (n == "kernel_variant") {
        if (value < -1 || value > 1) return fail(h, RT_ERR_INVALID_ARGUMENT, "kernel_variant must be -1 (auto), 0 or 1");
        h->kernel_variant = value;
    } else if (n == "persistent_blocks") {
        if (value < 1) return fail(h, RT_ERR_INVALID_ARGUMENT, "persistent_blocks must be >= 1");
        h->persistent_blocks = (uint32_t)value;
    } else if (n == "tile_feedback") {
        h->tile_feedback = value ? 1 : 0;
        h->history_valid = false;
    } else if (n == "primary_table") {
        h->use_primary = value ? 1 : 0;
    } else if (n == "primary_hits") {
        h->primary_hits = value ? 1 : 0;
        h->primary_valid = false;
        for (auto& st : h->slot_primary) st.valid = false;
    } else if (n == "max_device_mb") {
        if (value < 0) return fail(h, RT_ERR_INVALID_ARGUMENT, "max_device_mb must be >= 0 (0 = no cap)");
        h->max_device_bytes = (size_t)value << 20;
    } else if (n == "memo_in_table") {
        h->memo_in_table = value ? 1 : 0;
    } else if (n == "primary_per_slot") {
        h->primary_per_slot = value ? 1 : 0;
    } else if (n == "tile_feedback_period") {
        if (value < 1) return fail(h, RT_ERR_INVALID_ARGUMENT, "tile_feedback_period must be >= 1");
        h->tile_feedback_period = value;
        h->history_valid = false;
    } else if (n == "vote_eighths") {
        if (value < -1 || value > 8) return fail(h, RT_ERR_INVALID_ARGUMENT, "vote_eighths must be -1 (automatic) or 0..8");
        h->vote_eighths = value;
    } else if (n == "vote_patience") {
        if (value < -1) return fail(h, RT_ERR_INVALID_ARGUMENT, "vote_patience must be -1 (automatic) or >= 0");
        h->vote_patience = value;
    } else if (n == "pixel_cache") {
        if (value < 0 || value > 2) return fail(h, RT_ERR_INVALID_ARGUMENT, "pixel_cache must be 0, 1 or 2 (memo in global memory)");
        h->pixel_cache_opt = value;
    } else if (n == "tlas") {
        h->use_tlas = value ? 1 : 0;
    } else if (n == "stack_wide") {
        if (value < -1 || value > 1) return fail(h, RT_ERR_INVALID_ARGUMENT, "stack_wide must be -1 (auto), 0 or 1");
        h->force_stack_wide = value;
    } else if (n == "forest") {
        h->use_forest = value ? 1 : 0;
    } else if (n == "flat2") {
        h->use_flat2 = value ? 1 : 0;
    } else if (n == "tlas_min") {
        if (value < 2) return fail(h, RT_ERR_INVALID_ARGUMENT, "tlas_min must be >= 2");
        h->tlas_min = value;
    } else if (n == "cull_roots") {
        if (value < -1 || value > 1) return fail(h, RT_ERR_INVALID_ARGUMENT, "cull_roots must be -1 (auto), 0 or 1");
        h->cull_roots = value;
    } else if (n == "cross_prune") {
        h->cross_prune = value ? 1 : 0;
        h->primary_valid = false;   // (the primary hits in the tables were found by the other walk)
        for (auto& st : h->slot_primary) st.valid = false;
    } else if (n == "lds_scene") {
        h->force_global = value ? 0 : 1;
    } else if (n == "defer_min_nodes") {
        if (value < 1) return fail(h, RT_ERR_INVALID_ARGUMENT, "defer_min_nodes must be >= 1 (takes effect at the next rt_upload_scene)");
        h->defer_min_nodes = value;
    } else if (n == "sort_rounds") {
        if (value < -1 || value > 64) return fail(h, RT_ERR_INVALID_ARGUMENT, "sort_rounds must be -1 (automatic), 0 (off) or 1 .. 64");
        h->sort_rounds = value;
    } else if (!RT_EXPERIMENTS && (n == "lds_top" || n == "wavefront" || n == "hybrid" || n == "lds_tlas")) {
        // (0 = "off" is what this build does anyway: scripts that reset their options keep working)
        if (value != 0) return fail(h, RT_ERR_INVALID_ARGUMENT, "option " + n + " belongs to the measured-slower experiments: "
                                    "build with -DRT_EXPERIMENTS=1 (tools/build_variant.sh exp -DRT_EXPERIMENTS=1)");
    } else if (n == "lds_top") {
        if (value < -1 || value > 2048) return fail(h, RT_ERR_INVALID_ARGUMENT, "lds_top must be -1 (auto), 0 (off) or a record count <= 2048");
        h->lds_top = value;
    } else if (n == "wavefront") {
        if (value < 0 || value > 1) return fail(h, RT_ERR_INVALID_ARGUMENT, "wavefront must be 0 (off) or 1 (whenever legal)");
        h->wavefront = value;
    } else if (n == "pipeline") {
        if (value < -1 || value > rt_handle::PIPE_MAX) return fail(h, RT_ERR_INVALID_ARGUMENT, "pipeline must be -1 (automatic), 0 (off) or 2 .. 8 (frames in flight)");
        h->pipeline = value == 1 ? -1 : value;
    } else if (n == "pipeline_when_idle") {
        h->pipeline_when_idle = value ? 1 : 0;
    } else if (n == "frame_ahead") {
        if (value < -1 || value == 1 || value > (int)RT_MAX_BATCH_FRAMES)
            return fail(h, RT_ERR_INVALID_ARGUMENT, "frame_ahead must be -1 (automatic), 0 (off) or 2 .. 64 (frames per batch)");
        h->frame_ahead = value;
        h->frame_ahead_failed = false;
    } else if (n == "fast_miss") {
        h->fast_miss = value ? 1 : 0;
    } else if (n == "park_levels") {
        h->park_levels = value ? 1 : 0;
    } else if (n == "hybrid") {
        h->hybrid = value ? 1 : 0;
    } else if (n == "lds_tlas") {
        if (value < 0 || value > 2) return fail(h, RT_ERR_INVALID_ARGUMENT, "lds_tlas must be 0 (off), 1 (when it costs no occupancy) or 2 (whenever it fits)");
        h->lds_tlas = value;
    } else if (n == "multi_rccl") {
        if (value < 0 || value > 2) return fail(h, RT_ERR_INVALID_ARGUMENT, "multi_rccl must be 0 (peer copies), 1 (RCCL between distinct devices) or 2 (RCCL always)");
        h->multi_rccl = value;
    } else if (n == "specialise") {
        h->specialise = value ? 1 : 0;
    } else if (n == "batch_tile_major") {
        h->batch_tile_major = value ? 1 : 0;
    } else if (n == "batch_frames") {
        if (value < 1 || value > (int)RT_MAX_BATCH_FRAMES) return fail(h, RT_ERR_INVALID_ARGUMENT, "batch_frames must be 1..64");
        h->batch_frames_opt = value;
    } else {
        return fail(h, RT_ERR_INVALID_ARGUMENT, "unknown option " + n);
    }
    return RT_OK;
}

int rt_set_counters(rt_handle* h, int enabled) {
    if (!h) return RT_ERR_INVALID_ARGUMENT;
    h->count_tests = enabled ? 1 : 0;
    h->generation += 1;
    return RT_OK;
}

uint64_t rt_strip_texels(uint32_t width, uint32_t height, uint32_t rank, uint32_t world) {
    if (world == 0 || rank >= world) return 0;
    uint32_t strips = (height + 7) / 8;
    uint32_t local = strips / world + (rank < strips % world ? 1 : 0);
    return (uint64_t)local * 8u * width;
}

// One launch of the hot path: a single frame (n_batch == 0: wgsl `main`, blending in place), or a batch
// of n_batch >= 2 consecutive frames (Params.frames advancing by one each, app.rs:44-53) sampled in one
// persistent launch over (frame, tile) work items and blended in frame order by a dense second kernel.
// Sum of the recorded launch times (HIP events around every launch since rt_reset_timing).  The pairs are recorded on
// whichever stream the launch ran on -- the pipelined frames' internal streams too -- and synchronising the handle's
// stream only proves that the KERNELS before a pair's second event are done (the blends wait for them), not that the
// runtime has retired the event itself: every pair is synchronised before it is read (hipEventElapsedTime on an event
// that is not ready is hipErrorNotReady; seen once in 850 fuzz scenes with four frames in flight).
static hipError_t harvest_event_times(rt_handle* h, double& sum_ms) {
    for (size_t i = 0; i < h->ev_used; ++i) {
        hipError_t e = hipEventSynchronize(h->ev_pool[i].second);
        if (e != hipSuccess) return e;
        float ms = 0.0f;
        e = hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second);
        if (e != hipSuccess) return e;
        sum_ms += ms;
    }
    return hipSuccess;
}

// The wavefront sequence's host side lives with the other measured-slower features: compiled with
// -DRT_EXPERIMENTS=1 only; the product build has the two stubs below and never sets WavefrontPlan::on.
struct WavefrontPlan { bool on = false; uint32_t frame_slots = 0; uint64_t slots = 0, rounds = 0; };
#if RT_EXPERIMENTS
#include "experiments/rt_api_wavefront.inl"
#else
static int wavefront_prepare(rt_handle*, const rt_params*, const RenderArgs&, uint32_t, uint32_t, WavefrontPlan&, bool&) { return RT_OK; }
static int wavefront_run(rt_handle*, const RenderArgs&, const WavefrontPlan&) { return RT_OK; }
#endif

static int render_impl(rt_handle* h, const rt_params* params, uint32_t rank, uint32_t world, uint32_t n_batch = 0,
                       bool blend_later = false) {
    if (!h || !params) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    h->ahead.valid = false;  // (whatever launches here breaks a sequence of frames rendered ahead: render_single)
    if (!h->have_scene) return fail(h, RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    if (world == 0 || rank >= world) return fail(h, RT_ERR_INVALID_ARGUMENT, "bad rank/world");
    if (params->width == 0 || params->height == 0) return fail(h, RT_ERR_INVALID_ARGUMENT, "empty image");
    const uint64_t need_texels = world == 1 ? (uint64_t)params->width * params->height
                                            : rt_strip_texels(params->width, params->height, rank, world);
    if (need_texels > h->image_texels)
        return fail(h, RT_ERR_CAPACITY, "image larger than the bound image buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = wait_pending_copy(h); rc != RT_OK) return rc;
    RenderArgs a{};
    a.params = *params;
    a.camera = h->camera;
    if (n_batch) {
        if (n_batch > RT_MAX_BATCH_FRAMES) return fail(h, RT_ERR_INVALID_ARGUMENT, "batch too large");
        const uint64_t scratch_stride = need_texels;
        if (h->batch_scratch_texels < scratch_stride * n_batch) {
            // Room for the largest batch the options allow, not just this one: a later, larger batch of the same shape
            // (a warm-up of 5 frames, then 20) must not re-allocate -- a hipMalloc of hundreds of megabytes costs tens of
            // milliseconds -- in the middle of a host's frame sequence.
            const uint64_t slots = std::max<uint64_t>(n_batch, std::min<uint64_t>((uint64_t)h->batch_frames_opt, RT_MAX_BATCH_FRAMES));
            const size_t old_bytes = h->batch_scratch_texels * sizeof(float4);
            if (!fits_cap(h, scratch_stride * n_batch * sizeof(float4), old_bytes))
                return fail(h, RT_ERR_OUT_OF_MEMORY, "option max_device_mb leaves no room for the scratch images of this batch");
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            free_dev(h->batch_scratch);
            h->batch_scratch_texels = 0;
            if (fits_cap(h, scratch_stride * slots * sizeof(float4)) &&
                hipMalloc((void**)&h->batch_scratch, scratch_stride * slots * sizeof(float4)) == hipSuccess) {
                h->batch_scratch_texels = scratch_stride * slots;
            } else {   // (not that much memory: what this batch needs)
                (void)hipGetLastError();
                HIP_TRY(h, hipMalloc((void**)&h->batch_scratch, scratch_stride * n_batch * sizeof(float4)));
                h->batch_scratch_texels = scratch_stride * n_batch;
            }
            h->scratch_w = 0;  // (new memory: zero it below)
        }
        // The padding rows of a ragged last strip are never rendered: they must blend as zeros, not as the samples an
        // earlier batch of another shape left there.  Zero ALL of the scratch whenever the shape of its frames changes
        // (slot k of any later batch of this shape starts at k strides: the number of frames does not matter).
        if (h->scratch_w != params->width || h->scratch_h != params->height || h->scratch_rank != rank ||
            h->scratch_world != world) {
            HIP_TRY(h, hipMemsetAsync(h->batch_scratch, 0, h->batch_scratch_texels * sizeof(float4), h->stream));
            h->scratch_w = params->width; h->scratch_h = params->height; h->scratch_rank = rank;
            h->scratch_world = world; h->scratch_n = n_batch;
        }
        a.batch_frames = n_batch;
        a.batch_stride = scratch_stride;
        a.batch_tile_major = (uint32_t)h->batch_tile_major;
    }
    for (int k = 0; k < 3; ++k) {  // see pixel_cache_begin: the same IEEE operations, unfused
        volatile float rz = h->camera.cam_to_world[0][k] * 0.0f, uz = h->camera.cam_to_world[1][k] * 0.0f;
        volatile float t = h->camera.cam_to_world[3][k] + rz;
        a.memo_ro[k] = t + uz;
    }
    {
        // (i32 arithmetic wraps in the shader; frames = -1 gives +inf there too)
        volatile float w = 1.0f / (float)(int32_t)((uint32_t)params->frames + 1u);
        volatile float r = 1.0f - w;
        a.blend_weight = w;
        a.blend_rest = r;
    }
    {
        const int32_t n = params->rays_per_pixel;
        a.spp_reciprocal = (n > 0 && (n & (n - 1)) == 0) ? 1.0f / (float)n : 0.0f;
    }
    a.blob = h->blob;
    a.lay = h->lay;
    // (an explicitly requested deferred-walk sequence -- option "sort_rounds" > 0, tests -- reads the scene in place: the
    // parking instantiations exist for global-memory scenes only; the automatic setting never defers a mesh of a scene
    // that fits the LDS)
    a.lds_scene = (h->lds_scene && !h->force_global && !(h->sort_rounds > 0 && h->have_defer)) ? 1u : 0u;
    a.cull_roots = (h->roots_are_unions && (h->cull_roots == 1 || (h->cull_roots < 0 && h->n_meshes >= 16))) ? 1u : 0u;
    a.many_mesh = (h->has_tlas || (a.cull_roots && !h->has_forest)) ? 1u : 0u;
    a.forest_cull = h->cull_roots != 0 ? 1u : 0u;
    a.cross_prune = h->cross_prune != 0 ? 1u : 0u;
    {
        uint32_t ds, dv;
        memcpy(&ds, &h->camera.defocus_strength, 4);
        memcpy(&dv, &h->camera.diverge_strength, 4);
        a.simple = (h->specialise && h->plain_materials && ds == 0u && dv == 0u) ? 1u : 0u;  // (both strengths +0)
    }
    // the per-lane primary-ray cache is used when it still leaves room for 4 workgroups per CU
    a.textures = h->textures;
    a.srgb_lut = h->srgb_lut;
    a.image = n_batch ? h->batch_scratch : h->image;
    a.counters = h->counters;
    a.n_meshes = h->n_meshes;
    a.n_spheres = h->n_spheres;
    a.n_textures = h->n_textures;
    a.stack_entries = h->stack_entries;
    a.tlas_entries = h->tlas_entries;
    a.stack_wide = (h->stack_must_wide || (h->force_stack_wide < 0 ? h->stack_wide : h->force_stack_wide != 0)) ? 1u : 0u;
    // (the global-memory kernels park a pending mesh hit of 5 dwords in the stack column, intersect_scene)
    if (!a.lds_scene && a.stack_entries < (a.stack_wide ? 3u : 5u)) a.stack_entries = a.stack_wide ? 3u : 5u;
    a.n_items = h->n_items;
    a.strip_rank = rank;
    a.strip_world = world;
    a.tiles_x = (params->width + 7) / 8;
    uint32_t strips = (params->height + 7) / 8;
    a.tiles_y = strips / world + (rank < strips % world ? 1 : 0);
    a.count_tests = (uint32_t)h->count_tests;
    // auto: with about one tile per resident wave there is nothing to refill from, and the
    // plain one-wave-per-tile dispatch is a little faster (tools/strip_scaling.py)
    const uint32_t resident_waves = h->persistent_blocks * WAVES_PER_BLOCK;
    a.kernel_variant = h->kernel_variant >= 0 ? (uint32_t)h->kernel_variant
                                              : ((uint64_t)a.tiles_x * a.tiles_y * 4 <= (uint64_t)resident_waves * 5 ? 1u : 0u);
    if (n_batch) a.kernel_variant = 0;  // (frame, tile) work items are the persistent kernel's
    if (a.count_tests) a.kernel_variant = 0;  // (the counter instantiations exist for the persistent kernel only)
    // (all fields that size the LDS are set by now)
    a.pixel_cache = 0;
    a.pixel_cache_mem = nullptr;
    if (h->pixel_cache_opt == 1) {
        a.pixel_cache = 1;
        if (render_lds_bytes(a) > LDS_BUDGET_BYTES) a.pixel_cache = 0;  // no room in LDS
    }
    // LDS-staged top of the big mesh's BVH (global-memory scenes): as many of its breadth-first numbered
    // records as fit beside the stacks without costing a workgroup per CU, or what option "lds_top" says
    a.top_base = h->top_base;
    a.top_count = 0;
    if (RT_EXPERIMENTS && !a.lds_scene && h->top_available && h->lds_top != 0) {
        const size_t used = render_lds_bytes(a);
        uint32_t fit = used < LDS_BUDGET_BYTES ? (uint32_t)((LDS_BUDGET_BYTES - used) / WIDE_REC_BYTES) : 0u;
        if (h->lds_top > 0) fit = (uint32_t)h->lds_top;
        a.top_count = std::min(fit, h->top_available);
    }
    a.tlas_lds = 0;
    if (RT_EXPERIMENTS && !a.lds_scene && a.many_mesh && h->has_tlas && h->lds_tlas != 0 && params->debug_flag == 0) {
        // 1: as many of the breadth-first numbered records (the top levels first) as fit without costing a workgroup per
        // CU; 2: the whole tree if the LDS can hold it at all
        const size_t used = render_lds_bytes(a), need = (size_t)h->n_tlas_records * WIDE_REC_BYTES;
        const uint32_t per_cu_now = std::min<uint32_t>(BLOCKS_PER_CU, (uint32_t)((160u * 1024u) / (used ? used : 1)));
        const size_t room = (160u * 1024u) / per_cu_now > used ? (160u * 1024u) / per_cu_now - used : 0;
        if (h->lds_tlas == 2) {
            if (used + need <= 64u * 1024u) a.tlas_lds = h->n_tlas_records;
        } else {
            a.tlas_lds = std::min<uint32_t>(h->n_tlas_records, (uint32_t)(room / WIDE_REC_BYTES));
        }
    }
    a.persistent_blocks = h->persistent_blocks;
    {
        // workgroups that fit a CU's 160 KiB of LDS (BLOCKS_PER_CU when the register budget is the limit)
        const size_t lds = render_lds_bytes(a);
        uint32_t per_cu = lds ? (uint32_t)((160u * 1024u) / lds) : BLOCKS_PER_CU;
        if (per_cu > BLOCKS_PER_CU) per_cu = BLOCKS_PER_CU;
        if (per_cu < 1u) per_cu = 1u;
        const uint32_t fit = (h->persistent_blocks / BLOCKS_PER_CU) * per_cu;
        if (fit < a.persistent_blocks && fit > 0) a.persistent_blocks = fit;
    }
    if (h->pixel_cache_opt && a.pixel_cache == 0 && a.kernel_variant == 0 && params->debug_flag == 0 &&
        (h->pixel_cache_words >= (size_t)h->persistent_blocks * WAVES_PER_BLOCK * PIXEL_MEMO_DWORDS * 64u ||
         fits_cap(h, (size_t)h->persistent_blocks * WAVES_PER_BLOCK * PIXEL_MEMO_DWORDS * 64u * sizeof(uint32_t), h->pixel_cache_words * sizeof(uint32_t)))) {
        // persistent kernel: a fixed number of waves, so the cache can live in global memory
        const size_t need = (size_t)h->persistent_blocks * WAVES_PER_BLOCK * PIXEL_MEMO_DWORDS * 64u;
        if (h->pixel_cache_words < need) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            free_dev(h->pixel_cache_mem);
            HIP_TRY(h, hipMalloc((void**)&h->pixel_cache_mem, need * sizeof(uint32_t)));
            h->pixel_cache_words = need;
        }
        a.pixel_cache = 2;
        a.pixel_cache_mem = h->pixel_cache_mem;
    }
    a.fast_miss = h->fast_miss != 0 && a.pixel_cache != 0u ? 1u : 0u;
    // how many rounds of deferred walks this launch would run (0: none)
    const size_t park_records = (size_t)need_texels * (n_batch ? n_batch : 1u);
    const size_t park_bytes = ((park_records + 63) / 64) * (size_t)PARK_PLANES * 64u * sizeof(float4);  // per queue
    uint32_t n_rounds = h->sort_rounds > 0 ? (uint32_t)h->sort_rounds : 0u;
    if (h->sort_rounds < 0 && h->have_defer) {
        // (work of the launch in units of one 1920 x 1080 frame at 16 samples per pixel)
        const double units = (double)park_records * (double)(params->rays_per_pixel > 0 ? params->rays_per_pixel : 0) / (1920.0 * 1080.0 * 16.0);
        // (tuned on the config 3 and config 5 stand-ins: the longer the walks -- the bigger the mesh's BVH --, the
        // earlier a round pays for its fixed cost)
        if (h->defer_internal >= 400000u)
            // (config 5 stand-in at 3840 x 2160, 64 spp, 16 frames per launch = 256 units: 102.9 / 98.4 / 96.8 / 96.8 / 97.3 ms
            // per frame with 8 / 12 / 16 / 24 / 32 rounds)
            // (32 frames per launch = 512 units: 95.6 ms with 16 rounds, 95.0 with 24)
            n_rounds = units >= 384.0 ? 24u : units >= 96.0 ? 16u : units >= 48.0 ? 12u : units >= 24.0 ? 8u : units >= 12.0 ? 4u : units >= 4.0 ? 3u : units >= 2.0 ? 2u : 0u;
        else
            n_rounds = units >= 24.0 ? 6u : units >= 12.0 ? 4u : units >= 8.0 ? 3u : 0u;
        if (n_rounds && h->park_capacity < park_records) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || 2 * park_bytes > free_b / 2) n_rounds = 0;  // (both queues within half of what is free: 288 GB are there to be used)
            if (!fits_cap(h, 2 * park_bytes, 2 * (((h->park_capacity + 63) / 64) * (size_t)PARK_PLANES * 64u * sizeof(float4)))) n_rounds = 0;  // (option max_device_mb)
        }
    }
    bool rounds = n_rounds > 0 && h->have_defer && a.many_mesh == 0 && a.kernel_variant == 0 && params->debug_flag == 0 &&
                  params->rays_per_pixel > 0 && a.lds_scene == 0u;
    // The intersection vote (path_begin).  The more a traversal costs beside the rest of an iteration, the longer it pays
    // to let the lanes on memoised primary segments catch up first: 6/8 of the lanes or 3 iterations when the scene is in
    // LDS and walked by the few-mesh kernels (config 2: 1.221 ms per frame; 1.222 with 7/8 and 16, 1.366 with 8/8), 7/8 or
    // 16 iterations otherwise (sponza-sized stand-in 10.69 -> 10.21 ms, 200-mesh stand-in 5.00 -> 4.85), every lane or 16
    // iterations in a deferred-walk sequence, whose resumed pixels arrive in every phase (config 3 stand-in 5.86 -> 5.45,
    // config 5 geometry 3.35 -> 3.22; profiles/r03_experiments/ab_*_vote*.txt).
    {
        const bool costly = a.lds_scene == 0u || a.many_mesh != 0u;
        a.vote_eighths = h->vote_eighths >= 0 ? (uint32_t)h->vote_eighths : rounds ? 8u : costly ? 7u : 6u;
        a.vote_patience = h->vote_patience >= 0 ? (uint32_t)h->vote_patience : (rounds || costly) ? 16u : 3u;
    }
    const bool wavefront_wanted = RT_EXPERIMENTS && h->wavefront != 0 && a.many_mesh != 0 && !h->any_deep && params->debug_flag == 0 && params->rays_per_pixel > 0;
    // Pipelined single frames: a plain one-frame launch (no batch, no sequence of launches).
    // S is the stream this frame's sampling launch and its bookkeeping run on.
    const int pipeline_opt = h->pipeline < 0 ? automatic_pipeline_depth(world) : h->pipeline;
    if (!h->queues_noted && pipeline_opt != 0 && (hw_queues_requested() < 5 || pipeline_opt + 1 > hw_queues_requested())) {
        // Not an error, said once (rt_last_error after a call that returned RT_OK): the pipeline is limited by, or
        // oversubscribes, the hardware queues the host asked the runtime for.
        h->queues_noted = true;
        h->err = "note: GPU_MAX_HW_QUEUES is " + std::to_string(hw_queues_requested()) + " (read when the first handle was created): " +
                 (h->pipeline < 0 ? "the frame pipeline runs " + std::to_string(pipeline_opt) + " frames deep instead of 4"
                                  : "option pipeline = " + std::to_string(pipeline_opt) + " needs " + std::to_string(pipeline_opt + 1) + " queues, its streams will share queues and serialise") +
                 "; export GPU_MAX_HW_QUEUES=8 (12 for strip shares of >= 4 ranks) before the HIP runtime initialises (INTEGRATION.md section 3)";
    }
    const uint32_t pipe_depth = pipeline_opt >= 2 ? (uint32_t)pipeline_opt : 2u;
    bool pipe = pipeline_opt != 0 && n_batch == 0 && params->debug_flag == 0 && params->rays_per_pixel > 0 &&
                !rounds && !wavefront_wanted;   // (strips too: the gather reads the image behind the blend, on the handle's stream)
    if (pipe && h->pipe_scratch_texels < need_texels) {   // (option max_device_mb: the pipeline's scratch images have to fit)
        size_t held = 0;
        for (int k = 0; k < rt_handle::PIPE_MAX; ++k) if (h->pipe_scratch[k]) held += h->pipe_scratch_texels * sizeof(float4);
        if (!fits_cap(h, (size_t)pipe_depth * need_texels * sizeof(float4), held)) pipe = false;
    } else if (pipe) {
        uint32_t have = 0;
        for (uint32_t k = 0; k < pipe_depth; ++k) have += h->pipe_scratch[k] ? 1u : 0u;
        if (!fits_cap(h, (size_t)(pipe_depth - have) * h->pipe_scratch_texels * sizeof(float4))) pipe = false;
    }
    bool took_idle_path = false;  // this frame would have been pipelined, found nothing in flight and takes the plain launch
    bool wrote_tables = false;    // this launch rewrites a table later launches read (tile order, tile costs, primary table)
    if (pipe && h->pipeline_when_idle == 0) {
        // Every pipelined frame ends with its blend on the handle's stream: an idle stream means no frame is in flight,
        // i.e. this frame has nothing to overlap with (yet).  It takes the plain launch; the frames a host issues while
        // this one runs are pipelined behind it.
        const hipError_t q = hipStreamQuery(h->stream);
        (void)hipGetLastError();  // (hipErrorNotReady is an answer, not a failure: do not leave it for the launchers' hipGetLastError)
        if (q == hipSuccess) { pipe = false; took_idle_path = true; }
        else if (q != hipErrorNotReady) return fail(h, RT_ERR_DEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(q));
    }
    const uint32_t pslot = h->pipe_seq % pipe_depth;
    hipStream_t S = h->stream;
    bool pipe_barrier = false;  // this frame rewrites shared tables (tile order, primary rays): the other stream's frame has to be done
    if (pipe) {
        h->pipe_seq = (h->pipe_seq + 1) % 840u;   // (a multiple of every depth)
        for (int k = 0; k < (int)pipe_depth; ++k) {   // (no more streams than frames in flight: they share the process's hardware queues)
            if (!h->pipe_stream[k]) HIP_TRY(h, hipStreamCreateWithFlags(&h->pipe_stream[k], hipStreamNonBlocking));
            if (!h->pipe_sampled[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_sampled[k], hipEventDisableTiming));
            if (!h->pipe_blended[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_blended[k], hipEventDisableTiming));
            if (!h->pipe_work[k]) {
                HIP_TRY(h, hipMalloc((void**)&h->pipe_work[k], 64 * sizeof(uint32_t)));
                HIP_TRY(h, hipMemsetAsync(h->pipe_work[k], 0, 64 * sizeof(uint32_t), h->pipe_stream[k]));
                h->pipe_work_slot[k] = 0;
            }
        }
        if (!h->pipe_book) HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_book, hipEventDisableTiming));
        if (!h->pipe_main) HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_main, hipEventDisableTiming));
        if (h->pipe_scratch_texels < need_texels) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));   // (the blends on this stream wait for every sampling launch)
            for (int k = 0; k < rt_handle::PIPE_MAX; ++k) {
                free_dev(h->pipe_scratch[k]);
                h->pipe_scratch[k] = nullptr;
                h->pipe_blended_set[k] = false;
            }
            h->pipe_scratch_texels = need_texels;
        }
        if (!h->pipe_scratch[pslot]) {
            HIP_TRY(h, hipMalloc((void**)&h->pipe_scratch[pslot], h->pipe_scratch_texels * sizeof(float4)));
            h->pipe_layout[pslot][4] = 0u;  // (nothing zeroed yet)
        }
        S = h->pipe_stream[pslot];
        // this frame's scratch image is free once the frame before last has been blended out of it; the tables the last
        // bookkeeping frame rewrote are complete; whatever the handle's stream did outside the pipeline is complete
        if (h->pipe_blended_set[pslot]) HIP_TRY(h, hipStreamWaitEvent(S, h->pipe_blended[pslot], 0));
        {
            // The blend covers all need_texels, the sampling launch writes the pixels of the frame only: the padding rows of
            // a ragged last strip (world > 1, height % 8 != 0) must blend as zeros, as in the batch path -- a fresh
            // allocation, or samples an earlier frame of another shape left there, would otherwise reach the strip
            // buffer's padding rows (the assembled frame never reads them).
            const uint32_t key[5] = {params->width, params->height, rank, world, (uint32_t)need_texels};
            if (memcmp(h->pipe_layout[pslot], key, sizeof(key)) != 0) {
                HIP_TRY(h, hipMemsetAsync(h->pipe_scratch[pslot], 0, need_texels * sizeof(float4), S));
                memcpy(h->pipe_layout[pslot], key, sizeof(key));
            }
        }
        if (h->pipe_book_set) HIP_TRY(h, hipStreamWaitEvent(S, h->pipe_book, 0));
        // (not for a frame that took the plain launch because nothing was in flight and rewrote no table: the host SAW the
        // stream idle before it, so everything older is complete; this frame's samples go to a scratch image, and its blend
        // follows that launch in the handle's stream's order anyway -- so the second frame of a burst overlaps the first)
        if (h->pipe_main_set && h->pipe_main_need) HIP_TRY(h, hipStreamWaitEvent(S, h->pipe_main, 0));
    }
    auto barrier_other = [&]() -> hipError_t {   // before rewriting a shared table: the other stream's sampling launch is done
        wrote_tables = true;
        if (!pipe || pipe_barrier) return hipSuccess;
        pipe_barrier = true;
        if (h->pipe_main_set) {   // (... and whatever the handle's stream last launched outside the pipeline)
            const hipError_t e = hipStreamWaitEvent(S, h->pipe_main, 0);
            if (e != hipSuccess) return e;
        }
        for (uint32_t k = 0; k < (uint32_t)rt_handle::PIPE_MAX; ++k)
            if (k != pslot && h->pipe_sampled_set[k]) {
                const hipError_t e = hipStreamWaitEvent(S, h->pipe_sampled[k], 0);
                if (e != hipSuccess) return e;
            }
        return hipSuccess;
    };
    // Primary-ray table for the memo: recomputed when the camera or the frame size changed
    a.primary = nullptr;
    a.primary_complete = 0u;
    const bool camera_moved = h->last_frame_camera_valid && memcmp(&h->last_frame_camera, &h->camera, sizeof(rt_camera_uniform)) != 0;
    if (params->debug_flag == 0 && params->rays_per_pixel > 0) {
        h->last_frame_camera = h->camera;
        h->last_frame_camera_valid = true;
    }
    // (a batch keeps the table whatever the camera did: its frames share it)
    const size_t table_texels = (size_t)((params->width + 7) / 8) * ((params->height + 7) / 8) * 64;  // whole 8x8 tiles
    // (option max_device_mb: a table that would have to be allocated beyond the cap is done without -- every pixel then
    // computes its own memo; a slot table likewise falls back to the shared one)
    const bool table_fits = h->primary_texels >= table_texels || fits_cap(h, table_texels * 64u, h->primary_texels * 64u);
    if (h->use_primary && table_fits && a.pixel_cache != 0 && params->debug_flag == 0 && params->rays_per_pixel > 0 && !(camera_moved && n_batch == 0)) {
        const size_t texels = table_texels;
        // (the counter kernels re-intersect every segment, so a launch with counters neither needs nor fills the hits)
        const bool want_hits = h->primary_hits != 0 && a.count_tests == 0u;
        const bool shared_fits = h->primary && h->primary_texels >= texels && h->primary_valid && h->primary_w == params->width &&
                                 h->primary_h == params->height && h->primary_rank == rank && h->primary_world == world &&
                                 (!want_hits || h->primary_with_hits) &&
                                 memcmp(&h->primary_camera, &h->camera, sizeof(rt_camera_uniform)) == 0;
        if (pipe && !shared_fits && h->primary_per_slot != 0 && pslot < (uint32_t)rt_handle::PIPE_MAX &&
            (h->slot_primary[pslot].texels >= texels || fits_cap(h, texels * 64u, h->slot_primary[pslot].texels * 64u))) {
            rt_handle::SlotTable& st = h->slot_primary[pslot];
            if (st.texels < texels) {
                HIP_TRY(h, hipStreamSynchronize(S));   // (the slot's previous frame read the old one)
                free_dev(st.table);
                st.texels = 0;
                st.valid = false;
                HIP_TRY(h, hipMalloc((void**)&st.table, texels * 64));
                st.texels = texels;
            }
            if (!st.valid || st.w != params->width || st.h != params->height || st.rank != rank || st.world != world ||
                (want_hits && !st.with_hits) || memcmp(&st.camera, &h->camera, sizeof(rt_camera_uniform)) != 0) {
                a.primary = nullptr;
                HIP_TRY(h, launch_primary(a, st.table, want_hits, S));
                st.valid = true;
                st.w = params->width; st.h = params->height; st.rank = rank; st.world = world;
                st.with_hits = want_hits;
                st.camera = h->camera;
            }
            a.primary = st.table;
            a.primary_complete = st.with_hits ? 1u : 0u;
        } else {
        if (h->primary_texels < texels) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            for (int k = 0; k < rt_handle::PIPE_MAX; ++k)   // (pipelined frames read the table on their own streams)
                if (h->pipe_stream[k]) HIP_TRY(h, hipStreamSynchronize(h->pipe_stream[k]));
            free_dev(h->primary);
            HIP_TRY(h, hipMalloc((void**)&h->primary, texels * 64));
            h->primary_texels = texels;
            h->primary_valid = false;
        }
        // (the counter kernels re-intersect every segment, so a launch with counters neither needs nor fills the hits)
        const bool with_hits = h->primary_hits != 0 && a.count_tests == 0u;
        if (!h->primary_valid || h->primary_w != params->width || h->primary_h != params->height ||
            h->primary_rank != rank || h->primary_world != world || (with_hits && !h->primary_with_hits) ||
            memcmp(&h->primary_camera, &h->camera, sizeof(rt_camera_uniform)) != 0) {
            HIP_TRY(h, barrier_other());
            a.primary = nullptr;
            HIP_TRY(h, launch_primary(a, h->primary, with_hits, S));
            h->primary_valid = true;
            h->primary_w = params->width;
            h->primary_h = params->height;
            h->primary_rank = rank;
            h->primary_world = world;
            h->primary_with_hits = with_hits;
            h->primary_camera = h->camera;
        }
        a.primary = h->primary;
        a.primary_complete = h->primary_with_hits ? 1u : 0u;
        }
    }
    // With a complete table a lane's memo EQUALS its table entry for the pixel's whole life: the kernels whose memo has no
    // room in LDS read the entry in place (pixel_cache = 4) instead of copying it into a buffer in global memory --
    // that copy was most of what the global-memory kernels wrote and half of what they read.
    if (a.pixel_cache == 2u && a.lds_scene == 0u && a.primary != nullptr && a.primary_complete != 0u && h->memo_in_table != 0) {
        a.pixel_cache = 4u;
        a.pixel_cache_mem = nullptr;
    }
    // The global-memory memo is per resident wave and stateful across a pixel's samples: one per concurrent launch.
    // Slot 0 has its own too -- a plain launch on the handle's stream (pixel_cache_mem) may still be running beside
    // the pipelined frame of slot 0 when pipe_main_need is false (a frame that found the stream idle).
    if (pipe && a.pixel_cache == 2u) {
        const size_t need = (size_t)h->persistent_blocks * WAVES_PER_BLOCK * PIXEL_MEMO_DWORDS * 64u;
        if (h->pipe_memo_words[pslot] < need) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            free_dev(h->pipe_memo[pslot]);
            HIP_TRY(h, hipMalloc((void**)&h->pipe_memo[pslot], need * sizeof(uint32_t)));
            h->pipe_memo_words[pslot] = need;
        }
        a.pixel_cache_mem = h->pipe_memo[pslot];
    }
    // a fresh tile counter per launch: a ring of 64, zeroed in one go each time it wraps (launches on
    // one stream are ordered, so every earlier user of the ring is done by then)
    if (pipe) {   // (a ring per pipe stream: its memset only touches launches of its own, in-order stream)
        h->pipe_work_slot[pslot] = (h->pipe_work_slot[pslot] + 1) & 63u;
        if (h->pipe_work_slot[pslot] == 0u) HIP_TRY(h, hipMemsetAsync(h->pipe_work[pslot], 0, 64 * sizeof(uint32_t), S));
        a.work_counter = h->pipe_work[pslot] + h->pipe_work_slot[pslot];
    } else {
        h->work_slot = (h->work_slot + 1) & 63u;
        if (h->work_slot == 0u) HIP_TRY(h, hipMemsetAsync(h->work_counters, 0, 64 * sizeof(uint32_t), h->stream));
        a.work_counter = h->work_counters + h->work_slot;
    }
    // Tile-cost feedback (persistent kernel, path-trace frames only): the tiles are handed out in the
    // order of the rays each took in an earlier frame of the same shape, heaviest first.  Progressive
    // accumulation re-renders the same view, so an order stays good: it is refreshed every
    // `tile_feedback_period` frames (costs are recorded in the frame before a refresh), not every frame.
    const uint32_t n_tiles = a.tiles_x * a.tiles_y;
    a.tile_order = nullptr;
    a.tile_cost = nullptr;
    // Wavefront sequence (option "wavefront", experiments build only: experiments/rt_api_wavefront.inl)
    WavefrontPlan wf{};
    if (int rc = wavefront_prepare(h, params, a, n_tiles, n_batch, wf, rounds); rc != RT_OK) return rc;
    const bool wavefront = wf.on;
    if (rounds && h->park_capacity < park_records) {
        // the two park queues (an automatic sequence that cannot have them falls back to the plain launch)
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        free_dev(h->park_queue[0]);
        free_dev(h->park_queue[1]);
        h->park_capacity = 0;
        const hipError_t e0 = hipMalloc((void**)&h->park_queue[0], park_bytes);
        const hipError_t e1 = e0 == hipSuccess ? hipMalloc((void**)&h->park_queue[1], park_bytes) : e0;
        if (e1 != hipSuccess) {
            (void)hipGetLastError();
            free_dev(h->park_queue[0]);
            free_dev(h->park_queue[1]);
            if (h->sort_rounds > 0) return fail(h, RT_ERR_OUT_OF_MEMORY, "sort_rounds: no device memory for the park queues");
            rounds = false;
        } else {
            h->park_capacity = park_records;
        }
    }
    if (h->tile_feedback && params->debug_flag == 0 && n_tiles <= h->tile_capacity && n_tiles > 0 && a.kernel_variant == 0) {
        const bool same_shape = h->history_valid && h->hist_w == params->width && h->hist_h == params->height &&
                                h->hist_rank == rank && h->hist_world == world;
        if (!same_shape) {
            h->have_order = false;
            h->costs_ready = false;
        }
        if (h->costs_ready && (!h->have_order || h->order_age >= (uint32_t)h->tile_feedback_period)) {
            const long long per_tile = 64ll * (params->rays_per_pixel > 0 ? params->rays_per_pixel : 0) *
                                       (params->number_of_bounces >= 0 ? params->number_of_bounces + 1 : 0);
            const uint32_t max_cost = per_tile > 0xffffffffll ? 0xffffffffu : (uint32_t)per_tile;
            HIP_TRY(h, barrier_other());
            HIP_TRY(h, launch_tile_order(h->tile_cost[h->cost_slot], n_tiles, max_cost, h->tile_order, S));
            h->have_order = true;
            h->order_age = 0;
            h->costs_ready = false;
        }
        const uint32_t frames_now = n_batch ? n_batch : 1u;
        if (h->have_order) a.tile_order = h->tile_order;
        if (!h->have_order || h->order_age + frames_now >= (uint32_t)h->tile_feedback_period) {
            h->cost_slot ^= 1;
            a.tile_cost = h->tile_cost[h->cost_slot];
            wrote_tables = true;
            HIP_TRY(h, hipMemsetAsync(a.tile_cost, 0, (size_t)n_tiles * sizeof(uint32_t), S));
            h->costs_ready = true;
        }
        h->order_age += frames_now;
        h->history_valid = true;
        h->hist_w = params->width;
        h->hist_h = params->height;
        h->hist_rank = rank;
        h->hist_world = world;
    }
    if (h->ev_used == h->ev_pool.size()) {
        if (h->ev_pool.size() >= 4096) {
            // the pool is full: add the recorded times to the running total, then reuse the events
            HIP_TRY(h, harvest_event_times(h, h->ev_ms_harvested));
            h->ev_used = 0;
        } else {
            hipEvent_t s0 = nullptr, s1 = nullptr;
            HIP_TRY(h, hipEventCreate(&s0));
            HIP_TRY(h, hipEventCreate(&s1));
            h->ev_pool.emplace_back(s0, s1);
        }
    }
    {
        const uint32_t tile_blocks = (n_tiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
        h->last_launch[0] = (uint32_t)render_lds_bytes(a);
        h->last_launch[1] = a.kernel_variant == 1 || params->debug_flag != 0 ? tile_blocks : std::min(a.persistent_blocks, tile_blocks);
        h->last_launch[2] = a.lds_scene;
        h->last_launch[3] = (a.many_mesh ? 1u : 0u) | ((a.simple && !a.many_mesh && !a.count_tests) ? 2u : 0u) |
                            (a.kernel_variant == 1 ? 4u : 0u) | (rounds ? 8u : 0u);
    }
    auto& ev = h->ev_pool[h->ev_used++];
    if (pipe && pipe_barrier) {   // the tables this frame rewrote: later frames of the other stream wait for them
        HIP_TRY(h, hipEventRecord(h->pipe_book, S));
        h->pipe_book_set = true;
    }
    HIP_TRY(h, hipEventRecord(ev.first, S));
    if (wavefront) {
        if (int rc = wavefront_run(h, a, wf); rc != RT_OK) return rc;
    } else if (pipe) {
        // sample into this slot's scratch image on its stream; blend in frame order on the handle's stream
        a.image = h->pipe_scratch[pslot];
        a.batch_frames = 1;
        a.batch_stride = need_texels;
        a.batch_tile_major = 0;
        HIP_TRY(h, launch_render(a, S));
        HIP_TRY(h, hipEventRecord(h->pipe_sampled[pslot], S));
        h->pipe_sampled_set[pslot] = true;
        HIP_TRY(h, hipEventRecord(ev.second, S));
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->pipe_sampled[pslot], 0));
        BlendArgs b{};
        b.image = h->image;
        b.scratch = h->pipe_scratch[pslot];
        b.texels = need_texels;
        b.stride = need_texels;
        b.n = 1;
        b.frames0 = params->frames;
        b.weight[0] = a.blend_weight;   // (wgsl:157-158 with the host's two operations, as in a one-frame launch)
        b.rest[0] = a.blend_rest;
        HIP_TRY(h, launch_blend_frames(b, h->stream));
        HIP_TRY(h, hipEventRecord(h->pipe_blended[pslot], h->stream));
        h->pipe_blended_set[pslot] = true;
    } else if (!rounds) {
        HIP_TRY(h, launch_render(a, h->stream));
    } else {
        // launch 0 takes the tiles and parks every pixel in front of its first entry into the big mesh; then, round
        // after round, rt_walk_kernel walks the mesh for the parked rays and a render launch resumes those pixels
        // (parking them again at their next entry); the last render launch does not park: it walks what is left
        // inline.  Everything is ordered on the stream; a launch whose queue is empty ends at once.
        if (!h->park_counts) HIP_TRY(h, hipMalloc((void**)&h->park_counts, 72 * sizeof(uint32_t)));
        HIP_TRY(h, hipMemsetAsync(h->park_counts, 0, 72 * sizeof(uint32_t), h->stream));
        a.defer_mesh = h->defer_mesh;
        a.defer_xform = h->defer_xform;
#if RT_WALK2
        a.walk2 = h->walk2;
        a.walk2_base = h->walk2_base;
#endif
        auto fresh_counter = [&]() -> hipError_t {
            h->work_slot = (h->work_slot + 1) & 63u;
            if (h->work_slot == 0u) {
                hipError_t e = hipMemsetAsync(h->work_counters, 0, 64 * sizeof(uint32_t), h->stream);
                if (e != hipSuccess) return e;
            }
            a.work_counter = h->work_counters + h->work_slot;
            return hipSuccess;
        };
        const uint32_t R = n_rounds;
        a.park = 1;
        a.park_levels = (uint32_t)h->park_levels;
        a.q_in = nullptr;
        a.q_in_count = nullptr;
        a.q_out = h->park_queue[0];
        a.q_out_count = h->park_counts;
        // Hybrid launches: the render launches that PARK never walk the big mesh, so they run the LDS-scene kernels on
        // the small blob (everything but the big mesh's BVH and triangles) with the small meshes' stack depth; a winner on
        // the big mesh reads its shading record from the full blob.  The last launch (park = 0) and rt_walk_kernel
        // read the full blob as before.
        RenderArgs ah = a;
        bool hybrid = RT_EXPERIMENTS && h->hybrid != 0 && h->small_ok && !a.lds_scene && a.count_tests == 0 && a.kernel_variant == 0;
        if (hybrid) {
            ah.blob = h->small_blob;
            ah.lay = h->small_lay;
            ah.lds_scene = 1;
            ah.hybrid = 1;
            ah.big_blob = h->blob;
            ah.big_shade_off = h->lay.shade_off;
            ah.top_count = 0;
            ah.stack_entries = h->small_stack_entries ? h->small_stack_entries : 1u;
            ah.stack_wide = 0;
            ah.pixel_cache = h->pixel_cache_opt ? 1u : 0u;
            if (ah.pixel_cache && render_lds_bytes(ah) > LDS_BUDGET_BYTES) {
                // (the global-memory memo buffer; the plain launches may read theirs in place from the table -- pixel_cache 4 --,
                // which the LDS-scene kernels of the hybrid launches have not compiled in: a parked pixel's memo is rebuilt from
                // the table on resume either way, RenderArgs::primary_complete)
                ah.pixel_cache = (a.pixel_cache == 2u || a.pixel_cache == 4u) && h->pixel_cache_mem ? 2u : 0u;
                ah.pixel_cache_mem = h->pixel_cache_mem;
            }
            if (!ah.pixel_cache) ah.primary = nullptr;
            ah.fast_miss = h->fast_miss != 0 && ah.pixel_cache != 0u ? 1u : 0u;
            if (render_lds_bytes(ah) > LDS_BUDGET_BYTES) hybrid = false;
            ah.persistent_blocks = h->persistent_blocks;
            ah.park_levels = 0;  // (the small blob has no record of the big mesh's BVH)
        }
        if (!hybrid) ah = a;
        auto sync_queues = [&]() {  // (the queue fields move with the rounds: keep the hybrid copy's in step)
            ah.park = a.park; ah.q_in = a.q_in; ah.q_in_count = a.q_in_count; ah.q_out = a.q_out; ah.q_out_count = a.q_out_count;
            ah.work_counter = a.work_counter; ah.defer_mesh = a.defer_mesh; ah.defer_xform = a.defer_xform;
        };
        sync_queues();
        HIP_TRY(h, launch_render(ah, h->stream));
        for (uint32_t r = 0; r < R; ++r) {
            const bool last = r + 1 == R;
            a.q_in = h->park_queue[r & 1u];
            a.q_in_count = h->park_counts + r;
            HIP_TRY(h, fresh_counter());
            {
                RenderArgs w = a;
                w.top_count = 0;  // (nothing is staged into the walk kernel's LDS)
                HIP_TRY(h, launch_walk(w, h->compute_units, h->stream));
            }
            HIP_TRY(h, fresh_counter());
            a.park = last ? 0u : 1u;
            a.q_out = last ? nullptr : h->park_queue[(r + 1) & 1u];
            a.q_out_count = last ? nullptr : h->park_counts + r + 1;
            sync_queues();
            HIP_TRY(h, launch_render(last ? a : ah, h->stream));
        }
    }
    if (n_batch && !blend_later) {
        BlendArgs b{};
        b.image = h->image;
        b.scratch = h->batch_scratch;
        b.texels = need_texels;
        b.stride = need_texels;
        b.n = n_batch;
        b.frames0 = params->frames;
        for (uint32_t k = 0; k < n_batch; ++k) {  // wgsl:157-158 per frame: the same two IEEE operations
            volatile float w = 1.0f / (float)(int32_t)((uint32_t)params->frames + k + 1u);
            volatile float r = 1.0f - w;
            b.weight[k] = w;
            b.rest[k] = r;
        }
        HIP_TRY(h, launch_blend_frames(b, h->stream));
    }
    if (!pipe) {
        HIP_TRY(h, hipEventRecord(ev.second, h->stream));
        if (h->pipe_stream[0]) {   // (pipelined frames may follow: their streams wait for this launch)
            HIP_TRY(h, hipEventRecord(h->pipe_main, h->stream));
            h->pipe_main_set = true;
            h->pipe_main_need = !(took_idle_path && !wrote_tables);
        }
    }
    h->launches_total += 1;
    if (blend_later) {   // (frames rendered ahead: the call's own frame; the others count as asked for when their calls come)
        h->frames_total += 1u;
        h->frames_speculative += n_batch - 1u;
    } else {
        h->frames_total += n_batch ? n_batch : 1u;
    }
    if (params->debug_flag == 0 && params->rays_per_pixel > 0) {
        // pixels this call renders (strips are clipped to the image height)
        unsigned long long rows = 0;
        for (uint32_t st = rank; st < strips; st += world) {
            uint32_t y0 = st * 8, y1 = y0 + 8 < params->height ? y0 + 8 : params->height;
            rows += y1 - y0;
        }
        h->paths_total += rows * params->width * (unsigned long long)params->rays_per_pixel * (n_batch ? n_batch : 1u);
    }
    return RT_OK;
}

// n consecutive frames; batches of "batch_frames" frames per launch.  Debug views, single frames and
// frames without samples take the one-frame path (nothing to overlap).
static int render_single(rt_handle* h, const rt_params* params, uint32_t rank, uint32_t world);
static int render_frames_impl(rt_handle* h, const rt_params* params, uint32_t n_frames, uint32_t rank, uint32_t world) {
    if (!h || !params) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    if (n_frames == 1) return render_single(h, params, rank, world);  // (one frame per call: rt_render's path)
    rt_params p = *params;
    uint32_t done = 0;
    // batches of equal size (20 frames at 16 per launch: 10 + 10, not 16 + 4)
    uint32_t cap = (uint32_t)h->batch_frames_opt;
    if (h->max_device_bytes != 0 && world != 0 && rank < world) {   // (option max_device_mb: as many scratch frames as fit; 1 = plain launches)
        const uint64_t texels = world == 1 ? (uint64_t)params->width * params->height : rt_strip_texels(params->width, params->height, rank, world);
        const size_t other = optional_bytes(h) - h->batch_scratch_texels * sizeof(float4);
        const size_t room = h->max_device_bytes > other ? h->max_device_bytes - other : 0;
        const uint64_t fit = texels ? room / (texels * sizeof(float4)) : cap;
        if (fit < cap) cap = fit >= 2 ? (uint32_t)fit : 1u;
    }
    const uint32_t launches = (n_frames + cap - 1) / cap;
    const uint32_t per = launches ? (n_frames + launches - 1) / launches : 0;
    while (done < n_frames) {
        uint32_t n = n_frames - done;
        if (n > per) n = per;
        if (params->debug_flag != 0 || params->rays_per_pixel <= 0) n = 1;
        p.frames = (int32_t)((uint32_t)params->frames + done);
        const int rc = render_impl(h, &p, rank, world, n >= 2 ? n : 0u);
        if (rc != RT_OK) return rc;
        done += n;
    }
    return RT_OK;
}

// ---- one frame per call, frames rendered ahead (option "frame_ahead") ---------------------------------------------
static bool same_frame_params(const rt_params& a, const rt_params& b, int32_t b_frames) {  // (the padding is not compared)
    return a.width == b.width && a.height == b.height && a.number_of_bounces == b.number_of_bounces &&
           a.rays_per_pixel == b.rays_per_pixel && a.skybox == b.skybox && a.frames == b_frames && a.accumulate == b.accumulate &&
           a.debug_flag == b.debug_flag && a.debug_scale == b.debug_scale;
}

// How many frames a call that continues an accumulation renders at once (0: just its own).  Automatic: only when the
// host runs ahead of the device (`host_waits` false: the call found the stream busy) -- a host that waits for every frame
// gets its frame from a launch of its own, never behind frames it has not asked for --, and then only for scenes staged
// in LDS (rays of known cost) and shares so small that a launch of their own leaves lanes idle -- in units of a
// config-2 frame (1920 x 1080, 8 spp, 5 segments: 1.13 ms), batches of about 4 ms: 28 frames for a strip share of eight
// ranks (0.201 ms per frame pipelined -> 0.16), 14 for one of four (0.404 -> 0.30), 7 for one of two (0.647 -> 0.59),
// and none for the whole frame, whose pipelined launches (1.15 ms) a batch of three (1.2 ms) does not beat
// (tools/strip_scaling.py, profiles/r04_strip_scaling.txt).
static uint32_t ahead_depth(const rt_handle* h, const rt_params* params, uint64_t need_texels, bool host_waits) {
    if (params->debug_flag != 0 || params->rays_per_pixel <= 0 || h->count_tests != 0 || params->frames < 1) return 0;
    if (h->frame_ahead >= 0) return h->frame_ahead >= 2 ? std::min<uint32_t>((uint32_t)h->frame_ahead, RT_MAX_BATCH_FRAMES) : 0u;
    if (h->frame_ahead_failed || host_waits) return 0;
    const double segments = (double)need_texels * (double)params->rays_per_pixel *
                            (double)((params->number_of_bounces < 0 ? 0 : params->number_of_bounces) + 1);
    // (Scenes read from global memory: rays of unknown cost -- the rule of round 4, "up to 8 frames and about 33 ms by a
    // work estimate", guessed a frame time and needed a timing probe to take the guess back; a host that wants batches on
    // such a scene asks for them: frame_ahead = 8 gives config 5's geometry 4.43 -> 3.40 ms per call.)
    if (!h->lds_scene || h->force_global) return 0;
    const double ms = segments / (1920.0 * 1080.0 * 8.0 * 5.0) * 1.13;
    const double d = 4.0 / (ms > 1e-3 ? ms : 1e-3);
    const uint32_t n = d >= (double)RT_MAX_BATCH_FRAMES ? RT_MAX_BATCH_FRAMES : (uint32_t)d;
    return n >= 6u ? n : 0u;
}

// blend slot `slot` of the frames rendered ahead into the image: wgsl:157-158 with the frame's own weight
static int blend_ahead_slot(rt_handle* h, uint32_t slot, int32_t frames) {
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = wait_pending_copy(h); rc != RT_OK) return rc;
    BlendArgs b{};
    b.image = h->image;
    b.scratch = h->batch_scratch + (size_t)slot * h->ahead.texels;
    b.texels = h->ahead.texels;
    b.stride = h->ahead.texels;
    b.n = 1;
    b.frames0 = frames;
    {
        volatile float w = 1.0f / (float)(int32_t)((uint32_t)frames + 1u);
        volatile float r = 1.0f - w;
        b.weight[0] = w;
        b.rest[0] = r;
    }
    HIP_TRY(h, launch_blend_frames(b, h->stream));
    if (h->pipe_stream[0]) {  // (pipelined frames may follow: their blends are ordered behind this one)
        HIP_TRY(h, hipEventRecord(h->pipe_main, h->stream));
        h->pipe_main_set = true;
        h->pipe_main_need = true;
    }
    return RT_OK;
}

static int render_single(rt_handle* h, const rt_params* params, uint32_t rank, uint32_t world) {
    if (!h || !params) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    auto remember = [&]() {
        h->last_single.valid = true;
        h->last_single.params = *params;
        h->last_single.rank = rank;
        h->last_single.world = world;
        h->last_single.generation = h->generation;
    };
    if (h->ahead.valid) {
        const int32_t want = (int32_t)((uint32_t)h->ahead.base.frames + h->ahead.next);
        if (h->ahead.generation == h->generation && h->ahead.rank == rank && h->ahead.world == world &&
            same_frame_params(*params, h->ahead.base, want)) {
            const int rc = blend_ahead_slot(h, h->ahead.next, params->frames);
            if (rc != RT_OK) {
                h->ahead.valid = false;
                return rc;
            }
            h->ahead.next += 1;
            h->frames_total += 1;   // (a frame rendered ahead has now been asked for)
            if (h->frames_speculative > 0) h->frames_speculative -= 1;
            if (h->ahead.next == h->ahead.n) h->ahead.valid = false;
            remember();
            return RT_OK;
        }
        h->ahead.valid = false;  // the host went another way: what is left of the batch is dropped (it was never blended)
        h->ahead_ramp = 2;
    }
    const bool continues = h->last_single.valid && h->last_single.generation == h->generation &&
                           h->last_single.rank == rank && h->last_single.world == world &&
                           same_frame_params(*params, h->last_single.params, (int32_t)((uint32_t)h->last_single.params.frames + 1u));
    if (continues && h->have_scene && world != 0 && rank < world && params->width != 0 && params->height != 0) {
        const uint64_t need_texels = world == 1 ? (uint64_t)params->width * params->height
                                                : rt_strip_texels(params->width, params->height, rank, world);
        // (a host that WAITS for its frames finds the stream idle; one that runs ahead of the device finds it busy)
        const bool host_waits = h->frame_ahead < 0 && !(hipSetDevice(h->device) == hipSuccess && hipStreamQuery(h->stream) == hipErrorNotReady);
        (void)hipGetLastError();  // (hipErrorNotReady is an answer, not a failure: not for the launchers' hipGetLastError)
        uint32_t d = ahead_depth(h, params, need_texels, host_waits);
        if (h->frame_ahead < 0 && d >= 2) {
            // (room for the full depth at once: the batches on the way up would each re-allocate the scratch images -- a
            // stream synchronisation apiece)
            if (h->batch_scratch_texels < need_texels * d && fits_cap(h, need_texels * d * sizeof(float4), h->batch_scratch_texels * sizeof(float4)) &&
                hipSetDevice(h->device) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess) {
                float4* bigger = nullptr;
                if (hipMalloc((void**)&bigger, need_texels * d * sizeof(float4)) == hipSuccess) {
                    free_dev(h->batch_scratch);
                    h->batch_scratch = bigger;
                    h->batch_scratch_texels = need_texels * d;
                    h->scratch_w = 0;  // (new memory: render_impl zeroes it)
                } else {
                    (void)hipGetLastError();
                }
            }
            if (d > h->ahead_ramp) d = h->ahead_ramp;
            h->ahead_ramp = h->ahead_ramp >= RT_MAX_BATCH_FRAMES / 2 ? RT_MAX_BATCH_FRAMES : h->ahead_ramp * 2;
        }
        if (d >= 2 && need_texels <= h->image_texels) {
            int rc = render_impl(h, params, rank, world, d, true);
            if (rc == RT_OK) {
                h->ahead.valid = true;
                h->ahead.base = *params;
                h->ahead.n = d;
                h->ahead.next = 0;
                h->ahead.rank = rank;
                h->ahead.world = world;
                h->ahead.generation = h->generation;
                h->ahead.texels = need_texels;
                rc = blend_ahead_slot(h, 0, params->frames);
                h->ahead.next = 1;
                if (rc != RT_OK) h->ahead.valid = false;
                else remember();
                return rc;
            }
            // (no room for the batch: this handle renders its frames one by one from now on)
            if (h->frame_ahead < 0) h->frame_ahead_failed = true;
            (void)hipGetLastError();
        }
    }
    if (!continues) h->ahead_ramp = 2;
    const int rc = render_impl(h, params, rank, world);
    if (rc == RT_OK) remember();
    else h->last_single.valid = false;
    return rc;
}

int rt_render(rt_handle* h, const rt_params* params) { return render_single(h, params, 0, 1); }

int rt_render_frames(rt_handle* h, const rt_params* params, uint32_t n_frames) {
    return render_frames_impl(h, params, n_frames, 0, 1);
}

int rt_render_strips(rt_handle* h, const rt_params* params, uint32_t rank, uint32_t world) {
    return render_single(h, params, rank, world);
}

int rt_render_strips_frames(rt_handle* h, const rt_params* params, uint32_t n_frames, uint32_t rank, uint32_t world) {
    return render_frames_impl(h, params, n_frames, rank, world);
}

int rt_assemble_strips(rt_handle* h, const void* gathered_device, uint32_t width, uint32_t height, uint32_t world) {
    if (!h || !gathered_device || world == 0) return fail(h, RT_ERR_INVALID_ARGUMENT, "bad arguments");
    if ((uint64_t)width * height > h->image_texels)
        return fail(h, RT_ERR_CAPACITY, "image larger than the bound image buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = wait_pending_copy(h); rc != RT_OK) return rc;
    unsigned long long pad = rt_strip_texels(width, height, 0, world);
    HIP_TRY(h, launch_assemble((const float4*)gathered_device, h->image, width, height, world, pad, h->stream));
    return RT_OK;
}

// ---- multi-GPU frame in one process ------------------------------------------------------------
// Gather transport between distinct devices: RCCL (grouped ncclSend / ncclRecv = a gather: every peer
// sends over its own xGMI link to the root), loaded on first use -- librccl is ~570 MB and no other
// entry point needs it.
static int render_multi_impl(rt_handle** per_gpu, int n_gpus, const rt_params* params, uint32_t n_frames,
                             float* rgba32f_out) {
    if (!per_gpu || n_gpus < 1 || !params) return RT_ERR_INVALID_ARGUMENT;
    for (int r = 0; r < n_gpus; ++r)
        if (!per_gpu[r]) return RT_ERR_INVALID_ARGUMENT;
    rt_handle* root = per_gpu[0];
    if (n_frames == 0) return RT_OK;
    const uint32_t world = (uint32_t)n_gpus;
    const uint64_t pad = rt_strip_texels(params->width, params->height, 0, world);
    const uint64_t frame_texels = (uint64_t)params->width * params->height;
    // transport: RCCL between distinct devices (option "multi_rccl" = 1, default), device-to-device
    // copies when several handles share a device (tests) or when the option is 0; 2 forces RCCL on a
    // one-handle "node" (exercises the RCCL path on a single-GPU box)
    bool distinct = true;
    for (int r = 0; r < n_gpus; ++r)
        for (int q = 0; q < r; ++q)
            if (per_gpu[r]->device == per_gpu[q]->device) distinct = false;
    bool use_rccl = distinct && ((root->multi_rccl == 1 && n_gpus > 1) || root->multi_rccl == 2);
    if (root->multi_rccl == 2 && !distinct)
        return fail(root, RT_ERR_INVALID_ARGUMENT, "multi_rccl = 2 (RCCL always) needs one device per handle: a communicator cannot "
                                                   "hold two ranks of one device (multi_rccl = 1 uses device-to-device copies there)");
    // root-side staging, grown on demand
    HIP_TRY(root, hipSetDevice(root->device));
    if (root->multi_gathered_texels < pad * world) {
        HIP_TRY(root, hipStreamSynchronize(root->stream));
        free_dev(root->multi_gathered);
        HIP_TRY(root, hipMalloc((void**)&root->multi_gathered, pad * world * sizeof(float4)));
        root->multi_gathered_texels = pad * world;
    }
    if (root->multi_frame_texels < frame_texels) {
        HIP_TRY(root, hipStreamSynchronize(root->stream));
        free_dev(root->multi_frame);
        HIP_TRY(root, hipMalloc((void**)&root->multi_frame, frame_texels * sizeof(float4)));
        root->multi_frame_texels = frame_texels;
    }
    for (int r = 0; r < n_gpus; ++r) {  // every rank's "the root has copied my strips" event: of the root's device
        rt_handle* h = per_gpu[r];
        if (h->multi_copied && h->multi_copied_device == root->device) continue;
        if (h->multi_copied) {
            if (h->multi_copy_pending) (void)hipEventSynchronize(h->multi_copied);
            h->multi_copy_pending = false;
            (void)hipEventDestroy(h->multi_copied);
            h->multi_copied = nullptr;
        }
        HIP_TRY(root, hipEventCreateWithFlags(&h->multi_copied, hipEventDisableTiming));
        h->multi_copied_device = root->device;
    }
    if (use_rccl) {
        std::vector<int> devs(n_gpus);
        for (int r = 0; r < n_gpus; ++r) devs[r] = per_gpu[r]->device;
        if (root->multi_comm_devices != devs) {
            // Automatic mode (multi_rccl = 1) falls back to device-to-device copies when librccl cannot be loaded or
            // the communicators cannot be made (the reason stays in rt_last_error); multi_rccl = 2 insists.
            std::string why;
            if (!g_rccl.load()) {
                why = g_rccl.why;
            } else {
                multi_drop_comms(root);
                std::vector<ncclComm_t> comms(n_gpus);
                const ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_gpus, devs.data());
                if (r != ncclSuccess) {
                    why = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
                } else {
                    for (ncclComm_t c : comms) root->multi_comms.push_back((void*)c);
                    root->multi_comm_devices = devs;
                }
            }
            if (!why.empty()) {
                if (root->multi_rccl == 2) return fail(root, RT_ERR_DEVICE, why + " (option multi_rccl = 0 selects device-to-device copies)");
                root->err = "rt_render_multi: " + why + "; gathering with device-to-device copies instead";
                root->multi_rccl = 0;  // (do not try again on every frame)
                use_rccl = false;
            }
        }
    }
    if (!use_rccl) {
        // peer access for the device-to-device copies (without it the runtime stages them through the host)
        for (int r = 1; r < n_gpus; ++r) {
            const int peer = per_gpu[r]->device;
            if (peer == root->device || root->multi_peers_enabled.count(peer)) continue;
            int can = 0;
            HIP_TRY(root, hipDeviceCanAccessPeer(&can, root->device, peer));
            if (can) {
                hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                else if (e != hipSuccess) return fail(root, RT_ERR_DEVICE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            }
            root->multi_peers_enabled.insert(peer);
        }
    }
    // 1. every GPU renders its strips (asynchronous, in parallel across devices).  render_impl makes a
    //    rank's stream wait until the root has finished copying that rank's previous frame.
    for (int r = 0; r < n_gpus; ++r) {
        int rc = render_frames_impl(per_gpu[r], params, n_frames, (uint32_t)r, world);
        if (rc != RT_OK) {
            if (per_gpu[r] != root) root->err = "rank " + std::to_string(r) + ": " + per_gpu[r]->err;
            return rc;
        }
    }
    // 2. gather: one transfer per rank into the root's [world][pad] buffer
    auto rank_texels = [&](int r) {
        // world == 1: the "strips" are the frame itself (no padding of a ragged last strip)
        return world == 1 ? frame_texels : rt_strip_texels(params->width, params->height, (uint32_t)r, world);
    };
    if (use_rccl) {
        std::vector<GatherLeg> legs;
        for (int r = 0; r < n_gpus; ++r) {
            rt_handle* h = per_gpu[r];
            legs.push_back(GatherLeg{h->image, root->multi_gathered + (size_t)r * pad, (size_t)rank_texels(r) * 4, r, h->device,
                                     (ncclComm_t)root->multi_comms[r], h->stream});
        }
        auto set_device = [](int device, std::string& err) {
            const hipError_t e = hipSetDevice(device);
            if (e != hipSuccess && err.empty()) err = std::string("hipSetDevice: ") + hipGetErrorString(e);
            return e == hipSuccess ? 0 : 1;
        };
        std::string gerr;
        if (!rccl_grouped_gather(g_rccl, legs, root->device, (ncclComm_t)root->multi_comms[0], root->stream, set_device, gerr)) {
            // the group has been closed (rt_rccl.h); a half-issued gather may have left a send without its receive on
            // a stream: the communicators are aborted and made again by the next call
            multi_drop_comms(root, true);
            return fail(root, RT_ERR_DEVICE, "rt_render_multi gather: " + gerr);
        }
        // (a rank's send is ordered on its own stream before its next render: no extra event needed)
    } else {
        for (int r = 0; r < n_gpus; ++r) {
            rt_handle* h = per_gpu[r];
            const uint64_t n = rank_texels(r);
            if (n == 0) continue;
            HIP_TRY(h, hipSetDevice(h->device));
            if (!h->multi_event) HIP_TRY(h, hipEventCreateWithFlags(&h->multi_event, hipEventDisableTiming));
            HIP_TRY(h, hipEventRecord(h->multi_event, h->stream));
            HIP_TRY(root, hipSetDevice(root->device));
            HIP_TRY(root, hipStreamWaitEvent(root->stream, h->multi_event, 0));
            if (h->device == root->device)
                HIP_TRY(root, hipMemcpyAsync(root->multi_gathered + (size_t)r * pad, h->image, n * sizeof(float4),
                                             hipMemcpyDeviceToDevice, root->stream));
            else
                HIP_TRY(root, hipMemcpyPeerAsync(root->multi_gathered + (size_t)r * pad, root->device, h->image, h->device,
                                                 n * sizeof(float4), root->stream));
            if (h->stream != root->stream) {
                // the rank's next render overwrites h->image: it has to wait for this copy
                HIP_TRY(root, hipEventRecord(h->multi_copied, root->stream));
                h->multi_copy_pending = true;
            }
        }
    }
    // 3. assemble on the root and (optionally) read back
    HIP_TRY(root, hipSetDevice(root->device));
    HIP_TRY(root, launch_assemble(root->multi_gathered, root->multi_frame, params->width, params->height, world, pad,
                                  root->stream));
    if (rgba32f_out) {
        HIP_TRY(root, hipMemcpyAsync(rgba32f_out, root->multi_frame, frame_texels * sizeof(float4),
                                     hipMemcpyDeviceToHost, root->stream));
        HIP_TRY(root, hipStreamSynchronize(root->stream));
    }
    return RT_OK;
}

int rt_render_multi(rt_handle** per_gpu, int n_gpus, const rt_params* params, float* rgba32f_out) {
    return render_multi_impl(per_gpu, n_gpus, params, 1, rgba32f_out);
}

int rt_render_multi_frames(rt_handle** per_gpu, int n_gpus, const rt_params* params, uint32_t n_frames, float* rgba32f_out) {
    return render_multi_impl(per_gpu, n_gpus, params, n_frames, rgba32f_out);
}

int rt_read_multi_frame(rt_handle* root, float* rgba32f_out, size_t bytes) {
    if (!root || !rgba32f_out) return fail(root, RT_ERR_INVALID_ARGUMENT, "null argument");
    if (!root->multi_frame || bytes > root->multi_frame_texels * sizeof(float4))
        return fail(root, RT_ERR_INVALID_ARGUMENT, "no assembled frame of that size");
    HIP_TRY(root, hipSetDevice(root->device));
    HIP_TRY(root, hipMemcpyAsync(rgba32f_out, root->multi_frame, bytes, hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(root, hipStreamSynchronize(root->stream));
    return RT_OK;
}

int rt_synchronize(rt_handle* h) {
    if (!h) return RT_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RT_OK;
}

int rt_read_image(rt_handle* h, float* out, size_t bytes) {
    if (!h || !out) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes > h->image_texels * sizeof(float4))
        return fail(h, RT_ERR_INVALID_ARGUMENT, "read larger than the image");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = wait_pending_copy(h); rc != RT_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(out, h->image, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RT_OK;
}

// The frame of the last call, kept aside in stream order (a device-to-device copy, ~15 us for 33 MB): the handle's stream
// goes on rendering, rt_read_snapshot brings the copy to the host on a stream of its own (the copy engines run beside the
// kernels), so that a host that shows every frame pays max(render, read) per frame instead of their sum.
int rt_snapshot_image(rt_handle* h, size_t bytes) {
    if (!h) return RT_ERR_INVALID_ARGUMENT;
    if (bytes == 0 || bytes > h->image_texels * sizeof(float4))
        return fail(h, RT_ERR_INVALID_ARGUMENT, "snapshot empty or larger than the image");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = wait_pending_copy(h); rc != RT_OK) return rc;
    // (each object on its own: a creation that failed half-way must not leave later calls with null events)
    if (!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!h->snapshot_taken) HIP_TRY(h, hipEventCreateWithFlags(&h->snapshot_taken, hipEventDisableTiming));
    if (!h->snapshot_read) HIP_TRY(h, hipEventCreateWithFlags(&h->snapshot_read, hipEventDisableTiming));
    if (h->snapshot_capacity < bytes) {
        HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        free_dev(h->snapshot);
        h->snapshot_capacity = h->snapshot_bytes = 0;
        HIP_TRY(h, hipMalloc((void**)&h->snapshot, bytes));
        h->snapshot_capacity = bytes;
        h->snapshot_read_pending = false;
    }
    // (a read of the previous snapshot that is still under way keeps its bytes: the new copy waits for it)
    if (h->snapshot_read_pending) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->snapshot_read, 0));
    h->snapshot_read_pending = false;
    HIP_TRY(h, hipMemcpyAsync(h->snapshot, h->image, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipEventRecord(h->snapshot_taken, h->stream));
    h->snapshot_bytes = bytes;
    return RT_OK;
}

// Blocking read of the last snapshot (not of the image: frames rendered since the snapshot are not waited for).
int rt_read_snapshot(rt_handle* h, float* out, size_t bytes) {
    if (!h || !out) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    if (h->snapshot_bytes == 0) return fail(h, RT_ERR_INVALID_ARGUMENT, "rt_snapshot_image has not been called");
    if (bytes > h->snapshot_bytes) return fail(h, RT_ERR_INVALID_ARGUMENT, "read larger than the snapshot");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->snapshot_taken, 0));
    HIP_TRY(h, hipMemcpyAsync(out, h->snapshot, bytes, hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(h, hipEventRecord(h->snapshot_read, h->copy_stream));
    h->snapshot_read_pending = true;
    HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
    return RT_OK;
}

int rt_write_image(rt_handle* h, const float* in, size_t bytes) {
    if (!h || !in) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes > h->image_texels * sizeof(float4))
        return fail(h, RT_ERR_INVALID_ARGUMENT, "write larger than the image");
    h->generation += 1;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = wait_pending_copy(h); rc != RT_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->image, in, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RT_OK;
}

int rt_get_stats(rt_handle* h, rt_stats* out) {
    if (!h || !out) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    Counters c{};
    // (on the handle's stream, never the null stream: a process that has touched the null stream holds one more hardware
    // queue, and with main + three pipeline streams + null the pipelined frames' streams start sharing queues -- measured:
    // 1.30 -> 1.42 ms per frame after the first rt_get_stats)
    HIP_TRY(h, hipMemcpyAsync(&c, h->counters, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    memset(out, 0, sizeof(*out));
    out->segments = c.segments;
    out->paths = h->paths_total;
    out->node_tests = c.node_tests;
    out->triangle_tests = c.triangle_tests;
    out->segments_reused = c.reused;
    // counters accumulate over all launches since rt_reset_timing
    double total = h->ev_ms_harvested;
    HIP_TRY(h, harvest_event_times(h, total));
    out->kernel_ms = (float)total;
    out->launches = (uint32_t)h->launches_total;
    out->frames = (uint32_t)h->frames_total;
    out->frames_speculative = (uint32_t)h->frames_speculative;
    return RT_OK;
}

int rt_last_launch(rt_handle* h, uint32_t out[6]) {
    if (!h || !out) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    for (int k = 0; k < 4; ++k) out[k] = h->last_launch[k];
    out[4] = (uint32_t)((optional_bytes(h) + (1u << 20) - 1) >> 20);   // MiB held on the library's own initiative (option max_device_mb)
    out[5] = (uint32_t)(h->max_device_bytes >> 20);                     // the cap (0 = none)
    return RT_OK;
}

int rt_reset_timing(rt_handle* h) {
    if (!h) return RT_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemsetAsync(h->counters, 0, sizeof(Counters), h->stream));
    // (a pipelined frame's sampling launch runs on an internal stream that is not ordered behind this stream's
    // memset: the zeroing is complete before the call returns, so no later launch's counter adds can race with it)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->ev_used = 0;
    h->ev_ms_harvested = 0.0;
    h->launches_total = h->frames_total = h->frames_speculative = 0;
    h->paths_total = 0;
    return RT_OK;
}

int rt_set_stream(rt_handle* h, void* hip_stream) {
    if (!h) return RT_ERR_INVALID_ARGUMENT;
    h->generation += 1;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = wait_pending_copy(h); rc != RT_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, harvest_event_times(h, h->ev_ms_harvested));  // (the old stream is idle: its recorded times are final)
    h->ev_used = 0;
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return RT_OK;
}

int rt_bind_image(rt_handle* h, void* device_ptr, uint64_t texels) {
    if (!h) return RT_ERR_INVALID_ARGUMENT;
    h->generation += 1;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = wait_pending_copy(h); rc != RT_OK) return rc;  // (the caller may free the old buffer after this call)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (device_ptr) {
        h->image = (float4*)device_ptr;
        h->image_texels = texels;
    } else {
        h->image = h->own_image;
        h->image_texels = (size_t)h->max_width * h->max_height;
    }
    return RT_OK;
}

#if defined(RT_DIAG) || defined(RT_WAVE_TIMES)
int rt_diag_wave_times(rt_handle* h, unsigned long long* out) {
    if (!h || !out) return RT_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, rtd::diag_wave_times(out));
    return RT_OK;
}
#endif
#if defined(RT_DIAG) || defined(RT_DIAGT)
int rt_diag_read(rt_handle* h, unsigned long long* out64, int reset) {
    if (!h || !out64) return RT_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, rtd::diag_read(out64, reset != 0));
    return RT_OK;
}
#endif

#if RT_TEST_ENTRIES
// The test entry points (include/rt_test_abi.h) are NOT part of the product library: they are compiled only with
// -DRT_TEST_ENTRIES=1, into ray_tracer_2_amd/librt2_mi355x_test.so (ray_tracer_2_amd/build.py: build_test_library) --
// the same sources and flags otherwise.
// tests/test_gpu_device_units.py: evaluate the kernels' arithmetic building blocks on the device, element-wise over host arrays.
int rt_test_device_units(rt_handle* h, int fn, const float* x, const float* y, float* out, uint64_t n) {
    if (!h || !x || !y || !out) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)(n ? n : 1) * sizeof(float);
    HIP_TRY(h, hipMalloc((void**)&dx, bytes));
    HIP_TRY(h, hipMalloc((void**)&dy, bytes));
    HIP_TRY(h, hipMalloc((void**)&dout, bytes));
    HIP_TRY(h, hipMemcpyAsync(dx, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dy, y, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, launch_units(fn, dx, dy, dout, n, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out, dout, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_dev(dx);
    free_dev(dy);
    free_dev(dout);
    return RT_OK;
}

int rt_test_sweep(rt_handle* h, int which, uint64_t* out3) {
    if (!h || !out3 || which < 0 || which > 4) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned long long* d = nullptr;
    HIP_TRY(h, hipMalloc((void**)&d, 3 * sizeof(unsigned long long)));
    HIP_TRY(h, hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), h->stream));
    HIP_TRY(h, launch_sweep(which, d, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out3, d, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_dev(d);
    return RT_OK;
}

int rt_test_device_sample_texture(rt_handle* h, const rt_texture_desc* tex, const float* uv, float* rgba_out, uint64_t n) {
    if (!h || !tex || !tex->rgba8 || !uv || !rgba_out || tex->width == 0 || tex->height == 0)
        return fail(h, RT_ERR_INVALID_ARGUMENT, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    uint8_t* dt = nullptr;
    float *duv = nullptr, *dout = nullptr;
    const size_t tb = (size_t)tex->width * tex->height * 4;
    HIP_TRY(h, hipMalloc((void**)&dt, tb));
    HIP_TRY(h, hipMalloc((void**)&duv, (size_t)(n ? n : 1) * 2 * sizeof(float)));
    HIP_TRY(h, hipMalloc((void**)&dout, (size_t)(n ? n : 1) * 4 * sizeof(float)));
    HIP_TRY(h, hipMemcpyAsync(dt, tex->rgba8, tb, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(duv, uv, n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, launch_units_texture(dt, tex->width, tex->height, h->srgb_lut, duv, dout, n, h->stream));
    HIP_TRY(h, hipMemcpyAsync(rgba_out, dout, n * 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_dev(dt);
    free_dev(duv);
    free_dev(dout);
    return RT_OK;
}

// Test-only: raw copies of the wavefront sequence's buffers after the last launch (0 path state, 1 hit records,
// 2 the two slot lists, 3 the per-round list counts), for tests/tools that check the sequence slot by slot.
int rt_test_read_wavefront(rt_handle* h, int which, void* out, uint64_t bytes) {
    if (!h || !out) return fail(h, RT_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t blocks = h->wf_capacity / 64;
    const void* src = nullptr;
    size_t have = 0;
    switch (which) {
        case 0: src = h->wf_state; have = blocks * WF_STATE_PLANES * 64 * sizeof(float4); break;
        case 1: src = h->wf_hit; have = blocks * WF_HIT_PLANES * 64 * sizeof(float4); break;
        case 2: src = h->wf_lists; have = 2 * h->wf_capacity * sizeof(uint32_t); break;
        case 3: src = h->wf_counts; have = h->wf_counts_capacity * sizeof(uint32_t); break;
        case 4: src = h->park_counts; have = h->park_counts ? 72 * sizeof(uint32_t) : 0; break;
        case 5: case 6:   // the park records of the last sequence's even / odd rounds (rt_device.h: PARK_PLANES)
            src = h->park_queue[which - 5];
            have = src ? ((h->park_capacity + 63) / 64) * (size_t)PARK_PLANES * 64u * sizeof(float4) : 0;
            break;
        default: return fail(h, RT_ERR_INVALID_ARGUMENT, "which must be 0..6");
    }
    if (!src || bytes > have) return fail(h, RT_ERR_INVALID_ARGUMENT, "no wavefront buffer of that size");
    HIP_TRY(h, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RT_OK;
}

// Test-only, no GPU needed: the automatic depth of option "frame_ahead" (ahead_depth) for a one-frame call that continues an
// accumulation -- scene staged in LDS or read from global memory, texels of the call's share, samples per pixel, bounces,
// and whether the host counts as one that waits for its frames.  tests/test_frame_ahead_policy.py pins the documented
// figures with it.
int rt_test_frame_ahead_depth(int lds_scene, uint64_t texels, int rays_per_pixel, int number_of_bounces, int host_waits) {
    rt_handle h;
    h.lds_scene = lds_scene != 0;
    rt_params p{};
    p.rays_per_pixel = rays_per_pixel;
    p.number_of_bounces = number_of_bounces;
    p.frames = 1;
    return (int)ahead_depth(&h, &p, texels, host_waits != 0);
}

// Test-only: the grouped gather of rt_render_multi against the RCCL-shaped library at `lib_path`, on fake buffers and
// with no HIP call (so that it runs on a machine without a GPU): n_ranks communicators from ncclCommInitAll, one
// rccl_grouped_gather over them; on failure the communicators are aborted, as render_multi_impl does.  The error text
// goes to rt_last_error(NULL).  tests/test_rccl_group.py drives it against tests/cpp/fake_rccl.c.
int rt_test_rccl_gather(const char* lib_path, int n_ranks) {
    if (!lib_path || n_ranks < 1 || n_ranks > 64) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "bad arguments");
    RcclApi api;
    if (!api.load(lib_path)) return fail(nullptr, RT_ERR_IO, api.why);
    std::vector<ncclComm_t> comms((size_t)n_ranks);
    std::vector<int> devs((size_t)n_ranks);
    for (int r = 0; r < n_ranks; ++r) devs[(size_t)r] = r;
    ncclResult_t r0 = api.CommInitAll(comms.data(), n_ranks, devs.data());
    if (r0 != ncclSuccess) {
        const std::string why = std::string("ncclCommInitAll: ") + api.GetErrorString(r0);
        api.unload();
        return fail(nullptr, RT_ERR_DEVICE, why);
    }
    static float fake_send[64][4], fake_recv[64][4];
    std::vector<GatherLeg> legs;
    for (int r = 0; r < n_ranks; ++r)
        legs.push_back(GatherLeg{fake_send[r], fake_recv[r], 4, r, r, comms[(size_t)r], nullptr});
    std::string gerr;
    const bool ok = rccl_grouped_gather(api, legs, 0, comms[0], nullptr, [](int, std::string&) { return 0; }, gerr);
    for (ncclComm_t c : comms) {
        if (ok) (void)api.CommDestroy(c);
        else api.drop(c);
    }
    api.unload();
    return ok ? RT_OK : fail(nullptr, RT_ERR_DEVICE, "rt_render_multi gather: " + gerr);
}
#endif  // RT_TEST_ENTRIES

void* rt_device_image(rt_handle* h) { return h ? (void*)h->image : nullptr; }
void* rt_stream(rt_handle* h) { return h ? (void*)h->stream : nullptr; }

}  // extern "C"
