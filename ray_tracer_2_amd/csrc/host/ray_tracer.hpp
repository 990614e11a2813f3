// ray_tracer.hpp -- C++ host-side mirror of the reference's `RayTracer`
// (src/rendering/ray_tracer.rs:32-435), written above the C ABI of
// include/rt_abi.h because no Rust toolchain exists in the build image
// (INTEGRATION.md shows the Rust binding).  Same method names, same order of
// use, same capacity limits; errors are rt_* status codes (the reference
// panics), readable through last_error().
#ifndef RT_RAY_TRACER_HPP
#define RT_RAY_TRACER_HPP

#include <string>

#include "../../../include/rt_abi.h"
#include "scene.h"

namespace rt2 {

// ≙ Params::{update, reset_frame, default} (src/core/app.rs:42-91): the frame
// counter is the seed of the per-pixel RNG streams (wgsl:475).
struct FrameParams : rt_params {
    FrameParams();
    bool update(bool is_moving);  // app.rs:43-54
    void reset_frame();           // app.rs:55-57
};

class RayTracer {
   public:
    RayTracer() = default;
    ~RayTracer();
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    // ≙ RayTracer::new (ray_tracer.rs:49) + create_gpu_resources (:316): the
    // device objects (stream, RGBA32F image of max_width x max_height).
    int create_gpu_resources(int device_ordinal, uint32_t max_width, uint32_t max_height);
    // ≙ load_scene_gpu_resources (:237): textures of a freshly loaded scene.
    int load_scene_gpu_resources(const Scene& scene);
    // ≙ update_buffers (:397): scene arrays + SceneUniform.  Call on change;
    // rebuilds the BVH lazily like Scene::bvh_nodes (scene.rs:272-278).
    int update_buffers(Scene& scene);
    // ≙ render (:420): one frame, asynchronous.
    int render(const rt_params& params);
    // ≙ the texture->buffer copy of save_render_to_file (app.rs:341-407).
    int read_image(float* rgba32f, size_t bytes);
    // the display path (include/rt_abi.h): keep the frame just rendered aside, fetch it after the next render was queued
    int snapshot_image(size_t bytes);
    int read_snapshot(float* rgba32f, size_t bytes);
    int stats(rt_stats* out);
    const char* last_error() const;
    rt_handle* handle() const { return h_; }

   private:
    rt_handle* h_ = nullptr;
};

}  // namespace rt2

#endif
