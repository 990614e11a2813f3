// ray_tracer.cpp -- see ray_tracer.hpp.
#include "ray_tracer.hpp"

#include <vector>

struct rt_scene;  // defined in scene_capi.cpp
namespace rt2 {
const Scene& scene_of(const rt_scene* s);
}

namespace rt2 {

FrameParams::FrameParams() {  // app.rs:76-91
    width = 1920;
    height = 1080;
    number_of_bounces = 5;
    rays_per_pixel = 1;
    skybox = 0;
    frames = 0;
    accumulate = 1;
    debug_flag = 0;
    debug_scale = 0;
    _p1[0] = _p1[1] = _p1[2] = 0.0f;
}

bool FrameParams::update(bool is_moving) {
    if (is_moving) {
        reset_frame();
        return true;
    }
    if (accumulate == 1) {
        frames += 1;
        return false;
    }
    reset_frame();
    return true;
}

void FrameParams::reset_frame() { frames = -1; }

RayTracer::~RayTracer() {
    if (h_) rt_destroy(h_);
}

int RayTracer::create_gpu_resources(int device_ordinal, uint32_t max_width, uint32_t max_height) {
    if (h_) {
        rt_destroy(h_);
        h_ = nullptr;
    }
    return rt_create(device_ordinal, max_width, max_height, &h_);
}

static int upload_textures(rt_handle* h, const Scene& scene) {
    std::vector<rt_texture_desc> descs(scene.textures.size());
    for (size_t i = 0; i < descs.size(); ++i)
        descs[i] = rt_texture_desc{scene.textures[i].rgba.data(), scene.textures[i].width,
                                   scene.textures[i].height};
    return rt_upload_textures(h, descs.data(), (uint32_t)descs.size());
}

static int upload_arrays(rt_handle* h, const Scene& scene) {
    rt_scene_uniform u = scene.to_uniform();
    return rt_upload_scene(h, &u, scene.spheres.data(), (uint32_t)scene.spheres.size(),
                           scene.mesh_uniforms.data(), (uint32_t)scene.mesh_uniforms.size(),
                           scene.triangles.data(), (uint32_t)scene.triangles.size(),
                           scene.nodes.data(), (uint32_t)scene.nodes.size());
}

int RayTracer::load_scene_gpu_resources(const Scene& scene) {
    if (!h_) return RT_ERR_INVALID_ARGUMENT;
    return upload_textures(h_, scene);
}

int RayTracer::update_buffers(Scene& scene) {
    if (!h_) return RT_ERR_INVALID_ARGUMENT;
    if (!scene.built_bvh && !scene.meshes.empty()) scene.build_per_mesh(Quality::High);  // scene.rs:272-278
    return upload_arrays(h_, scene);
}

int RayTracer::render(const rt_params& params) { return h_ ? rt_render(h_, &params) : RT_ERR_INVALID_ARGUMENT; }
int RayTracer::read_image(float* rgba, size_t bytes) { return h_ ? rt_read_image(h_, rgba, bytes) : RT_ERR_INVALID_ARGUMENT; }
int RayTracer::snapshot_image(size_t bytes) { return h_ ? rt_snapshot_image(h_, bytes) : RT_ERR_INVALID_ARGUMENT; }
int RayTracer::read_snapshot(float* rgba, size_t bytes) { return h_ ? rt_read_snapshot(h_, rgba, bytes) : RT_ERR_INVALID_ARGUMENT; }
int RayTracer::stats(rt_stats* out) { return h_ ? rt_get_stats(h_, out) : RT_ERR_INVALID_ARGUMENT; }
const char* RayTracer::last_error() const { return rt_last_error(h_); }

}  // namespace rt2

extern "C" int rt_upload_built_scene(rt_handle* h, const rt_scene* s) {
    if (!h || !s) return RT_ERR_INVALID_ARGUMENT;
    const rt2::Scene& scene = rt2::scene_of(s);
    int rc = rt2::upload_textures(h, scene);
    if (rc != RT_OK) return rc;
    return rt2::upload_arrays(h, scene);
}
