// Drives the C++ mirror of the reference's RayTracer (csrc/host/ray_tracer.hpp ≙
// src/rendering/ray_tracer.rs:32-435) and FrameParams (≙ Params, src/core/app.rs:27-91) the way the
// reference's App does: new/create_gpu_resources once, load_scene_gpu_resources + update_buffers when a
// scene arrives (app.rs:135-142, 160-162), then per redraw Params::update + render (app.rs:300-313), and
// finally the texture read-back of save_render_to_file (app.rs:341-407).
//
// Built by __graft_entry__.build() into tests/_build/librt2_class_driver.so and called in-process from
// tests/test_gpu_cpp_class.py: on this GPU pool a process that has initialised the GPU must not exec
// another program, so the "test program" is a library with one entry point instead of an executable.
#include <cstdio>
#include <vector>
#include <cstring>
#include <string>

#include "../../ray_tracer_2_amd/csrc/host/ray_tracer.hpp"

#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::snprintf(err, err_len, "line %d: %s", __LINE__, #cond);       \
            return -__LINE__;                                                  \
        }                                                                      \
    } while (0)

extern "C" int rt2_class_driver(const char* scene_name, const char* assets_dir, uint32_t width, uint32_t height,
                                int bounces, int spp, int n_frames, float* rgba32f_out, unsigned long long* segments_out,
                                char* err, size_t err_len) {
    using namespace rt2;
    // ---- Params::default / update / reset_frame (app.rs:42-91) ----
    FrameParams p;
    CHECK(p.width == 1920 && p.height == 1080 && p.number_of_bounces == 5 && p.rays_per_pixel == 1);
    CHECK(p.skybox == 0 && p.frames == 0 && p.accumulate == 1 && p.debug_flag == 0 && p.debug_scale == 0);
    CHECK(sizeof(FrameParams) == 48 && sizeof(rt_params) == 48);
    CHECK(p.update(false) == false && p.frames == 1);           // accumulating: the counter advances
    CHECK(p.update(true) == true && p.frames == -1);            // moving: reset_frame
    CHECK(p.update(false) == false && p.frames == 0);           // the first still frame stores (frames = 0)
    p.accumulate = 0;
    CHECK(p.update(false) == true && p.frames == -1);           // accumulation off: reset every frame
    p.accumulate = 1;
    p.reset_frame();
    CHECK(p.frames == -1);

    // ---- a scene from the library (scene.rs:1003) ----
    Scene scene;
    std::string why;
    if (!load_builtin_scene(scene_name, assets_dir, ImageDecoder(decode_png_file), scene, why)) {
        std::snprintf(err, err_len, "load_builtin_scene: %s", why.c_str());
        return -1;
    }
    scene.built_bvh = false;   // let update_buffers build it, as Scene::bvh_nodes does lazily (scene.rs:272-278)

    RayTracer tracer;
    CHECK(tracer.render(p) == RT_ERR_INVALID_ARGUMENT);         // no device objects yet: a status, not a panic
    int rc = tracer.create_gpu_resources(0, width, height);
    if (rc != RT_OK) {
        std::snprintf(err, err_len, "create_gpu_resources: %s", tracer.last_error());
        return rc;
    }
    CHECK(tracer.render(p) == RT_ERR_NO_SCENE);
    CHECK(tracer.load_scene_gpu_resources(scene) == RT_OK);
    CHECK(tracer.update_buffers(scene) == RT_OK);
    CHECK(scene.built_bvh || scene.meshes.empty());

    p.width = width;
    p.height = height;
    p.number_of_bounces = bounces;
    p.rays_per_pixel = spp;
    p.skybox = 1;
    for (int f = 0; f < n_frames; ++f) {
        p.update(false);                                        // frames = 0, 1, 2, ...
        CHECK(p.frames == f);
        rc = tracer.render(p);
        if (rc != RT_OK) {
            std::snprintf(err, err_len, "render: %s", tracer.last_error());
            return rc;
        }
    }
    CHECK(tracer.read_image(rgba32f_out, (size_t)width * height * 16) == RT_OK);
    {   // the display path returns the same frame
        std::vector<float> snap((size_t)width * height * 4);
        CHECK(tracer.snapshot_image(snap.size() * 4) == RT_OK);
        CHECK(tracer.read_snapshot(snap.data(), snap.size() * 4) == RT_OK);
        CHECK(std::memcmp(snap.data(), rgba32f_out, snap.size() * 4) == 0);
    }
    rt_stats st;
    CHECK(tracer.stats(&st) == RT_OK);
    // (a call that continues an accumulation may render the next frames with its own: option "frame_ahead")
    CHECK(st.launches >= 1 && st.launches <= (uint32_t)n_frames && st.frames >= (uint32_t)n_frames);
    *segments_out = st.segments;
    return RT_OK;
}
