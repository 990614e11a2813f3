// host_sanitizer_driver.cpp -- TEST INFRASTRUCTURE (tests/test_host_malformed_inputs.py): the host loaders
// (csrc/host/obj_loader.cpp, png_decode.cpp, scene.cpp, bvh.cpp) behind the C API of csrc/host/scene_capi.cpp, built
// with AddressSanitizer + UndefinedBehaviorSanitizer on the CPU, fed one file per line of a manifest:
//     <mode> <path>        mode = obj (rt_scene_add_obj with its MTL, then rt_scene_build) | png (decode_png_file)
// Prints "BEGIN <path>" before and "END <rc>" after each file, so that a sanitizer abort names its input.
// The reference panics on a file it cannot read (src/core/asset.rs:72-75,118); here every malformed input has to come
// back as an error code.
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>

#include "../../include/rt_abi.h"
#include "../../ray_tracer_2_amd/csrc/host/scene.h"

namespace rt2 {
// (the device-side plane search lives in csrc/rt_bvh_search.hip; not part of this CPU build)
LevelSearch make_device_level_search(int, const float*, size_t) { throw std::runtime_error("HIP: no device in the sanitizer build"); }
}  // namespace rt2

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream manifest(argv[1]);
    std::string mode, path;
    while (manifest >> mode && std::getline(manifest >> std::ws, path)) {
        std::printf("BEGIN %s\n", path.c_str());
        std::fflush(stdout);
        int rc = 0;
        if (mode == "obj") {
            rt_scene* s = nullptr;
            rc = rt_scene_create(&s);
            if (rc == RT_OK) {
                const size_t slash = path.find_last_of('/');
                const std::string dir = slash == std::string::npos ? "." : path.substr(0, slash);
                const std::string file = slash == std::string::npos ? path : path.substr(slash + 1);
                rc = rt_scene_add_obj(s, dir.c_str(), file.c_str(), nullptr, 1, nullptr);
                if (rc == RT_OK) rc = rt_scene_build(s, 2);
                if (rc == RT_OK) {
                    // what an upload would read: every array through its accessor
                    volatile uint32_t sink = rt_scene_num_meshes(s) + rt_scene_num_triangles(s) + rt_scene_num_nodes(s) + rt_scene_num_textures(s);
                    (void)sink;
                }
                rt_scene_destroy(s);
            }
        } else if (mode == "png") {
            rt2::Image img;
            rc = rt2::decode_png_file(path, img) ? RT_OK : RT_ERR_IO;
            if (rc == RT_OK && img.rgba.size() != (size_t)img.width * img.height * 4) rc = -100;  // (a decoder that lies about its output)
        } else {
            rc = -101;
        }
        std::printf("END %d\n", rc);
        std::fflush(stdout);
    }
    return 0;
}
