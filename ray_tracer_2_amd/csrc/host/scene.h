// scene.h -- host-side scene model mirroring src/scene/** of the reference:
// SceneDefinition -> Scene (spheres, mesh instances, camera, textures) ->
// the POD arrays of include/rt_abi.h.  Names follow the reference.
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "bvh.h"

namespace rt2 {

// ≙ MaterialUniform::default() (material.rs:19-36)
rt_material material_uniform_default();
// ≙ MaterialDefinition::default() / ::new() (material.rs:75-115) as uniforms
rt_material material_definition_default();
rt_material material_definition_new();

struct Image {  // ≙ image::RgbaImage
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;
};

struct MeshData {  // mesh.rs:8-12
    std::vector<Vertex> vertices;
    std::vector<uint32_t> indices;
};

struct MeshInstance {  // mesh.rs:14-20
    std::string label;
    std::shared_ptr<MeshData> data;
    Transform transform;
    rt_material material;
};

struct Camera {  // camera.rs:24-35 (controller omitted: UI)
    Transform transform;
    float fov = 90.0f, aspect = 16.0f / 9.0f, near_plane = 0.01f, far_plane = 1000.0f,
          focus_dist = 1.0f, defocus_strength = 0.0f, diverge_strength = 0.0f;
    rt_camera_uniform to_uniform() const;  // camera.rs:81-91
};

// Decodes an image file to RGBA8 (no flip).  Installed by the embedding host
// (the reference uses the `image` crate, asset.rs:77); the built-in decoder
// handles PNG.
typedef std::function<bool(const std::string& path, Image& out)> ImageDecoder;

class AssetManager {  // src/core/asset.rs:25-30
   public:
    explicit AssetManager(std::string assets_dir) : assets_dir_(std::move(assets_dir)) {}
    // ≙ load_texture (asset.rs:60-85): decode, flip horizontally, next index
    int load_texture(const std::string& path, std::string& err);
    int add_texture(Image img, const std::string& key);
    // ≙ load_model_with_material (asset.rs:86-100)
    bool load_model(const std::string& path, const Transform& t, bool use_mtl,
                    const rt_material& material, std::vector<MeshInstance>& out, std::string& err);
    const std::vector<Image>& textures() const { return textures_; }
    ImageDecoder decoder;
    const std::string& assets_dir() const { return assets_dir_; }

   private:
    std::string assets_dir_;
    std::vector<std::string> texture_keys_;
    std::vector<Image> textures_;
};

class Scene {  // scene.rs:148-156
   public:
    Camera camera;
    std::vector<rt_sphere> spheres;
    std::vector<MeshInstance> meshes;
    // bvh_data (bvh.rs:110-115)
    std::vector<rt_packed_triangle> triangles;
    std::vector<rt_node> nodes;
    std::vector<rt_mesh_uniform> mesh_uniforms;
    std::vector<Image> textures;
    bool built_bvh = false;

    // ≙ BVH::build_per_mesh (bvh.rs:152-207)
    // device >= 0: meshes of at least `device_min_tris` triangles have their SAH searches done on that
    // GPU (csrc/rt_bvh_search.hip); the result is the same, bit for bit
    void build_per_mesh(Quality q, int device = -2, size_t device_min_tris = 16384);  // -2: plain bvh_build
    // ≙ Scene::to_uniform (scene.rs:985-1001)
    rt_scene_uniform to_uniform() const;
    // n x n barycentric split of every mesh triangle (stand-in geometry)
    void subdivide_meshes(uint32_t n);
};

// ≙ Scene::from_name + instantiate_scene for the built-in library
// (scene.rs:280-983).  Returns false and sets err on failure.
bool load_builtin_scene(const std::string& name, const std::string& assets_dir,
                        const ImageDecoder& decoder, Scene& out, std::string& err);

bool decode_png_file(const std::string& path, Image& out);
void flip_horizontal(Image& img);

}  // namespace rt2

#endif
