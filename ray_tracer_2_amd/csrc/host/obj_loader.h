// obj_loader.h -- OBJ/MTL reader with the model-splitting, triangulation and
// multi-index behaviour of tobj 4.0.3 (Cargo.lock pin of the reference; the
// crate is not vendored, so this restates its published behaviour -- SURVEY.md
// 8a-6 / 8c) as configured by the reference:
//   tobj::load_obj(path, &LoadOptions { triangulate: true, single_index: false,
//                  ..Default::default() })          (src/core/asset.rs:110-118)
#ifndef RT_OBJ_LOADER_H
#define RT_OBJ_LOADER_H

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace rt2 {

struct ObjMesh {  // ≙ tobj::Mesh (multi-index form)
    std::vector<float> positions;           // 3 per model-local position
    std::vector<float> normals;             // 3 per model-local normal
    std::vector<float> texcoords;           // 2 per model-local texcoord
    std::vector<uint32_t> indices;          // into positions
    std::vector<uint32_t> normal_indices;   // into normals (parallel to indices)
    std::vector<uint32_t> texcoord_indices; // into texcoords (parallel to indices)
    int material_id = -1;                   // ≙ Option<usize>
};

struct ObjModel {  // ≙ tobj::Model
    ObjMesh mesh;
    std::string name;
};

struct ObjMaterial {  // ≙ tobj::Material (fields the reference reads)
    std::string name;
    bool has_diffuse = false, has_specular = false, has_shininess = false,
         has_optical_density = false, has_illum = false;
    float diffuse[3] = {0, 0, 0};
    float specular[3] = {0, 0, 0};
    float shininess = 0;
    float optical_density = 0;
    int illumination_model = 0;
    bool has_diffuse_texture = false;
    std::string diffuse_texture;
    std::map<std::string, std::string> unknown_param;
};

struct ObjLoadResult {
    std::vector<ObjModel> models;
    std::vector<ObjMaterial> materials;
    bool materials_ok = true;  // ≙ the Result of the material load
    std::string error;         // non-empty => load_obj returned Err
};

ObjLoadResult load_obj_file(const std::string& path);
// Parse from memory; mtl_dir is where `mtllib` names are resolved.
ObjLoadResult load_obj_text(const std::string& text, const std::string& mtl_dir);
bool load_mtl_text(const std::string& text, std::vector<ObjMaterial>& out,
                   std::map<std::string, size_t>& name_map);

}  // namespace rt2

#endif
