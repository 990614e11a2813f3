// scene.cpp -- see scene.h.  Restates, in order: src/core/asset.rs:60-330
// (texture + model loading), src/core/bvh.rs:152-207 (per-mesh build and
// offsets), src/scene/scene.rs:179-271 (instantiate_scene), :280-983 (scene
// library), :985-1001 (to_uniform), src/scene/camera.rs:67-91.
//
// Intentional divergences: the reference's mesh
// cache is keyed by OBJ group name and races under rayon, so two models that
// share a name may alias; here every tobj model keeps its own geometry
// ("no aliasing" is the canonical scene).  Texture indices are assigned in
// material order instead of thread-scheduling order.  Models without faces
// are dropped (the reference would emit a mesh with zero BVH nodes).
#include "scene.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "obj_loader.h"

namespace rt2 {

rt_material material_uniform_default() {  // material.rs:19-36
    rt_material m{};
    m.color[0] = m.color[1] = m.color[2] = 0.7f;
    m.color[3] = 1.0f;
    m.smoothness = 0.9f;
    m.specular = 0.0f;
    m.ior = 1.0f;
    m.flag = RT_MATERIAL_DEFAULT;
    m.diffuse_index = -1;
    m.normal_index = -1;
    return m;
}

rt_material material_definition_default() {  // material.rs:75-92
    rt_material m{};
    m.color[0] = m.color[1] = m.color[2] = 0.7f;
    m.color[3] = 1.0f;
    for (int i = 0; i < 4; ++i) m.specular_color[i] = 1.0f;
    m.smoothness = 1.0f;
    m.specular = 0.0f;
    m.ior = 1.0f;
    m.flag = RT_MATERIAL_DEFAULT;
    m.diffuse_index = -1;
    m.normal_index = -1;
    return m;
}

rt_material material_definition_new() {  // material.rs:96-111
    rt_material m{};
    for (int i = 0; i < 4; ++i) m.color[i] = m.emission_color[i] = m.specular_color[i] = 1.0f;
    m.smoothness = 0.0f;
    m.specular = 0.1f;
    m.ior = 0.0f;
    m.flag = RT_MATERIAL_DEFAULT;
    m.diffuse_index = -1;
    m.normal_index = -1;
    return m;
}

rt_camera_uniform Camera::to_uniform() const {  // camera.rs:81-91
    const float rads_per_deg = 3.14159274101257324f / 180.0f;  // f32::to_radians
    float plane_height = focus_dist * std::tan((fov * 0.5f) * rads_per_deg) * 2.0f;
    float plane_width = plane_height * aspect;
    rt_camera_uniform u{};
    Mat4 m = transform.to_matrix();
    memcpy(u.cam_to_world, m.c, sizeof(m.c));
    u.view_params[0] = plane_width;
    u.view_params[1] = plane_height;
    u.view_params[2] = focus_dist;
    u.defocus_strength = defocus_strength;
    u.diverge_strength = diverge_strength;
    return u;
}

void flip_horizontal(Image& img) {
    for (uint32_t y = 0; y < img.height; ++y) {
        uint8_t* row = img.rgba.data() + (size_t)y * img.width * 4;
        for (uint32_t x = 0; x < img.width / 2; ++x) {
            uint8_t tmp[4];
            memcpy(tmp, row + x * 4, 4);
            memcpy(row + x * 4, row + (img.width - 1 - x) * 4, 4);
            memcpy(row + (img.width - 1 - x) * 4, tmp, 4);
        }
    }
}

int AssetManager::add_texture(Image img, const std::string& key) {
    if (textures_.size() >= RT_MAX_TEXTURES) return -1;  // asset.rs:61-64
    textures_.push_back(std::move(img));
    texture_keys_.push_back(key);
    return (int)textures_.size() - 1;
}

int AssetManager::load_texture(const std::string& path, std::string& err) {  // asset.rs:60-85
    if (textures_.size() == RT_MAX_TEXTURES) return -1;
    for (size_t i = 0; i < texture_keys_.size(); ++i)
        if (texture_keys_[i] == path) return (int)i;
    Image img;
    std::string full = assets_dir_ + "/" + path;
    bool ok = decoder ? decoder(full, img) : decode_png_file(full, img);
    if (!ok) {
        err = "cannot load texture " + full;
        return -2;
    }
    flip_horizontal(img);  // asset.rs:77
    return add_texture(std::move(img), path);
}

namespace {

float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }  // f32::clamp(0,1)

bool parse_f32_word(const std::string& w, float& out) {
    if (w.empty()) return false;
    char* end = nullptr;
    out = strtof(w.c_str(), &end);
    return end && *end == '\0';
}

}  // namespace

bool AssetManager::load_model(const std::string& path, const Transform& transform, bool use_mtl,
                              const rt_material& material, std::vector<MeshInstance>& out,
                              std::string& err) {
    ObjLoadResult obj = load_obj_file(assets_dir_ + "/" + path);  // asset.rs:108-118
    if (!obj.error.empty()) {
        err = "Failed to load OBJ File: " + obj.error;
        return false;
    }
    std::map<size_t, rt_material> material_map;
    if (use_mtl && obj.materials_ok) {  // asset.rs:124-206
        std::map<std::string, int> texture_refs;
        for (const ObjMaterial& m : obj.materials) {
            if (m.has_diffuse_texture && !texture_refs.count(m.diffuse_texture)) {
                int r = load_texture(m.diffuse_texture, err);
                if (r == -2) return false;
                texture_refs[m.diffuse_texture] = r;
            }
            auto nd = m.unknown_param.find("map_Disp");
            if (nd != m.unknown_param.end() && !texture_refs.count(nd->second)) {
                int r = load_texture(nd->second, err);
                if (r == -2) return false;
                texture_refs[nd->second] = r;
            }
        }
        for (size_t i = 0; i < obj.materials.size(); ++i) {
            const ObjMaterial& m = obj.materials[i];
            float color[3] = {0.7f, 0.7f, 0.7f}, spec[3] = {1.0f, 1.0f, 1.0f};
            if (m.has_diffuse) memcpy(color, m.diffuse, 12);
            if (m.has_specular) memcpy(spec, m.specular, 12);
            int illum = m.has_illum ? m.illumination_model : 0;
            int flag = (illum == 4 || illum == 6 || illum == 9) ? RT_MATERIAL_GLASS
                                                                : RT_MATERIAL_DEFAULT;
            int diffuse_index = -1, normal_index = -1;
            if (m.has_diffuse_texture) {
                flag = RT_MATERIAL_TEXTURE;
                diffuse_index = texture_refs[m.diffuse_texture];
            }
            auto nd = m.unknown_param.find("map_Disp");
            if (nd != m.unknown_param.end()) {
                flag = RT_MATERIAL_TEXTURE;
                normal_index = texture_refs[nd->second];
            }
            float emission_strength = 0.0f;
            float ecol[3] = {0, 0, 0};
            auto ke = m.unknown_param.find("Ke");
            if (ke != m.unknown_param.end()) {
                std::vector<float> vals;
                size_t p = 0;
                const std::string& s = ke->second;
                while (p < s.size()) {
                    while (p < s.size() && isspace((unsigned char)s[p])) ++p;
                    size_t q = p;
                    while (q < s.size() && !isspace((unsigned char)s[q])) ++q;
                    float f;
                    if (q > p && parse_f32_word(s.substr(p, q - p), f)) vals.push_back(f);
                    p = q;
                }
                if (vals.size() == 3) {
                    emission_strength = fmax32(fmax32(vals[0], vals[1]), vals[2]);
                    float d = emission_strength == 0.0f ? 1.0f : emission_strength;
                    ecol[0] = vals[0] / d;
                    ecol[1] = vals[1] / d;
                    ecol[2] = vals[2] / d;
                }
            }
            rt_material mat = material_uniform_default();
            for (int k = 0; k < 3; ++k) {
                mat.color[k] = color[k];
                mat.emission_color[k] = ecol[k];
                mat.specular_color[k] = spec[k];
            }
            mat.color[3] = mat.emission_color[3] = mat.specular_color[3] = 1.0f;
            mat.emission_strength = emission_strength * 2.0f;
            float ns = m.has_shininess ? m.shininess : 0.0f;
            mat.smoothness = clamp01(std::sqrt(ns / 100.0f));
            mat.specular = clamp01(fmax32(fmax32(spec[0], spec[1]), spec[2]));
            mat.ior = m.has_optical_density ? m.optical_density : 1.0f;
            mat.flag = flag;
            mat.diffuse_index = diffuse_index;
            mat.normal_index = normal_index;
            material_map[i] = mat;
        }
    }

    for (ObjModel& m : obj.models) {  // asset.rs:208-327
        if (m.mesh.indices.empty()) continue;
        auto data = std::make_shared<MeshData>();
        size_t num_vertices = m.mesh.positions.size() / 3;
        std::vector<Vec3> calculated(num_vertices);
        auto P = [&](size_t i) {
            return Vec3{m.mesh.positions[3 * i], m.mesh.positions[3 * i + 1],
                        m.mesh.positions[3 * i + 2]};
        };
        if (m.mesh.normals.empty()) {  // asset.rs:224-261
            for (size_t t = 0; t + 2 < m.mesh.indices.size(); t += 3) {
                size_t i0 = m.mesh.indices[t], i1 = m.mesh.indices[t + 1],
                       i2 = m.mesh.indices[t + 2];
                Vec3 v0 = P(i0), v1 = P(i1), v2 = P(i2);
                Vec3 e1 = v1 - v0, e2 = v2 - v1;
                Vec3 n = cross(e1, e2);
                calculated[i0] = calculated[i0] + n;
                calculated[i1] = calculated[i1] + n;
                calculated[i2] = calculated[i2] + n;
            }
            for (Vec3& n : calculated) {
                float len = length(n);
                if (len > 0.0f) n = n / len;
            }
        }
        size_t n_idx = m.mesh.indices.size();
        data->vertices.resize(n_idx);
        bool has_n = !m.mesh.normals.empty();
        bool has_ni = !m.mesh.normal_indices.empty();
        bool has_t = !m.mesh.texcoords.empty() && !m.mesh.texcoord_indices.empty();
        if ((has_n && has_ni && m.mesh.normal_indices.size() != n_idx) ||
            (has_t && m.mesh.texcoord_indices.size() != n_idx)) {
            err = "model '" + m.name + "': faces mix vertices with and without vn/vt";
            return false;
        }
        for (size_t j = 0; j < n_idx; ++j) {  // asset.rs:262-310
            size_t pi = m.mesh.indices[j];
            Vertex v;
            v.pos = P(pi);
            if (has_n && has_ni) {
                size_t ni = m.mesh.normal_indices[j];
                v.normal = Vec3{m.mesh.normals[3 * ni], m.mesh.normals[3 * ni + 1],
                                m.mesh.normals[3 * ni + 2]};
            } else if (has_n) {
                if (3 * pi + 2 >= m.mesh.normals.size()) {
                    err = "model '" + m.name + "': normal index out of range";
                    return false;
                }
                v.normal = Vec3{m.mesh.normals[3 * pi], m.mesh.normals[3 * pi + 1],
                                m.mesh.normals[3 * pi + 2]};
            } else {
                v.normal = calculated[pi];
            }
            if (has_t) {
                size_t ti = m.mesh.texcoord_indices[j];
                v.uv[0] = m.mesh.texcoords[2 * ti];
                v.uv[1] = m.mesh.texcoords[2 * ti + 1];
            }
            data->vertices[j] = v;
        }
        data->indices.resize(n_idx);
        for (size_t j = 0; j < n_idx; ++j) data->indices[j] = (uint32_t)j;

        MeshInstance inst;
        inst.label = m.name;
        inst.transform = transform;
        inst.data = data;
        if (use_mtl && m.mesh.material_id >= 0 && material_map.count((size_t)m.mesh.material_id))
            inst.material = material_map[(size_t)m.mesh.material_id];
        else
            inst.material = material_uniform_default();
        if (!use_mtl) inst.material = material;  // asset.rs:94-98
        out.push_back(std::move(inst));
    }
    return true;
}

void Scene::build_per_mesh(Quality q, int device, size_t device_min_tris) {  // bvh.rs:152-207
    triangles.clear();
    nodes.clear();
    mesh_uniforms.clear();
    size_t triangle_offset = 0, node_offset = 0;
    for (const MeshInstance& mi : meshes) {
        BvhResult r;
        if (device >= -1 && q == Quality::High && mi.data->indices.size() / 3 >= device_min_tris) {
            // (device -1: the level-wise build with the searches on the host, for validation)
            const std::vector<float> tri9 = bvh_search_data(mi.data->vertices, mi.data->indices);
            r = bvh_build_levels(mi.data->vertices, mi.data->indices, q,
                                 device >= 0 ? make_device_level_search(device, tri9.data(), tri9.size() / 9)
                                             : make_host_level_search(tri9.data(), tri9.size() / 9));
        } else {
            r = bvh_build(mi.data->vertices, mi.data->indices, q);
        }
        Mat4 m2w = mi.transform.to_matrix();
        Mat4 w2m = mat4_inverse(m2w);
        rt_mesh_uniform u{};
        memcpy(u.world_to_model, w2m.c, sizeof(w2m.c));
        memcpy(u.model_to_world, m2w.c, sizeof(m2w.c));
        u.node_offset = (uint32_t)node_offset;
        u.triangle_offset = (uint32_t)triangle_offset;
        u.triangles = (uint32_t)r.triangles.size();
        u.material = mi.material;
        mesh_uniforms.push_back(u);
        triangles.insert(triangles.end(), r.triangles.begin(), r.triangles.end());
        nodes.insert(nodes.end(), r.nodes.begin(), r.nodes.end());
        triangle_offset += r.triangles.size();
        node_offset += r.nodes.size();
    }
    built_bvh = true;
}

rt_scene_uniform Scene::to_uniform() const {  // scene.rs:985-1001
    rt_scene_uniform u{};
    uint32_t nv = 0, ni = 0;
    for (const MeshInstance& m : meshes) {
        nv += (uint32_t)m.data->vertices.size();
        ni += (uint32_t)m.data->indices.size();
    }
    u.spheres = (uint32_t)spheres.size();
    u.n_vertices = nv;
    u.n_indices = ni;
    u.meshes = (uint32_t)meshes.size();
    u.camera = camera.to_uniform();
    u.nodes = (uint32_t)nodes.size();
    return u;
}

void Scene::subdivide_meshes(uint32_t n) {
    if (n <= 1) return;
    for (MeshInstance& mi : meshes) {
        auto nd = std::make_shared<MeshData>();
        const MeshData& d = *mi.data;
        float inv = 1.0f / (float)n;
        auto lerp3 = [&](const Vertex& a, const Vertex& b, const Vertex& c, uint32_t i, uint32_t j) {
            // barycentric point (i, j, n - i - j) / n
            float wb = (float)i * inv, wc = (float)j * inv, wa = (float)(n - i - j) * inv;
            Vertex v;
            v.pos = (a.pos * wa + b.pos * wb) + c.pos * wc;
            Vec3 nn = (a.normal * wa + b.normal * wb) + c.normal * wc;
            float len = length(nn);
            v.normal = len > 0.0f ? nn / len : nn;
            v.uv[0] = (a.uv[0] * wa + b.uv[0] * wb) + c.uv[0] * wc;
            v.uv[1] = (a.uv[1] * wa + b.uv[1] * wb) + c.uv[1] * wc;
            return v;
        };
        for (size_t t = 0; t + 2 < d.indices.size(); t += 3) {
            const Vertex &a = d.vertices[d.indices[t]], &b = d.vertices[d.indices[t + 1]],
                         &c = d.vertices[d.indices[t + 2]];
            for (uint32_t i = 0; i < n; ++i) {
                for (uint32_t j = 0; i + j < n; ++j) {
                    Vertex p00 = lerp3(a, b, c, i, j), p10 = lerp3(a, b, c, i + 1, j),
                           p01 = lerp3(a, b, c, i, j + 1);
                    nd->vertices.push_back(p00);
                    nd->vertices.push_back(p10);
                    nd->vertices.push_back(p01);
                    if (i + j + 1 < n) {
                        Vertex p11 = lerp3(a, b, c, i + 1, j + 1);
                        nd->vertices.push_back(p10);
                        nd->vertices.push_back(p11);
                        nd->vertices.push_back(p01);
                    }
                }
            }
        }
        nd->indices.resize(nd->vertices.size());
        for (size_t k = 0; k < nd->indices.size(); ++k) nd->indices[k] = (uint32_t)k;
        mi.data = nd;
    }
    built_bvh = false;
}

// ------------------------------------------------------------------------
// Scene library (scene.rs:280-983)
// ------------------------------------------------------------------------
namespace {

struct Def {  // ≙ SceneDefinition + instantiate_scene, entity by entity
    Scene& scene;
    AssetManager& assets;
    std::string& err;
    bool ok = true;
    int mesh_counter = 0;  // entity index for "mesh_{i}" labels (scene.rs:236)
    int entity_index = 0;

    void sphere(Vec3 c, float r, rt_material m) {  // scene.rs:79-85,219-221
        rt_sphere s{};
        s.pos[0] = c.x; s.pos[1] = c.y; s.pos[2] = c.z;
        s.radius = r;
        s.material = m;
        scene.spheres.push_back(s);
        ++entity_index;
    }
    void mesh_data(Transform t, std::vector<Vertex> verts, std::vector<uint32_t> idx,
                   rt_material m) {  // scene.rs:234-244
        MeshInstance mi;
        mi.label = "mesh_" + std::to_string(entity_index);
        mi.transform = t;
        mi.data = std::make_shared<MeshData>();
        mi.data->vertices = std::move(verts);
        mi.data->indices = std::move(idx);
        mi.material = m;
        scene.meshes.push_back(std::move(mi));
        ++entity_index;
    }
    void mesh_file(Transform t, const std::string& path, bool use_mtl, rt_material m) {
        if (!ok) return;
        if (!assets.load_model(path, t, use_mtl, m, scene.meshes, err)) ok = false;
        ++entity_index;
    }
};

Vertex V(float x, float y, float z, Vec3 n) {
    Vertex v;
    v.pos = {x, y, z};
    v.normal = n;
    return v;
}

rt_material with_color(rt_material m, float r, float g, float b, float a) {
    m.color[0] = r; m.color[1] = g; m.color[2] = b; m.color[3] = a;
    return m;
}
rt_material with_emissive(rt_material m, float r, float g, float b, float a, float s) {
    m.emission_color[0] = r; m.emission_color[1] = g; m.emission_color[2] = b;
    m.emission_color[3] = a;
    m.emission_strength = s;
    return m;
}
rt_material with_specular(rt_material m, float r, float g, float b, float a, float s) {
    m.specular_color[0] = r; m.specular_color[1] = g; m.specular_color[2] = b;
    m.specular_color[3] = a;
    m.specular = s;
    return m;
}
rt_material with_smooth(rt_material m, float s) { m.smoothness = s; return m; }
rt_material with_glass(rt_material m, float ior) { m.ior = ior; m.flag = RT_MATERIAL_GLASS; return m; }

const Vec3 X{1, 0, 0}, Y{0, 1, 0}, Z{0, 0, 1};
const std::vector<uint32_t> IDX_A{0, 1, 2, 0, 2, 3}, IDX_B{2, 1, 0, 3, 2, 0};

std::vector<Vertex> quad_y(float x0, float x1, float y, float z0, float z1, Vec3 n) {
    return {V(x0, y, z0, n), V(x1, y, z0, n), V(x1, y, z1, n), V(x0, y, z1, n)};
}

}  // namespace

bool load_builtin_scene(const std::string& name, const std::string& assets_dir,
                        const ImageDecoder& decoder, Scene& scene, std::string& err) {
    AssetManager assets(assets_dir);
    assets.decoder = decoder;
    scene = Scene();
    Def d{scene, assets, err};
    const rt_material NEW = material_definition_new();
    Camera cam;  // CameraDescriptor::default() camera.rs:50-66

    if (name == "cornell_box") {  // scene.rs:911-933
        cam.transform = Transform::cam({0, 1, 2}, {0, 1, 0});
        rt_material m = material_definition_default();
        m.flag = RT_MATERIAL_GLASS;  // texture_from_obj(), unused because use_mtl
        d.mesh_file(Transform{}, "CornellBox-Original.obj", true, m);
    } else if (name == "texture_test") {  // scene.rs:280-309
        cam.transform = Transform::cam({0, 0, -1}, {0, 0, 0});
        rt_material m{};
        m.color[0] = 1; m.color[3] = 1;
        for (int i = 0; i < 4; ++i) m.specular_color[i] = 1;
        m.specular = 0.05f;
        m.ior = 1.0f;
        m.flag = RT_MATERIAL_TEXTURE;
        m.normal_index = -1;
        m.diffuse_index = assets.load_texture("earthmap.png", err);
        if (m.diffuse_index == -2) return false;
        d.sphere({0, 0, 0}, 1.0f, m);
    } else if (name == "obj_test") {  // scene.rs:310-364
        cam.transform = Transform::cam({5, 0, 0}, {1, 0, 0});
        cam.fov = 45; cam.near_plane = 0.1f; cam.far_plane = 100; cam.focus_dist = 1.0f;
        d.mesh_file(Transform{}, "dragon.obj", false, NEW);
        d.mesh_data(Transform{}, {V(0.5f, 0, -1, X), V(0.5f, 1, -1, X), V(0, 1, 1, X), V(0.2f, 0, 1, X)},
                    IDX_A, with_emissive(with_color(NEW, 1, 1, 0, 1), 1, 0, 0, 1, 0.4f));
        d.sphere({1.8f, 0.1f, 1.0f}, 0.6f, with_color(NEW, 1, 0, 0, 1));
        d.sphere({1.0f, 0.5f, 1.0f}, 0.3f, with_color(NEW, 1, 0, 0, 1));
        d.sphere({0, -10, 0}, 10.0f, with_color(NEW, 1, 0, 0, 1));
    } else if (name == "room") {  // scene.rs:445-573
        cam.transform = Transform::cam({0, 1, 3}, {0, 1, 2});
        cam.fov = 45; cam.near_plane = 0.1f; cam.far_plane = 100; cam.focus_dist = 0.1f;
        d.mesh_data(Transform{}, quad_y(-2, 2, 0, -2, 2, Y), IDX_B, with_color(NEW, 1, 0, 0, 1));
        d.mesh_data(Transform{}, quad_y(-2, 2, 4, -2, 2, -Y), IDX_A, with_color(NEW, 0, 0.3f, 0.3f, 1));
        d.mesh_data(Transform{}, {V(-2, 0, -2, X), V(-2, 4, -2, X), V(-2, 4, 2, X), V(-2, 0, 2, X)}, IDX_A,
                    with_smooth(with_specular(NEW, 1, 1, 1, 1, 1.0f), 1.0f));
        d.mesh_data(Transform{}, {V(2, 0, -2, -X), V(2, 0, 2, -X), V(2, 4, 2, -X), V(2, 4, -2, -X)}, IDX_A,
                    with_smooth(with_specular(NEW, 1, 1, 1, 1, 0.99f), 0.99f));
        d.mesh_data(Transform{}, {V(-2, 0, 2, -Z), V(2, 0, 2, -Z), V(2, 4, 2, -Z), V(-2, 4, 2, -Z)}, IDX_B,
                    with_smooth(with_specular(with_color(NEW, 0.2f, 0.2f, 0.82f, 1), 1, 1, 1, 1, 0.99f), 0.99f));
        d.mesh_data(Transform{}, quad_y(-0.4f, 0.4f, 3.98f, -0.4f, 0.4f, -Y), IDX_A,
                    with_emissive(NEW, 1, 1, 1, 1, 3.0f));
        d.sphere({0.4f, 1.0f, 0.0f}, 0.3f, with_glass(with_color(NEW, 0.4f, 0.9f, 0.4f, 1), 1.34f));
        d.sphere({-0.4f, 1.0f, 0.0f}, 0.4f, with_specular(with_color(NEW, 0.7f, 0.7f, 0.7f, 1), 1, 1, 1, 1, 0.2f));
    } else if (name == "room_2") {  // scene.rs:574-757
        cam.transform = Transform::cam({0, 1.28f, 13.5f}, {0, 1.28f, 12.5f});
        cam.fov = 26; cam.near_plane = 0.1f; cam.far_plane = 100; cam.focus_dist = 8.6f;
        cam.defocus_strength = 100.0f; cam.diverge_strength = 1.5f;
        const float width = 3.0f, depth = 2.0f, height = 4.0f;
        rt_material dragon = with_specular(with_smooth(with_color(NEW, 0.96078f, 0.11372f, 0.4039f, 1), 0.8f), 1, 1, 1, 1, 0.015f);
        Transform t1; t1.pos = {0, 1.2f, -0.6f}; t1.rot = quat_from_euler(EulerRot::XYX, 0, -1.5708f, 0); t1.scale = {4.7f, 4.7f, 4.7f};
        d.mesh_file(t1, "Dragon_80K.obj", false, dragon);
        Transform t2; t2.pos = {0, 7.2f, 2.0f}; t2.rot = quat_from_euler(EulerRot::XYX, 0, -1.5708f, 0); t2.scale = {1, 1, 1};
        d.mesh_file(t2, "Dragon_80K.obj", false, dragon);
        d.mesh_data(Transform{}, quad_y(-10, 10, -0.01f, -10, 10, Y), IDX_B, with_color(NEW, 0.4f, 0.4f, 0.64313f, 1));
        d.mesh_data(Transform{}, quad_y(-10, 10, 8.5f, -10, 10, -Y), IDX_A,
                    with_specular(with_smooth(with_color(NEW, 0.898f, 0.87f, 0.815f, 1), 0.877f), 1, 1, 1, 1, 0.327f));
        d.mesh_data(Transform{}, quad_y(-width, width, 0, -depth, depth, Y), IDX_B, with_color(NEW, 0.898f, 0.87f, 0.815f, 1));
        d.mesh_data(Transform{}, quad_y(-width, width, height, -depth, depth, -Y), IDX_A, with_color(NEW, 1.0f, 0.9647f, 0.9019f, 1));
        d.mesh_data(Transform{}, {V(-width, 0, -depth, X), V(-width, height, -depth, X), V(-width, height, depth, X), V(-width, 0, depth, X)},
                    IDX_A, with_color(NEW, 0.0705f, 0.596f, 0.2078f, 1));
        d.mesh_data(Transform{}, {V(width, 0, -depth, -X), V(width, 0, depth, -X), V(width, height, depth, -X), V(width, height, -depth, -X)},
                    IDX_A, with_color(NEW, 0.7725f, 0.12156f, 0.188235f, 1));
        d.mesh_data(Transform{}, {V(-width, 0, -depth, Z), V(width, 0, -depth, Z), V(width, height, -depth, Z), V(-width, height, -depth, Z)},
                    IDX_A, with_color(NEW, 0.1254f, 0.41176f, 0.8274f, 1));
        d.mesh_data(Transform{}, quad_y(-0.8f, 0.8f, height - 0.02f, -0.8f, 0.8f, -Y), IDX_A,
                    with_emissive(NEW, 1.0f, 0.8588f, 0.3529f, 1.0f, 60.0f));
        d.sphere({0, 1.0f, 4.4f}, 1.15f, with_glass(with_smooth(with_specular(NEW, 1, 1, 1, 1, 0.517f), 1.0f), 1.6f));
    } else if (name == "metal") {  // scene.rs:758-801
        cam.transform = Transform::cam({0, 0, 3}, {0, 0, -1});
        cam.fov = 45; cam.near_plane = 0.1f; cam.far_plane = 100; cam.focus_dist = 0.1f;
        d.sphere({0, -100.5f, -1}, 100.0f, with_color(NEW, 0.8f, 0.8f, 0.0f, 1));
        d.sphere({0, 0, -1}, 0.5f, with_color(NEW, 0.7f, 0.3f, 0.3f, 1));
        d.sphere({-1, 0, -1}, 0.5f, with_glass(with_color(NEW, 0.8f, 0.8f, 0.8f, 1), 1.3f));
        d.sphere({1, 0, -1}, 0.5f, with_specular(with_color(NEW, 0.8f, 0.6f, 0.2f, 1), 1, 1, 1, 1, 0.15f));
    } else if (name == "balls") {  // scene.rs:802-863
        cam.transform = Transform::cam({3.089f, 1.53f, -3.0f}, {-2, -1, 2});
        cam.fov = 45; cam.near_plane = 0.1f; cam.far_plane = 100; cam.focus_dist = 0.1f;
        d.sphere({-3.64f, -0.42f, 0.8028f}, 0.75f, with_color(with_specular(NEW, 1, 1, 1, 1, 0.7f), 1, 1, 1, 1));
        d.sphere({-2.54f, -0.72f, 0.5f}, 0.6f, with_specular(with_color(NEW, 1, 0, 0, 1), 1, 0, 0, 1, 0.5f));
        d.sphere({-1.27f, -0.72f, 1.0f}, 0.5f, with_specular(with_color(NEW, 0, 1, 0, 1), 0, 1, 0, 1, 0.2f));
        d.sphere({-0.5f, -0.9f, 1.55f}, 0.35f, with_color(NEW, 0, 0, 1, 1));
        d.sphere({-3.46f, -15.88f, 2.76f}, 15.0f, with_color(NEW, 0.5f, 0.0f, 0.8f, 1));
        d.sphere({-7.44f, -0.72f, 20.0f}, 15.0f, with_emissive(with_color(NEW, 0.1f, 0.1f, 0.1f, 0.0f), 1, 1, 1, 1, 1.0f));
    } else if (name == "random_balls" || name.rfind("random_balls:", 0) == 0) {  // scene.rs:365-444
        // The reference draws this scene from rand::rng() (ThreadRng, seeded by the OS, scene.rs:403), so no two
        // of its runs agree and none can be reproduced.  Documented divergence: the same construction -- the
        // same draws in the same order, the same thresholds and ranges -- from a SEEDED generator, the shader's
        // own PCG variant (wgsl:195-200); "random_balls:<seed>" picks the seed (default 0).  f32 draws follow
        // rand 0.9's StandardUniform (24 random bits in [0, 1)); random_range(a..b) is a + (b - a) * that.
        uint32_t state = 0;
        if (name.size() > 13) state = (uint32_t)std::strtoul(name.c_str() + 13, nullptr, 10);
        auto next_u32 = [&]() {
            state = state * 747796405u + 2891336453u;
            uint32_t r = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
            return (r >> 22u) ^ r;
        };
        auto unit = [&]() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); };
        auto range = [&](float lo, float hi) { return lo + (hi - lo) * unit(); };
        cam.transform = Transform::cam({13, 2, 3}, {0, 0, 0});
        cam.fov = 20; cam.near_plane = 0.1f; cam.far_plane = 100; cam.focus_dist = 10.0f;
        d.sphere({0, -1000, 0}, 1000.0f, with_color(NEW, 0.5f, 0.5f, 0.5f, 1));
        d.sphere({0, 1, 0}, 1.0f, with_glass(NEW, 1.5f));
        d.sphere({-4, 1, 0}, 1.0f, with_color(NEW, 0.4f, 0.2f, 0.1f, 1));
        d.sphere({4, 1, 0}, 1.0f, with_smooth(with_specular(with_color(NEW, 0.7f, 0.6f, 0.5f, 1), 0.7f, 0.6f, 0.5f, 1, 1.0f), 1.0f));
        for (int a = -11; a < 11; ++a)
            for (int b = -11; b < 11; ++b) {
                const float mat = unit();
                const float cx = (float)a + 0.9f * unit(), cz = (float)b + 0.9f * unit();
                const float dx = cx - 4.0f, dy = 0.2f - 0.2f, dz = cz - 0.0f;
                if (std::sqrt((dx * dx + dy * dy) + dz * dz) > 0.9f) {
                    if (mat < 0.8f) {
                        const float r = unit(), g = unit(), bl = unit();
                        d.sphere({cx, 0.2f, cz}, 0.2f, with_color(NEW, r, g, bl, 1));
                    } else if (mat < 0.95f) {
                        const float r = range(0.5f, 1.0f), g = range(0.5f, 1.0f), bl = range(0.5f, 1.0f);
                        const float fuzz = range(0.0f, 0.5f);
                        d.sphere({cx, 0.2f, cz}, 0.2f, with_specular(with_color(NEW, r, g, bl, 1), 1, 1, 1, 1, fuzz));
                    } else {
                        d.sphere({cx, 0.2f, cz}, 0.2f, with_glass(NEW, 1.3f));
                    }
                }
            }
    } else if (name == "sponza" || name == "bugatti") {  // scene.rs:864-910, 934-983
        const bool sp = name == "sponza";
        cam.transform = sp ? Transform::cam({0, 4, 0}, {0, 4, 1}) : Transform::cam({0, 0, 0}, {0, 0, 1});
        rt_material m = material_definition_default();
        m.flag = RT_MATERIAL_GLASS;
        Transform t; t.scale = {0.05f, 0.05f, 0.05f};
        d.mesh_file(t, sp ? "sponza.obj" : "f1/f1.obj", true, m);
        Transform q; q.pos = {-15, 60, 0};
        q.rot = quat_rotation_x(3.14159274101257324f / 2.0f);
        q.scale = {40, 20, 1};
        std::vector<Vertex> quad = {V(-1, -1, 0, Z), V(1, -1, 0, Z), V(1, 1, 0, Z), V(-1, 1, 0, Z)};  // mesh.rs:22-31
        quad[1].uv[0] = 1; quad[2].uv[0] = 1; quad[2].uv[1] = 1; quad[3].uv[1] = 1;
        d.mesh_data(q, quad, IDX_A, with_emissive(material_definition_default(), 1, 1, 1, 1, 4.0f));
        rt_material s = material_definition_default();
        for (int i = 0; i < 4; ++i) s.emission_color[i] = s.color[i] = s.specular_color[i] = 1.0f;
        s.emission_strength = 10.0f;
        s.smoothness = 0.0f;
        s.specular = 0.0f;
        d.sphere({5, 2, 0}, 2.0f, s);
    } else {
        err = "unknown scene '" + name + "'";
        return false;
    }
    if (!d.ok) return false;
    // Camera::new clamps the focus distance (camera.rs:75)
    cam.focus_dist = fmax32(cam.focus_dist, 1.0f);
    scene.camera = cam;
    scene.build_per_mesh(Quality::High);  // scene.rs:260
    // create_texture_array (asset.rs:32-47): loaded textures at their indices
    scene.textures = assets.textures();
    return true;
}

}  // namespace rt2
