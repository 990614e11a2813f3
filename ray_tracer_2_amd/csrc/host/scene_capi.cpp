// scene_capi.cpp -- section 3 of include/rt_abi.h: the C ABI over the host
// scene pipeline (scene.h).  No device code; usable without a GPU.
#include <cmath>
#include <cstring>
#include <new>

#include "scene.h"

using namespace rt2;

struct rt_scene {
    Scene scene;
    std::unique_ptr<AssetManager> assets;
    std::string assets_dir;
    std::string err;
};

namespace rt2 {
const Scene& scene_of(const rt_scene* s) { return s->scene; }
}  // namespace rt2

namespace {

Transform from_abi(const rt_transform* t) {
    Transform r;
    if (!t) return r;
    r.pos = {t->pos[0], t->pos[1], t->pos[2]};
    r.rot = Quat{t->rot[0], t->rot[1], t->rot[2], t->rot[3]};
    r.scale = {t->scale[0], t->scale[1], t->scale[2]};
    return r;
}

AssetManager& assets_of(rt_scene* s, const char* dir) {
    std::string d = dir ? dir : "";
    if (!s->assets || s->assets_dir != d) {
        // keep already loaded textures when the directory is unchanged
        auto a = std::make_unique<AssetManager>(d);
        if (s->assets)
            for (const Image& im : s->assets->textures()) a->add_texture(im, "");
        s->assets = std::move(a);
        s->assets_dir = d;
    }
    return *s->assets;
}

}  // namespace

extern "C" {

int rt_scene_create(rt_scene** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return RT_ERR_OUT_OF_MEMORY;
    *out = s;
    return RT_OK;
}

int rt_scene_load_builtin(const char* name, const char* assets_dir, rt_scene** out) {
    if (!name || !out) return RT_ERR_INVALID_ARGUMENT;
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return RT_ERR_OUT_OF_MEMORY;
    *out = s;
    s->assets_dir = assets_dir ? assets_dir : "";
    try {
        if (!load_builtin_scene(name, s->assets_dir, ImageDecoder(), s->scene, s->err))
            return s->err.rfind("unknown scene", 0) == 0 ? RT_ERR_INVALID_ARGUMENT : RT_ERR_IO;
    } catch (const std::exception& e) {
        s->err = e.what();
        return RT_ERR_OUT_OF_MEMORY;
    }
    return RT_OK;
}

int rt_scene_set_camera(rt_scene* s, const rt_camera_desc* c) {
    if (!s || !c) return RT_ERR_INVALID_ARGUMENT;
    Camera cam;
    cam.transform = from_abi(&c->transform);
    cam.fov = c->fov;
    cam.aspect = c->aspect;
    cam.near_plane = c->near_plane;
    cam.far_plane = c->far_plane;
    cam.focus_dist = std::fmax(c->focus_dist, 1.0f);  // camera.rs:75
    cam.defocus_strength = c->defocus_strength;
    cam.diverge_strength = c->diverge_strength;
    if (cam.focus_dist == 0.0f) {  // camera.rs:82 assert
        s->err = "Focus Distance cannot be zero";
        return RT_ERR_INVALID_ARGUMENT;
    }
    s->scene.camera = cam;
    return RT_OK;
}

void rt_transform_cam(const float origin[3], const float look_at[3], rt_transform* out) {
    Transform t = Transform::cam({origin[0], origin[1], origin[2]}, {look_at[0], look_at[1], look_at[2]});
    out->pos[0] = t.pos.x; out->pos[1] = t.pos.y; out->pos[2] = t.pos.z;
    out->rot[0] = t.rot.x; out->rot[1] = t.rot.y; out->rot[2] = t.rot.z; out->rot[3] = t.rot.w;
    out->scale[0] = out->scale[1] = out->scale[2] = 1.0f;
}

int rt_scene_add_sphere(rt_scene* s, const float centre[3], float radius, const rt_material* m) {
    if (!s || !centre || !m) return RT_ERR_INVALID_ARGUMENT;
    if (s->scene.spheres.size() >= RT_MAX_SPHERES) {
        s->err = "more than 500 spheres";
        return RT_ERR_CAPACITY;
    }
    rt_sphere sp{};
    memcpy(sp.pos, centre, 12);
    sp.radius = radius;
    sp.material = *m;
    s->scene.spheres.push_back(sp);
    return RT_OK;
}

int rt_scene_add_obj(rt_scene* s, const char* assets_dir, const char* path, const rt_transform* t,
                     int use_mtl, const rt_material* m) {
    if (!s || !path) return RT_ERR_INVALID_ARGUMENT;
    try {
        AssetManager& a = assets_of(s, assets_dir);
        rt_material mat = m ? *m : material_uniform_default();
        if (!a.load_model(path, from_abi(t), use_mtl != 0, mat, s->scene.meshes, s->err))
            return RT_ERR_IO;
        s->scene.textures = a.textures();
        s->scene.built_bvh = false;
    } catch (const std::exception& e) {
        s->err = e.what();
        return RT_ERR_OUT_OF_MEMORY;
    }
    return RT_OK;
}

int rt_scene_add_mesh_data(rt_scene* s, const float* v8, uint32_t n_vertices, const uint32_t* indices,
                           uint32_t n_indices, const rt_transform* t, const rt_material* m) {
    if (!s || !v8 || !indices || n_indices % 3 != 0) return RT_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < n_indices; ++i)
        if (indices[i] >= n_vertices) {
            s->err = "vertex index out of range";
            return RT_ERR_INDEX_RANGE;
        }
    MeshInstance mi;
    mi.label = "mesh_" + std::to_string(s->scene.meshes.size() + s->scene.spheres.size());
    mi.transform = from_abi(t);
    mi.material = m ? *m : material_uniform_default();
    mi.data = std::make_shared<MeshData>();
    mi.data->vertices.resize(n_vertices);
    for (uint32_t i = 0; i < n_vertices; ++i) {
        Vertex& v = mi.data->vertices[i];
        const float* p = v8 + (size_t)i * 8;
        v.pos = {p[0], p[1], p[2]};
        v.normal = {p[3], p[4], p[5]};
        v.uv[0] = p[6];
        v.uv[1] = p[7];
    }
    mi.data->indices.assign(indices, indices + n_indices);
    s->scene.meshes.push_back(std::move(mi));
    s->scene.built_bvh = false;
    return RT_OK;
}

int rt_scene_add_texture_rgba8(rt_scene* s, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    if (!s || !rgba8 || !w || !h) return RT_ERR_INVALID_ARGUMENT;
    AssetManager& a = assets_of(s, s->assets_dir.c_str());
    Image im;
    im.width = w;
    im.height = h;
    im.rgba.assign(rgba8, rgba8 + (size_t)w * h * 4);
    int idx = a.add_texture(std::move(im), "");
    if (idx < 0) {
        s->err = "Cannot load more than 64 textures";
        return RT_ERR_CAPACITY;
    }
    s->scene.textures = a.textures();
    return idx;
}

int rt_scene_build(rt_scene* s, int quality) {
    if (!s || quality < 0 || quality > 2) return RT_ERR_INVALID_ARGUMENT;
    if (s->scene.meshes.size() > RT_MAX_MESHES) {
        s->err = "more than 400 meshes";
        return RT_ERR_CAPACITY;
    }
    try {
        s->scene.build_per_mesh((Quality)quality);
    } catch (const std::exception& e) {
        s->err = e.what();
        return RT_ERR_OUT_OF_MEMORY;
    }
    if (s->scene.triangles.size() > RT_MAX_TRIANGLES || s->scene.nodes.size() > RT_MAX_NODES) {
        s->err = "scene exceeds the triangle/node capacity";
        return RT_ERR_CAPACITY;
    }
    return RT_OK;
}

int rt_scene_build_device(rt_scene* s, int quality, int device, uint32_t min_triangles) {
    if (!s || quality < 0 || quality > 2 || device < -1) return RT_ERR_INVALID_ARGUMENT;
    if (s->scene.meshes.size() > RT_MAX_MESHES) {
        s->err = "more than 400 meshes";
        return RT_ERR_CAPACITY;
    }
    try {
        s->scene.build_per_mesh((Quality)quality, device, min_triangles ? (size_t)min_triangles : 16384);
    } catch (const std::exception& e) {
        s->err = e.what();
        return strncmp(e.what(), "HIP", 3) == 0 ? RT_ERR_DEVICE : RT_ERR_OUT_OF_MEMORY;
    }
    if (s->scene.triangles.size() > RT_MAX_TRIANGLES || s->scene.nodes.size() > RT_MAX_NODES) {
        s->err = "scene exceeds the triangle/node capacity";
        return RT_ERR_CAPACITY;
    }
    return RT_OK;
}

int rt_scene_subdivide_meshes(rt_scene* s, uint32_t n) {
    if (!s || n == 0) return RT_ERR_INVALID_ARGUMENT;
    try {
        s->scene.subdivide_meshes(n);
    } catch (const std::exception& e) {
        s->err = e.what();
        return RT_ERR_OUT_OF_MEMORY;
    }
    return RT_OK;
}

int rt_scene_get_uniform(const rt_scene* s, rt_scene_uniform* out) {
    if (!s || !out) return RT_ERR_INVALID_ARGUMENT;
    *out = s->scene.to_uniform();
    return RT_OK;
}

uint32_t rt_scene_num_spheres(const rt_scene* s) { return s ? (uint32_t)s->scene.spheres.size() : 0; }
uint32_t rt_scene_num_meshes(const rt_scene* s) { return s ? (uint32_t)s->scene.mesh_uniforms.size() : 0; }
uint32_t rt_scene_num_triangles(const rt_scene* s) { return s ? (uint32_t)s->scene.triangles.size() : 0; }
uint32_t rt_scene_num_nodes(const rt_scene* s) { return s ? (uint32_t)s->scene.nodes.size() : 0; }
uint32_t rt_scene_num_textures(const rt_scene* s) { return s ? (uint32_t)s->scene.textures.size() : 0; }
const rt_sphere* rt_scene_spheres(const rt_scene* s) { return s ? s->scene.spheres.data() : nullptr; }
const rt_mesh_uniform* rt_scene_meshes(const rt_scene* s) { return s ? s->scene.mesh_uniforms.data() : nullptr; }
const rt_packed_triangle* rt_scene_triangles(const rt_scene* s) { return s ? s->scene.triangles.data() : nullptr; }
const rt_node* rt_scene_nodes(const rt_scene* s) { return s ? s->scene.nodes.data() : nullptr; }

int rt_scene_get_texture(const rt_scene* s, uint32_t i, rt_texture_desc* out) {
    if (!s || !out || i >= s->scene.textures.size()) return RT_ERR_INVALID_ARGUMENT;
    const Image& im = s->scene.textures[i];
    out->rgba8 = im.rgba.data();
    out->width = im.width;
    out->height = im.height;
    return RT_OK;
}

const char* rt_scene_mesh_label(const rt_scene* s, uint32_t i) {
    if (!s || i >= s->scene.meshes.size()) return "";
    return s->scene.meshes[i].label.c_str();
}

uint32_t rt_scene_num_mesh_instances(const rt_scene* s) { return s ? (uint32_t)s->scene.meshes.size() : 0; }

int rt_scene_mesh_data(const rt_scene* s, uint32_t i, float* v8, uint32_t* n_vertices, uint32_t* indices,
                       uint32_t* n_indices, rt_transform* t, rt_material* m) {
    if (!s || i >= s->scene.meshes.size()) return RT_ERR_INVALID_ARGUMENT;
    const MeshInstance& mi = s->scene.meshes[i];
    if (n_vertices) *n_vertices = (uint32_t)mi.data->vertices.size();
    if (n_indices) *n_indices = (uint32_t)mi.data->indices.size();
    if (v8) {
        for (size_t k = 0; k < mi.data->vertices.size(); ++k) {
            const Vertex& v = mi.data->vertices[k];
            float* p = v8 + k * 8;
            p[0] = v.pos.x; p[1] = v.pos.y; p[2] = v.pos.z;
            p[3] = v.normal.x; p[4] = v.normal.y; p[5] = v.normal.z;
            p[6] = v.uv[0]; p[7] = v.uv[1];
        }
    }
    if (indices) memcpy(indices, mi.data->indices.data(), mi.data->indices.size() * sizeof(uint32_t));
    if (t) {
        t->pos[0] = mi.transform.pos.x; t->pos[1] = mi.transform.pos.y; t->pos[2] = mi.transform.pos.z;
        t->rot[0] = mi.transform.rot.x; t->rot[1] = mi.transform.rot.y; t->rot[2] = mi.transform.rot.z;
        t->rot[3] = mi.transform.rot.w;
        t->scale[0] = mi.transform.scale.x; t->scale[1] = mi.transform.scale.y; t->scale[2] = mi.transform.scale.z;
    }
    if (m) *m = mi.material;
    return RT_OK;
}

const char* rt_scene_last_error(const rt_scene* s) { return s ? s->err.c_str() : "null scene"; }

void rt_scene_destroy(rt_scene* s) { delete s; }

int rt_export_rgba8(const float* rgba, uint32_t width, uint32_t height, uint8_t* out) {
    // app.rs:408-460: per channel powf(1/2.2), clamp(0,1), *255 as u8
    // (truncating, NaN -> 0); x reversed then flipped back, rows flipped.
    if (!rgba || !out) return RT_ERR_INVALID_ARGUMENT;
    for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            const float* p = rgba + ((size_t)y * width + x) * 4;
            uint8_t* o = out + ((size_t)(height - 1 - y) * width + x) * 4;
            for (int c = 0; c < 4; ++c) {
                float v = std::pow(p[c], 1.0f / 2.2f);
                // f32::clamp propagates NaN; `as u8` maps NaN to 0
                if (v != v) { o[c] = 0; continue; }
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                o[c] = (uint8_t)(v * 255.0f);
            }
        }
    }
    return RT_OK;
}

}  // extern "C"
