// obj_loader.cpp -- see obj_loader.h.  Behaviour notes (tobj 4.0.3):
//  * `o` and `g` are treated alike: if faces are pending they are flushed as a
//    model carrying the *previous* name, then the name changes.
//  * `usemtl X` flushes pending faces as a model when X differs from the
//    current material, keeping the current name (so one group can yield
//    several models, and a `g` that follows its faces names the *next* model:
//    CornellBox-Original.obj yields 8 models, two of them named "leftWall").
//  * end of file flushes whatever is pending under the current name.
//  * negative indices are relative to the element counts at that line.
//  * triangulate: quads -> (0,1,2),(0,2,3); polygons -> fans; points and
//    lines become degenerate triangles (a,a,a) / (a,b,b).
//  * multi-index export: positions / normals / texcoords are re-indexed per
//    model in first-use order.
#include "obj_loader.h"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

namespace rt2 {
namespace {

const uint32_t MISSING = 0xffffffffu;

struct VertexIndices {
    uint32_t v = MISSING, vt = MISSING, vn = MISSING;
};
typedef std::vector<VertexIndices> Face;

std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0, n = s.size();
    while (i < n) {
        while (i < n && isspace((unsigned char)s[i])) ++i;
        size_t j = i;
        while (j < n && !isspace((unsigned char)s[j])) ++j;
        if (j > i) out.emplace_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// Rust `str::parse::<f32>` is a correctly rounded decimal->binary32
// conversion, as glibc strtof is; reject trailing garbage like Rust does.
bool parse_f32(const std::string& w, float& out) {
    if (w.empty()) return false;
    char* end = nullptr;
    out = strtof(w.c_str(), &end);
    return end && *end == '\0';
}

bool parse_floatn(const std::vector<std::string>& words, size_t first, std::vector<float>& vals,
                  size_t n) {
    size_t sz = vals.size();
    for (size_t i = first; i < words.size() && vals.size() - sz < n; ++i) {
        float f;
        if (!parse_f32(words[i], f)) return false;
        vals.push_back(f);
    }
    return vals.size() - sz == n;
}

bool parse_index(const std::string& tok, size_t count, uint32_t& out) {
    if (tok.empty()) {
        out = MISSING;
        return true;
    }
    char* end = nullptr;
    long long v = strtoll(tok.c_str(), &end, 10);
    if (!end || *end != '\0') return false;
    if (v < 0) {
        long long r = (long long)count + v;
        if (r < 0) return false;
        out = (uint32_t)r;
    } else {
        if (v == 0) return false;  // OBJ indices are 1-based
        out = (uint32_t)(v - 1);
    }
    return true;
}

bool parse_face(const std::vector<std::string>& words, std::vector<Face>& faces, size_t pos_sz,
                size_t tex_sz, size_t norm_sz) {
    Face f;
    for (size_t i = 1; i < words.size(); ++i) {
        const std::string& w = words[i];
        VertexIndices vi;
        size_t s1 = w.find('/');
        std::string a = w.substr(0, s1), b, c;
        if (s1 != std::string::npos) {
            size_t s2 = w.find('/', s1 + 1);
            b = w.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
            if (s2 != std::string::npos) c = w.substr(s2 + 1);
        }
        if (a.empty()) return false;
        if (!parse_index(a, pos_sz, vi.v)) return false;
        if (!parse_index(b, tex_sz, vi.vt)) return false;
        if (!parse_index(c, norm_sz, vi.vn)) return false;
        f.push_back(vi);
    }
    if (f.empty()) return false;
    faces.push_back(f);
    return true;
}

struct Exporter {
    const std::vector<float>&pos, &tex, &nrm;
    ObjMesh mesh;
    std::unordered_map<uint32_t, uint32_t> pmap, tmap, nmap;
    bool ok = true;
    Exporter(const std::vector<float>& p, const std::vector<float>& t, const std::vector<float>& n)
        : pos(p), tex(t), nrm(n) {}
    void add(const VertexIndices& v) {
        auto it = pmap.find(v.v);
        if (it != pmap.end()) {
            mesh.indices.push_back(it->second);
        } else {
            if ((size_t)v.v * 3 + 2 >= pos.size()) {
                ok = false;
                return;
            }
            uint32_t next = (uint32_t)pmap.size();
            mesh.positions.push_back(pos[v.v * 3]);
            mesh.positions.push_back(pos[v.v * 3 + 1]);
            mesh.positions.push_back(pos[v.v * 3 + 2]);
            mesh.indices.push_back(next);
            pmap.emplace(v.v, next);
        }
        if (!tex.empty() && v.vt != MISSING) {
            auto jt = tmap.find(v.vt);
            if (jt != tmap.end()) {
                mesh.texcoord_indices.push_back(jt->second);
            } else {
                if ((size_t)v.vt * 2 + 1 >= tex.size()) {
                    ok = false;
                    return;
                }
                uint32_t next = (uint32_t)tmap.size();
                mesh.texcoords.push_back(tex[v.vt * 2]);
                mesh.texcoords.push_back(tex[v.vt * 2 + 1]);
                mesh.texcoord_indices.push_back(next);
                tmap.emplace(v.vt, next);
            }
        }
        if (!nrm.empty() && v.vn != MISSING) {
            auto jt = nmap.find(v.vn);
            if (jt != nmap.end()) {
                mesh.normal_indices.push_back(jt->second);
            } else {
                if ((size_t)v.vn * 3 + 2 >= nrm.size()) {
                    ok = false;
                    return;
                }
                uint32_t next = (uint32_t)nmap.size();
                mesh.normals.push_back(nrm[v.vn * 3]);
                mesh.normals.push_back(nrm[v.vn * 3 + 1]);
                mesh.normals.push_back(nrm[v.vn * 3 + 2]);
                mesh.normal_indices.push_back(next);
                nmap.emplace(v.vn, next);
            }
        }
    }
};

// ≙ tobj export_faces_multi_index with triangulate = true
bool export_faces(const std::vector<float>& pos, const std::vector<float>& tex,
                  const std::vector<float>& nrm, const std::vector<Face>& faces, int mat_id,
                  ObjMesh& out) {
    Exporter ex(pos, tex, nrm);
    for (const Face& f : faces) {
        switch (f.size()) {
            case 1:
                ex.add(f[0]); ex.add(f[0]); ex.add(f[0]);
                break;
            case 2:
                ex.add(f[0]); ex.add(f[1]); ex.add(f[1]);
                break;
            case 3:
                ex.add(f[0]); ex.add(f[1]); ex.add(f[2]);
                break;
            case 4:
                ex.add(f[0]); ex.add(f[1]); ex.add(f[2]);
                ex.add(f[0]); ex.add(f[2]); ex.add(f[3]);
                break;
            default: {
                const VertexIndices& a = f[0];
                size_t b = 1;
                for (size_t c = 2; c < f.size(); ++c) {
                    ex.add(a); ex.add(f[b]); ex.add(f[c]);
                    b = c;
                }
            }
        }
        if (!ex.ok) return false;
    }
    out = std::move(ex.mesh);
    out.material_id = mat_id;
    return true;
}

std::string rest_after_keyword(const std::string& line) {
    // ≙ line.split_once(' ').1.trim()
    size_t a = 0;
    while (a < line.size() && isspace((unsigned char)line[a])) ++a;
    size_t sp = line.find(' ', a);
    if (sp == std::string::npos) {
        sp = line.find('\t', a);
        if (sp == std::string::npos) return "";
    }
    return trim(line.substr(sp + 1));
}

bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

}  // namespace

bool load_mtl_text(const std::string& text, std::vector<ObjMaterial>& out,
                   std::map<std::string, size_t>& name_map) {
    std::istringstream in(text);
    std::string raw;
    ObjMaterial cur;
    bool have = false;
    auto flush = [&]() {
        if (have) {
            name_map[cur.name] = out.size();
            out.push_back(cur);
        }
    };
    while (std::getline(in, raw)) {
        if (!raw.empty() && raw.back() == '\r') raw.pop_back();
        std::string line = trim(raw);
        std::vector<std::string> w = split_ws(line);
        if (w.empty() || w[0][0] == '#') continue;
        const std::string& k = w[0];
        if (k == "newmtl") {
            flush();
            cur = ObjMaterial();
            cur.name = trim(line.substr(6));
            if (cur.name.empty()) return false;
            have = true;
        } else if (k == "Ka") {
            std::vector<float> v;
            if (!parse_floatn(w, 1, v, 3)) return false;
        } else if (k == "Kd") {
            std::vector<float> v;
            if (!parse_floatn(w, 1, v, 3)) return false;
            cur.has_diffuse = true;
            memcpy(cur.diffuse, v.data(), 12);
        } else if (k == "Ks") {
            std::vector<float> v;
            if (!parse_floatn(w, 1, v, 3)) return false;
            cur.has_specular = true;
            memcpy(cur.specular, v.data(), 12);
        } else if (k == "Ns") {
            std::vector<float> v;
            if (!parse_floatn(w, 1, v, 1)) return false;
            cur.has_shininess = true;
            cur.shininess = v[0];
        } else if (k == "Ni") {
            std::vector<float> v;
            if (!parse_floatn(w, 1, v, 1)) return false;
            cur.has_optical_density = true;
            cur.optical_density = v[0];
        } else if (k == "d") {
            std::vector<float> v;
            if (!parse_floatn(w, 1, v, 1)) return false;
        } else if (k == "map_Kd") {
            std::string t = trim(line.substr(6));
            if (t.empty()) return false;
            cur.has_diffuse_texture = true;
            cur.diffuse_texture = t;
        } else if (k == "map_Ka" || k == "map_Ks" || k == "map_Ns" || k == "map_ns" ||
                   k == "bump" || k == "map_bump" || k == "map_Bump" || k == "map_d") {
            // parsed into fields the reference never reads
        } else if (k == "illum") {
            if (w.size() < 2) return false;
            char* end = nullptr;
            long v = strtol(w[1].c_str(), &end, 10);
            if (!end || *end != '\0' || v < 0 || v > 255) return false;
            cur.has_illum = true;
            cur.illumination_model = (int)v;
        } else {
            // ≙ unknown_param.insert(key, rest-of-line.trim())
            cur.unknown_param[k] = trim(line.substr(k.size()));
        }
    }
    flush();
    return true;
}

ObjLoadResult load_obj_text(const std::string& text, const std::string& mtl_dir) {
    ObjLoadResult res;
    std::vector<float> tmp_pos, tmp_tex, tmp_nrm, tmp_color;
    std::vector<Face> tmp_faces;
    std::string name = "unnamed_object";
    int mat_id = -1;
    std::map<std::string, size_t> mat_map;

    auto flush_model = [&](const std::string& nm) -> bool {
        ObjModel m;
        if (!export_faces(tmp_pos, tmp_tex, tmp_nrm, tmp_faces, mat_id, m.mesh)) return false;
        m.name = nm;
        res.models.push_back(std::move(m));
        tmp_faces.clear();
        return true;
    };

    std::istringstream in(text);
    std::string raw;
    while (std::getline(in, raw)) {
        if (!raw.empty() && raw.back() == '\r') raw.pop_back();
        std::vector<std::string> w = split_ws(raw);
        if (w.empty() || w[0] == "#") continue;
        const std::string& k = w[0];
        if (k == "v") {
            if (!parse_floatn(w, 1, tmp_pos, 3)) {
                res.error = "PositionParseError";
                return res;
            }
            parse_floatn(w, 4, tmp_color, 3);  // optional vertex colour, unused
        } else if (k == "vt") {
            if (!parse_floatn(w, 1, tmp_tex, 2)) {
                res.error = "TexcoordParseError";
                return res;
            }
        } else if (k == "vn") {
            if (!parse_floatn(w, 1, tmp_nrm, 3)) {
                res.error = "NormalParseError";
                return res;
            }
        } else if (k == "f" || k == "l") {
            if (!parse_face(w, tmp_faces, tmp_pos.size() / 3, tmp_tex.size() / 2,
                            tmp_nrm.size() / 3)) {
                res.error = "FaceParseError";
                return res;
            }
        } else if (k == "o" || k == "g") {
            if (!tmp_faces.empty()) {
                if (!flush_model(name)) {
                    res.error = "FaceVertexOutOfBounds";
                    return res;
                }
            }
            std::string line = trim(raw);
            name = trim(line.substr(1));
            if (name.empty()) name = "unnamed_object";
        } else if (k == "mtllib") {
            std::string lib = rest_after_keyword(raw);
            std::string mtl_text;
            std::string p = mtl_dir.empty() ? lib : (mtl_dir + "/" + lib);
            std::vector<ObjMaterial> mats;
            std::map<std::string, size_t> map;
            if (read_file(p, mtl_text) && load_mtl_text(mtl_text, mats, map)) {
                size_t off = res.materials.size();
                for (auto& m : mats) res.materials.push_back(m);
                for (auto& kv : map) mat_map[kv.first] = kv.second + off;
            } else {
                res.materials_ok = false;
            }
        } else if (k == "usemtl") {
            std::string mat_name = rest_after_keyword(raw);
            if (mat_name.empty()) {
                res.error = "MaterialParseError";
                return res;
            }
            auto it = mat_map.find(mat_name);
            int new_mat = it == mat_map.end() ? -1 : (int)it->second;
            if (mat_id != new_mat && !tmp_faces.empty()) {
                if (!flush_model(name)) {
                    res.error = "FaceVertexOutOfBounds";
                    return res;
                }
            }
            mat_id = new_mat;
        }
        // anything else is ignored
    }
    if (!flush_model(name)) res.error = "FaceVertexOutOfBounds";
    return res;
}

ObjLoadResult load_obj_file(const std::string& path) {
    std::string text;
    if (!read_file(path, text)) {
        ObjLoadResult r;
        r.error = "OpenFileFailed: " + path;
        return r;
    }
    std::string dir;
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) dir = path.substr(0, slash);
    return load_obj_text(text, dir);
}

}  // namespace rt2
