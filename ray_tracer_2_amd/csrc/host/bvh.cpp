// bvh.cpp -- see bvh.h.  Line references are to src/core/bvh.rs.
#include "bvh.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>

namespace rt2 {
namespace {

struct BuildTri {  // bvh.rs:11-17
    Vec3 centroid, mn, mx;
    int32_t i;
};

struct Builder {
    std::vector<BuildTri> tris;
    std::vector<rt_node> nodes;
    Quality quality;

    static void fit_bounds(float mn[3], float mx[3], const BuildTri& t) {  // :292-297
        for (int a = 0; a < 3; ++a) {
            mn[a] = fmin32(mn[a], t.mn[a]);
            mx[a] = fmax32(mx[a], t.mx[a]);
        }
    }

    static float node_cost(const rt_node& n) {  // :69-73
        float ex = n.aabb_max[0] - n.aabb_min[0];
        float ey = n.aabb_max[1] - n.aabb_min[1];
        float ez = n.aabb_max[2] - n.aabb_min[2];
        float half_area = (ex * ey + ey * ez) + ex * ez;
        return half_area * (float)n.count;
    }

    float evaluate_sah(int axis, float pos, size_t start, size_t count) const {  // :352-370
        const float inf = std::numeric_limits<float>::infinity();
        Vec3 lmn{inf, inf, inf}, lmx{-inf, -inf, -inf}, rmn{inf, inf, inf}, rmx{-inf, -inf, -inf};
        float left_count = 0.0f, right_count = 0.0f;
        size_t end = start + count;
        for (size_t i = start; i < end; ++i) {
            const BuildTri& t = tris[i];
            if (t.centroid[axis] < pos) {
                left_count += 1.0f;
                lmn = vmin(lmn, t.mn);
                lmx = vmax(lmx, t.mx);
            } else {
                right_count += 1.0f;
                rmn = vmin(rmn, t.mn);
                rmx = vmax(rmx, t.mx);
            }
        }
        Vec3 le = lmx - lmn, re = rmx - rmn;  // Aabb::half_area :87-90
        float lha = (le.x * le.y + le.y * le.z) + le.x * le.z;
        float rha = (re.x * re.y + re.y * re.z) + re.x * re.z;
        return left_count * lha + right_count * rha;
    }

    float find_best_split(const rt_node& node, int& axis, float& split_pos, size_t start,
                          size_t count) const {  // :299-351
        const float inf = std::numeric_limits<float>::infinity();
        if (node.count <= 1) {
            axis = 0;
            split_pos = 0.0f;
            return inf;
        }
        float bounds[3];
        for (int a = 0; a < 3; ++a) bounds[a] = node.aabb_max[a] - node.aabb_min[a];
        switch (quality) {
            case Quality::Low: {
                axis = (bounds[0] > bounds[1] && bounds[0] > bounds[2]) ? 0
                       : (bounds[1] > bounds[2] ? 1 : 2);
                split_pos = node.aabb_min[axis] + bounds[axis] * 0.5f;
                return evaluate_sah(axis, split_pos, start, count);
            }
            case Quality::High: {
                float best_cost = inf;
                float max_axis = fmax32(bounds[0], fmax32(bounds[1], bounds[2]));
                for (int a = 0; a < 3; ++a) {
                    float axis_size = bounds[a];
                    float axis_min = node.aabb_min[a];
                    if (axis_size == 0.0f) continue;
                    // `as u32` saturates; NaN -> 0
                    float c = std::ceil(axis_size / max_axis * 50.0f);
                    uint32_t n_tests;
                    if (!(c == c) || c <= 0.0f) n_tests = 0;
                    else if (c >= 4294967296.0f) n_tests = 0xffffffffu;
                    else n_tests = (uint32_t)c;
                    if (n_tests < 1) n_tests = 1;
                    if (n_tests > 50) n_tests = 50;
                    for (uint32_t i = 0; i < n_tests; ++i) {
                        float split_t = (float)(i + 1) / ((float)n_tests + 1.0f);
                        float test_pos = axis_min + axis_size * split_t;
                        float cost = evaluate_sah(a, test_pos, start, count);
                        if (cost < best_cost) {
                            split_pos = test_pos;
                            axis = a;
                            best_cost = cost;
                        }
                    }
                }
                return best_cost;
            }
            default: return inf;
        }
    }

    void subdivide(size_t node_idx, size_t start, size_t n_tris, uint64_t depth) {  // :372-470
        float parent_cost = node_cost(nodes[node_idx]);
        int axis = 0;
        float split_pos = 0.0f;
        float cost = find_best_split(nodes[node_idx], axis, split_pos, start, n_tris);
        if (cost < parent_cost && depth < 32) {
            float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            float rmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            size_t left_count = 0;
            for (size_t i = start; i < start + n_tris; ++i) {
                const BuildTri& t = tris[i];
                if (t.centroid[axis] < split_pos) {
                    fit_bounds(lmn, lmx, t);
                    std::swap(tris[start + left_count], tris[i]);
                    left_count += 1;
                } else {
                    fit_bounds(rmn, rmx, t);
                }
            }
            uint32_t right_count = (uint32_t)(n_tris - left_count);
            uint32_t left_first = (uint32_t)start;
            uint32_t right_first = left_first + (uint32_t)left_count;
            uint32_t left_index = (uint32_t)nodes.size();
            uint32_t right_index = left_index + 1;
            rt_node l{}, r{};
            for (int a = 0; a < 3; ++a) {
                l.aabb_min[a] = lmn[a];
                l.aabb_max[a] = lmx[a];
                r.aabb_min[a] = rmn[a];
                r.aabb_max[a] = rmx[a];
            }
            l.first = left_first;
            l.count = (uint32_t)left_count;
            r.first = right_first;
            r.count = right_count;
            nodes.push_back(l);
            nodes.push_back(r);
            nodes[node_idx].left = left_index;
            nodes[node_idx].right = right_index;
            nodes[node_idx].count = 0;
            subdivide(left_index, start, left_count, depth + 1);
            subdivide(right_index, start + left_count, right_count, depth + 1);
        }
    }
};

void put3(float dst[3], Vec3 v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

}  // namespace

BvhResult bvh_build(const std::vector<Vertex>& vertices, const std::vector<uint32_t>& indices,
                    Quality quality) {
    BvhResult out;
    size_t n_tris = indices.size() / 3;
    if (n_tris == 0) return out;  // BVH::empty() :216-218
    Builder b;
    b.quality = quality;
    b.tris.resize(n_tris);
    for (size_t j = 0; j < n_tris; ++j) {  // :221-242
        size_t i = j * 3;
        Vec3 v1 = vertices[indices[i]].pos, v2 = vertices[indices[i + 1]].pos,
             v3 = vertices[indices[i + 2]].pos;
        BuildTri t;
        t.centroid = ((v1 + v2) + v3) * (1.0f / 3.0f);
        t.mx = vmax(v1, vmax(v2, v3));
        t.mn = vmin(v1, vmin(v2, v3));
        t.i = (int32_t)i;
        b.tris[j] = t;
    }
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};  // :244-250
    for (const BuildTri& t : b.tris) Builder::fit_bounds(mn, mx, t);
    rt_node root{};
    for (int a = 0; a < 3; ++a) {
        root.aabb_min[a] = mn[a];
        root.aabb_max[a] = mx[a];
    }
    root.count = (uint32_t)n_tris;
    b.nodes.push_back(root);
    if (quality == Quality::Disabled) {
        // :270-273 returns before packing: a Disabled build has nodes but no
        // packed triangles.  Kept as is.
        out.nodes = std::move(b.nodes);
        return out;
    }
    b.subdivide(0, 0, n_tris, 0);
    out.triangles.resize(n_tris);
    for (size_t k = 0; k < n_tris; ++k) {  // :278-287, PackedTriangle::new :37-52
        const BuildTri& t = b.tris[k];
        const Vertex& a = vertices[indices[t.i]];
        const Vertex& bb = vertices[indices[t.i + 1]];
        const Vertex& c = vertices[indices[t.i + 2]];
        rt_packed_triangle p;
        put3(p.v1, a.pos); put3(p.v2, bb.pos); put3(p.v3, c.pos);
        put3(p.n1, a.normal); put3(p.n2, bb.normal); put3(p.n3, c.normal);
        p.uv10 = a.uv[0]; p.uv11 = a.uv[1];
        p.uv20 = bb.uv[0]; p.uv21 = bb.uv[1];
        p.uv30 = c.uv[0]; p.uv31 = c.uv[1];
        out.triangles[k] = p;
    }
    out.nodes = std::move(b.nodes);
    return out;
}

// ---------------------------------------------------------------------------------------------
// level-wise build
// ---------------------------------------------------------------------------------------------
namespace {
std::vector<BuildTri> make_build_tris(const std::vector<Vertex>& vertices, const std::vector<uint32_t>& indices) {
    size_t n_tris = indices.size() / 3;
    std::vector<BuildTri> tris(n_tris);
    for (size_t j = 0; j < n_tris; ++j) {  // bvh.rs:221-242
        size_t i = j * 3;
        Vec3 v1 = vertices[indices[i]].pos, v2 = vertices[indices[i + 1]].pos, v3 = vertices[indices[i + 2]].pos;
        BuildTri t;
        t.centroid = ((v1 + v2) + v3) * (1.0f / 3.0f);
        t.mx = vmax(v1, vmax(v2, v3));
        t.mn = vmin(v1, vmin(v2, v3));
        t.i = (int32_t)i;
        tris[j] = t;
    }
    return tris;
}
}  // namespace

std::vector<float> bvh_search_data(const std::vector<Vertex>& vertices, const std::vector<uint32_t>& indices) {
    std::vector<BuildTri> tris = make_build_tris(vertices, indices);
    std::vector<float> out(tris.size() * 9);
    for (size_t j = 0; j < tris.size(); ++j) {
        float* o = out.data() + j * 9;
        o[0] = tris[j].centroid.x; o[1] = tris[j].centroid.y; o[2] = tris[j].centroid.z;
        o[3] = tris[j].mn.x; o[4] = tris[j].mn.y; o[5] = tris[j].mn.z;
        o[6] = tris[j].mx.x; o[7] = tris[j].mx.y; o[8] = tris[j].mx.z;
    }
    return out;
}

LevelSearch make_host_level_search(const float* tri9, size_t n_tris) {
    std::vector<float> data(tri9, tri9 + n_tris * 9);
    return [data](const uint32_t* order, size_t n, const std::vector<SplitQuery>& qs, std::vector<SplitResult>& out) {
        Builder b;
        b.quality = Quality::High;
        b.tris.resize(n);
        for (size_t p = 0; p < n; ++p) {
            const float* d = data.data() + (size_t)order[p] * 9;
            b.tris[p].centroid = Vec3{d[0], d[1], d[2]};
            b.tris[p].mn = Vec3{d[3], d[4], d[5]};
            b.tris[p].mx = Vec3{d[6], d[7], d[8]};
            b.tris[p].i = (int32_t)order[p] * 3;
        }
        out.resize(qs.size());
        for (size_t k = 0; k < qs.size(); ++k) {
            rt_node node{};
            memcpy(node.aabb_min, qs[k].aabb_min, 12);
            memcpy(node.aabb_max, qs[k].aabb_max, 12);
            node.count = qs[k].count;
            int axis = 0;
            float pos = 0.0f;
            out[k].cost = b.find_best_split(node, axis, pos, qs[k].start, qs[k].count);
            out[k].axis = axis;
            out[k].pos = pos;
        }
    };
}

BvhResult bvh_build_levels(const std::vector<Vertex>& vertices, const std::vector<uint32_t>& indices, Quality quality,
                           const LevelSearch& search) {
    if (quality != Quality::High || !search) return bvh_build(vertices, indices, quality);
    BvhResult out;
    size_t n_tris = indices.size() / 3;
    if (n_tris == 0) return out;
    Builder b;
    b.quality = quality;
    b.tris = make_build_tris(vertices, indices);
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};  // bvh.rs:244-250
    for (const BuildTri& t : b.tris) Builder::fit_bounds(mn, mx, t);
    // nodes in breadth-first order while building; renumbered to the reference's order at the end
    struct Open { uint32_t node, start, count, depth; };
    std::vector<rt_node> nodes;
    rt_node root{};
    for (int a = 0; a < 3; ++a) {
        root.aabb_min[a] = mn[a];
        root.aabb_max[a] = mx[a];
    }
    root.count = (uint32_t)n_tris;
    nodes.push_back(root);
    std::vector<Open> level{Open{0, 0, (uint32_t)n_tris, 0}};
    std::vector<uint32_t> order(n_tris);
    std::vector<SplitQuery> qs;
    std::vector<SplitResult> rs;
    while (!level.empty()) {
        // find_best_split of every node of the level (count <= 1: infinite cost, bvh.rs:301-303)
        qs.clear();
        std::vector<uint32_t> asked;
        for (uint32_t k = 0; k < level.size(); ++k) {
            if (level[k].count <= 1 || level[k].depth >= 32) continue;  // (depth 32 never splits: :384)
            SplitQuery q;
            q.start = level[k].start;
            q.count = level[k].count;
            memcpy(q.aabb_min, nodes[level[k].node].aabb_min, 12);
            memcpy(q.aabb_max, nodes[level[k].node].aabb_max, 12);
            qs.push_back(q);
            asked.push_back(k);
        }
        if (qs.empty()) break;
        for (size_t p = 0; p < n_tris; ++p) order[p] = (uint32_t)(b.tris[p].i / 3);
        search(order.data(), n_tris, qs, rs);
        std::vector<Open> next;
        for (size_t j = 0; j < asked.size(); ++j) {
            const Open o = level[asked[j]];
            const float parent_cost = Builder::node_cost(nodes[o.node]);
            const int axis = rs[j].axis;
            const float split_pos = rs[j].pos;
            if (!(rs[j].cost < parent_cost)) continue;  // bvh.rs:384
            // the partition of subdivide (bvh.rs:385-400), operation for operation
            float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            float rmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            size_t left_count = 0;
            for (size_t i = o.start; i < (size_t)o.start + o.count; ++i) {
                const BuildTri& t = b.tris[i];
                if (t.centroid[axis] < split_pos) {
                    Builder::fit_bounds(lmn, lmx, t);
                    std::swap(b.tris[o.start + left_count], b.tris[i]);
                    left_count += 1;
                } else {
                    Builder::fit_bounds(rmn, rmx, t);
                }
            }
            rt_node l{}, r{};
            for (int a = 0; a < 3; ++a) {
                l.aabb_min[a] = lmn[a];
                l.aabb_max[a] = lmx[a];
                r.aabb_min[a] = rmn[a];
                r.aabb_max[a] = rmx[a];
            }
            l.first = o.start;
            l.count = (uint32_t)left_count;
            r.first = o.start + (uint32_t)left_count;
            r.count = o.count - (uint32_t)left_count;
            const uint32_t li = (uint32_t)nodes.size();
            nodes.push_back(l);
            nodes.push_back(r);
            nodes[o.node].left = li;
            nodes[o.node].right = li + 1;
            nodes[o.node].count = 0;
            next.push_back(Open{li, l.first, l.count, o.depth + 1});
            next.push_back(Open{li + 1, r.first, r.count, o.depth + 1});
        }
        level.swap(next);
    }
    // The reference numbers nodes in the order subdivide creates them: both children of a node,
    // then the left subtree, then the right one (bvh.rs:402-468).
    std::vector<uint32_t> renum(nodes.size(), 0xffffffffu);
    std::vector<uint32_t> stack{0u};
    uint32_t next_index = 1;
    renum[0] = 0;
    while (!stack.empty()) {
        const uint32_t n = stack.back();
        stack.pop_back();
        if (nodes[n].count == 0 && (nodes[n].left != 0 || nodes[n].right != 0)) {
            renum[nodes[n].left] = next_index;
            renum[nodes[n].right] = next_index + 1;
            next_index += 2;
            stack.push_back(nodes[n].right);
            stack.push_back(nodes[n].left);
        }
    }
    out.nodes.resize(nodes.size());
    for (size_t n = 0; n < nodes.size(); ++n) {
        rt_node v = nodes[n];
        if (v.count == 0 && (v.left != 0 || v.right != 0)) {
            v.left = renum[v.left];
            v.right = renum[v.right];
        }
        out.nodes[renum[n]] = v;
    }
    out.triangles.resize(n_tris);
    for (size_t k = 0; k < n_tris; ++k) {  // bvh.rs:278-287
        const BuildTri& t = b.tris[k];
        const Vertex& a = vertices[indices[t.i]];
        const Vertex& bb = vertices[indices[t.i + 1]];
        const Vertex& c = vertices[indices[t.i + 2]];
        rt_packed_triangle p;
        put3(p.v1, a.pos); put3(p.v2, bb.pos); put3(p.v3, c.pos);
        put3(p.n1, a.normal); put3(p.n2, bb.normal); put3(p.n3, c.normal);
        p.uv10 = a.uv[0]; p.uv11 = a.uv[1];
        p.uv20 = bb.uv[0]; p.uv21 = bb.uv[1];
        p.uv30 = c.uv[0]; p.uv31 = c.uv[1];
        out.triangles[k] = p;
    }
    return out;
}

}  // namespace rt2
