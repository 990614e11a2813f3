// rt_kernel.hip -- the hot path: hand-written HIP for gfx950 (MI355X) of the
// per-pixel path-trace loop of shaders/ray_tracer.wgsl (reference).
//
// Mapping to CDNA4
//   * one lane per pixel, one 64-lane wavefront per 8x8 pixel tile (the
//     reference's @workgroup_size(8,8), wgsl:145); a block is one wave.
//   * blockIdx -> tile mapping is XCD-aware: consecutive tiles go to the same
//     XCD so neighbouring tiles share that XCD's L2 (matters once the BVH no
//     longer fits L1).
//   * the mesh loop (wgsl:369) is wave-uniform: mesh records, root nodes and
//     the triangles of root-leaf meshes are read through scalar loads (SGPR
//     operands), only diverged BVH levels use per-lane vector loads.
//   * per-lane BVH stacks live in LDS, lane-interleaved (entry k of lane l at
//     dword k*64 + l: conflict-free), with the near child kept in a register.
//   * a path is a small state machine (ray generation / segment / shading) so
//     lanes whose path ended start their pixel's next sample at once instead
//     of idling until the longest path of the wave ends.  The samples of one
//     pixel share one sequential RNG stream (wgsl:475,487-497), so they
//     cannot be spread over lanes.
//   * MFMA is deliberately unused: there is no dense contraction here.
//
// Arithmetic contract: every floating-point operation below is the IEEE
// binary32 operation the shader text prescribes, in the shader's order, with
// contraction disabled (-ffp-contract=off); implementation-defined builtins
// come from rt_transc.h / rt_texture.h.  Re-layouts and hoists are limited to
// ones that provably produce the same bits (noted inline).  The CPU oracle
// (oracle/shader_oracle.cpp) is an independent literal restatement; the
// parity tests require bit-identical images.
#include <hip/hip_runtime.h>

#include "rt_device.h"
#include "rt_texture.h"
#include "rt_transc.h"

namespace rtd {
namespace {

#define DEV __device__ __forceinline__

struct f3 {
    float x, y, z;
};
struct f4 {
    float x, y, z, w;
};

DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
DEV f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
DEV f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
DEV f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
DEV f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
DEV f4 operator*(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
DEV f4 operator+(f4 a, float s) { return {a.x + s, a.y + s, a.z + s, a.w + s}; }

constexpr float INF = 0x1p+127f;  // wgsl:132
constexpr float EPSILON = 1e-5f;  // wgsl:131

DEV float min_(float a, float b) { return __builtin_fminf(a, b); }
DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }
DEV float sign_(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV f3 cross3(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
DEV f3 normalize3(f3 a) { return a / rtm::sqrt_(dot3(a, a)); }
DEV f3 mix3(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
DEV f4 mix4(f4 a, f4 b, float t) { return a * (1.0f - t) + b * t; }
DEV f3 reflect3(f3 I, f3 N) { return I - (2.0f * dot3(N, I)) * N; }
DEV f3 refract3(f3 I, f3 N, float eta) {
    float d = dot3(N, I);
    float k = 1.0f - (eta * eta) * (1.0f - d * d);
    if (k < 0.0f) return {0.0f, 0.0f, 0.0f};
    return eta * I - (eta * d + rtm::sqrt_(k)) * N;
}
DEV float smoothstep_(float lo, float hi, float x) {
    float t = min_(max_((x - lo) / (hi - lo), 0.0f), 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}
// (mat4 * vec4(v, w)).xyz, column-major m[col*4 + row]
DEV f3 mat_xyz(const float* __restrict__ m, f3 v, float w) {
    f3 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * w;
    return r;
}

// ---- RNG (wgsl:164-206) ---------------------------------------------------
DEV uint32_t next_random_number(uint32_t& s) {
    s = s * 747796405u + 2891336453u;
    uint32_t r = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (r >> 22u) ^ r;
}
// f32(r) / 4294967295.0: the divisor rounds to 2^32 in f32 and a division by
// a power of two is exact, so this equals the multiplication by 2^-32 bit for
// bit (tests/test_kernel_units.py checks it over the full u32 range).
DEV float rand_(uint32_t& s) { return (float)next_random_number(s) * 0x1p-32f; }
DEV float rand_normal_dist(uint32_t& s) {
    float theta = 6.28318500518798828125f * rand_(s);  // f32(2.0 * 3.1415926)
    float rho = rtm::sqrt_(-2.0f * rtm::log_(rand_(s)));
    return rho * rtm::cos_(theta);
}
DEV f3 rand_unit_sphere(uint32_t& s) {
    float x = rand_normal_dist(s);
    float y = rand_normal_dist(s);
    float z = rand_normal_dist(s);
    return normalize3(f3{x, y, z});
}
DEV void rand_in_unit_disk(uint32_t& s, float& ox, float& oy) {
    float angle = (rand_(s) * 2.0f) * 3.1415926f;
    float c = rtm::cos_(angle), sn = rtm::sin_(angle);
    float r = rtm::sqrt_(rand_(s));
    ox = c * r;
    oy = sn * r;
}

// ---- intersection ---------------------------------------------------------
struct MeshBest {  // closest triangle hit inside the mesh being traversed
    float t, u, v, w, det;
    uint32_t tri;  // mesh-local triangle index
};

// wgsl:258-290 against the pre-laid-out record (rt_device.h): edge_ab, edge_ac
// and their cross product are the same IEEE operations wgsl:261-263 evaluate,
// hoisted to upload time.  Shading outputs are deferred to the winner.
DEV void tri_test(f3 lo, f3 ld, float4 q0, float4 q1, float4 q2, bool cull, uint32_t idx,
                  MeshBest& b) {
    f3 v1{q0.x, q0.y, q0.z}, n{q0.w, q1.w, q2.w};
    f3 eab{q1.x, q1.y, q1.z}, eac{q2.x, q2.y, q2.z};
    f3 ao = lo - v1;
    f3 dao = cross3(ao, ld);
    float det = -dot3(ld, n);
    bool keep = cull ? (det >= 1e-8f) : (rtm::abs_(det) >= 1e-8f);
    if (keep) {
        float inv = 1.0f / det;
        float dst = dot3(ao, n) * inv;
        float u = dot3(eac, dao) * inv;
        float v = -dot3(eab, dao) * inv;
        float w = (1.0f - u) - v;
        if (dst > EPSILON && u >= 0.0f && v >= 0.0f && w >= 0.0f && dst < b.t) {
            b.t = dst;
            b.u = u;
            b.v = v;
            b.w = w;
            b.det = det;
            b.tri = idx;
        }
    }
}

// wgsl:337-351
DEV float aabb_dist(f3 lo, f3 inv, float4 bmin, float4 bmax, float t) {
    float t1x = (bmin.x - lo.x) * inv.x, t1y = (bmin.y - lo.y) * inv.y, t1z = (bmin.z - lo.z) * inv.z;
    float t2x = (bmax.x - lo.x) * inv.x, t2y = (bmax.y - lo.y) * inv.y, t2z = (bmax.z - lo.z) * inv.z;
    float t_near = max_(max_(min_(t1x, t2x), min_(t1y, t2y)), min_(t1z, t2z));
    float t_far = min_(min_(max_(t1x, t2x), max_(t1y, t2y)), max_(t1z, t2z));
    bool did_hit = t_far >= t_near && t_near < t && t_far > 0.0f;
    return did_hit ? t_near : INF;
}

// wgsl:292-335 for one mesh.  `stack` points at this lane's LDS column
// (stride 64 dwords).  Visit order per lane is the shader's: far child
// pushed, near child visited next (kept in a register instead of a push/pop
// pair), nodes popped without re-testing.
template <bool STATS>
DEV void traverse_mesh(const RenderArgs& a, const DMesh* __restrict__ m, f3 lo, f3 ld, f3 inv,
                       uint32_t* stack, MeshBest& best, int& node_tests, int& tri_tests) {
    const float4* __restrict__ nodes = a.nodes + (size_t)m->node_offset * 3;
    const float4* __restrict__ tris = a.tri_isect + (size_t)m->tri_offset * 3;
    const bool cull = (m->flags & DMESH_GLASS) == 0;
    const uint32_t root_count = m->root_count;
    if (root_count > 0) {
        // Root is a leaf: every lane tests the same triangles (uniform
        // addresses -> scalar loads).
        const uint32_t first = __float_as_uint(nodes[0].z);
        if (STATS) tri_tests += (int)root_count;
        for (uint32_t j = 0; j < root_count; ++j) {
            const float4* t = tris + (size_t)(first + j) * 3;
            tri_test(lo, ld, t[0], t[1], t[2], cull, first + j, best);
        }
        return;
    }
    uint32_t cur = 0, sp = 0;
    for (;;) {
        const float4 h = nodes[(size_t)cur * 3];
        const uint32_t count = __float_as_uint(h.w);
        if (count > 0) {
            const uint32_t first = __float_as_uint(h.z);
            if (STATS) tri_tests += (int)count;
            for (uint32_t j = 0; j < count; ++j) {
                const float4* t = tris + (size_t)(first + j) * 3;
                tri_test(lo, ld, t[0], t[1], t[2], cull, first + j, best);
            }
            if (sp == 0) break;
            --sp;
            cur = stack[sp * 64];
        } else {
            const uint32_t ia = __float_as_uint(h.x), ib = __float_as_uint(h.y);
            const float4* na = nodes + (size_t)ia * 3;
            const float4* nb = nodes + (size_t)ib * 3;
            float da = aabb_dist(lo, inv, na[1], na[2], best.t);
            float db = aabb_dist(lo, inv, nb[1], nb[2], best.t);
            if (STATS) node_tests += 2;
            bool left_closer = da < db;
            float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
            uint32_t near_i = left_closer ? ia : ib, far_i = left_closer ? ib : ia;
            if (far_d < best.t) {
                stack[sp * 64] = far_i;
                ++sp;
            }
            if (near_d < best.t) {
                cur = near_i;
            } else {
                if (sp == 0) break;
                --sp;
                cur = stack[sp * 64];
            }
        }
    }
}

struct Hit {
    bool hit;
    float dst;
    f3 point, normal;
    float u, v;       // texture coordinates
    bool backface;
    int object;       // >= 0: mesh index, < 0: sphere -(index) - 1
};

// wgsl:353-396 (+ ray_sphere :223-256).  Per-object outputs that only the
// overall winner needs (normals, uv) are computed once after the loops from
// the same inputs, which yields the same bits.
template <bool STATS>
DEV Hit intersect_scene(const RenderArgs& a, f3 ro, f3 rd, uint32_t* stack, int& node_tests,
                        int& tri_tests) {
    float closest = INF;
    int object = 0;
    bool any = false;
    // spheres
    float s_dst = 0.0f;
    bool s_inside = false;
    for (uint32_t i = 0; i < a.n_spheres; ++i) {
        const DSphere sp = a.spheres[i];
        f3 oc = ro - f3{sp.cx, sp.cy, sp.cz};
        float qa = dot3(rd, rd);
        float qb = 2.0f * dot3(oc, rd);
        float qc = dot3(oc, oc) - sp.radius * sp.radius;
        float disc = qb * qb - (4.0f * qa) * qc;
        if (disc >= 0.0f) {
            float s = rtm::sqrt_(disc);
            float dst_near = max_(0.0f, (-qb - s) / (2.0f * qa));
            float dst_far = (-qb + s) / (2.0f * qa);
            if (dst_far >= 0.001f) {
                bool inside = dst_near == 0.0f;
                float dst = inside ? dst_far : dst_near;
                if (dst < closest) {
                    closest = dst;
                    any = true;
                    object = -(int)i - 1;
                    s_dst = dst;
                    s_inside = inside;
                }
            }
        }
    }
    // meshes
    f3 lo{0, 0, 0}, ld{0, 0, 0}, inv{0, 0, 0};
    MeshBest win{};  // winner's triangle data
    f3 win_point{0, 0, 0};
    for (uint32_t i = 0; i < a.n_meshes; ++i) {
        const DMesh* __restrict__ m = a.meshes + i;
        if ((m->flags & DMESH_SAME_XFORM) == 0) {
            // identical matrix => identical local ray: reuse (same bits)
            lo = mat_xyz(m->w2m, ro, 1.0f);
            ld = normalize3(mat_xyz(m->w2m, rd, 0.0f));
            inv = f3{1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z};
        }
        MeshBest b;
        b.t = INF;
        b.tri = 0xffffffffu;
        b.u = b.v = b.w = b.det = 0.0f;
        traverse_mesh<STATS>(a, m, lo, ld, inv, stack, b, node_tests, tri_tests);
        if (b.tri != 0xffffffffu) {
            f3 lhp = lo + ld * b.t;
            f3 whp = mat_xyz(m->m2w, lhp, 1.0f);
            f3 dv = ro - whp;
            float wdst = rtm::sqrt_(dot3(dv, dv));
            if (wdst < closest) {
                closest = wdst;
                any = true;
                object = (int)i;
                win = b;
                win_point = whp;
            }
        }
    }
    Hit h;
    h.hit = any;
    h.dst = closest;
    h.object = object;
    h.point = f3{0, 0, 0};
    h.normal = f3{0, 0, 0};
    h.u = h.v = 0.0f;
    h.backface = false;
    if (any) {
        if (object >= 0) {
            const DMesh* __restrict__ m = a.meshes + object;
            const float4* __restrict__ sh = a.tri_shade + ((size_t)m->tri_offset + win.tri) * 4;
            float4 s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3];
            f3 n1{s0.x, s0.y, s0.z}, n2{s1.x, s1.y, s1.z}, n3{s2.x, s2.y, s2.z};
            f3 ln = normalize3((n1 * win.w + n2 * win.u) + n3 * win.v) * sign_(win.det);
            h.normal = normalize3(mat_xyz(m->m2w, ln, 0.0f));
            h.backface = win.det < 0.0f;
            h.point = win_point;
            // uv = (uv1 * w + uv2 * u) + uv3 * v, uv1 = (u10,u11), uv2 = (u20,u21), uv3 = (u30,u31)
            h.u = (s0.w * win.w + s2.w * win.u) + s3.y * win.v;
            h.v = (s1.w * win.w + s3.x * win.u) + s3.z * win.v;
        } else {
            const DSphere sp = a.spheres[-object - 1];
            f3 c{sp.cx, sp.cy, sp.cz};
            h.point = ro + rd * s_dst;
            f3 n = normalize3(h.point - c);
            h.normal = s_inside ? -n : n;
            h.backface = s_inside;
            const float pi = 3.1415926f;
            float theta = rtm::acos_(-h.normal.y);
            float phi = rtm::atan2_(-h.normal.z, -h.normal.x) + pi;
            h.u = phi / (2.0f * pi);
            h.v = theta / pi;
        }
    }
    return h;
}

DEV f4 sample_texture(const RenderArgs& a, int index, float u, float v) {
    float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (index >= 0 && (uint32_t)index < a.n_textures) {
        const DTexture t = a.textures[index];
        rtm::TexView tv{t.rgba8, t.width, t.height};
        rtm::sample_bilinear(tv, a.srgb_lut, u, v, out);
    }
    return f4{out[0], out[1], out[2], out[3]};
}

// wgsl:214-221
DEV f4 environment_light(f3 dir) {
    const f4 SKY_HORIZON{1.0f, 1.0f, 1.0f, 0.0f};
    const f4 SKY_ZENITH{0.0788092f, 0.36480793f, 0.7264151f, 0.0f};
    const f4 GROUND{0.35f, 0.3f, 0.35f, 0.0f};
    float sky_t = rtm::pow_(smoothstep_(0.0f, 0.4f, dir.y), 0.35f);
    float g2s = smoothstep_(-0.01f, 0.0f, dir.y);
    f4 sky = mix4(SKY_HORIZON, SKY_ZENITH, sky_t);
    float sun = rtm::pow_(max_(0.0f, dot3(dir, f3{0.1f, 1.0f, 0.1f})), 500.0f) * 0.1f;
    return mix4(GROUND, sky, g2s) + sun * (g2s >= 1.0f ? 1.0f : 0.0f);
}

// wgsl:208-212
DEV float reflectance(float cos_theta, float ior) {
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0 *= r0;
    return r0 + (1.0f - r0) * rtm::pow_(1.0f - cos_theta, 5.0f);
}

// Blocks are dealt round-robin over the 8 XCDs (block b -> XCD b % 8).  Give
// each XCD a contiguous run of tiles: bijection for any grid size.
DEV uint32_t xcd_tile(uint32_t b, uint32_t nb) {
    uint32_t q = nb >> 3, r = nb & 7u;
    uint32_t x = b & 7u;
    uint32_t start = x * q + (x < r ? x : r);
    return start + (b >> 3);
}

struct PixelCoord {
    uint32_t x, y, out_row;
    bool valid;
};

DEV PixelCoord pixel_of_lane(const RenderArgs& a) {
    uint32_t tile = xcd_tile(blockIdx.x, gridDim.x);
    uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    uint32_t lane = threadIdx.x & 63u;
    uint32_t lx = lane & 7u, ly = lane >> 3;
    PixelCoord p;
    p.x = tx * 8u + lx;
    uint32_t strip = ty * a.strip_world + a.strip_rank;  // global 8-row strip
    p.y = strip * 8u + ly;
    p.out_row = a.strip_world > 1u ? ty * 8u + ly : p.y;
    p.valid = p.x < a.params.width && p.y < a.params.height;
    return p;
}

// wgsl:154-161
DEV void store_texel(const RenderArgs& a, const PixelCoord& p, f4 cur) {
    float4* texel = a.image + (size_t)p.out_row * a.params.width + p.x;
    if (a.params.frames >= 1) {
        float4 prev = *texel;
        float weight = 1.0f / (float)(a.params.frames + 1);
        float om = 1.0f - weight;
        *texel = make_float4(prev.x * om + cur.x * weight, prev.y * om + cur.y * weight,
                             prev.z * om + cur.z * weight, prev.w * om + cur.w * weight);
    } else {
        *texel = make_float4(cur.x, cur.y, cur.z, cur.w);
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// wgsl `main` + `frag` + `trace` (wgsl:144-162, 473-500, 398-471)
// ---------------------------------------------------------------------------
template <bool STATS>
__global__ void __launch_bounds__(64) rt_render_kernel(const RenderArgs a) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stack = lds_stack + (threadIdx.x & 63u);

    const PixelCoord px = pixel_of_lane(a);
    const float sx = (float)a.params.width, sy = (float)a.params.height;
    const float fx = (float)px.x, fy = (float)px.y;
    const int32_t fr = a.params.frames;
    const uint32_t absf = fr < 0 ? 0u - (uint32_t)fr : (uint32_t)fr;
    uint32_t rng = (uint32_t)(fy * sx + fx) + absf * 719393u;  // wgsl:475

    const float* __restrict__ c2w = &a.camera.cam_to_world[0][0];
    const f3 cam_origin{c2w[12], c2w[13], c2w[14]};
    const f3 cam_right{c2w[0], c2w[1], c2w[2]};
    const f3 cam_up{c2w[4], c2w[5], c2w[6]};
    const float uvx = fx / (sx - 1.0f), uvy = fy / (sy - 1.0f);
    const f3 local_focus = f3{uvx - 0.5f, uvy - 0.5f, 1.0f} *
                           f3{a.camera.view_params[0], a.camera.view_params[1], a.camera.view_params[2]};
    const f3 focus_point = mat_xyz(c2w, local_focus, 1.0f);

    const int32_t spp = a.params.rays_per_pixel;
    const int32_t nb = a.params.number_of_bounces;

    f4 total{0, 0, 0, 0};
    f4 light{0, 0, 0, 0}, T{1, 1, 1, 1};
    f3 ro{0, 0, 0}, rd{0, 0, 1};
    int32_t j = 0, seg = 0;
    bool fresh = true;
    bool active = px.valid && spp > 0;
    unsigned long long n_segments = 0;
    int node_tests = 0, tri_tests = 0;

    while (active) {
        if (fresh) {  // wgsl:487-495: next sample of this pixel
            float jx, jy;
            rand_in_unit_disk(rng, jx, jy);
            jx = jx * a.camera.defocus_strength / sx;
            jy = jy * a.camera.defocus_strength / sx;
            ro = (cam_origin + cam_right * jx) + cam_up * jy;
            float kx, ky;
            rand_in_unit_disk(rng, kx, ky);
            kx = kx * a.camera.diverge_strength / sx;
            ky = ky * a.camera.diverge_strength / sx;
            f3 jfp = (focus_point + cam_right * kx) + cam_up * ky;
            rd = normalize3(jfp - ro);
            rd = normalize3(rd);  // wgsl:400
            T = f4{1, 1, 1, 1};
            light = f4{0, 0, 0, 0};
            seg = 0;
            fresh = false;
        }
        bool end_path = true;
        if (seg <= nb) {
            Hit hit = intersect_scene<STATS>(a, ro, rd, stack, node_tests, tri_tests);
            n_segments += 1;
            if (!hit.hit) {
                if (a.params.skybox != 0) light = light + T * environment_light(rd);
            } else {
                const rt_material* __restrict__ mat =
                    hit.object >= 0 ? a.mesh_materials + hit.object : a.sphere_materials + (-hit.object - 1);
                const int flag = mat->flag;
                ro = hit.point;
                if (flag == RT_MATERIAL_GLASS) {  // wgsl:414-436
                    if (hit.backface) {
                        float as = mat->absorption_strength;
                        float ex = ((-hit.dst) * mat->absorption[0]) * as;
                        float ey = ((-hit.dst) * mat->absorption[1]) * as;
                        float ez = ((-hit.dst) * mat->absorption[2]) * as;
                        T = f4{T.x * rtm::exp_(ex), T.y * rtm::exp_(ey), T.z * rtm::exp_(ez), 1.0f};
                    }
                    float mior = mat->ior;
                    float ior = hit.backface ? mior : (1.0f / mior);
                    f3 reflect_dir = reflect3(rd, hit.normal);
                    f3 refract_dir = refract3(rd, hit.normal, ior);
                    float cos_theta = min_(dot3(-rd, hit.normal), 1.0f);
                    float sin_theta = rtm::sqrt_(1.0f - cos_theta * cos_theta);
                    bool cannot_refract = ior * sin_theta > 1.0f;
                    bool follow_reflection = cannot_refract;
                    if (!cannot_refract) follow_reflection = reflectance(cos_theta, ior) > rand_(rng);
                    f3 diffuse_dir = normalize3(hit.normal + rand_unit_sphere(rng));
                    reflect_dir = normalize3(mix3(diffuse_dir, reflect_dir, mat->specular));
                    refract_dir = normalize3(mix3(-diffuse_dir, refract_dir, mat->smoothness));
                    rd = follow_reflection ? reflect_dir : refract_dir;
                    ro = hit.point + (1e-4f * hit.normal) * sign_(dot3(hit.normal, rd));
                } else {  // wgsl:437-460
                    bool is_spec = mat->specular >= rand_(rng);
                    f3 sph = rand_unit_sphere(rng);
                    f3 diffuse_dir = sph * sign_(dot3(hit.normal, sph));
                    f3 specular_dir = reflect3(rd, hit.normal);
                    float es = mat->emission_strength;
                    f4 emitted{mat->emission_color[0] * es, mat->emission_color[1] * es,
                               mat->emission_color[2] * es, mat->emission_color[3] * es};
                    rd = normalize3(mix3(diffuse_dir, specular_dir, mat->smoothness * (is_spec ? 1.0f : 0.0f)));
                    light = light + emitted * T;
                    f4 color;
                    if (flag == RT_MATERIAL_TEXTURE && mat->diffuse_index != -1) {
                        color = sample_texture(a, mat->diffuse_index, hit.u, hit.v);
                    } else {
                        color = f4{mat->color[0], mat->color[1], mat->color[2], mat->color[3]};
                    }
                    f4 spec{mat->specular_color[0], mat->specular_color[1], mat->specular_color[2],
                            mat->specular_color[3]};
                    T = T * (is_spec ? spec : color);
                }
                float p = max_(T.x, max_(T.y, T.z));  // wgsl:462-466
                bool die = rand_(rng) >= p;
                if (!die) {
                    T = T * (1.0f / p);
                    seg += 1;
                    end_path = seg > nb;
                }
            }
        }
        if (end_path) {  // wgsl:496
            total = total + light;
            j += 1;
            fresh = true;
            active = j < spp;
        }
    }

    if (px.valid) {
        float n = (float)spp;
        store_texel(a, px, f4{total.x / n, total.y / n, total.z / n, total.w / n});
    }
    if (a.counters) {
        atomicAdd(&a.counters->segments, n_segments);
        if (STATS) {
            atomicAdd(&a.counters->node_tests, (unsigned long long)node_tests);
            atomicAdd(&a.counters->triangle_tests, (unsigned long long)tri_tests);
        }
    }
}

// ---------------------------------------------------------------------------
// wgsl debug_trace (wgsl:502-573): one primary ray, no RNG.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) rt_debug_kernel(const RenderArgs a) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stack = lds_stack + (threadIdx.x & 63u);
    const PixelCoord px = pixel_of_lane(a);
    if (!px.valid) return;
    const float sx = (float)a.params.width, sy = (float)a.params.height;
    const float fx = (float)px.x, fy = (float)px.y;
    const float* __restrict__ c2w = &a.camera.cam_to_world[0][0];
    const f3 cam_origin{c2w[12], c2w[13], c2w[14]};
    const float uvx = fx / (sx - 1.0f), uvy = fy / (sy - 1.0f);
    const f3 local_focus = f3{uvx - 0.5f, uvy - 0.5f, 1.0f} *
                           f3{a.camera.view_params[0], a.camera.view_params[1], a.camera.view_params[2]};
    const f3 focus_point = mat_xyz(c2w, local_focus, 1.0f);
    f3 rd = normalize3(focus_point - cam_origin);
    int s0 = 0, s1 = 0;
    Hit hit = intersect_scene<true>(a, cam_origin, rd, stack, s0, s1);
    const float scale = (float)a.params.debug_scale;
    f4 out{1.0f, 0.0f, 1.0f, 1.0f};
    switch (a.params.debug_flag) {
        case 5: {
            float d = (float)s0 / scale;
            out = d > 1.0f ? f4{1, 0, 0, 1} : f4{d, d, d, 1};
            break;
        }
        case 6: {
            float t = (float)s1 / scale;
            out = t > 1.0f ? f4{1, 0, 0, 1} : f4{t, t, t, 1};
            break;
        }
        case 2: {
            float d = hit.dst / scale;
            out = hit.hit ? f4{d, d, d, 1} : f4{0, 0, 0, 0};
            break;
        }
        case 1: {
            if (!hit.hit) {
                out = f4{0, 0, 0, 0};
            } else {
                const rt_material* mat =
                    hit.object >= 0 ? a.mesh_materials + hit.object : a.sphere_materials + (-hit.object - 1);
                if (mat->flag == RT_MATERIAL_TEXTURE && mat->normal_index != -1) {
                    f4 x = sample_texture(a, mat->normal_index, hit.u, hit.v);
                    out = f4{0.5f * (2.0f * x.x - 1.0f) + 0.5f, 0.5f * (2.0f * x.y - 1.0f) + 0.5f,
                             0.5f * (2.0f * x.z - 1.0f) + 0.5f, 1.0f};
                } else {
                    out = f4{hit.normal.x * 0.5f + 0.5f, hit.normal.y * 0.5f + 0.5f,
                             hit.normal.z * 0.5f + 0.5f, 1.0f};
                }
            }
            break;
        }
        case 7: {
            float d = (float)s0 / scale, t = (float)s1 / scale;
            out = f4{t, 0.0f, d, 1.0f};
            break;
        }
        case 4: {
            if (!hit.hit) {
                out = f4{0, 0, 0, 0};
            } else {
                float s = scale / 100.0f, d = hit.dst;
                out = d > s ? f4{0, 1, 0, 1} : f4{d, d, d, 1};
            }
            break;
        }
        case 3: {
            out = hit.hit ? f4{hit.u, hit.v, 0.0f, 1.0f} : f4{0, 0, 0, 0};
            break;
        }
        default: break;
    }
    store_texel(a, px, out);
    if (a.counters) {
        atomicAdd(&a.counters->segments, 1ull);
        atomicAdd(&a.counters->node_tests, (unsigned long long)s0);
        atomicAdd(&a.counters->triangle_tests, (unsigned long long)s1);
    }
}

// Scatter gathered strips (rank-major, each rank padded to `pad_texels`) into
// the full frame: strip s of the frame is local strip s / world of rank
// s % world.
__global__ void rt_assemble_kernel(const float4* __restrict__ gathered, float4* __restrict__ image,
                                   uint32_t width, uint32_t height, uint32_t world,
                                   unsigned long long pad_texels) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n = (unsigned long long)width * height;
    if (i >= n) return;
    uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i % width);
    uint32_t strip = y >> 3, ly = y & 7u;
    uint32_t rank = strip % world, ls = strip / world;
    image[i] = gathered[(unsigned long long)rank * pad_texels + (unsigned long long)(ls * 8u + ly) * width + x];
}

// Launchers (called from rt_api.hip)
hipError_t launch_render(const RenderArgs& a, hipStream_t stream) {
    uint32_t nblocks = a.tiles_x * a.tiles_y;
    if (nblocks == 0) return hipSuccess;
    size_t lds = (size_t)(a.stack_entries ? a.stack_entries : 1u) * 64u * sizeof(uint32_t);
    if (a.params.debug_flag != 0) {
        hipLaunchKernelGGL(rt_debug_kernel, dim3(nblocks), dim3(64), lds, stream, a);
    } else if (a.count_tests) {
        hipLaunchKernelGGL(rt_render_kernel<true>, dim3(nblocks), dim3(64), lds, stream, a);
    } else {
        hipLaunchKernelGGL(rt_render_kernel<false>, dim3(nblocks), dim3(64), lds, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_assemble(const float4* gathered, float4* image, uint32_t width, uint32_t height,
                           uint32_t world, unsigned long long pad_texels, hipStream_t stream) {
    unsigned long long n = (unsigned long long)width * height;
    if (n == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(rt_assemble_kernel, dim3(blocks), dim3(256), 0, stream, gathered, image, width,
                       height, world, pad_texels);
    return hipGetLastError();
}

}  // namespace rtd
