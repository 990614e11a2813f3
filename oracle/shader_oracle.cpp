// shader_oracle.cpp -- CPU ORACLE (test infrastructure, not product code).
//
// A literal, function-by-function restatement of the reference's compute
// shader shaders/ray_tracer.wgsl in scalar IEEE binary32 C++: same statements,
// same evaluation order, no FMA contraction (-ffp-contract=off), WGSL builtins
// written out from their WGSL-spec definitions.  It is
// deliberately independent of the HIP kernel's restructured code; the two only
// share the *definitions* of what WGSL leaves implementation-defined
// (ray_tracer_2_amd/csrc/rt_transc.h: log/cos/sin/exp/pow/acos/atan2;
// rt_texture.h: bilinear sRGB sampling).
//
// PARITY UNPINNED: the reference has no tests, golden vectors or fixtures, and
// neither its Rust host nor its WGSL shader can be built or run in the build
// container (no cargo/rustc, no naga/wgpu, no Vulkan ICD).  The
// oracle is pinned only by analytic checks (tests/test_oracle_golden.py)
// and by the fixtures it generated itself (tests/golden/, made by
// tests/golden/make_golden.py).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.  Every function cites the wgsl line range it restates.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rt_abi.h"
#include "../ray_tracer_2_amd/csrc/rt_srgb_lut.h"
#include "../ray_tracer_2_amd/csrc/rt_texture.h"
#include "../ray_tracer_2_amd/csrc/rt_transc.h"

namespace orc {

// ---- vector types with WGSL component-wise semantics ----------------------
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

static inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
static inline vec2 operator-(vec2 a, float b) { return {a.x - b, a.y - b}; }
static inline vec2 operator*(vec2 a, float b) { return {a.x * b, a.y * b}; }
static inline vec2 operator/(vec2 a, float b) { return {a.x / b, a.y / b}; }
static inline vec2 operator/(vec2 a, vec2 b) { return {a.x / b.x, a.y / b.y}; }

static inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline vec3 operator*(vec3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
static inline vec3 operator*(float a, vec3 b) { return {a * b.x, a * b.y, a * b.z}; }
static inline vec3 operator/(vec3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
static inline vec3 operator/(float a, vec3 b) { return {a / b.x, a / b.y, a / b.z}; }

static inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline vec4 operator+(vec4 a, float b) { return {a.x + b, a.y + b, a.z + b, a.w + b}; }
static inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline vec4 operator*(vec4 a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }
static inline vec4 operator/(vec4 a, float b) { return {a.x / b, a.y / b, a.z / b, a.w / b}; }

// ---- WGSL builtins (appendix A1) ------------------------------------------
static inline float wmin(float a, float b) { return __builtin_fminf(a, b); }
static inline float wmax(float a, float b) { return __builtin_fmaxf(a, b); }
static inline vec3 wmin(vec3 a, vec3 b) { return {wmin(a.x, b.x), wmin(a.y, b.y), wmin(a.z, b.z)}; }
static inline vec3 wmax(vec3 a, vec3 b) { return {wmax(a.x, b.x), wmax(a.y, b.y), wmax(a.z, b.z)}; }
static inline float wabs(float a) { return rtm::abs_(a); }
static inline float wsign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float length(vec3 a) { return rtm::sqrt_(dot(a, a)); }
// normalize(): precision is implementation-defined in WGSL; this build's canonical form is v * (1 / length(v)) with a
// correctly rounded reciprocal ("Canonical arithmetic")
static inline vec3 normalize(vec3 a) { return a * (1.0f / length(a)); }
static inline float distance(vec3 a, vec3 b) { return length(a - b); }
static inline vec3 reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
static inline vec3 refract(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - (eta * eta) * (1.0f - d * d);
    if (k < 0.0f) return {0.0f, 0.0f, 0.0f};
    return eta * I - (eta * d + rtm::sqrt_(k)) * N;
}
static inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline vec3 mix(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
static inline vec4 mix(vec4 a, vec4 b, float t) { return a * (1.0f - t) + b * t; }
static inline float clamp01(float x) { return wmin(wmax(x, 0.0f), 1.0f); }
static inline float smoothstep(float lo, float hi, float x) {
    float t = clamp01((x - lo) / (hi - lo));
    return (t * t) * (3.0f - 2.0f * t);
}
// mat4x4 * vec4, column-major: ((c0*x + c1*y) + c2*z) + c3*w
static inline vec3 mat_mul_xyz(const float m[4][4], vec3 v, float w) {
    vec3 r;
    r.x = ((m[0][0] * v.x + m[1][0] * v.y) + m[2][0] * v.z) + m[3][0] * w;
    r.y = ((m[0][1] * v.x + m[1][1] * v.y) + m[2][1] * v.z) + m[3][1] * w;
    r.z = ((m[0][2] * v.x + m[1][2] * v.y) + m[2][2] * v.z) + m[3][2] * w;
    return r;
}

// ---- shader structs (wgsl:84-105) -----------------------------------------
struct Ray {
    vec3 origin{0, 0, 0}, dir{0, 0, 0}, inv_dir{0, 0, 0};
    vec4 transmittance{0, 0, 0, 0};
    uint32_t bounces = 0;
};

struct Hit {
    bool hit = false;
    float dst = 0.0f;
    vec3 hit_point{0, 0, 0}, normal{0, 0, 0};
    vec2 uv{0, 0};
    bool backface = false;
    rt_material material{};
    int mesh = -1;  // oracle-only bookkeeping for transcripts
    int tri = -1;
};

static const float SRGB_LUT[256] = {RT_SRGB_LUT_VALUES};

struct Ctx {
    rt_params params;
    rt_scene_uniform scene;
    const rt_sphere* spheres;
    const rt_mesh_uniform* meshes;
    const rt_packed_triangle* triangles;
    const rt_node* nodes;
    const rtm::TexView* textures;
    uint32_t n_textures;
};

struct Transcript {  // per-segment record for golden path transcripts
    int32_t hit_mesh;
    int32_t hit_tri;
    float dst;
    uint32_t rng_after;
};
struct TranscriptBuf {
    Transcript* rec;
    uint32_t cap, n;
};

static const vec4 SKY_HORIZON{1.0f, 1.0f, 1.0f, 0.0f};                       // wgsl:126
static const vec4 SKY_ZENITH{0.0788092f, 0.36480793f, 0.7264151f, 0.0f};     // wgsl:127
static const vec4 GROUND_COLOR{0.35f, 0.3f, 0.35f, 0.0f};                    // wgsl:128
static const float SUN_INTENSITY = 0.1f, SUN_FOCUS = 500.0f, EPSILON = 1e-5f;
static const float INF = 0x1p+127f;                                          // wgsl:132

// wgsl:195-200
static inline uint32_t next_random_number(uint32_t* seed) {
    *seed = *seed * 747796405u + 2891336453u;
    uint32_t result = ((*seed >> ((*seed >> 28u) + 4u)) ^ *seed) * 277803737u;
    result = (result >> 22u) ^ result;
    return result;
}
// wgsl:164-166 -- the literal 4294967295.0 rounds to 2^32 in f32
static inline float rand_(uint32_t* seed) { return (float)next_random_number(seed) / 4294967295.0f; }
// wgsl:181-185
static inline float rand_normal_dist(uint32_t* seed) {
    float theta = (float)(2.0 * 3.1415926) * rand_(seed);
    float rho = rtm::sqrt_(-2.0f * rtm::log_(rand_(seed)));
    return rho * rtm::cos_(theta);
}
// wgsl:168-174 and :187-193 (identical bodies)
static inline vec3 rand_unit_sphere(uint32_t* seed) {
    float x = rand_normal_dist(seed);
    float y = rand_normal_dist(seed);
    float z = rand_normal_dist(seed);
    return normalize(vec3{x, y, z});
}
static inline vec3 rand_direction(uint32_t* seed) { return rand_unit_sphere(seed); }
// wgsl:176-179
static inline vec3 rand_hemisphere(vec3 normal, uint32_t* seed) {
    vec3 dir = rand_unit_sphere(seed);
    return dir * wsign(dot(normal, dir));
}
// wgsl:202-206
static inline vec2 rand_in_unit_disk(uint32_t* seed) {
    float angle = (rand_(seed) * 2.0f) * 3.1415926f;
    vec2 p{rtm::cos_(angle), rtm::sin_(angle)};
    return p * rtm::sqrt_(rand_(seed));
}
// wgsl:208-212
static inline float reflectance(float cos_theta, float ior) {
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0 *= r0;
    return r0 + (1.0f - r0) * rtm::pow_(1.0f - cos_theta, 5.0f);
}
// wgsl:214-221
static inline vec4 get_environment_light(const Ray& ray) {
    float sky_gradient_t = rtm::pow_(smoothstep(0.0f, 0.4f, ray.dir.y), 0.35f);
    float ground_to_sky_t = smoothstep(-0.01f, 0.0f, ray.dir.y);
    vec4 sky_gradient = mix(SKY_HORIZON, SKY_ZENITH, sky_gradient_t);
    float sun = rtm::pow_(wmax(0.0f, dot(ray.dir, vec3{0.1f, 1.0f, 0.1f})), SUN_FOCUS) * SUN_INTENSITY;
    vec4 composite = mix(GROUND_COLOR, sky_gradient, ground_to_sky_t) +
                     sun * (ground_to_sky_t >= 1.0f ? 1.0f : 0.0f);
    return composite;
}

// wgsl:223-256
static Hit ray_sphere(const Ray& ray, vec3 centre, float radius, bool /*cull_backface*/) {
    Hit hit;
    hit.dst = INF;
    vec3 offset_ray_origin = ray.origin - centre;
    float a = dot(ray.dir, ray.dir);
    float b = 2.0f * dot(offset_ray_origin, ray.dir);
    float c = dot(offset_ray_origin, offset_ray_origin) - radius * radius;
    float discriminant = b * b - (4.0f * a) * c;
    if (discriminant >= 0.0f) {
        float s = rtm::sqrt_(discriminant);
        float dst_near = wmax(0.0f, (-b - s) / (2.0f * a));
        float dst_far = (-b + s) / (2.0f * a);
        if (dst_far >= 0.001f) {
            bool is_inside = dst_near == 0.0f;
            hit.hit = true;
            hit.dst = is_inside ? dst_far : dst_near;
            hit.hit_point = ray.origin + ray.dir * hit.dst;
            vec3 n = normalize(hit.hit_point - centre);
            hit.normal = is_inside ? -n : n;
            hit.backface = is_inside;
            float theta = rtm::acos_(-hit.normal.y);
            const float pi = 3.1415926f;
            float phi = rtm::atan2_(-hit.normal.z, -hit.normal.x) + pi;
            hit.uv = vec2{phi / (2.0f * pi), theta / pi};
        }
    }
    return hit;
}

// wgsl:258-290
static Hit ray_triangle(const Ray& ray, const rt_packed_triangle& tri, bool cull_backface) {
    Hit hit;
    hit.hit = false;
    vec3 v1{tri.v1[0], tri.v1[1], tri.v1[2]}, v2{tri.v2[0], tri.v2[1], tri.v2[2]},
        v3{tri.v3[0], tri.v3[1], tri.v3[2]};
    vec3 edge_ab = v2 - v1;
    vec3 edge_ac = v3 - v1;
    vec3 normal = cross(edge_ab, edge_ac);
    vec3 ao = ray.origin - v1;
    vec3 dao = cross(ao, ray.dir);
    float determinant = -dot(ray.dir, normal);
    bool keep = cull_backface ? (determinant >= 1e-8f) : (wabs(determinant) >= 1e-8f);
    if (!keep) return hit;
    float inverse_determinant = 1.0f / determinant;
    float dst = dot(ao, normal) * inverse_determinant;
    float u = dot(edge_ac, dao) * inverse_determinant;
    float v = -dot(edge_ab, dao) * inverse_determinant;
    float w = (1.0f - u) - v;
    if (dst > EPSILON && u >= 0.0f && v >= 0.0f && w >= 0.0f) {
        vec3 n1{tri.n1[0], tri.n1[1], tri.n1[2]}, n2{tri.n2[0], tri.n2[1], tri.n2[2]},
            n3{tri.n3[0], tri.n3[1], tri.n3[2]};
        hit.hit = true;
        hit.normal = normalize((n1 * w + n2 * u) + n3 * v) * wsign(determinant);
        hit.backface = determinant < 0.0f;
        hit.hit_point = ray.origin + ray.dir * dst;
        hit.dst = dst;
        hit.uv = (vec2{tri.uv10, tri.uv11} * w + vec2{tri.uv20, tri.uv21} * u) +
                 vec2{tri.uv30, tri.uv31} * v;
    }
    return hit;
}

// wgsl:337-351
static inline float ray_aabb_dist(const Ray& ray, vec3 b_min, vec3 b_max, float t) {
    vec3 t1 = (b_min - ray.origin) * ray.inv_dir;
    vec3 t2 = (b_max - ray.origin) * ray.inv_dir;
    vec3 tmin = wmin(t1, t2);
    vec3 tmax = wmax(t1, t2);
    float t_near = wmax(wmax(tmin.x, tmin.y), tmin.z);
    float t_far = wmin(wmin(tmax.x, tmax.y), tmax.z);
    bool did_hit = t_far >= t_near && t_near < t && t_far > 0.0f;
    if (did_hit) return t_near;
    return INF;
}

// wgsl:292-335.  Out-of-range stack indices are clamped (naga "Restrict").
static thread_local uint32_t g_max_stack_index = 0;  // oracle-only: lets tests see that the stack overflowed

// Oracle-only census for the hypothesis behind the product's cross-mesh pruning: for every
// triangle test that reports a hit, how the hit's parameter t compares with the entry distance the slab test computes
// for the LEAF box that holds the triangle (the boxes above it can only be entered earlier: nested boxes have nested
// slab intervals).  In exact arithmetic entry <= t; the pruning is exact as long as entry <= 1.125 t.
static std::atomic<int> g_census_on{0};
static std::atomic<uint64_t> g_census[6];            // hits, entry > t, > t (1 + 1e-6), > t (1 + 1e-4), > 1.01 t, > 1.125 t
static std::atomic<uint32_t> g_census_max_bits{0};   // largest entry / t seen (float bits; ratios are positive)
static inline float ray_aabb_dist(const Ray& ray, vec3 b_min, vec3 b_max, float t);
static void census_hit(const Ray& ray, const rt_node& leaf, float t) {
    const float entry = ray_aabb_dist(ray, vec3{leaf.aabb_min[0], leaf.aabb_min[1], leaf.aabb_min[2]},
                                      vec3{leaf.aabb_max[0], leaf.aabb_max[1], leaf.aabb_max[2]}, INF);
    g_census[0].fetch_add(1, std::memory_order_relaxed);
    if (!(entry < INF) || !(entry > t)) return;   // (a root leaf's box is never tested by the shader: it may be missed)
    const float ratio = entry / t;
    g_census[1].fetch_add(1, std::memory_order_relaxed);
    if (ratio > 1.0f + 1e-6f) g_census[2].fetch_add(1, std::memory_order_relaxed);
    if (ratio > 1.0f + 1e-4f) g_census[3].fetch_add(1, std::memory_order_relaxed);
    if (ratio > 1.01f) g_census[4].fetch_add(1, std::memory_order_relaxed);
    if (ratio > 1.125f) g_census[5].fetch_add(1, std::memory_order_relaxed);
    uint32_t bits, old = g_census_max_bits.load(std::memory_order_relaxed);
    memcpy(&bits, &ratio, 4);
    while (bits > old && !g_census_max_bits.compare_exchange_weak(old, bits, std::memory_order_relaxed)) {}
}

static Hit ray_BVH(const Ctx& c, const Ray& ray, float ray_length, uint32_t node_offset,
                   uint32_t tri_offset, bool cull_backface, int32_t stats[2]) {
    Hit closest_hit;
    closest_hit.hit = false;
    closest_hit.dst = ray_length;
    uint32_t stack[32];
    auto slot = [](uint32_t i) { return i < 32u ? i : 31u; };
    uint32_t stack_index = 0u;
    stack[slot(stack_index)] = node_offset + 0u;
    stack_index += 1u;
    while (stack_index > 0u) {
        stack_index -= 1u;
        const rt_node node = c.nodes[stack[slot(stack_index)]];
        if (node.count > 0u) {
            stats[1] += (int32_t)node.count;
            for (uint32_t j = 0u; j < node.count; j += 1u) {
                const rt_packed_triangle& tri = c.triangles[tri_offset + node.first + j];
                Hit hit = ray_triangle(ray, tri, cull_backface);
                if (hit.hit && g_census_on.load(std::memory_order_relaxed)) census_hit(ray, node, hit.dst);
                if (hit.hit && hit.dst < closest_hit.dst) {
                    closest_hit = hit;
                    closest_hit.tri = (int)(tri_offset + node.first + j);
                }
            }
        } else {
            uint32_t child_index_a = node_offset + node.left;
            uint32_t child_index_b = node_offset + node.right;
            const rt_node& child_a = c.nodes[child_index_a];
            const rt_node& child_b = c.nodes[child_index_b];
            float dst_a = ray_aabb_dist(ray, vec3{child_a.aabb_min[0], child_a.aabb_min[1], child_a.aabb_min[2]},
                                        vec3{child_a.aabb_max[0], child_a.aabb_max[1], child_a.aabb_max[2]},
                                        closest_hit.dst);
            float dst_b = ray_aabb_dist(ray, vec3{child_b.aabb_min[0], child_b.aabb_min[1], child_b.aabb_min[2]},
                                        vec3{child_b.aabb_max[0], child_b.aabb_max[1], child_b.aabb_max[2]},
                                        closest_hit.dst);
            stats[0] += 2;
            bool left_is_closer = dst_a < dst_b;
            float near_dst = left_is_closer ? dst_a : dst_b;
            float far_dst = !left_is_closer ? dst_a : dst_b;
            uint32_t near_idx = left_is_closer ? child_index_a : child_index_b;
            uint32_t far_idx = !left_is_closer ? child_index_a : child_index_b;
            if (far_dst < closest_hit.dst) { stack[slot(stack_index)] = far_idx; stack_index += 1u; }
            if (near_dst < closest_hit.dst) { stack[slot(stack_index)] = near_idx; stack_index += 1u; }
            if (stack_index > g_max_stack_index) g_max_stack_index = stack_index;
        }
    }
    return closest_hit;
}

// wgsl:353-396
static Hit calculate_ray_collions(const Ctx& c, const Ray& ray, int32_t stats[2]) {
    Hit closest_hit;
    closest_hit.hit = false;
    closest_hit.dst = INF;
    for (uint32_t i = 0u; i < c.scene.spheres; i += 1u) {
        bool cull_backface = c.spheres[i].material.flag != RT_MATERIAL_GLASS;
        Hit hit = ray_sphere(ray, vec3{c.spheres[i].pos[0], c.spheres[i].pos[1], c.spheres[i].pos[2]},
                             c.spheres[i].radius, cull_backface);
        if (hit.hit && hit.dst < closest_hit.dst) {
            closest_hit = hit;
            closest_hit.material = c.spheres[i].material;
            closest_hit.mesh = -2 - (int)i;  // spheres are numbered -2, -3, ...
        }
    }
    Ray local_ray;
    local_ray.transmittance = vec4{0, 0, 0, 0};
    local_ray.bounces = 0u;
    for (uint32_t i = 0u; i < c.scene.meshes; i += 1u) {
        const rt_mesh_uniform& mesh = c.meshes[i];
        local_ray.origin = mat_mul_xyz(mesh.world_to_model, ray.origin, 1.0f);
        local_ray.dir = normalize(mat_mul_xyz(mesh.world_to_model, ray.dir, 0.0f));
        local_ray.inv_dir = 1.0f / local_ray.dir;
        bool cull_backface = mesh.material.flag != RT_MATERIAL_GLASS;
        Hit hit = ray_BVH(c, local_ray, INF, mesh.node_offset, mesh.triangle_offset, cull_backface, stats);
        if (hit.hit) {
            vec3 local_hit_point = local_ray.origin + local_ray.dir * hit.dst;
            vec3 world_hit_point = mat_mul_xyz(mesh.model_to_world, local_hit_point, 1.0f);
            float world_dst = distance(ray.origin, world_hit_point);
            if (world_dst < closest_hit.dst) {
                closest_hit.hit = true;
                closest_hit.backface = hit.backface;
                closest_hit.normal = normalize(mat_mul_xyz(mesh.model_to_world, hit.normal, 0.0f));
                closest_hit.hit_point = world_hit_point;
                closest_hit.dst = world_dst;
                closest_hit.material = mesh.material;
                closest_hit.uv = hit.uv;
                closest_hit.mesh = (int)i;
                closest_hit.tri = hit.tri;
            }
        }
    }
    return closest_hit;
}

static inline vec4 texture_sample_level(const Ctx& c, int index, vec2 uv) {
    float out[4] = {0, 0, 0, 0};
    if (index >= 0 && (uint32_t)index < c.n_textures)
        rtm::sample_bilinear(c.textures[index], SRGB_LUT, uv.x, uv.y, out);
    // indices in [n_textures, 64) are the reference's 1x1 zero dummies
    return vec4{out[0], out[1], out[2], out[3]};
}

// wgsl:398-471
static vec4 trace(const Ctx& c, const Ray& incident_ray, uint32_t* seed, uint64_t* segments,
                  int32_t stats_total[2], TranscriptBuf* tb) {
    Ray ray = incident_ray;
    ray.dir = normalize(ray.dir);
    ray.transmittance = vec4{1.0f, 1.0f, 1.0f, 1.0f};
    vec4 incoming_light{0, 0, 0, 0};
    int32_t _stats[2] = {0, 0};
    for (int32_t i = (int32_t)ray.bounces; i <= c.params.number_of_bounces; i += 1) {
        Hit hit = calculate_ray_collions(c, ray, _stats);
        *segments += 1;
        if (!hit.hit) {
            if (tb && tb->n < tb->cap) tb->rec[tb->n++] = Transcript{-1, -1, INF, *seed};
            if (c.params.skybox != 0) {
                incoming_light = incoming_light + ray.transmittance * get_environment_light(ray);
            }
            break;
        }
        ray.origin = hit.hit_point;
        const rt_material& m = hit.material;
        if (m.flag == RT_MATERIAL_GLASS) {
            if (hit.backface) {
                vec3 absorb{m.absorption[0], m.absorption[1], m.absorption[2]};
                vec3 e = ((-hit.dst) * absorb) * m.absorption_strength;
                vec3 t3{ray.transmittance.x, ray.transmittance.y, ray.transmittance.z};
                vec3 x = t3 * vec3{rtm::exp_(e.x), rtm::exp_(e.y), rtm::exp_(e.z)};
                ray.transmittance = vec4{x.x, x.y, x.z, 1.0f};
            }
            float ior = hit.backface ? m.ior : (1.0f / m.ior);
            vec3 reflect_dir = reflect(ray.dir, hit.normal);
            vec3 refract_dir = refract(ray.dir, hit.normal, ior);
            float cos_theta = wmin(dot(-ray.dir, hit.normal), 1.0f);
            float sin_theta = rtm::sqrt_(1.0f - cos_theta * cos_theta);
            bool cannot_refract = ior * sin_theta > 1.0f;
            bool follow_reflection = cannot_refract || reflectance(cos_theta, ior) > rand_(seed);
            vec3 diffuse_dir = normalize(hit.normal + rand_direction(seed));
            reflect_dir = normalize(mix(diffuse_dir, reflect_dir, m.specular));
            refract_dir = normalize(mix(-diffuse_dir, refract_dir, m.smoothness));
            ray.dir = follow_reflection ? reflect_dir : refract_dir;
            ray.origin = hit.hit_point + (1e-4f * hit.normal) * wsign(dot(hit.normal, ray.dir));
        } else {
            bool is_specular_bounce = m.specular >= rand_(seed);
            vec3 normal = hit.normal;  // wgsl:439-447: the normal-map branch is dead code
            vec3 diffuse_dir = rand_hemisphere(normal, seed);
            vec3 specular_dir = reflect(ray.dir, normal);
            vec4 emitted_light = vec4{m.emission_color[0], m.emission_color[1], m.emission_color[2],
                                      m.emission_color[3]} * m.emission_strength;
            ray.dir = normalize(mix(diffuse_dir, specular_dir, m.smoothness * (is_specular_bounce ? 1.0f : 0.0f)));
            incoming_light = incoming_light + emitted_light * ray.transmittance;
            vec4 color;
            if (m.flag == RT_MATERIAL_TEXTURE && m.diffuse_index != -1) {
                color = texture_sample_level(c, m.diffuse_index, hit.uv);
            } else {
                color = vec4{m.color[0], m.color[1], m.color[2], m.color[3]};
            }
            vec4 spec{m.specular_color[0], m.specular_color[1], m.specular_color[2], m.specular_color[3]};
            ray.transmittance = ray.transmittance * (is_specular_bounce ? spec : color);
        }
        float p = wmax(ray.transmittance.x, wmax(ray.transmittance.y, ray.transmittance.z));
        bool die = rand_(seed) >= p;
        if (tb && tb->n < tb->cap) tb->rec[tb->n++] = Transcript{hit.mesh, hit.tri, hit.dst, *seed};
        if (die) break;
        ray.transmittance = ray.transmittance * (1.0f / p);
        ray.inv_dir = 1.0f / ray.dir;
    }
    stats_total[0] += _stats[0];
    stats_total[1] += _stats[1];
    return incoming_light;
}

// wgsl:502-573
static vec4 debug_trace(const Ctx& c, vec2 pos_, vec2 size, int32_t stats_total[2]) {
    int32_t stats[2] = {0, 0};
    Ray ray;
    vec3 cam_origin{c.scene.camera.cam_to_world[3][0], c.scene.camera.cam_to_world[3][1],
                    c.scene.camera.cam_to_world[3][2]};
    vec2 uv = pos_ / (size - 1.0f);
    vec2 uvc = uv - 0.5f;
    vec3 local_focus_point = vec3{uvc.x, uvc.y, 1.0f} *
                             vec3{c.scene.camera.view_params[0], c.scene.camera.view_params[1],
                                  c.scene.camera.view_params[2]};
    vec3 focus_point = mat_mul_xyz(c.scene.camera.cam_to_world, local_focus_point, 1.0f);
    ray.origin = cam_origin;
    ray.dir = normalize(focus_point - ray.origin);
    ray.inv_dir = 1.0f / ray.dir;
    Hit hit = calculate_ray_collions(c, ray, stats);
    stats_total[0] += stats[0];
    stats_total[1] += stats[1];
    float scale = (float)c.params.debug_scale;
    switch (c.params.debug_flag) {
        case 5: {
            float d = (float)stats[0] / scale;
            if (d > 1.0f) return vec4{1.0f, 0.0f, 0.0f, 1.0f};
            return vec4{d, d, d, 1.0f};
        }
        case 6: {
            float t = (float)stats[1] / scale;
            if (t > 1.0f) return vec4{1.0f, 0.0f, 0.0f, 1.0f};
            return vec4{t, t, t, 1.0f};
        }
        case 2: {
            if (!hit.hit) return vec4{0, 0, 0, 0};
            float d = hit.dst / scale;
            return vec4{d, d, d, 1.0f};
        }
        case 1: {
            if (!hit.hit) return vec4{0, 0, 0, 0};
            vec3 n;
            if (hit.material.flag == RT_MATERIAL_TEXTURE && hit.material.normal_index != -1) {
                vec4 x = texture_sample_level(c, hit.material.normal_index, hit.uv);
                vec3 t = 2.0f * vec3{x.x, x.y, x.z} - vec3{1.0f, 1.0f, 1.0f};
                n = 0.5f * t + vec3{0.5f, 0.5f, 0.5f};
            } else {
                n = hit.normal * 0.5f + vec3{0.5f, 0.5f, 0.5f};
            }
            return vec4{n.x, n.y, n.z, 1.0f};
        }
        case 7: {
            float d = (float)stats[0] / scale;
            float t = (float)stats[1] / scale;
            return vec4{t, 0.0f, d, 1.0f};
        }
        case 4: {
            if (!hit.hit) return vec4{0, 0, 0, 0};
            float s = scale / 100.0f;
            float d = hit.dst;
            if (d > s) return vec4{0.0f, 1.0f, 0.0f, 1.0f};
            return vec4{d, d, d, 1.0f};
        }
        case 3: {
            if (!hit.hit) return vec4{0, 0, 0, 0};
            return vec4{hit.uv.x, hit.uv.y, 0.0f, 1.0f};
        }
        default: return vec4{1.0f, 0.0f, 1.0f, 1.0f};
    }
}

// wgsl:473-500
static vec4 frag(const Ctx& c, vec2 pos_, vec2 size, uint64_t* segments, int32_t stats_total[2],
                 TranscriptBuf* tb) {
    vec2 pixel_coord = pos_;
    int32_t fr = c.params.frames;
    uint32_t absf = fr < 0 ? (uint32_t)0 - (uint32_t)fr : (uint32_t)fr;
    uint32_t rng_state = (uint32_t)(pixel_coord.y * size.x + pixel_coord.x) + absf * 719393u;
    if (c.params.debug_flag != 0) return debug_trace(c, pos_, size, stats_total);
    vec2 uv = pos_ / (size - 1.0f);
    const rt_camera_uniform& cam = c.scene.camera;
    vec3 cam_origin{cam.cam_to_world[3][0], cam.cam_to_world[3][1], cam.cam_to_world[3][2]};
    vec2 uvc = uv - 0.5f;
    vec3 local_focus_point = vec3{uvc.x, uvc.y, 1.0f} * vec3{cam.view_params[0], cam.view_params[1], cam.view_params[2]};
    vec3 focus_point = mat_mul_xyz(cam.cam_to_world, local_focus_point, 1.0f);
    vec3 cam_right{cam.cam_to_world[0][0], cam.cam_to_world[0][1], cam.cam_to_world[0][2]};
    vec3 cam_up{cam.cam_to_world[1][0], cam.cam_to_world[1][1], cam.cam_to_world[1][2]};
    vec4 total_incoming_light{0, 0, 0, 0};
    for (int32_t j = 0; j < c.params.rays_per_pixel; j += 1) {
        vec2 defocus_jitter = rand_in_unit_disk(&rng_state) * cam.defocus_strength / size.x;
        Ray ray;
        ray.origin = (cam_origin + cam_right * defocus_jitter.x) + cam_up * defocus_jitter.y;
        vec2 diverge_jitter = rand_in_unit_disk(&rng_state) * cam.diverge_strength / size.x;
        vec3 jittered_focus_point = (focus_point + cam_right * diverge_jitter.x) + cam_up * diverge_jitter.y;
        ray.dir = normalize(jittered_focus_point - ray.origin);
        total_incoming_light = total_incoming_light + trace(c, ray, &rng_state, segments, stats_total, tb);
    }
    return total_incoming_light / (float)c.params.rays_per_pixel;
}

// wgsl:144-162
static void main_invocation(const Ctx& c, uint32_t gx, uint32_t gy, float* image, uint64_t* segments,
                            int32_t stats_total[2], TranscriptBuf* tb) {
    vec2 pos_{(float)gx, (float)gy};
    vec2 size{(float)c.params.width, (float)c.params.height};
    vec4 current_sample = frag(c, pos_, size, segments, stats_total, tb);
    float* texel = image + ((size_t)gy * c.params.width + gx) * 4;
    if (c.params.frames >= 1) {
        vec4 prev{texel[0], texel[1], texel[2], texel[3]};
        float weight = 1.0f / (float)(c.params.frames + 1);
        vec4 nc = prev * (1.0f - weight) + current_sample * weight;
        texel[0] = nc.x; texel[1] = nc.y; texel[2] = nc.z; texel[3] = nc.w;
    } else {
        texel[0] = current_sample.x; texel[1] = current_sample.y;
        texel[2] = current_sample.z; texel[3] = current_sample.w;
    }
}

}  // namespace orc

extern "C" {

typedef struct oracle_stats {
    uint64_t segments;
    uint64_t node_tests;
    uint64_t triangle_tests;
} oracle_stats;

typedef struct oracle_transcript {
    int32_t hit_mesh, hit_tri;
    float dst;
    uint32_t rng_after;
} oracle_transcript;

// Renders the listed rows of the frame into `image` (full-frame RGBA32F,
// read-modify-write when params->frames >= 1) on n_threads threads, one row
// per task.
int oracle_render_rows(const rt_params* params, const rt_scene_uniform* scene, const rt_sphere* spheres,
                       const rt_mesh_uniform* meshes, const rt_packed_triangle* triangles,
                       const rt_node* nodes, const rt_texture_desc* textures, uint32_t n_textures,
                       float* image, const uint32_t* rows, uint32_t n_rows, int n_threads,
                       oracle_stats* stats_out) {
    if (!params || !scene || !image || (n_rows && !rows)) return -1;
    std::vector<rtm::TexView> tv(n_textures);
    for (uint32_t i = 0; i < n_textures; ++i) tv[i] = rtm::TexView{textures[i].rgba8, textures[i].width, textures[i].height};
    orc::Ctx c{*params, *scene, spheres, meshes, triangles, nodes, tv.data(), n_textures};
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint32_t> next(0);
    std::atomic<uint64_t> seg(0), nt(0), tt(0);
    auto worker = [&]() {
        uint64_t segments = 0, n_tests = 0, t_tests = 0;
        for (;;) {
            uint32_t k = next.fetch_add(1);
            if (k >= n_rows) break;
            uint32_t y = rows[k];
            if (y >= params->height) continue;
            for (uint32_t x = 0; x < params->width; ++x) {
                int32_t st[2] = {0, 0};
                orc::main_invocation(c, x, y, image, &segments, st, nullptr);
                n_tests += (uint64_t)st[0];
                t_tests += (uint64_t)st[1];
            }
        }
        seg += segments; nt += n_tests; tt += t_tests;
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (stats_out) {
        stats_out->segments = seg; stats_out->node_tests = nt; stats_out->triangle_tests = tt;
    }
    return 0;
}

// Rows [row_begin, row_end).
int oracle_render(const rt_params* params, const rt_scene_uniform* scene, const rt_sphere* spheres,
                  const rt_mesh_uniform* meshes, const rt_packed_triangle* triangles,
                  const rt_node* nodes, const rt_texture_desc* textures, uint32_t n_textures,
                  float* image, uint32_t row_begin, uint32_t row_end, int n_threads,
                  oracle_stats* stats_out) {
    if (!params) return -1;
    if (row_end > params->height) row_end = params->height;
    std::vector<uint32_t> rows;
    for (uint32_t y = row_begin; y < row_end; ++y) rows.push_back(y);
    return oracle_render_rows(params, scene, spheres, meshes, triangles, nodes, textures, n_textures, image,
                              rows.data(), (uint32_t)rows.size(), n_threads, stats_out);
}

// One pixel with a per-segment transcript; returns the number of records.
// `rgba_out` receives frag()'s value (before accumulation).
int oracle_trace_pixel(const rt_params* params, const rt_scene_uniform* scene, const rt_sphere* spheres,
                       const rt_mesh_uniform* meshes, const rt_packed_triangle* triangles,
                       const rt_node* nodes, const rt_texture_desc* textures, uint32_t n_textures,
                       uint32_t x, uint32_t y, float rgba_out[4], oracle_transcript* rec, uint32_t cap) {
    std::vector<rtm::TexView> tv(n_textures);
    for (uint32_t i = 0; i < n_textures; ++i) tv[i] = rtm::TexView{textures[i].rgba8, textures[i].width, textures[i].height};
    orc::Ctx c{*params, *scene, spheres, meshes, triangles, nodes, tv.data(), n_textures};
    static_assert(sizeof(orc::Transcript) == sizeof(oracle_transcript), "layout");
    orc::TranscriptBuf tb{reinterpret_cast<orc::Transcript*>(rec), cap, 0};
    uint64_t segments = 0;
    int32_t st[2] = {0, 0};
    orc::vec4 v = orc::frag(c, orc::vec2{(float)x, (float)y}, orc::vec2{(float)params->width, (float)params->height},
                            &segments, st, &tb);
    rgba_out[0] = v.x; rgba_out[1] = v.y; rgba_out[2] = v.z; rgba_out[3] = v.w;
    return (int)tb.n;
}

// Largest stack_index ray_BVH reached on the calling thread since the last reset (> 32 means
// the shader's 32-entry stack overflowed and index clamping took effect).
uint32_t oracle_max_stack_index(int reset) {
    uint32_t v = orc::g_max_stack_index;
    if (reset) orc::g_max_stack_index = 0;
    return v;
}

// Census of triangle hits against their leaf boxes (see census_hit): enable != 0 switches it on and clears it, 0 switches
// it off; out[7] = {hits, entry > t, > t (1 + 1e-6), > t (1 + 1e-4), > 1.01 t, > 1.125 t, largest entry / t}.
void oracle_census(int enable, double out[7]) {
    if (out) {
        for (int k = 0; k < 6; ++k) out[k] = (double)orc::g_census[k].load();
        float m;
        const uint32_t b = orc::g_census_max_bits.load();
        memcpy(&m, &b, 4);
        out[6] = m;
    }
    if (enable) {
        for (auto& c : orc::g_census) c.store(0);
        orc::g_census_max_bits.store(0);
    }
    orc::g_census_on.store(enable ? 1 : 0);
}

uint32_t oracle_next_random_number(uint32_t* state) { return orc::next_random_number(state); }
float oracle_rand(uint32_t* state) { return orc::rand_(state); }

// fn: 0 log, 1 cos, 2 sin, 3 exp, 4 exp2, 5 log2, 6 pow(x,y), 7 acos, 8 atan2(x=y_arg, y=x_arg),
//     9 sqrt, 10 x/y
void oracle_transc(int fn, const float* x, const float* y, float* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = rtm::log_(x[i]); break;
            case 1: out[i] = rtm::cos_(x[i]); break;
            case 2: out[i] = rtm::sin_(x[i]); break;
            case 3: out[i] = rtm::exp_(x[i]); break;
            case 4: out[i] = rtm::exp2_(x[i]); break;
            case 5: out[i] = rtm::log2_(x[i]); break;
            case 6: out[i] = rtm::pow_(x[i], y[i]); break;
            case 7: out[i] = rtm::acos_(x[i]); break;
            case 8: out[i] = rtm::atan2_(x[i], y[i]); break;
            case 9: out[i] = rtm::sqrt_(x[i]); break;
            case 10: out[i] = x[i] / y[i]; break;
            // (11-16: the RNG and helpers, state / integer values passed as bit patterns)
            case 11: { uint32_t s = rtm::f2u(x[i]); out[i] = orc::rand_(&s); break; }
            case 12: { uint32_t s = rtm::f2u(x[i]); out[i] = rtm::u2f(orc::next_random_number(&s)); break; }
            case 13: {  // sign bits of cos and sin as the shader's functions give them (wgsl:202-206)
                const uint32_t c = rtm::f2u(rtm::cos_(x[i])) >> 31, sn = rtm::f2u(rtm::sin_(x[i])) >> 31;
                out[i] = rtm::u2f(c | (sn << 1));
                break;
            }
            case 14: { uint32_t s = rtm::f2u(x[i]); out[i] = orc::rand_normal_dist(&s); break; }
            case 15: out[i] = (float)rtm::f2u(x[i]) / 4294967295.0f; break;  // wgsl:165
            case 16: out[i] = orc::normalize(orc::vec3{x[i], y[i], x[i] * y[i]}).x; break;
            default: out[i] = 0.0f;
        }
    }
}

void oracle_sample_texture(const rt_texture_desc* t, const float* uv, float* out, size_t n) {
    rtm::TexView tv{t->rgba8, t->width, t->height};
    for (size_t i = 0; i < n; ++i) rtm::sample_bilinear(tv, orc::SRGB_LUT, uv[2 * i], uv[2 * i + 1], out + 4 * i);
}

// The pixel loop of save_render_to_file (src/core/app.rs:408-460), step by step as the reference does
// it: rows top to bottom of the buffer, x REVERSED inside a row, per channel
// (v.powf(1.0 / 2.2).clamp(0.0, 1.0) * 255.0) as u8 -- `as u8` truncates, saturates and maps NaN to 0 --
// then flip_horizontal_in_place and flip_vertical_in_place (image 0.25.8).  The byte result depends on
// the platform's powf (Rust's f32::powf is libm's powf, as is this one); the product's rt_export_rgba8
// folds the two x reversals into none and writes the rows flipped directly.
void oracle_export_rgba8(const float* rgba, uint32_t width, uint32_t height, uint8_t* out) {
    std::vector<uint8_t> data;
    data.reserve((size_t)width * height * 4);
    auto to_byte = [](float v) -> uint8_t {
        float p = powf(v, 1.0f / 2.2f);
        // f32::clamp: NaN stays NaN (app.rs:443-446)
        float c = p != p ? p : (p < 0.0f ? 0.0f : (p > 1.0f ? 1.0f : p));
        float s = c * 255.0f;
        if (s != s) return 0;          // Rust float -> int casts: NaN -> 0, saturating
        if (s <= 0.0f) return 0;
        if (s >= 255.0f) return 255;
        return (uint8_t)s;
    };
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t xr = 0; xr < width; ++xr) {
            const uint32_t x = width - 1 - xr;  // (0..RENDER_SIZE.0).rev()
            const float* p = rgba + ((size_t)y * width + x) * 4;
            for (int c = 0; c < 4; ++c) data.push_back(to_byte(p[c]));
        }
    // flip_horizontal_in_place, then flip_vertical_in_place
    auto px = [&](uint32_t x, uint32_t y) { return data.data() + ((size_t)y * width + x) * 4; };
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width / 2; ++x)
            for (int c = 0; c < 4; ++c) std::swap(px(x, y)[c], px(width - 1 - x, y)[c]);
    for (uint32_t y = 0; y < height / 2; ++y)
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 4; ++c) std::swap(px(x, y)[c], px(x, height - 1 - y)[c]);
    memcpy(out, data.data(), data.size());
}

int oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
