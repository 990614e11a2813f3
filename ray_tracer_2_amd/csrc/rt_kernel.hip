// rt_kernel.hip -- the hot path: hand-written HIP for gfx950 (MI355X) of the
// per-pixel path-trace loop of shaders/ray_tracer.wgsl (reference).
//
// Mapping to CDNA4
//   * one lane per pixel; a wavefront starts on one 8x8 pixel tile (the
//     reference's @workgroup_size(8,8), wgsl:145); workgroups are 4 waves.
//   * the scene (mesh records, BVH, triangles, materials) is one blob of
//     16-byte words (rt_device.h).  When it fits the LDS budget every
//     workgroup stages it into LDS once with coalesced 16-byte loads and all
//     traversal reads are LDS reads (~64 cycles instead of an L1/L2 round
//     trip); otherwise it is read in place and wave-uniform reads (the mesh
//     loop, wgsl:369, root-leaf triangles) become scalar loads.
//   * BVH internal nodes are re-laid-out "wide": both children's boxes and
//     kinds sit in the parent's 64-byte record, so a node visit is one memory
//     round trip, not three dependent ones.
//   * per-lane BVH stacks live in LDS, lane-interleaved (conflict-free), with
//     the near child kept in registers.
//   * a path is a small state machine (ray generation / segment / shading) so
//     lanes whose path ended start their pixel's next sample at once; the
//     persistent variant refills finished lanes with new pixels using
//     __ballot + mbcnt prefix counts.  The samples of one pixel share one
//     sequential RNG stream (wgsl:475,487-497) and stay on one lane.
//   * blockIdx -> tile mapping of the tile variant is XCD-aware.
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

// waves per SIMD the render kernels are compiled for (register budget 512 / N)

// Diagnostic build only (-DRT_DIAG=1, tools/diag.py): per-section wave visits
// and active-lane sums, to price divergence.  Compiled out of the product.
#if defined(RT_DIAG) || defined(RT_WAVE_TIMES)
__device__ unsigned long long g_wave_times[2 * 8192];  // (start, end) s_memrealtime per persistent wave
#endif
#if defined(RT_DIAG) || defined(RT_DIAGT)
__device__ unsigned long long g_diag[128];
#endif
#if defined(RT_DIAGT)
// Timing build (-DRT_DIAGT=1, tools/diag.py --time): wave-cycles per code section.
// (accumulated per wave in LDS, flushed once at the end of the persistent kernel: global atomics
// per section would serialise at L2 and distort everything)
__shared__ unsigned long long g_tacc[4][24];
#define TIC(v) const unsigned long long v = __builtin_readcyclecounter()
#define TOC(v, id)                                                                                  \
    do {                                                                                            \
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(1)))                          \
            g_tacc[threadIdx.x >> 6][id] += (unsigned long long)(__builtin_readcyclecounter() - v); \
    } while (0)
#else
#define TIC(v) do { } while (0)
#define TOC(v, id) do { } while (0)
#endif
#if defined(RT_DIAG)
#define DIAG(id)                                                                  \
    do {                                                                          \
        unsigned long long m_ = __ballot(1);                                      \
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m_)) {               \
            atomicAdd(&g_diag[2 * (id)], 1ull);                                   \
            atomicAdd(&g_diag[2 * (id) + 1], (unsigned long long)__popcll(m_));   \
        }                                                                         \
    } while (0)
#else
#define DIAG(id) do { } while (0)
#endif

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
// 1.0f / x, correctly rounded, in fewer instructions than the compiler's IEEE division (11: two v_div_scale, v_rcp, six
// fma, v_div_fmas, v_div_fixup).  For a normal x whose reciprocal is normal and away from the denormals, v_rcp_f32
// (1 ULP) plus one Newton step in fma arithmetic lands on the correctly rounded quotient -- not by argument but by
// enumeration: rt_sweep_kernel compares the two for EVERY float in that range on the device
// (tests/test_gpu_device_units.py::test_rcp_is_the_ieee_division_for_every_float).  If a lane of the wave holds anything
// else (zero, denormal, huge, inf, NaN), the whole wave takes the compiler's division.
DEV bool rcp_in_range(float x) { return rtm::abs_(x) >= 0x1p-120f && rtm::abs_(x) <= 0x1p+120f; }  // (false for NaN)
DEV float rcp_core(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return rtm::fma_(r, rtm::fma_(-x, r, 1.0f), r);
}
DEV float rcp_(float x) {
#if defined(RT_IEEE_DIV)
    return 1.0f / x;
#else
    if (__ballot(!rcp_in_range(x)) != 0ull) return 1.0f / x;
    return rcp_core(x);
#endif
}
// sqrt(x), correctly rounded: the compiler's own expansion (v_sqrt_f32, then the two neighbours tried with an exact fma
// residual) without its scaling for denormal inputs and its class check -- for a normal positive x away from both
// ends; enumerated against __builtin_sqrtf on the device like rcp_.  The compiler's expansion for the whole wave otherwise.
DEV bool sqrt_in_range(float x) { return x >= 0x1p-100f && x <= 0x1p+100f; }  // (false for NaN, zero, negatives)
DEV float sqrt_core(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float rb = rtm::fma_(-below, s, x), ra = rtm::fma_(-above, s, x);
    float r = 0.0f >= rb ? below : s;
    r = 0.0f < ra ? above : r;
    return r;
}
DEV float sqrt_dev(float x) {
#if defined(RT_IEEE_DIV)
    return rtm::sqrt_(x);
#else
    if (__ballot(!sqrt_in_range(x)) != 0ull) return rtm::sqrt_(x);
    return sqrt_core(x);
#endif
}
// WGSL leaves normalize()'s precision to the implementation ("inherited from e / length(e)", with 2.5 ULP for a
// division); drivers multiply by an inverse square root.  Canonical here (round 3): ONE correctly rounded reciprocal
// of the correctly rounded length, then three multiplications -- 19 instructions fewer than three IEEE divisions,
// five or so times per segment.  The oracle's normalize() is the same two-step form.
// (SQ: the short square root too.  Measured per kernel family -- tools/experiments/README.md: it is worth 1.5 % to the
// kernels that read the scene from global memory and costs the LDS-scene kernels 0.7 %, so the callers pass !LDS.)
template <bool SQ = false>
DEV f3 normalize3(f3 a) { return a * rcp_(SQ ? sqrt_dev(dot3(a, a)) : rtm::sqrt_(dot3(a, a))); }
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
template <class P>
DEV f3 mat_xyz(P m, f3 v, float w) {
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
template <bool SQ = false>
DEV f3 rand_unit_sphere(uint32_t& s) {
    float x = rand_normal_dist(s);
    float y = rand_normal_dist(s);
    float z = rand_normal_dist(s);
    return normalize3<SQ>(f3{x, y, z});
}
DEV void rand_in_unit_disk(uint32_t& s, float& ox, float& oy) {
    float angle = (rand_(s) * 2.0f) * 3.1415926f;
    float c = rtm::cos_(angle), sn = rtm::sin_(angle);
    float r = rtm::sqrt_(rand_(s));
    ox = c * r;
    oy = sn * r;
}

// Camera jitter of wgsl:488,492: rand_in_unit_disk() * strength / size.x.
// When strength is +0.0 (the default camera) the value is (cos*r)*0/width =
// +-0 and only the sign of cos / sin survives; the two RNG draws still happen.
DEV void disk_jitter(uint32_t& s, float strength, float sx, float& jx, float& jy) {
    if (__float_as_uint(strength) == 0u) {  // wave-uniform
        float angle = (rand_(s) * 2.0f) * 3.1415926f;
        const uint32_t sb = rtm::trig_signbits(angle);
        (void)next_random_number(s);  // the radius draw; sqrt(rand) >= +0 cannot change a sign
        jx = __uint_as_float((sb & 1u) << 31);  // +-0
        jy = __uint_as_float((sb & 2u) << 30);
    } else {
        rand_in_unit_disk(s, jx, jy);
        jx = jx * strength / sx;
        jy = jy * strength / sx;
    }
}

// ---- scene memory ----------------------------------------------------------
extern __shared__ float4 lds_mem[];

template <bool LDS>
DEV float4 ld4(const RenderArgs& a, uint32_t byte_off) {
    // every record of the blob is 16-byte aligned: lets neighbouring loads share one address
    // register with immediate offsets instead of masking each address
    __builtin_assume((byte_off & 15u) == 0u);
    if constexpr (LDS) {
        return lds_mem[byte_off >> 4];
    } else {
        return a.blob[byte_off >> 4];
    }
}
template <bool LDS>
DEV float ldf(const RenderArgs& a, uint32_t byte_off) {
    if constexpr (LDS) {
        return reinterpret_cast<const float*>(lds_mem)[byte_off >> 2];
    } else {
        return reinterpret_cast<const float*>(a.blob)[byte_off >> 2];
    }
}
template <bool LDS>
DEV int ldi(const RenderArgs& a, uint32_t byte_off) {
    return __float_as_int(ldf<LDS>(a, byte_off));
}
DEV uint32_t fbits(float f) { return __float_as_uint(f); }

// Built-and-measured-slower features (LDS-staged BVH / tree tops, hybrid launches, the wavefront sequence: DESIGN.md
// sections 5.4, 5.5) are only compiled with -DRT_EXPERIMENTS=1 (tools/build_variant.sh exp -DRT_EXPERIMENTS=1); the
// product library has none of their code.
// The two record fetches of the global-memory kernels keep their "is this record staged in LDS?" test even in the product
// build, where nothing is ever staged (RenderArgs::top_count = tlas_lds = 0: the test is wave-uniform and always false).
// Measured, not argued: without the two tests hipcc 7.2 schedules the node-fetch loops differently and the kernels are
// slower -- sponza-sized stand-in 5.67 -> 5.92 ms per frame, config 3 stand-in 5.26 -> 5.37 (either test alone: no gain;
// profiles/r04_dead_branch_ab.txt, tools/experiments/README.md).  -DRT_TOP_BRANCH=0 -DRT_TLAS_BRANCH=0 compiles them out.
// The pre-step of path_step (a memoised primary segment is taken in the same iteration as the segment behind it): config 2
// 1.1416 -> 1.1235 ms per frame, sponza-sized stand-in 5.679 -> 5.652, config 3 stand-in unchanged (tools/experiments/README.md).
#ifndef RT_PRESTEP
#define RT_PRESTEP 1
#endif
// (not in the launches of a deferred-walk sequence: there its second copy of path_end costs 16 B more scratch per lane
// than it saves -- config 3 stand-in 5.276 -> 5.260 ms per frame without it, config 5 geometry 3.152 -> 3.116)
#ifndef RT_PRESTEP_PARK
#define RT_PRESTEP_PARK 0
#endif
#ifndef RT_TOP_BRANCH
#define RT_TOP_BRANCH 1
#endif
#ifndef RT_TLAS_BRANCH
#define RT_TLAS_BRANCH 1
#endif
#if RT_EXPERIMENTS || RT_TOP_BRANCH || RT_TLAS_BRANCH
// float4 offset of the LDS-staged BVH top (global-memory kernels): behind the wave regions and cost tables
DEV uint32_t top_lds_off16(const RenderArgs& a);
DEV uint32_t tlas_lds_off16(const RenderArgs& a);
#endif

// The wide record `idx` (absolute index): from the staged scene or from global memory (RT_EXPERIMENTS: or from the
// LDS-staged top of the big mesh's BVH, option "lds_top").
template <bool LDS>
DEV void load_wide(const RenderArgs& a, uint32_t idx, float4& q0, float4& q1, float4& q2, float4& q3) {
#if RT_TOP_BRANCH
    if constexpr (!LDS) {
        const uint32_t rel = idx - a.top_base;
        if (rel < a.top_count) {
            const float4* p = lds_mem + top_lds_off16(a) + rel * 4u;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
            return;
        }
    }
#endif
    const uint32_t wo = a.lay.wide_off + idx * WIDE_REC_BYTES;
    q0 = ld4<LDS>(a, wo); q1 = ld4<LDS>(a, wo + 16); q2 = ld4<LDS>(a, wo + 32); q3 = ld4<LDS>(a, wo + 48);
}

// Top-level tree record `e` (many-mesh kernels): from the staged scene or from global memory (RT_EXPERIMENTS: or from
// the LDS-staged copy of the tree, option "lds_tlas").
template <bool LDS>
DEV void load_tlas(const RenderArgs& a, uint32_t e, float4& q0, float4& q1, float4& q2, float4& q3) {
#if RT_TLAS_BRANCH
    if constexpr (!LDS) {
        if (e < a.tlas_lds) {  // (the first tlas_lds records -- the top levels, numbered breadth-first -- are staged)
            const float4* p = lds_mem + tlas_lds_off16(a) + e * 4u;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
            return;
        }
    }
#endif
    const uint32_t wo = a.lay.tlas_off + e * WIDE_REC_BYTES;
    q0 = ld4<LDS>(a, wo); q1 = ld4<LDS>(a, wo + 16); q2 = ld4<LDS>(a, wo + 32); q3 = ld4<LDS>(a, wo + 48);
}

// (mat4 * vec4(v, w)).xyz with the four columns as float4
DEV f3 mat_cols_xyz(float4 c0, float4 c1, float4 c2, float4 c3, f3 v, float w) {
    f3 r;
    r.x = ((c0.x * v.x + c1.x * v.y) + c2.x * v.z) + c3.x * w;
    r.y = ((c0.y * v.x + c1.y * v.y) + c2.y * v.z) + c3.y * w;
    r.z = ((c0.z * v.x + c1.z * v.y) + c2.z * v.z) + c3.z * w;
    return r;
}

// ---- intersection ---------------------------------------------------------
struct MeshBest {  // closest triangle hit inside the mesh being traversed
    // (w and the determinant are not kept: w is (1 - u) - v again where it is needed -- the operations of wgsl:280 --
    // and of the determinant only the sign is used after the test, wgsl:284-285; it cannot be zero, wgsl:268)
    float t, u, v;
    uint32_t tri;  // triangle index (absolute) | 0x80000000 when the determinant is negative; 0xffffffff: no hit
};

// wgsl:258-290 against the pre-laid-out record (rt_device.h): edge_ab, edge_ac
// and their cross product are the same IEEE operations wgsl:261-263 evaluate,
// hoisted to upload time.  Shading outputs are deferred to the winner.
template <int SEC>
DEV void tri_test(f3 lo, f3 ld, float4 q0, float4 q1, float4 q2, bool cull, uint32_t idx,
                  MeshBest& b) {
    DIAG(SEC);
    f3 n{q0.w, q1.w, q2.w};
    // The determinant only needs the ray direction and the (hoisted) normal:
    // evaluate it first so culled triangles cost five operations.
    float det = -dot3(ld, n);
    bool keep = cull ? (det >= 1e-8f) : (rtm::abs_(det) >= 1e-8f);
    if (keep) {
        DIAG(SEC + 1);
        f3 v1{q0.x, q0.y, q0.z};
        f3 eab{q1.x, q1.y, q1.z}, eac{q2.x, q2.y, q2.z};
        f3 ao = lo - v1;
        f3 dao = cross3(ao, ld);
        float inv = rcp_(det);
        float dst = dot3(ao, n) * inv;
        float u = dot3(eac, dao) * inv;
        float v = -dot3(eab, dao) * inv;
        float w = (1.0f - u) - v;
        if (dst > EPSILON && u >= 0.0f && v >= 0.0f && w >= 0.0f && dst < b.t) {
            b.t = dst;
            b.u = u;
            b.v = v;
            b.tri = idx | (det < 0.0f ? 0x80000000u : 0u);
        }
    }
}

// wgsl:337-351 on a box record (rt_device.h): qa = (min.x, max.x, min.y, max.y), qb = (min.z,
// max.z, idx, count).  (Written with 2-vectors -- one v_pk_add_f32 + one v_pk_mul_f32 per axis --
// this measured 1 % slower than the plain form below, and letting the SLP vectoriser pack f32
// arithmetic on its own 6 % slower: the operand pairs have to be assembled with moves.)
DEV float aabb_dist(f3 lo, f3 inv, float4 qa, float4 qb, float t) {
    const float t1x = (qa.x - lo.x) * inv.x, t2x = (qa.y - lo.x) * inv.x;
    const float t1y = (qa.z - lo.y) * inv.y, t2y = (qa.w - lo.y) * inv.y;
    const float t1z = (qb.x - lo.z) * inv.z, t2z = (qb.y - lo.z) * inv.z;
    float t_near = max_(max_(min_(t1x, t2x), min_(t1y, t2y)), min_(t1z, t2z));
    float t_far = min_(min_(max_(t1x, t2x), max_(t1y, t2y)), max_(t1z, t2z));
    bool did_hit = t_far >= t_near && t_near < t && t_far > 0.0f;
    return did_hit ? t_near : INF;
}

// Per-lane BVH stack in LDS (lane-interleaved columns).  An entry is one dword,
// leaf: 1 << 31 | count << 24 | first triangle, internal node: wide-record index -- or, for
// scenes with a leaf of more than 127 triangles (RenderArgs::stack_wide), two dwords (idx, count).
struct LaneStack {
    uint32_t* col;  // this lane's column
    bool wide;      // wave-uniform
};
DEV void stack_put(const LaneStack& st, uint32_t e, uint32_t idx, uint32_t cnt) {
    if (st.wide) {
        st.col[e * 128u] = idx;
        st.col[e * 128u + 64u] = cnt;
    } else {
        st.col[e * 64u] = cnt ? (0x80000000u | (cnt << 24) | idx) : idx;
    }
}
DEV void stack_get(const LaneStack& st, uint32_t e, uint32_t& idx, uint32_t& cnt) {
    if (st.wide) {
        idx = st.col[e * 128u];
        cnt = st.col[e * 128u + 64u];
    } else {
        const uint32_t v = st.col[e * 64u];
        const bool leaf = (v & 0x80000000u) != 0u;
        idx = leaf ? (v & 0x00ffffffu) : v;
        cnt = leaf ? ((v >> 24) & 0x7fu) : 0u;
    }
}
DEV uint32_t stack_dwords(const RenderArgs& a) {
    return (a.stack_entries ? a.stack_entries : 1u) * (a.stack_wide ? 128u : 64u);
}

// wgsl:292-335 for one mesh.  A traversal entry is (idx, count): count > 0 is
// a leaf holding triangles [idx, idx + count), count == 0 is the internal node
// whose wide record has mesh-local index idx.  `stack` points at this lane's
// LDS column (stride 64 dwords, two dwords per entry).  Visit order per lane
// is the shader's: far child pushed, near child visited next (kept in
// registers instead of a push/pop pair), entries popped without re-testing.
// "while-while": a lane first descends through internal nodes until it holds a
// leaf, then the wave tests leaf triangles together.
template <bool LDS, bool STATS>
DEV void traverse_mesh(const RenderArgs& a, uint32_t root_idx, uint32_t root_count, bool cull, bool deep, f3 lo,
                       f3 ld, f3 inv, uint32_t* stack,
                       MeshBest& best, int& node_tests, int& tri_tests) {
    const uint32_t tri0 = a.lay.tri_off;  // (indices are absolute)
    const LaneStack st{stack, a.stack_wide != 0u};
    if (root_count > 0) {
        // Root is a leaf: every lane tests the same triangles (uniform reads).
        TIC(t4);
        if (STATS) tri_tests += (int)root_count;
        for (uint32_t j = 0; j < root_count; ++j) {
            const uint32_t t = tri0 + (root_idx + j) * TRI_ISECT_BYTES;
            tri_test<4>(lo, ld, ld4<LDS>(a, t), ld4<LDS>(a, t + 16), ld4<LDS>(a, t + 32), cull, root_idx + j, best);
        }
        TOC(t4, 4);
        return;
    }
    TIC(t16);
    if (deep) {
        // BVH of height >= 32: the shader's `array<u32,32>` stack can overflow, and what it
        // then does is defined by naga's Restrict policy (out-of-range indices are clamped to
        // 31 while stack_index keeps counting).  Reproduce wgsl:297-333 literally.
        auto slot = [](uint32_t i) { return i < RT_BVH_STACK ? i : RT_BVH_STACK - 1u; };
        uint32_t stack_index = 0;
        stack_put(st, slot(0), root_idx, 0u);
        stack_index = 1;
        while (stack_index > 0) {
            stack_index -= 1;
            uint32_t idx, cnt;
            stack_get(st, slot(stack_index), idx, cnt);
            if (cnt > 0) {
                if (STATS) tri_tests += (int)cnt;
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t t = tri0 + (idx + j) * TRI_ISECT_BYTES;
                    tri_test<8>(lo, ld, ld4<LDS>(a, t), ld4<LDS>(a, t + 16), ld4<LDS>(a, t + 32), cull, idx + j, best);
                }
            } else {
                float4 q0, q1, q2, q3;
                load_wide<LDS>(a, idx, q0, q1, q2, q3);
                float da = aabb_dist(lo, inv, q0, q1, best.t);
                float db = aabb_dist(lo, inv, q2, q3, best.t);
                if (STATS) node_tests += 2;
                const bool left_closer = da < db;
                const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
                if (far_d < best.t) {
                    stack_put(st, slot(stack_index), fbits(left_closer ? q3.z : q1.z), fbits(left_closer ? q3.w : q1.w));
                    stack_index += 1;
                }
                if (near_d < best.t) {
                    stack_put(st, slot(stack_index), fbits(left_closer ? q1.z : q3.z), fbits(left_closer ? q1.w : q3.w));
                    stack_index += 1;
                }
            }
        }
        return;
    }
    uint32_t cur = root_idx, cur_count = 0, sp = 0;
    for (;;) {
        bool finished = false;
        while (cur_count == 0) {
            DIAG(7);
            float4 q0, q1, q2, q3;
            load_wide<LDS>(a, cur, q0, q1, q2, q3);
            float da = aabb_dist(lo, inv, q0, q1, best.t);
            float db = aabb_dist(lo, inv, q2, q3, best.t);
            if (STATS) node_tests += 2;
            const bool left_closer = da < db;
            const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
            const uint32_t near_i = fbits(left_closer ? q1.z : q3.z), near_c = fbits(left_closer ? q1.w : q3.w);
            const uint32_t far_i = fbits(left_closer ? q3.z : q1.z), far_c = fbits(left_closer ? q3.w : q1.w);
            if (far_d < best.t) {
                stack_put(st, sp, far_i, far_c);
                ++sp;
            }
            if (near_d < best.t) {
                cur = near_i;
                cur_count = near_c;
            } else {
                if (sp == 0) {
                    finished = true;
                    break;
                }
                --sp;
                stack_get(st, sp, cur, cur_count);
            }
        }
        if (finished) break;
        if (STATS) tri_tests += (int)cur_count;
        for (uint32_t j = 0; j < cur_count; ++j) {
            const uint32_t t = tri0 + (cur + j) * TRI_ISECT_BYTES;
            tri_test<8>(lo, ld, ld4<LDS>(a, t), ld4<LDS>(a, t + 16), ld4<LDS>(a, t + 32), cull, cur + j, best);
        }
        if (sp == 0) break;
        --sp;
        stack_get(st, sp, cur, cur_count);
    }
    TOC(t16, 16);
}

// wgsl:292-335 for a mesh whose root has two leaf children (ITEM_FLAT2), as straight-line code.  What the
// shader's loop does on such a tree: both child boxes are tested against the initial closest distance
// (INF); the far one is pushed if hit, the near one visited next if hit, and a popped entry is not tested
// again -- so the near leaf's triangles are tested iff its box is hit, then the far leaf's iff its box is hit,
// each triangle against the closest hit so far.  Same tests, same order, same counters; no stack, no loop
// over nodes, and every lane of the wave does it together.
template <bool LDS, bool STATS>
DEV void traverse_flat2(const RenderArgs& a, uint32_t root_rec, bool cull, f3 lo, f3 ld, f3 inv, MeshBest& best,
                        int& node_tests, int& tri_tests) {
    float4 q0, q1, q2, q3;
    load_wide<LDS>(a, root_rec, q0, q1, q2, q3);
    const float da = aabb_dist(lo, inv, q0, q1, INF), db = aabb_dist(lo, inv, q2, q3, INF);
    if (STATS) node_tests += 2;
    const bool left_closer = da < db;
    const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
    // (indices and counts of the two leaves are the same in every lane)
    const uint32_t a_i = __builtin_amdgcn_readfirstlane(fbits(q1.z)), a_c = __builtin_amdgcn_readfirstlane(fbits(q1.w));
    const uint32_t b_i = __builtin_amdgcn_readfirstlane(fbits(q3.z)), b_c = __builtin_amdgcn_readfirstlane(fbits(q3.w));
    const uint32_t first_i = left_closer ? a_i : b_i, second_i = left_closer ? b_i : a_i;
    const uint32_t first_c = near_d < INF ? (left_closer ? a_c : b_c) : 0u;
    const uint32_t second_c = far_d < INF ? (left_closer ? b_c : a_c) : 0u;
    if (STATS) tri_tests += (int)(first_c + second_c);
    const uint32_t tri0 = a.lay.tri_off, c_max = a_c > b_c ? a_c : b_c;
    for (uint32_t j = 0; j < c_max; ++j)
        if (j < first_c) {
            const uint32_t t = tri0 + (first_i + j) * TRI_ISECT_BYTES;
            tri_test<8>(lo, ld, ld4<LDS>(a, t), ld4<LDS>(a, t + 16), ld4<LDS>(a, t + 32), cull, first_i + j, best);
        }
    for (uint32_t j = 0; j < c_max; ++j)
        if (j < second_c) {
            const uint32_t t = tri0 + (second_i + j) * TRI_ISECT_BYTES;
            tri_test<8>(lo, ld, ld4<LDS>(a, t), ld4<LDS>(a, t + 16), ld4<LDS>(a, t + 32), cull, second_i + j, best);
        }
}

// A forest item (rt_device.h): meshes with an internal root that share one local space.  The
// shader visits them one after the other with all lanes in step; here every lane first marks
// the members whose root box its ray can hit, then walks ITS members back to back, so lanes
// busy in different meshes share the node-visit and triangle passes instead of taking turns.
// Per lane and per mesh the sequence of box and triangle tests is exactly traverse_mesh's;
// meshes are independent of each other (each starts from an infinite best distance,
// wgsl:292-296) and the caller's closest-hit update is order-free, so the result is the same.
// Skipping a member whose root box is missed is the monotonicity argument of intersect_scene
// (FOREST_CULLABLE members only, finite ray only).
template <bool LDS, bool STATS, bool SIMPLE, class Accept>
DEV void traverse_forest(const RenderArgs& a, uint32_t entry0, uint32_t n_members, f3 lo, f3 ld, f3 inv,
                         uint32_t* stack, Accept&& accept, int& node_tests, int& tri_tests) {
    const bool finite_ray = rtm::abs_(inv.x) < INF && rtm::abs_(inv.y) < INF && rtm::abs_(inv.z) < INF &&
                            rtm::abs_(lo.x) < INF && rtm::abs_(lo.y) < INF && rtm::abs_(lo.z) < INF;
    const uint32_t e0 = a.lay.forest_off + entry0 * FOREST_ENTRY_BYTES;
    uint32_t todo = 0;  // bit j: member j still to be traversed by this lane
    TIC(t5);
    for (uint32_t j = 0; j < n_members; ++j) {
        const uint32_t eo = e0 + j * FOREST_ENTRY_BYTES;
        const uint32_t flags = fbits(ld4<LDS>(a, eo).z);
        bool may_hit = true;
        if ((flags & FOREST_CULLABLE) != 0u && a.forest_cull != 0u)
            may_hit = !finite_ray || aabb_dist(lo, inv, ld4<LDS>(a, eo + 16), ld4<LDS>(a, eo + 32), INF) < INF;
        if (may_hit) todo |= 1u << j;
        else if (STATS) node_tests += 2;  // the shader's two root-level tests (wgsl:322)
    }
    TOC(t5, 5);
    const uint32_t tri0 = a.lay.tri_off;
    const LaneStack st{stack, a.stack_wide != 0u};
    uint32_t cur = 0, cur_count = 0, sp = 0, mesh = 0;
    bool have = false, cull = false;
    MeshBest b;
    b.t = INF;
    b.tri = 0xffffffffu;
    b.u = b.v = 0.0f;
    for (;;) {
        TIC(t8);
        if (!have) {
            if (b.tri != 0xffffffffu) {  // the mesh just left had a hit
                accept(mesh, b);
                b.tri = 0xffffffffu;
            }
            if (todo != 0u) {
                const uint32_t j = (uint32_t)__builtin_ctz(todo);
                todo &= todo - 1u;
                const float4 e = ld4<LDS>(a, e0 + j * FOREST_ENTRY_BYTES);
                cur = fbits(e.x);
                cur_count = 0;
                mesh = fbits(e.y);
                cull = SIMPLE || (fbits(e.z) & DMESH_GLASS) == 0u;
                b.t = INF;
                sp = 0;
                have = true;
            }
        }
        TOC(t8, 8);
        if (__ballot(have) == 0ull) break;
        TIC(t6);
        while (have && cur_count == 0) {  // descend to the next leaf
            DIAG(7);
            float4 q0, q1, q2, q3;
            load_wide<LDS>(a, cur, q0, q1, q2, q3);
            float da = aabb_dist(lo, inv, q0, q1, b.t);
            float db = aabb_dist(lo, inv, q2, q3, b.t);
            if (STATS) node_tests += 2;
            const bool left_closer = da < db;
            const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
            const uint32_t near_i = fbits(left_closer ? q1.z : q3.z), near_c = fbits(left_closer ? q1.w : q3.w);
            const uint32_t far_i = fbits(left_closer ? q3.z : q1.z), far_c = fbits(left_closer ? q3.w : q1.w);
            if (far_d < b.t) {
                stack_put(st, sp, far_i, far_c);
                ++sp;
            }
            if (near_d < b.t) {
                cur = near_i;
                cur_count = near_c;
            } else if (sp == 0) {
                have = false;
            } else {
                --sp;
                stack_get(st, sp, cur, cur_count);
            }
        }
        TOC(t6, 6);
        TIC(t7);
        if (have) {  // a leaf
            if (STATS) tri_tests += (int)cur_count;
            for (uint32_t j = 0; j < cur_count; ++j) {
                const uint32_t t = tri0 + (cur + j) * TRI_ISECT_BYTES;
                tri_test<8>(lo, ld, ld4<LDS>(a, t), ld4<LDS>(a, t + 16), ld4<LDS>(a, t + 32), cull, cur + j, b);
            }
            if (sp == 0) {
                have = false;
            } else {
                --sp;
                stack_get(st, sp, cur, cur_count);
            }
        }
        TOC(t7, 7);
    }
}

struct Hit {
    bool hit;
    float dst;
    f3 point, normal;
    float u, v;       // texture coordinates
    bool backface;
    uint32_t mat_off; // byte offset of the winner's material in the blob
    bool suspended;   // deferred walks: the segment stops before the big mesh (the caller parks the pixel)
};

// world-space hit point and distance of a mesh-local hit at parameter t (wgsl:380-381); m2w =
// byte offset of the four model_to_world columns
template <bool LDS>
DEV void world_hit(const RenderArgs& a, uint32_t m2w, f3 lo, f3 ld, f3 ro, float t, f3& whp, float& wdst) {
    const float4 c0 = ld4<LDS>(a, m2w), c1 = ld4<LDS>(a, m2w + 16), c2 = ld4<LDS>(a, m2w + 32), c3 = ld4<LDS>(a, m2w + 48);
    f3 lhp = lo + ld * t;
    whp = mat_cols_xyz(c0, c1, c2, c3, lhp, 1.0f);
    f3 dv = ro - whp;
    wdst = rtm::sqrt_(dot3(dv, dv));
}

// Closest-hit record of one scene intersection (wgsl:353-396).
struct Isect {
    float closest = INF;
    int object = 0;  // >= 0 mesh index, < 0 sphere -(index) - 1
    bool any = false;
    float s_dst = 0.0f;     // sphere winner: distance, inside flag
    bool s_inside = false;
    // mesh winner: its triangle, barycentrics and world hit point.  Kept compact (4 registers instead of MeshBest's
    // 6): w is (1 - u) - v again (the operations of wgsl:280) and of the determinant only the sign is used later
    // (wgsl:284-285; it cannot be zero, wgsl:268) -- it rides in the triangle index's top bit.
    float win_u = 0.0f, win_v = 0.0f;
    uint32_t win_tri = 0;   // | 0x80000000: determinant negative
    f3 win_point{0, 0, 0};
};

// closest-hit update of wgsl:383-391; equal distances go to the lower mesh index, which is
// what the shader's in-order loop with its strict `<` yields
struct CompactHit {  // what the mesh loop keeps of a mesh's closest triangle hit (see Isect)
    float t, u, v;
    uint32_t tri;  // | 0x80000000: determinant negative
};
DEV CompactHit compact(const MeshBest& b) { return CompactHit{b.t, b.u, b.v, b.tri}; }

DEV void isect_offer(Isect& I, uint32_t i, const CompactHit& b, f3 whp, float wdst) {
    if (wdst < I.closest || (wdst == I.closest && I.any && I.object >= 0 && (int)i < I.object)) {
        I.closest = wdst;
        I.any = true;
        I.object = (int)i;
        I.win_u = b.u;
        I.win_v = b.v;
        I.win_tri = b.tri;
        I.win_point = whp;
    }
}

// ray_sphere over all spheres (wgsl:223-256, 359-367)
template <bool LDS>
DEV void isect_spheres(const RenderArgs& a, f3 ro, f3 rd, Isect& I) {
    TIC(t2);
    for (uint32_t i = 0; i < a.n_spheres; ++i) {
        const float4 sp = ld4<LDS>(a, a.lay.sphere_off + i * SPHERE_BYTES);
        f3 oc = ro - f3{sp.x, sp.y, sp.z};
        float qa = dot3(rd, rd);
        float qb = 2.0f * dot3(oc, rd);
        float qc = dot3(oc, oc) - sp.w * sp.w;
        float disc = qb * qb - (4.0f * qa) * qc;
        if (disc >= 0.0f) {
            float s = rtm::sqrt_(disc);
            float dst_near = max_(0.0f, (-qb - s) / (2.0f * qa));
            float dst_far = (-qb + s) / (2.0f * qa);
            if (dst_far >= 0.001f) {
                bool inside = dst_near == 0.0f;
                float dst = inside ? dst_far : dst_near;
                if (dst < I.closest) {
                    I.closest = dst;
                    I.any = true;
                    I.object = -(int)i - 1;
                    I.s_dst = dst;
                    I.s_inside = inside;
                }
            }
        }
    }
    TOC(t2, 2);
}

// Per-object outputs that only the overall winner needs (normals, uv) are computed once after
// the loops from the same inputs, which yields the same bits.
// HYB (hybrid launches of a deferred-walk sequence, RenderArgs::hybrid): the staged blob lacks the deferred mesh's
// triangles; a winner on that mesh reads its shading record from the full blob.
template <bool LDS, bool SIMPLE = false, bool HYB = false>
DEV Hit isect_finish(const RenderArgs& a, const Isect& I, f3 ro, f3 rd) {
    Hit h;
    h.hit = I.any;
    h.dst = I.closest;
    h.mat_off = 0;
    h.point = f3{0, 0, 0};
    h.normal = f3{0, 0, 0};
    h.u = h.v = 0.0f;
    h.backface = false;
    h.suspended = false;
    TIC(t9);
    if (I.any) {
        DIAG(11);
        if (SIMPLE || I.object >= 0) {  // (SIMPLE: the scene has no spheres)
            const uint32_t mo = a.lay.mesh_off + (uint32_t)I.object * MESH_REC_BYTES;
            const uint32_t so = a.lay.shade_off + (I.win_tri & 0x7fffffffu) * TRI_SHADE_BYTES;
            const bool det_negative = (I.win_tri & 0x80000000u) != 0u;
            const float wu = I.win_u, wv = I.win_v, ww = (1.0f - wu) - wv;
            float4 s0, s1, s2, s3;
            if (HYB && (uint32_t)I.object == a.defer_mesh) {
                const float4* g = a.big_blob + ((a.big_shade_off + (I.win_tri & 0x7fffffffu) * TRI_SHADE_BYTES) >> 4);
                s0 = g[0]; s1 = g[1]; s2 = g[2]; s3 = g[3];
            } else {
                s0 = ld4<LDS>(a, so); s1 = ld4<LDS>(a, so + 16); s2 = ld4<LDS>(a, so + 32); s3 = ld4<LDS>(a, so + 48);
            }
            f3 n1{s0.x, s0.y, s0.z}, n2{s1.x, s1.y, s1.z}, n3{s2.x, s2.y, s2.z};
            f3 ln = normalize3<!LDS>((n1 * ww + n2 * wu) + n3 * wv) * (det_negative ? -1.0f : 1.0f);
            const float4 c0 = ld4<LDS>(a, mo + 64), c1 = ld4<LDS>(a, mo + 80), c2 = ld4<LDS>(a, mo + 96),
                         c3 = ld4<LDS>(a, mo + 112);
            h.normal = normalize3<!LDS>(mat_cols_xyz(c0, c1, c2, c3, ln, 0.0f));
            h.backface = det_negative;
            h.point = I.win_point;
            // uv = (uv1 * w + uv2 * u) + uv3 * v, uv1 = (u10,u11), uv2 = (u20,u21), uv3 = (u30,u31)
            h.u = (s0.w * ww + s2.w * wu) + s3.y * wv;
            h.v = (s1.w * ww + s3.x * wu) + s3.z * wv;
            h.mat_off = a.lay.mat_off + (uint32_t)I.object * MATERIAL_BYTES;
        } else {
            const uint32_t si = (uint32_t)(-I.object - 1);
            const float4 sp = ld4<LDS>(a, a.lay.sphere_off + si * SPHERE_BYTES);
            f3 c{sp.x, sp.y, sp.z};
            h.point = ro + rd * I.s_dst;
            f3 n = normalize3<!LDS>(h.point - c);
            h.normal = I.s_inside ? -n : n;
            h.backface = I.s_inside;
            const float pi = 3.1415926f;
            float theta = rtm::acos_(-h.normal.y);
            float phi = rtm::atan2_(-h.normal.z, -h.normal.x) + pi;
            h.u = phi / (2.0f * pi);
            h.v = theta / pi;
            h.mat_off = a.lay.mat_off + (a.n_meshes + si) * MATERIAL_BYTES;
        }
    }
    TOC(t9, 9);
    return h;
}

// wgsl:353-396 (+ ray_sphere :223-256).
// SIMPLE (few-mesh kernels only): the scene has no spheres, no glass and no textured material, and the camera no
// jitter (RenderArgs::simple, decided by the host per launch) -- the instantiation BASELINE configs 2, 3 and 5 run.
// The general code is the same code with those branches present; compiled out, they stop costing registers at the
// kernels' 96-VGPR ceiling and instruction-cache space.
template <bool LDS, bool STATS, bool TLAS, bool PARK = false, bool SIMPLE = false, bool HYB = false>
DEV Hit intersect_scene(const RenderArgs& a, f3 ro, f3 rd, uint32_t* stack, int& node_tests,
                        int& tri_tests, Isect& I_parked) {
    // this lane's TLAS stack column sits behind the wave's BVH stack columns
    uint32_t* tstack = stack + stack_dwords(a);
    bool suspended = false;
    Isect I;
    if constexpr (!SIMPLE) isect_spheres<LDS>(a, ro, rd, I);
    // meshes (wgsl:369-393), as items: single meshes and top-level trees over mesh root boxes
    f3 lo{0, 0, 0}, ld{0, 0, 0}, inv{0, 0, 0};
    bool cull_ok = false;
    // Cross-mesh pruning (many-mesh product kernels; RenderArgs::cross_prune).  The shader walks
    // every mesh from an infinite closest distance (wgsl:292-296) and compares WORLD distances afterwards (wgsl:381-383).
    // Here a local-space bound t_cut = closest * pa + pb is kept per local space such that a triangle hit at a local
    // parameter t >= t_cut / 1.125 provably fails `world_dst < closest.dst` (error budget at ITEM_NEW_XFORM below); boxes
    // whose entry distance is >= t_cut are not entered and a mesh's walk starts from t_cut instead of INF.  The counter
    // kernels (STATS: the shader's test counts, debug views) never prune.
    constexpr bool PRUNE = TLAS && !STATS;
    float pa = INF, pb = INF;  // (INF, INF: no pruning -- t_cut is INF whatever the closest distance)
    auto t_cut_of = [&](float closest) {
        const float tc = closest * pa + pb;
        return tc < INF ? tc : INF;  // (false for NaN and +inf)
    };
    auto accept_hit = [&](uint32_t i, const CompactHit& b) {
        DIAG(10);
        TIC(t19);
        f3 whp;
        float wdst;
        world_hit<LDS>(a, a.lay.mesh_off + i * MESH_REC_BYTES + 64u, lo, ld, ro, b.t, whp, wdst);
        isect_offer(I, i, b, whp, wdst);
        TOC(t19, 19);
    };
    auto accept_mesh_hit = [&](uint32_t i, const MeshBest& b) { accept_hit(i, compact(b)); };
    // Deferred offers.  The meshes every lane visits together (root-leaf and two-leaf items) are hit by a few lanes
    // each, and mostly by different lanes: offering each hit at once costs a sparsely populated pass of world_hit
    // per mesh.  A lane therefore keeps ONE hit pending and offers it when it gets another one (rare: a ray seldom
    // hits two of these meshes), when the local space changes, or at the end -- where most lanes offer together.
    // The offers are order-free (isect_offer breaks ties by mesh index), so nothing changes but the number of passes.
    // Where the pending hit lives: in registers in the LDS-scene kernels (their LDS pipe is the busy one: parked in
    // LDS it cost 1 % instead of saving 1.2 %), in the lane's BVH stack column in the global-memory kernels (5
    // dwords, RenderArgs::stack_entries is sized for it; nobody uses the column while these meshes are visited, and
    // the hit is offered before anything walks a BVH -- there the registers are the scarce resource: -1.6 %).
    CompactHit pend{INF, 0.0f, 0.0f, 0u};
    uint32_t pend_mesh = 0xffffffffu;
    bool have_pending = false;
    auto flush_pending = [&]() {
        if constexpr (LDS) {
            if (pend_mesh != 0xffffffffu) {
                accept_hit(pend_mesh, pend);
                pend_mesh = 0xffffffffu;
            }
        } else {
            if (have_pending) {
                CompactHit p;
                p.t = __uint_as_float(stack[0]); p.u = __uint_as_float(stack[64]); p.v = __uint_as_float(stack[128]);
                p.tri = stack[192];
                accept_hit(stack[256], p);
                have_pending = false;
            }
        }
    };
    auto offer_later = [&](uint32_t i, const MeshBest& b) {
        flush_pending();
        if constexpr (LDS) {
            pend = compact(b);
            pend_mesh = i;
        } else {
            stack[0] = __float_as_uint(b.t); stack[64] = __float_as_uint(b.u); stack[128] = __float_as_uint(b.v);
            stack[192] = b.tri;
            stack[256] = i;
            have_pending = true;
        }
    };
    auto visit_mesh = [&](uint32_t i, float4 hdr) {
        const uint32_t flags = fbits(hdr.x);
        MeshBest b;
        b.t = INF;
        b.tri = 0xffffffffu;
        b.u = b.v = 0.0f;
        if (!TLAS && fbits(hdr.z) == 0u) flush_pending();  // (a BVH walk uses the stack column / is where the register pressure peaks)
        traverse_mesh<LDS, STATS>(a, fbits(hdr.y), fbits(hdr.z),
                                  SIMPLE || (flags & DMESH_GLASS) == 0, (flags & DMESH_DEEP) != 0, lo, ld, inv, stack, b,
                                  node_tests, tri_tests);
        if (b.tri != 0xffffffffu) {
            if (!TLAS && fbits(hdr.z) != 0u) offer_later(i, b);  // root leaf: the wave visits it in step
            else accept_mesh_hit(i, b);
        }
    };
    for (uint32_t it = 0; it < a.n_items; ++it) {
        const float4 item = ld4<LDS>(a, a.lay.item_off + it * ITEM_BYTES);
        // item words are the same in every lane: keep them scalar
        const uint32_t kind = __builtin_amdgcn_readfirstlane(fbits(item.x));
        const uint32_t ia = __builtin_amdgcn_readfirstlane(fbits(item.y));
        if (kind & ITEM_NEW_XFORM) {
            if constexpr (!TLAS) flush_pending();  // (a pending hit belongs to the old local space)
            DIAG(3);
            TIC(t3);
            // meshes with bit-identical world_to_model share the local ray (same inputs, same bits)
            const uint32_t xo = a.lay.mesh_off + __builtin_amdgcn_readfirstlane(fbits(item.z)) * MESH_REC_BYTES;
            const float4 c0 = ld4<LDS>(a, xo), c1 = ld4<LDS>(a, xo + 16), c2 = ld4<LDS>(a, xo + 32),
                         c3 = ld4<LDS>(a, xo + 48);
            lo = mat_cols_xyz(c0, c1, c2, c3, ro, 1.0f);
            ld = normalize3<!LDS>(mat_cols_xyz(c0, c1, c2, c3, rd, 0.0f));
            inv = f3{rcp_(ld.x), rcp_(ld.y), rcp_(ld.z)};
            // the root-box arguments below are only proven for finite slab arithmetic
            if constexpr (TLAS)
                cull_ok = rtm::abs_(inv.x) < INF && rtm::abs_(inv.y) < INF && rtm::abs_(inv.z) < INF &&
                          rtm::abs_(lo.x) < INF && rtm::abs_(lo.y) < INF && rtm::abs_(lo.z) < INF;
            if constexpr (PRUNE) {
                pa = pb = INF;
                if (a.cross_prune != 0u && cull_ok) {
                    // The world distance the shader computes for a hit at local parameter t (world_hit: wgsl:380-381) is
                    //   wdst_c(t) = fl|ro - fl(M (lo + ld t))|,  M = model_to_world = (A | c).
                    // In exact arithmetic M (lo + ld t) = p0 + t g with p0 = A lo + c, g = A ld, so
                    //   |ro - M (lo + ld t)| >= t |g| - |ro - p0|.
                    // Rounding (u = 2^-24; S = max row sum of |A|, C = max |c_k|, L = max |lo_k|, R = max |ro_k|, |ld_k| <= 1 + 3u):
                    //   lhp = fl(lo + fl(ld t)):  per component <= 2.1u (t + L);  whp = fl(M lhp): <= 6.2u S (L + t) + 4.1u C;
                    //   dv = fl(ro - whp): <= 7.3u S (L + t) + 5.2u C + 1.1u R per component, sqrt(3) times that in norm;
                    //   dot + sqrt: relative 3u.  Hence
                    //   wdst_c(t) >= (t (|g| - 12.7u S) - |ro - p0| - 12.7u S L - 9.1u C - 2u R)(1 - 3u),
                    // and with g, p0, the two norms evaluated in binary32 below (another 5.2u S on g, 3u relative on each
                    // norm, 8.9u (S L + C) + 1.8u R on |ro - p0|) every term is covered by K = 2^-18 = 64u:
                    //   wdst_c(t) > closest   for every   t >= (closest + r0 + K (S L + C + R)) / (gn (1 - K) - K S) * (1 + 3K).
                    // t_cut is that bound times 1.125 (the slack that absorbs box / triangle inconsistencies)
                    // times (1 + 2^-16) (the (1 + 3K) above and the roundings of pa, pb and closest * pa + pb).
                    const float4 m0 = ld4<LDS>(a, xo + 64), m1 = ld4<LDS>(a, xo + 80), m2 = ld4<LDS>(a, xo + 96),
                                 m3 = ld4<LDS>(a, xo + 112), aux = ld4<LDS>(a, xo + 144);
                    const f3 g = mat_cols_xyz(m0, m1, m2, m3, ld, 0.0f);
                    const f3 d0 = ro - mat_cols_xyz(m0, m1, m2, m3, lo, 1.0f);
                    const float gn = rtm::sqrt_(dot3(g, g)), r0 = rtm::sqrt_(dot3(d0, d0));
                    const float L = max_(max_(rtm::abs_(lo.x), rtm::abs_(lo.y)), rtm::abs_(lo.z));
                    const float R = max_(max_(rtm::abs_(ro.x), rtm::abs_(ro.y)), rtm::abs_(ro.z));
                    const float K = 0x1p-18f, F = 1.125f + 0x1.2p-16f;  // F = 1.125 (1 + 2^-16), exact
                    const float D = gn * (1.0f - K) - K * aux.y;
                    const float B = r0 + K * ((aux.y * L + aux.z) + R);
                    if (D > 0.0f && B < INF) {  // (false for NaN)
                        const float fd = F / D;
                        pa = fd;
                        pb = fd * B;
                    }
                }
            }
            TOC(t3, 3);
        }
        if constexpr (!TLAS) if (kind & ITEM_FOREST) {
            // (the forest walk uses the stack column; and in the LDS-scene kernels a pending hit held across the
            // walk, where the register pressure peaks, was 7 more spilled dwords per lane: 95 MB of scratch write-back
            // per frame)
            flush_pending();
            traverse_forest<LDS, STATS, SIMPLE>(a, ia, __builtin_amdgcn_readfirstlane(fbits(item.w)), lo, ld, inv, stack,
                                                accept_mesh_hit, node_tests, tri_tests);
            continue;
        }
        if constexpr (!TLAS) {
            const float4 hdr = ld4<LDS>(a, a.lay.item_off + it * ITEM_BYTES + 16);
            if (PARK && (kind & ITEM_DEFER) != 0u && a.park != 0u) {
                // Deferred walk (RenderArgs::park): the segment stops here for the lanes whose ray can enter the
                // mesh; rt_walk_kernel walks it, the next launch offers its hit and finishes the segment.  A ray
                // that misses a root box which contains its children's boxes misses the mesh (the monotonicity
                // argument of root-box culling, finite rays only) and goes on as if it had walked it.
                flush_pending();
                bool may_hit = true;
                if ((kind & ITEM_DEFER_CULL) != 0u) {
                    const bool finite_ray = rtm::abs_(inv.x) < INF && rtm::abs_(inv.y) < INF && rtm::abs_(inv.z) < INF &&
                                            rtm::abs_(lo.x) < INF && rtm::abs_(lo.y) < INF && rtm::abs_(lo.z) < INF;
                    const uint32_t mo = a.lay.mesh_off + ia * MESH_REC_BYTES;
                    may_hit = !finite_ray || aabb_dist(lo, inv, ld4<LDS>(a, mo + 160), ld4<LDS>(a, mo + 176), INF) < INF;
                    if (STATS && !may_hit) node_tests += 2;  // the shader's two root-level tests (wgsl:322)
                    if (finite_ray && may_hit && a.park_levels != 0u) {
                        // The walk's first two levels, inline: until it reaches a leaf the walk's closest distance
                        // is INF, so these ARE its tests (wgsl:316-327) -- a ray that misses the root's children, or
                        // the children of those it hits, ends its walk there without a hit and need not park.
                        // (The root box is 12-22 % looser than its four grandchildren for the stand-ins' rays.)
                        float4 q0, q1, q2, q3;
                        load_wide<LDS>(a, fbits(hdr.y), q0, q1, q2, q3);
                        const bool hit_a = aabb_dist(lo, inv, q0, q1, INF) < INF, hit_b = aabb_dist(lo, inv, q2, q3, INF) < INF;
                        int tests = 2;
                        bool reaches = false;
                        if (hit_a) {
                            if (fbits(q1.w) != 0u) {
                                reaches = true;
                            } else {
                                float4 r0, r1, r2, r3;
                                load_wide<LDS>(a, fbits(q1.z), r0, r1, r2, r3);
                                reaches = aabb_dist(lo, inv, r0, r1, INF) < INF || aabb_dist(lo, inv, r2, r3, INF) < INF;
                                tests += 2;
                            }
                        }
                        if (hit_b && !reaches) {
                            if (fbits(q3.w) != 0u) {
                                reaches = true;
                            } else {
                                float4 r0, r1, r2, r3;
                                load_wide<LDS>(a, fbits(q3.z), r0, r1, r2, r3);
                                reaches = aabb_dist(lo, inv, r0, r1, INF) < INF || aabb_dist(lo, inv, r2, r3, INF) < INF;
                                tests += 2;
                            }
                        }
                        if (!reaches) {
                            may_hit = false;
                            if (STATS) node_tests += tests;
                        }
                    }
                }
                if (may_hit) suspended = true;
                continue;
            }
            if (kind & ITEM_FLAT2) {
                MeshBest b;
                b.t = INF;
                b.tri = 0xffffffffu;
                b.u = b.v = 0.0f;
                TIC(t18);
                traverse_flat2<LDS, STATS>(a, __builtin_amdgcn_readfirstlane(fbits(hdr.y)), SIMPLE || (fbits(hdr.x) & DMESH_GLASS) == 0, lo, ld, inv,
                                           b, node_tests, tri_tests);
                TOC(t18, 18);
                if (b.tri != 0xffffffffu) offer_later(ia, b);
            } else {
                visit_mesh(ia, hdr);
            }
        } else {
            // Many-mesh kernels: one traversal loop serves single meshes and top-level trees, so the mesh walk is
            // instantiated once.  The walk's state (see below):
            const LaneStack st{stack, a.stack_wide != 0u};
            const uint32_t tri0 = a.lay.tri_off;
            uint32_t cur = 0, cur_count = 0, sp = 0, mesh = 0, tsp = 0;
            bool have = false, cull = false;
            MeshBest b;
            b.t = INF;
            b.tri = 0xffffffffu;
            b.u = b.v = 0.0f;
            // (PRUNE) may this item's meshes be cut?  (wave-uniform)
            const bool prune_item = PRUNE && (kind & ITEM_PRUNE) != 0u;
            if ((kind & ITEM_TLAS) == 0u) {
                bool may_hit = true;
                const float4 hdr = ld4<LDS>(a, a.lay.item_off + it * ITEM_BYTES + 16);  // (flags, root, root count)
                const float seed = prune_item && fbits(hdr.z) == 0u ? t_cut_of(I.closest) : INF;
                if (a.cull_roots && fbits(hdr.z) == 0u) {
                    // Mesh-level culling that cannot change the result.  The shader
                    // never tests the root box, only its two children (wgsl:316-321) -- but the
                    // root box contains the child boxes and IEEE subtraction / multiplication /
                    // min / max are monotone, so with finite operands each child's slab interval
                    // lies inside the root's: a ray that misses the root box fails both child
                    // tests and the mesh contributes nothing.  (0 * inf = NaN would break
                    // monotonicity: cull_ok.)
                    const uint32_t mo = a.lay.mesh_off + ia * MESH_REC_BYTES;
                    const float4 rmin = ld4<LDS>(a, mo + 160), rmax = ld4<LDS>(a, mo + 176);
                    // (PRUNE: ... and a root box entered at or beyond t_cut holds nothing that can win)
                    may_hit = !cull_ok || aabb_dist(lo, inv, rmin, rmax, seed) < INF;
                    // (a culled mesh -- always one with an internal root -- still counts its two root-level tests)
                    if (STATS && !may_hit) node_tests += 2;
                }
                if (may_hit) {
                    const uint32_t flags = fbits(hdr.x);
                    if ((flags & DMESH_DEEP) != 0u) {
                        visit_mesh(ia, hdr);  // (the shader's literal stack: walked on its own)
                    } else {
                        mesh = ia;
                        cur = fbits(hdr.y);
                        cur_count = fbits(hdr.z);
                        cull = (flags & DMESH_GLASS) == 0u;
                        b.t = seed;
                        have = true;
                    }
                }
            } else {
                // Top-level tree over the root boxes of fbits(item.w) meshes (all with internal
                // roots, one shared local space).  Every tree box contains the root boxes below
                // it, so by the same monotonicity argument a ray that misses a tree box misses
                // every mesh below it.  Each mesh still counts its two root-level tests (wgsl:322).
                if (STATS) node_tests += 2 * (int)fbits(item.w);
                tstack[0] = ia;  // internal tree node
                tsp = 1;
            }
            // The top-level tree is an ITERATOR of candidate meshes, the walk is shared: a lane that has no mesh in
            // hand pops its tree stack until it finds one (a few box tests), then all lanes walk THEIR meshes in
            // common node-visit and triangle passes and fetch the next candidate as they finish (like a forest
            // item) -- instead of every lane walking a whole mesh to the end, the wave in step, each time one of its
            // tree pops is a mesh.  Per lane the meshes are visited in the same order and every mesh's walk is
            // traverse_mesh's, entry for entry; the counters are kept as before.  A tree's reference to a mesh
            // (TLAS_REF_*, rt_device.h) carries the mesh's index, its root record and its culling rule, so entering
            // a mesh costs no load (it was a dependent one, of the mesh header: 13 % of the loads on a ray's
            // critical path in the sponza-sized scene).
            // (Measured on the 200-mesh stand-in: 9.0 -> 7.9 ms per frame.  Going one step further -- tree nodes and
            // mesh nodes visited in the SAME passes, one walk over both levels -- was 9.0 again: the passes did not
            // merge, 45 per iteration instead of 35 + 14, because a wave-iteration lasts as long as its longest ray
            // and that ray's tree and mesh visits are sequential either way.)
            {
                for (;;) {
                    TIC(t8);
                    if (!have) {
                        if (b.tri != 0xffffffffu) {  // the mesh just left had a hit
                            accept_mesh_hit(mesh, b);
                            b.tri = 0xffffffffu;
                        }
                        while (tsp > 0 && !have) {
                            --tsp;
                            const uint32_t e = tstack[tsp * 64];
                            if (e & 0x80000000u) {
                                DIAG(18);
                                if (STATS) node_tests -= 2;  // counted above; the walk counts them again
                                mesh = (e >> TLAS_REF_MESH_SHIFT) & TLAS_REF_MESH_MASK;
                                cur = e & TLAS_REF_ROOT_MASK;
                                cur_count = 0;
                                cull = (e & TLAS_REF_GLASS) == 0u;
                                b.t = prune_item ? t_cut_of(I.closest) : INF;
                                sp = 0;
                                have = true;
                            } else {
                                DIAG(17);
                                float4 q0, q1, q2, q3;
                                load_tlas<LDS>(a, e, q0, q1, q2, q3);
                                if constexpr (PRUNE) {
                                    // the nearer box is visited first (the order of the mesh loop is free: isect_offer
                                    // breaks ties by mesh index), so that what lies behind a hit meets a small t_cut
                                    const float tc = prune_item ? t_cut_of(I.closest) : INF;
                                    const float da = aabb_dist(lo, inv, q0, q1, tc), db = aabb_dist(lo, inv, q2, q3, tc);
                                    const bool hit_a = !cull_ok || da < INF, hit_b = !cull_ok || db < INF;
                                    const bool a_first = da <= db;
                                    const uint32_t ea = fbits(q1.z) | (fbits(q1.w) ? 0x80000000u : 0u);
                                    const uint32_t eb = fbits(q3.z) | (fbits(q3.w) ? 0x80000000u : 0u);
                                    if (a_first ? hit_b : hit_a) {
                                        tstack[tsp * 64] = a_first ? eb : ea;
                                        ++tsp;
                                    }
                                    if (a_first ? hit_a : hit_b) {
                                        tstack[tsp * 64] = a_first ? ea : eb;
                                        ++tsp;
                                    }
                                } else {
                                    const bool hit_a = !cull_ok || aabb_dist(lo, inv, q0, q1, INF) < INF;
                                    const bool hit_b = !cull_ok || aabb_dist(lo, inv, q2, q3, INF) < INF;
                                    if (hit_b) {
                                        tstack[tsp * 64] = fbits(q3.z) | (fbits(q3.w) ? 0x80000000u : 0u);
                                        ++tsp;
                                    }
                                    if (hit_a) {
                                        tstack[tsp * 64] = fbits(q1.z) | (fbits(q1.w) ? 0x80000000u : 0u);
                                        ++tsp;
                                    }
                                }
                            }
                        }
                    }
                    TOC(t8, 8);
                    if (__ballot(have) == 0ull) break;
                    DIAG(19);
                    TIC(t6);
                    while (have && cur_count == 0) {  // descend to the next leaf (traverse_mesh's step)
                        DIAG(7);
                        float4 q0, q1, q2, q3;
                        load_wide<LDS>(a, cur, q0, q1, q2, q3);
                        float da = aabb_dist(lo, inv, q0, q1, b.t);
                        float db = aabb_dist(lo, inv, q2, q3, b.t);
                        if (STATS) node_tests += 2;
                        const bool left_closer = da < db;
                        const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
                        const uint32_t near_i = fbits(left_closer ? q1.z : q3.z), near_c = fbits(left_closer ? q1.w : q3.w);
                        const uint32_t far_i = fbits(left_closer ? q3.z : q1.z), far_c = fbits(left_closer ? q3.w : q1.w);
                        if (far_d < b.t) {
                            stack_put(st, sp, far_i, far_c);
                            ++sp;
                        }
                        if (near_d < b.t) {
                            cur = near_i;
                            cur_count = near_c;
                        } else if (sp == 0) {
                            have = false;
                        } else {
                            --sp;
                            stack_get(st, sp, cur, cur_count);
                        }
                    }
                    TOC(t6, 6);
                    TIC(t7);
                    if (have) {  // a leaf
                        if (STATS) tri_tests += (int)cur_count;
                        for (uint32_t j = 0; j < cur_count; ++j) {
                            const uint32_t t = tri0 + (cur + j) * TRI_ISECT_BYTES;
                            tri_test<8>(lo, ld, ld4<LDS>(a, t), ld4<LDS>(a, t + 16), ld4<LDS>(a, t + 32), cull, cur + j, b);
                        }
                        if (sp == 0) {
                            have = false;
                        } else {
                            --sp;
                            stack_get(st, sp, cur, cur_count);
                        }
                    }
                    TOC(t7, 7);
                }
            }
        }
    }
    if constexpr (!TLAS) flush_pending();
    if (PARK && suspended) {
        I_parked = I;
        Hit h;
        h.hit = false;
        h.suspended = true;
        return h;
    }
    return isect_finish<LDS, SIMPLE, HYB>(a, I, ro, rd);
}

// field byte offsets inside rt_material
enum : uint32_t {
    M_COLOR = 0, M_EMISSION = 16, M_SPECCOL = 32, M_ABSORB = 48, M_ABSORB_S = 64, M_EMISSION_S = 68,
    M_SMOOTH = 72, M_SPECULAR = 76, M_IOR = 80, M_FLAG = 84, M_DIFFUSE_IDX = 88, M_NORMAL_IDX = 92
};

DEV f4 sample_texture(const RenderArgs& a, int index, float u, float v) {
    float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    DIAG(26);
    if (index >= 0 && (uint32_t)index < a.n_textures) {
        // global-address-space pointers: global_load, not flat_load
        typedef const __attribute__((address_space(1))) uint32_t* GWords;
        typedef const __attribute__((address_space(1))) float* GFloats;
        const DTexture* tp = a.textures + index;
        const uint8_t* base = tp->rgba8;
        const uint32_t w = tp->width, h = tp->height;
        if (w != 0u && h != 0u && base != nullptr)
            rtm::sample_bilinear_words((GWords)(const void*)base, w, h, (GFloats)(const void*)a.srgb_lut, u, v, out);
    }
    return f4{out[0], out[1], out[2], out[3]};
}

// wgsl:214-221, literally (the definition; the render kernels call environment_light below)
DEV float sky_gradient_t_literal(float y) { return rtm::pow_(smoothstep_(0.0f, 0.4f, y), 0.35f); }
DEV float ground_to_sky_t_literal(float y) { return smoothstep_(-0.01f, 0.0f, y); }
DEV float sun_literal(float m) { return rtm::pow_(m, 500.0f) * 0.1f; }  // m = max(0, dot(dir, (0.1, 1, 0.1)))

// The same three values with the branches of their definitions taken apart -- same bits, ENUMERATED, not argued:
// rt_sweep_kernel (which = 2, 3, 4) compares each with its literal form above for every float in [-1.5, 1.5] / [0, 1.5]
// on the device (tests/test_gpu_device_units.py::test_sky_shortcuts_equal_the_literal_forms_for_every_float), and the
// literal forms are what the oracle evaluates.
//   smoothstep(lo, hi, y) clamps (y - lo) / (hi - lo) to [0, 1]: below lo it is 0, from hi on it is 1, and pow(0, 0.35) =
//   exp2(0.35 * log2(0)) = exp2(-inf) = 0, pow(1, 0.35) = exp2(0) = 1 (rt_transc.h);
//   pow(m, 500) = exp2(500 * log2(m)) and exp2_ returns +0 below -150: every m < 0.8 (log2 <= -0.32) gives +0.
// An escaping ray mostly looks down or along the horizon: the two logarithm / exponential pairs and the two IEEE divisions
// are then skipped for the whole wave (sky was 11 % of config 2's wave time, tools/diag.py --time).
DEV float sky_gradient_t(float y) {
    if (y <= 0.0f) return 0.0f;
    if (y >= 0.4f) return 1.0f;
    return sky_gradient_t_literal(y);  // (NaN comes here)
}
DEV float ground_to_sky_t(float y) {
    if (y >= 0.0f) return 1.0f;
    if (y <= -0.01f) return 0.0f;
    return ground_to_sky_t_literal(y);
}
DEV float sun_term(float m) {
    if (m < 0.8f) return 0.0f;
    return sun_literal(m);
}

// wgsl:214-221
DEV f4 environment_light(f3 dir) {
    const f4 SKY_HORIZON{1.0f, 1.0f, 1.0f, 0.0f};
    const f4 SKY_ZENITH{0.0788092f, 0.36480793f, 0.7264151f, 0.0f};
    const f4 GROUND{0.35f, 0.3f, 0.35f, 0.0f};
    float sky_t = sky_gradient_t(dir.y);
    float g2s = ground_to_sky_t(dir.y);
    f4 sky = mix4(SKY_HORIZON, SKY_ZENITH, sky_t);
    float sun = sun_term(max_(0.0f, dot3(dir, f3{0.1f, 1.0f, 0.1f})));
    return mix4(GROUND, sky, g2s) + sun * (g2s >= 1.0f ? 1.0f : 0.0f);
}

// wgsl:208-212
DEV float reflectance(float cos_theta, float ior) {
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0 *= r0;
    return r0 + (1.0f - r0) * rtm::pow_(1.0f - cos_theta, 5.0f);
}

// Blocks are dealt round-robin over the 8 XCDs (block b -> XCD b % 8).  Give
// each XCD a contiguous run of work: bijection for any grid size.
DEV uint32_t xcd_remap(uint32_t b, uint32_t nb) {
    uint32_t q = nb >> 3, r = nb & 7u;
    uint32_t x = b & 7u;
    uint32_t start = x * q + (x < r ? x : r);
    return start + (b >> 3);
}

// Pixel w (0..63) of local 8x8 tile `tile`: frame coordinates and output row.
struct PixelCoord {
    uint32_t x, y, out_row;
    bool valid;
};

template <class A>
DEV PixelCoord pixel_at(const A& a, uint32_t tx, uint32_t ty, uint32_t w);

template <class A>
DEV PixelCoord pixel_of(const A& a, uint32_t tile, uint32_t w) {
    return pixel_at(a, tile % a.tiles_x, tile / a.tiles_x, w);
}

// pixel w of the tile in column tx, local tile row ty
template <class A>
DEV PixelCoord pixel_at(const A& a, uint32_t tx, uint32_t ty, uint32_t w) {
    PixelCoord p;
    p.x = tx * 8u + (w & 7u);
    const uint32_t strip = ty * a.strip_world + a.strip_rank;  // global 8-row strip
    p.y = strip * 8u + (w >> 3);
    p.out_row = a.strip_world > 1u ? ty * 8u + (w >> 3) : p.y;
    p.valid = p.x < a.params.width && p.y < a.params.height;
    return p;
}

// wgsl:154-161
template <class A>
DEV void store_texel(const A& a, uint32_t x, uint32_t out_row, uint32_t batch_frame, f4 cur) {
    float4* texel = a.image + (size_t)out_row * a.params.width + x;
    if (a.batch_frames != 0u) {
        // frame batch: the sample goes to its frame's scratch image; rt_blend_frames_kernel blends in frame order
        texel[(size_t)batch_frame * a.batch_stride] = make_float4(cur.x, cur.y, cur.z, cur.w);
    } else if (a.params.frames >= 1) {
        float4 prev = *texel;
        // wgsl:157: weight = 1 / f32(frames + 1), evaluated once per launch by the host with the
        // same two operations (RenderArgs::blend_weight, ::blend_rest)
        const float weight = a.blend_weight, om = a.blend_rest;
        *texel = make_float4(prev.x * om + cur.x * weight, prev.y * om + cur.y * weight,
                             prev.z * om + cur.z * weight, prev.w * om + cur.w * weight);
    } else {
        *texel = make_float4(cur.x, cur.y, cur.z, cur.w);
    }
}

// LDS map of a workgroup: [scene blob (LDS kernels)] [4 wave regions] [4 tile-cost tables].
// A wave region is lane-interleaved (dword k of lane l at [k * 64 + l], conflict-free):
//   [primary-ray memo, PIXEL_MEMO_DWORDS x 64, only when pixel_cache == 1]
//   [lane state, LANE_STATE_DWORDS x 64] [BVH stack, stack_entries x 2 x 64] [TLAS stack, tlas_entries x 64]
// Everything is addressed from ONE per-lane pointer (the lane-state base) with immediate
// offsets, so the whole map costs a single VGPR.
DEV uint32_t lane_state_dwords(const RenderArgs& a) { return total_in_lds(a.lds_scene != 0u) ? LANE_STATE_DWORDS : 0u; }
DEV uint32_t wave_region_dwords(const RenderArgs& a) {
    return (a.pixel_cache == 1u ? PIXEL_MEMO_DWORDS * 64u : 0u) + lane_state_dwords(a) * 64u +
           stack_dwords(a) + a.tlas_entries * 64u;
}

#if RT_EXPERIMENTS || RT_TOP_BRANCH || RT_TLAS_BRANCH
DEV uint32_t top_lds_off16(const RenderArgs& a) { return (WAVES_PER_BLOCK * wave_region_dwords(a) + WAVES_PER_BLOCK * 8u * 3u) >> 2; }
DEV uint32_t tlas_lds_off16(const RenderArgs& a) { return top_lds_off16(a) + a.top_count * 4u; }
#endif

// Stage the scene blob into LDS (coalesced 16-byte loads, whole workgroup) and
// return this lane's base pointer (its lane state; see the map above).
template <bool LDS>
DEV uint32_t* block_prologue(const RenderArgs& a) {
    uint32_t base = 0;  // in float4 units
    if constexpr (LDS) {
        const uint32_t n16 = a.lay.bytes >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += BLOCK_THREADS) lds_mem[i] = a.blob[i];
        __syncthreads();
        base = n16;
    }
#if RT_EXPERIMENTS
    else if (a.top_count != 0u || a.tlas_lds != 0u) {
        // the top of the big mesh's BVH: top_count consecutive wide records
        const uint32_t src = (a.lay.wide_off + a.top_base * WIDE_REC_BYTES) >> 4, dst = top_lds_off16(a);
        for (uint32_t i = threadIdx.x; i < a.top_count * 4u; i += BLOCK_THREADS) lds_mem[dst + i] = a.blob[src + i];
        // the top-level tree(s): tlas_lds consecutive wide records (every ray's first dependent fetches)
        const uint32_t tsrc = a.lay.tlas_off >> 4, tdst = tlas_lds_off16(a);
        for (uint32_t i = threadIdx.x; i < a.tlas_lds * 4u; i += BLOCK_THREADS) lds_mem[tdst + i] = a.blob[tsrc + i];
        __syncthreads();
    }
#endif
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    return reinterpret_cast<uint32_t*>(lds_mem + base) + wave * wave_region_dwords(a) +
           (a.pixel_cache == 1u ? PIXEL_MEMO_DWORDS * 64u : 0u) + lane;
}
template <bool LDS>
DEV uint32_t* stack_of(uint32_t* lane_base) { return lane_base + (LDS ? LANE_STATE_DWORDS * 64u : 0u); }  // LDS = total_in_lds(...)

// The persistent kernel's per-wave tile-cost tables follow the wave regions.
template <bool LDS>
DEV uint32_t* cost_table_of_wave(const RenderArgs& a) {
    const uint32_t base = LDS ? (a.lay.bytes >> 4) : 0u;  // float4 units
    return reinterpret_cast<uint32_t*>(lds_mem + base) + WAVES_PER_BLOCK * wave_region_dwords(a) +
           (threadIdx.x >> 6) * (8u * 3u);
}

}  // namespace

// ---------------------------------------------------------------------------
// wgsl `main` + `frag` + `trace` (wgsl:144-162, 473-500, 398-471)
// ---------------------------------------------------------------------------
namespace {

struct CameraConsts {  // wave-uniform
    f3 origin, right, up;
    float sx, sy;
};

template <class A>
DEV CameraConsts camera_consts(const A& a) {
    CameraConsts c;
    c.origin = f3{a.camera.cam_to_world[3][0], a.camera.cam_to_world[3][1], a.camera.cam_to_world[3][2]};
    c.right = f3{a.camera.cam_to_world[0][0], a.camera.cam_to_world[0][1], a.camera.cam_to_world[0][2]};
    c.up = f3{a.camera.cam_to_world[1][0], a.camera.cam_to_world[1][1], a.camera.cam_to_world[1][2]};
    c.sx = (float)a.params.width;
    c.sy = (float)a.params.height;
    return c;
}

struct PixelState {
    uint32_t x, out_row;   // where the texel goes
    uint32_t rng;          // wgsl:475, one stream per pixel
    int32_t j;             // sample index (wgsl:487)
    // current path (wgsl:398-471); `total` (wgsl:486) lives in the lane's LDS state region
    // (LANE_STATE_DWORDS)
    f4 total;
    f3 ro, rd;
    f4 T, light;
    int32_t seg;
    bool fresh;
    uint32_t meta;         // rays of this pixel so far (bits 0-15, saturating) | cost-table slot (bits 16-18)
                           // | index of the pixel's frame inside a frame batch (bits 19-31)
};

// wgsl:479-482: the pixel's focus point
template <class A>
DEV f3 focus_point_of(const A& a, const CameraConsts& c, uint32_t x, uint32_t y) {
    const float fx = (float)x, fy = (float)y;
    const float uvx = fx / (c.sx - 1.0f), uvy = fy / (c.sy - 1.0f);
    const f3 local_focus = f3{uvx - 0.5f, uvy - 0.5f, 1.0f} *
                           f3{a.camera.view_params[0], a.camera.view_params[1], a.camera.view_params[2]};
    return mat_xyz(&a.camera.cam_to_world[0][0], local_focus, 1.0f);
}
// frame row of a pixel from its output row (they differ in the compact strip layout)
template <class A>
DEV uint32_t frame_row_of(const A& a, uint32_t out_row) {
    return a.strip_world > 1u ? ((out_row >> 3) * a.strip_world + a.strip_rank) * 8u + (out_row & 7u) : out_row;
}

// wgsl:475 for pixel (x, y) of the full frame; batch_frame = index of the frame inside a frame batch
// (Params.frames advances by one per frame, app.rs:44-53)
template <bool LDS, class A>
DEV void pixel_begin(const A& a, const CameraConsts& c, PixelState& s, uint32_t* ls, uint32_t x,
                     uint32_t y, uint32_t out_row, uint32_t batch_frame = 0u) {
    const float fx = (float)x, fy = (float)y;
    const int32_t fr = (int32_t)((uint32_t)a.params.frames + batch_frame);
    const uint32_t absf = fr < 0 ? 0u - (uint32_t)fr : (uint32_t)fr;
    s.rng = (uint32_t)(fy * c.sx + fx) + absf * 719393u;
    s.x = x;
    s.out_row = out_row;
    if constexpr (LDS) {
        ls[0] = 0u; ls[64] = 0u; ls[128] = 0u; ls[192] = 0u;  // total = 0
    }
    s.total = f4{0, 0, 0, 0};
    s.j = 0;
    s.fresh = true;
    s.seg = 0;
    s.T = f4{1, 1, 1, 1};
    s.light = f4{0, 0, 0, 0};
    s.ro = f3{0, 0, 0};
    s.rd = f3{0, 0, 1};
    s.meta = 0;
}

enum : uint32_t { MEMO_HIT = 1u, MEMO_BACKFACE = 2u, MEMO_RAY = 4u, MEMO_HIT_VALID = 8u };

// Kernel arguments for the rarely executed paths (taking a new pixel, finishing one, camera
// jitter): re-read from the kernarg segment where they are needed.  The asm keeps the compiler
// from hoisting those loads -- and every wave-uniform float computation that depends on them
// (there is no scalar float ALU: they would be parked in VGPRs) -- out of the render loop.
typedef const __attribute__((address_space(4))) RenderArgs ColdArgs;
DEV ColdArgs& cold_args() {
    ColdArgs* p = (ColdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
}

// This lane's primary-ray memo: in LDS right below the lane state (pixel_cache == 1), or in
// global memory (== 2; the pointer is rebuilt per use rather than kept in two VGPRs).  `f`
// is instantiated once per address space, so the LDS copy uses ds_ and the other global_
// instructions (a runtime-selected pointer would mean flat_ accesses).
// WF (wavefront kernels, pixel_cache == 3): `ls` IS the memo's address -- one 13-dword record per path slot in global
// memory, in blocks of 64 slots with the same lane-interleaved layout (dword k of slot i at [k * 64 + (i & 63)]).
template <bool WF = false, class F>
DEV void with_memo(const RenderArgs& a, uint32_t* ls, F&& f) {
    if constexpr (WF) {
        typedef __attribute__((address_space(1))) uint32_t* GlobalPtr;
        f((GlobalPtr)ls);
        return;
    }
    if (a.pixel_cache == 2u) {
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));
        typedef __attribute__((address_space(1))) uint32_t* GlobalPtr;
        f((GlobalPtr)a.pixel_cache_mem + (size_t)(blockIdx.x * WAVES_PER_BLOCK + (t >> 6)) * (PIXEL_MEMO_DWORDS * 64u) +
          (t & 63u));
    } else {
        typedef __attribute__((address_space(3))) uint32_t* LdsPtr;
        f((LdsPtr)(ls - PIXEL_MEMO_DWORDS * 64u));
    }
}

// Called when a lane takes a new pixel: computes the pixel's constant primary ray when the
// camera has no jitter and no -0 is involved (see path_step), else marks the memo empty.
DEV size_t primary_index(uint32_t width, uint32_t x, uint32_t y) {
    const uint32_t tiles_x = (width + 7u) >> 3;
    return ((size_t)(y >> 3) * tiles_x + (x >> 3)) * 64u + (((y & 7u) << 3) | (x & 7u));
}

// pixel_cache == 4 (round 5): the lane's memo IS its entry of the primary table.  With a complete table
// (RenderArgs::primary_complete) a pixel's memo equals its table entry for the pixel's whole life -- nothing ever rewrites
// it (memo_hit_store would store the same bits) --, so the kernels whose memo does not fit the LDS read the 13 dwords
// straight from the read-only table at every sample instead of copying them into a per-wave buffer in global memory
// first: that copy was written one dword per 32-byte sector (the buffer is lane-interleaved and lanes take pixels a few
// at a time) and did not survive in the L2 between a pixel's samples either -- 740 of the 774 MB the sponza-sized stand-in
// wrote per frame and 1.4 of its 2.5 GB of reads.  Same values, same operations on them.
struct TableMemo {
    typedef const __attribute__((address_space(1))) uint32_t* P;
    P p;
    DEV uint32_t operator[](uint32_t i) const { return p[i >> 6]; }  // (the callers index the lane-interleaved layout: k * 64)
};
// read-only access to the memo of the pixel `s` holds: the table entry (pixel_cache == 4) or with_memo's buffer
// (TABLE: compiled into the kernels that read the scene from global memory only -- the LDS-scene kernels keep their memo
// in LDS, and the extra branch cost the headline instantiation 8 B of scratch per lane)
template <bool WF = false, bool TABLE = true, class F>
DEV void with_memo_ro(const RenderArgs& a, uint32_t* ls, const PixelState& s, F&& f) {
    if constexpr (!WF && TABLE) {
        if (a.pixel_cache == 4u) {  // wave-uniform
            DIAG(20);
            ColdArgs& ca = cold_args();
            f(TableMemo{(TableMemo::P)a.primary + primary_index(ca.params.width, s.x, frame_row_of(ca, s.out_row)) * 16u});
            return;
        }
    }
    with_memo<WF>(a, ls, f);
}

// the memoised primary ray of pixel (x, y): direction, and whether the ray is constant at all
template <bool SQ = false, class A>
DEV f3 memo_ray_of(const A& ca, const CameraConsts& c, uint32_t x, uint32_t y, bool& constant_ray) {
    const f3 focus = focus_point_of(ca, c, x, y);
    auto not_neg_zero = [](float v) { return __float_as_uint(v) != 0x80000000u; };
    auto finite3 = [](f3 v) { return rtm::abs_(v.x) < INF && rtm::abs_(v.y) < INF && rtm::abs_(v.z) < INF; };
    constant_ray = __float_as_uint(ca.camera.defocus_strength) == 0u &&
                   __float_as_uint(ca.camera.diverge_strength) == 0u && finite3(c.right) && finite3(c.up) &&
                   not_neg_zero(c.origin.x) && not_neg_zero(c.origin.y) && not_neg_zero(c.origin.z) &&
                   not_neg_zero(focus.x) && not_neg_zero(focus.y) && not_neg_zero(focus.z);
    f3 rd{0, 0, 0};
    if (constant_ray) {
        // x + (+-0) + (+-0) is x for x != 0 and +0 for x = +0: the jitter signs cannot matter
        // (so the origin is the same for every pixel: RenderArgs::memo_ro, computed by the host
        // with these same operations)
        const f3 ro = (c.origin + c.right * 0.0f) + c.up * 0.0f;
        const f3 jfp = (focus + c.right * 0.0f) + c.up * 0.0f;
        rd = normalize3<SQ>(jfp - ro);
        rd = normalize3<SQ>(rd);  // wgsl:400
    }
    return rd;
}

// A pixel's memo from the primary table (see pixel_cache_begin)
template <bool WF = false, class A>
DEV void memo_from_table(const RenderArgs& a, const A& ca, const PixelState& s, uint32_t y, uint32_t* ls) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // (a plain vector: global_load_dwordx4 through the typed pointer)
    const __attribute__((address_space(1))) u32x4* v =
        (const __attribute__((address_space(1))) u32x4*)a.primary + primary_index(ca.params.width, s.x, y) * 4u;
    const u32x4 e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
    DIAG(25);
    with_memo<WF>(a, ls, [&](auto pc) {
        pc[0] = e0.x; pc[64] = e0.y; pc[2 * 64] = e0.z; pc[3 * 64] = e0.w;
        pc[4 * 64] = e1.x; pc[5 * 64] = e1.y; pc[6 * 64] = e1.z; pc[7 * 64] = e1.w;
        pc[8 * 64] = e2.x; pc[9 * 64] = e2.y; pc[10 * 64] = e2.z; pc[11 * 64] = e2.w;
        pc[12 * 64] = e3.x;
    });
}

template <bool WF = false, bool SQ = false, class A>
DEV void pixel_cache_begin(const RenderArgs& a, const A& ca, const CameraConsts& c, const PixelState& s, uint32_t* ls) {
    if (!a.pixel_cache || (SQ && a.pixel_cache == 4u)) return;  // (4: the memo is the pixel's table entry, read in place; SQ = a global-memory kernel)
    bool constant_ray;
    f3 rd;
    const uint32_t y = frame_row_of(ca, s.out_row);
    if (a.primary) {
        // Camera, frame size and scene have not changed since rt_primary_kernel filled the table: the pixel's whole memo
        // -- its constant primary ray and, with option "primary_hits", that ray's hit (a pure function of the ray and
        // the scene) -- computed there once, with every lane busy, instead of here with a handful and again in every
        // frame of the accumulation.  One 64-byte entry per pixel (the memo's 13 dwords, see path_begin), in the order in
        // which lanes take pixels (8x8 tile by tile, row-major inside): the lanes refilled together read neighbouring
        // entries.  Same inputs, same operations, same bits as the per-sample path.
        memo_from_table<WF>(a, ca, s, y, ls);
        return;
    }
    rd = memo_ray_of<SQ>(ca, c, s.x, y, constant_ray);
    with_memo<WF>(a, ls, [&](auto pc) {
        pc[0] = __float_as_uint(rd.x); pc[64] = __float_as_uint(rd.y); pc[128] = __float_as_uint(rd.z);
        pc[12 * 64] = constant_ray ? MEMO_RAY : 0u;
    });
}

// One iteration of the per-lane state machine = path_begin (start the next sample, decide what
// this lane does now) + the segment's hit (memo or scene intersection) + path_end (shading,
// russian roulette, end of path).  path_end returns true when the pixel's last sample ended.
enum : uint32_t {
    STEP_END = 0,       // bounce budget used up: the path ends without another segment
    STEP_WAIT = 1,      // wants a traversal but the wave voted against one now
    STEP_REUSE = 2,     // memoised primary ray: take its hit from the memo
    STEP_TRAVERSE = 3,  // intersect the scene
    STEP_RESUME = 4,    // deferred walks: the segment's hit is the parked closest-hit record + the big mesh's walk
};

template <bool STATS, bool SIMPLE = false, bool WF = false, bool SQ = false>
DEV uint32_t path_begin(const RenderArgs& a, PixelState& s, uint32_t* ls, uint32_t& starve) {
    const int32_t nb = a.params.number_of_bounces;
    // Primary-ray memo.  With defocus_strength = diverge_strength = +0 (the default camera) the
    // camera jitter is +-0, and unless a component of the camera origin or of the pixel's focus
    // point is -0 the sums `origin + right*j.x + up*j.y` do not depend on those signs: every
    // sample of the pixel starts with the same ray (pixel_cache_begin computes it once).  The
    // ray and its intersection (a pure function of the ray) are memoised per lane:
    // pc[0..2] = rd, pc[3..11] = hit record, pc[12] = mat_off | MEMO_* flags.  Same inputs, same
    // bits; the four RNG draws are still made.  Counter builds (STATS) re-intersect so that the
    // node/triangle test counters stay the shader's.
    const bool cache_on = a.pixel_cache != 0;  // wave-uniform
    bool reuse_hit = false;
    DIAG(0);
    TIC(t1);
    if (s.fresh) {  // wgsl:487-495: next sample of this pixel
        DIAG(1);
        uint32_t st = 0u;
        if (cache_on) {
            f3 mrd{0, 0, 0};
            with_memo_ro<WF, SQ>(a, ls, s, [&](auto pc) {  // (SQ = a kernel that reads the scene from global memory)
                st = pc[12 * 64];
                mrd = f3{__uint_as_float(pc[0]), __uint_as_float(pc[64]), __uint_as_float(pc[128])};
            });
            s.rd = mrd;
        }
        if (st & MEMO_RAY) {
            // the two disks' angle and radius draws: four steps of the generator's LCG
            // s -> s * 747796405 + 2891336453 (mod 2^32) in one, (A^4, C (A^3 + A^2 + A + 1)); exact
            s.rng = s.rng * 2200120369u + 878960812u;
            s.ro = f3{a.memo_ro[0], a.memo_ro[1], a.memo_ro[2]};
            reuse_hit = !STATS && (st & MEMO_HIT_VALID) != 0u;
        } else {
            ColdArgs& ca = cold_args();
            const CameraConsts c = camera_consts(ca);
            float jx, jy, kx, ky;
            // (SIMPLE: both strengths are +0 -- a pixel only gets here when a -0 is involved in its ray)
            disk_jitter(s.rng, SIMPLE ? 0.0f : ca.camera.defocus_strength, c.sx, jx, jy);
            disk_jitter(s.rng, SIMPLE ? 0.0f : ca.camera.diverge_strength, c.sx, kx, ky);
            s.ro = (c.origin + c.right * jx) + c.up * jy;
            const f3 focus = focus_point_of(ca, c, s.x, frame_row_of(ca, s.out_row));
            f3 jfp = (focus + c.right * kx) + c.up * ky;
            s.rd = normalize3<SQ>(jfp - s.ro);
            s.rd = normalize3<SQ>(s.rd);  // wgsl:400
        }
        s.T = f4{1, 1, 1, 1};
        s.light = f4{0, 0, 0, 0};
        s.seg = 0;
        s.fresh = false;
    }
    TOC(t1, 1);
    if (s.seg > nb) return STEP_END;
    // Intersection vote.  Lanes whose segment is a memoised primary ray need no traversal; the
    // others do.  The traversal is the expensive part of an iteration and the wave pays for
    // it whenever a single lane needs it, so it only runs when enough of the lanes here want
    // it (or somebody has already waited); the waiting lanes simply take their turn in the
    // next iteration, by which time the memoised-primary lanes have moved on to secondary
    // segments and want it too.  Pure scheduling: no lane's sequence of operations changes.
    if (cache_on && !WF) {  // (the wavefront kernels traverse in a kernel of their own: nothing to vote on)
        const uint32_t n_here = (uint32_t)__popcll(__ballot(true));
        const uint32_t n_want = (uint32_t)__popcll(__ballot(!reuse_hit));
        const bool run = n_want * 8u >= n_here * a.vote_eighths || starve >= a.vote_patience;
        starve = (n_want != 0u && !run) ? starve + 1u : 0u;
#if defined(RT_DIAG)
        if (n_want == 0u) { DIAG(28); } else if (run && n_want == n_here) { DIAG(29); } else if (run) { DIAG(30); } else { DIAG(31); }
#endif
        if (!reuse_hit && !run) return STEP_WAIT;  // nothing about this lane has changed
    }
    return reuse_hit ? STEP_REUSE : STEP_TRAVERSE;
}

template <bool WF = false, bool TABLE = true>
DEV void memo_hit_load(const RenderArgs& a, const PixelState& s, uint32_t* ls, Hit& hit) {
    TIC(t11);
    DIAG(21);
    with_memo_ro<WF, TABLE>(a, ls, s, [&](auto pc) {
        hit.dst = __uint_as_float(pc[3 * 64]);
        hit.point = f3{__uint_as_float(pc[4 * 64]), __uint_as_float(pc[5 * 64]), __uint_as_float(pc[6 * 64])};
        hit.normal = f3{__uint_as_float(pc[7 * 64]), __uint_as_float(pc[8 * 64]), __uint_as_float(pc[9 * 64])};
        hit.u = __uint_as_float(pc[10 * 64]);
        hit.v = __uint_as_float(pc[11 * 64]);
        const uint32_t w = pc[12 * 64];
        hit.mat_off = w & ~15u;  // material records are 16-byte aligned
        hit.hit = (w & MEMO_HIT) != 0u;
        hit.backface = (w & MEMO_BACKFACE) != 0u;
    });
    TOC(t11, 11);
}

// the memoised ray's hit goes into the memo the first time it is computed
template <bool STATS, bool WF = false, bool TABLE = true>
DEV void memo_hit_store(const RenderArgs& a, const PixelState& s, uint32_t* ls, const Hit& hit) {
    TIC(t13);
    // (pixel_cache == 4: the table entry already holds this hit -- a complete table has the hit of every memoised ray, so
    // such a ray is never traversed again and there is nothing to store)
    const bool cache_on = a.pixel_cache != 0 && !(TABLE && a.pixel_cache == 4u);
    if (cache_on && !STATS && s.seg == 0) {
        with_memo<WF>(a, ls, [&](auto pc) {
            if ((pc[12 * 64] & MEMO_RAY) == 0u) return;
            // the memoised ray's hit
            pc[3 * 64] = __float_as_uint(hit.dst);
            pc[4 * 64] = __float_as_uint(hit.point.x); pc[5 * 64] = __float_as_uint(hit.point.y); pc[6 * 64] = __float_as_uint(hit.point.z);
            pc[7 * 64] = __float_as_uint(hit.normal.x); pc[8 * 64] = __float_as_uint(hit.normal.y); pc[9 * 64] = __float_as_uint(hit.normal.z);
            pc[10 * 64] = __float_as_uint(hit.u);
            pc[11 * 64] = __float_as_uint(hit.v);
            pc[12 * 64] = (hit.hit ? (hit.mat_off & ~15u) | MEMO_HIT : 0u) | (hit.backface ? MEMO_BACKFACE : 0u) |
                          MEMO_RAY | MEMO_HIT_VALID;
        });
    }
    TOC(t13, 13);
}

// LDS: material reads from LDS; TOTAL_LDS: the pixel sum lives in the lane's LDS state
// FAST_MISS (the product's memo-using instantiations): a memoised primary ray that leaves the scene ends the pixel --
// every remaining sample is this same segment (the same ray, throughput 1, no light yet, the same sky; the draws of
// its zero-strength jitter are made and never read), so the lane adds the sample's light once per remaining sample
// right here, in order, instead of coming back for one iteration each; `more_reused` counts those segments.
template <bool LDS, bool TOTAL_LDS, bool SIMPLE = false, bool FAST_MISS = false>
DEV bool path_end(const RenderArgs& a, PixelState& s, uint32_t* ls, uint32_t mode, const Hit& hit,
                  uint32_t& n_segments, uint32_t* more_reused = nullptr) {
    const int32_t nb = a.params.number_of_bounces;
    bool end_path = true;
    uint32_t again = 0u;  // further samples of the pixel that are this sample again
    if (mode != STEP_END) {
        n_segments += 1;
        if ((s.meta & 0xffffu) != 0xffffu) s.meta += 1;
        if (!hit.hit) {
            DIAG(12);
            TIC(t10);
            if (a.params.skybox != 0) s.light = s.light + s.T * environment_light(s.rd);
            if constexpr (FAST_MISS) {
                if (a.fast_miss != 0u && s.seg == 0) {  // (pixel_cache != 0, wave-uniform)
                    uint32_t st = 0u;
                    with_memo_ro<false, !LDS>(a, ls, s, [&](auto pc) { st = pc[12 * 64]; });
                    if ((st & (MEMO_RAY | MEMO_HIT_VALID)) == (MEMO_RAY | MEMO_HIT_VALID))
                        again = (uint32_t)(a.params.rays_per_pixel - 1 - s.j);
                }
            }
            TOC(t10, 10);
        } else {
            TIC(t12);
            const uint32_t mo = hit.mat_off;
            const int flag = ldi<LDS>(a, mo + M_FLAG);
            f3 rd = s.rd;
            f4 T = s.T;
            s.ro = hit.point;
            if (!SIMPLE && flag == RT_MATERIAL_GLASS) {  // wgsl:414-436
                DIAG(14);
                if (hit.backface) {
                    const float4 ab = ld4<LDS>(a, mo + M_ABSORB);
                    float as = ldf<LDS>(a, mo + M_ABSORB_S);
                    float ex = ((-hit.dst) * ab.x) * as;
                    float ey = ((-hit.dst) * ab.y) * as;
                    float ez = ((-hit.dst) * ab.z) * as;
                    T = f4{T.x * rtm::exp_(ex), T.y * rtm::exp_(ey), T.z * rtm::exp_(ez), 1.0f};
                }
                float mior = ldf<LDS>(a, mo + M_IOR);
                float ior = hit.backface ? mior : (1.0f / mior);
                f3 reflect_dir = reflect3(rd, hit.normal);
                f3 refract_dir = refract3(rd, hit.normal, ior);
                float cos_theta = min_(dot3(-rd, hit.normal), 1.0f);
                float sin_theta = rtm::sqrt_(1.0f - cos_theta * cos_theta);
                bool cannot_refract = ior * sin_theta > 1.0f;
                bool follow_reflection = cannot_refract;
                if (!cannot_refract) follow_reflection = reflectance(cos_theta, ior) > rand_(s.rng);
                f3 diffuse_dir = normalize3<!LDS>(hit.normal + rand_unit_sphere<!LDS>(s.rng));
                reflect_dir = normalize3<!LDS>(mix3(diffuse_dir, reflect_dir, ldf<LDS>(a, mo + M_SPECULAR)));
                refract_dir = normalize3<!LDS>(mix3(-diffuse_dir, refract_dir, ldf<LDS>(a, mo + M_SMOOTH)));
                rd = follow_reflection ? reflect_dir : refract_dir;
                s.ro = hit.point + (1e-4f * hit.normal) * sign_(dot3(hit.normal, rd));
            } else {  // wgsl:437-460
                DIAG(13);
                const float4 msc = ld4<LDS>(a, mo + M_ABSORB_S);  // (absorb_s, emission_s, smoothness, specular)
                bool is_spec = msc.w >= rand_(s.rng);
                f3 sph = rand_unit_sphere<!LDS>(s.rng);
                f3 diffuse_dir = sph * sign_(dot3(hit.normal, sph));
                f3 specular_dir = reflect3(rd, hit.normal);
                const float4 ec = ld4<LDS>(a, mo + M_EMISSION);
                float es = msc.y;
                f4 emitted{ec.x * es, ec.y * es, ec.z * es, ec.w * es};
                rd = normalize3<!LDS>(mix3(diffuse_dir, specular_dir, msc.z * (is_spec ? 1.0f : 0.0f)));
                s.light = s.light + emitted * T;
                f4 color;
                const int diffuse_index = ldi<LDS>(a, mo + M_DIFFUSE_IDX);
                if (!SIMPLE && flag == RT_MATERIAL_TEXTURE && diffuse_index != -1) {
                    color = sample_texture(a, diffuse_index, hit.u, hit.v);
                } else {
                    const float4 cc = ld4<LDS>(a, mo + M_COLOR);
                    color = f4{cc.x, cc.y, cc.z, cc.w};
                }
                const float4 sc = ld4<LDS>(a, mo + M_SPECCOL);
                f4 spec{sc.x, sc.y, sc.z, sc.w};
                T = T * (is_spec ? spec : color);
            }
            s.rd = rd;
            float p = max_(T.x, max_(T.y, T.z));  // wgsl:462-466
            bool die = rand_(s.rng) >= p;
            if (!die) {
                T = T * rcp_(p);
                s.seg += 1;
                end_path = s.seg > nb;
            }
            s.T = T;
            TOC(t12, 12);
        }
    }
    if (end_path) {  // wgsl:496
        if constexpr (TOTAL_LDS) {
            f4 t{__uint_as_float(ls[0]), __uint_as_float(ls[64]), __uint_as_float(ls[128]), __uint_as_float(ls[192])};
            t = t + s.light;  // total += incoming_light
            if constexpr (FAST_MISS)
                for (uint32_t k = 0; k < again; ++k) t = t + s.light;
            ls[0] = __float_as_uint(t.x); ls[64] = __float_as_uint(t.y); ls[128] = __float_as_uint(t.z); ls[192] = __float_as_uint(t.w);
        } else {
            s.total = s.total + s.light;
            if constexpr (FAST_MISS)
                for (uint32_t k = 0; k < again; ++k) s.total = s.total + s.light;
        }
        s.j += 1;
        if constexpr (FAST_MISS) {
            if (again != 0u) {
                s.j += (int32_t)again;
                n_segments += again;
                *more_reused += again;
                const uint32_t rays = (s.meta & 0xffffu) + again;
                s.meta = (s.meta & 0xffff0000u) | (rays < 0xffffu ? rays : 0xffffu);
            }
        }
        s.fresh = true;
        return s.j >= a.params.rays_per_pixel;
    }
    return false;
}

// returns PATH_CONTINUE, PATH_PIXEL_DONE (the pixel's last sample ended) or PATH_PARK (the pixel was parked in front
// of the deferred mesh, RenderArgs::park)
enum : uint32_t { PATH_CONTINUE = 0, PATH_PIXEL_DONE = 1, PATH_PARK = 2 };
template <bool TOTAL_LDS>
DEV void park_store(const RenderArgs& a, uint32_t slot, const PixelState& s, uint32_t* ls, const Isect& I);
DEV void park_load_hit(const RenderArgs& a, uint32_t slot, Isect& I, CompactHit& walked);

template <bool LDS, bool STATS, bool TLAS, bool PARK = false, bool SIMPLE = false, bool HYB = false>
DEV uint32_t path_step(const RenderArgs& a, PixelState& s, uint32_t* ls, uint32_t& starve, uint32_t& n_segments,
                       bool& reused, bool& reused_pre, uint32_t& more_reused, int& node_tests, int& tri_tests, uint32_t resume_slot = 0xffffffffu) {
#if RT_PRESTEP
    // Pre-step: a lane at the start of a sample whose primary hit is memoised takes that segment NOW -- it needs no
    // traversal -- and goes on to its next segment in the same iteration, so that it votes for (and joins) this
    // iteration's traversal instead of shading here while the others wait and traversing one iteration later.  Pure
    // scheduling: the lane's operations are path_begin's fresh branch, memo_hit_load and path_end, in that order, as before.
    reused_pre = false;
    if constexpr (!STATS && RT_PRESTEP_PARK >= (PARK ? 1 : 0)) {
        if ((!PARK || resume_slot == 0xffffffffu) && a.pixel_cache != 0u && s.fresh && a.params.number_of_bounces >= 0) {
            uint32_t st = 0u;
            f3 rd{0, 0, 0};
            with_memo_ro<false, !LDS>(a, ls, s, [&](auto pc) {
                st = pc[12 * 64];
                rd = f3{__uint_as_float(pc[0]), __uint_as_float(pc[64]), __uint_as_float(pc[128])};
            });
            if ((st & (MEMO_RAY | MEMO_HIT_VALID)) == (MEMO_RAY | MEMO_HIT_VALID)) {
                s.rd = rd;
                s.rng = s.rng * 2200120369u + 878960812u;  // (the two disks' four draws: path_begin)
                s.ro = f3{a.memo_ro[0], a.memo_ro[1], a.memo_ro[2]};
                s.T = f4{1, 1, 1, 1};
                s.light = f4{0, 0, 0, 0};
                s.seg = 0;
                s.fresh = false;
                Hit mh;
                mh.hit = false;
                mh.suspended = false;
                memo_hit_load<false, !LDS>(a, s, ls, mh);
                reused_pre = true;
                if (path_end<LDS, total_in_lds(LDS), SIMPLE, true>(a, s, ls, STEP_REUSE, mh, n_segments, &more_reused)) return PATH_PIXEL_DONE;
            }
        }
    }
#endif
    // (a resumed pixel was parked behind path_begin: its segment has begun)
    const uint32_t mode = PARK && resume_slot != 0xffffffffu ? (uint32_t)STEP_RESUME : path_begin<STATS, SIMPLE, false, !LDS>(a, s, ls, starve);
    // segments served from the memo: the caller counts them per wave, OUTSIDE its `if (active)` (a ballot + a scalar
    // add with every lane of the wave present -- counted in here, under the divergent branch, the sum lived in the
    // active lanes only and was lost for every iteration lane 0 sat out)
    reused = mode == STEP_REUSE;
    if (mode == STEP_WAIT) return PATH_CONTINUE;
    Hit hit;
    hit.hit = false;
    hit.suspended = false;
    if (mode == STEP_REUSE) {
        memo_hit_load<false, !LDS>(a, s, ls, hit);
    } else if (mode == STEP_TRAVERSE) {
        TIC(t0);
        Isect I;
        hit = intersect_scene<LDS, STATS, TLAS, PARK, SIMPLE, HYB>(a, s.ro, s.rd, stack_of<total_in_lds(LDS)>(ls), node_tests, tri_tests, I);
        TOC(t0, 0);
        if constexpr (PARK && !TLAS) {
            if (a.park != 0u) {  // (wave-uniform)
                // park the suspended lanes' pixels: consecutive records, one counter update per wave
                const unsigned long long here = __ballot(true), parking = __ballot(hit.suspended);
                TIC(t20);
                if (parking != 0ull) {
                    uint32_t slot0 = 0;
                    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(here)) slot0 = atomicAdd(a.q_out_count, (uint32_t)__popcll(parking));
                    slot0 = __builtin_amdgcn_readlane(slot0, __builtin_ctzll(here));
                    if (hit.suspended) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(parking >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)parking, 0u));
                        park_store<total_in_lds(LDS)>(a, slot0 + rank, s, ls, I);
                        TOC(t20, 20);
                        return PATH_PARK;
                    }
                }
                TOC(t20, 20);
            }
        }
        memo_hit_store<STATS, false, !LDS>(a, s, ls, hit);
    } else if (PARK && !TLAS && mode == STEP_RESUME) {
        // behind the deferred mesh's walk: offer the walk's hit (the local ray: the operations of ITEM_NEW_XFORM),
        // finish the segment
        TIC(t22);
        Isect I;
        CompactHit walked;
        park_load_hit(a, resume_slot, I, walked);
        if (walked.tri != 0xffffffffu) {
            const uint32_t xo = a.lay.mesh_off + a.defer_xform * MESH_REC_BYTES;
            const float4 c0 = ld4<LDS>(a, xo), c1 = ld4<LDS>(a, xo + 16), c2 = ld4<LDS>(a, xo + 32), c3 = ld4<LDS>(a, xo + 48);
            const f3 lo = mat_cols_xyz(c0, c1, c2, c3, s.ro, 1.0f);
            const f3 ld = normalize3<!LDS>(mat_cols_xyz(c0, c1, c2, c3, s.rd, 0.0f));
            f3 whp;
            float wdst;
            world_hit<LDS>(a, a.lay.mesh_off + a.defer_mesh * MESH_REC_BYTES + 64u, lo, ld, s.ro, walked.t, whp, wdst);
            isect_offer(I, a.defer_mesh, walked, whp, wdst);
        }
        hit = isect_finish<LDS, SIMPLE, HYB>(a, I, s.ro, s.rd);
        memo_hit_store<STATS, false, !LDS>(a, s, ls, hit);
        TOC(t22, 22);
    }
    return path_end<LDS, total_in_lds(LDS), SIMPLE, !STATS>(a, s, ls, mode, hit, n_segments, &more_reused) ? PATH_PIXEL_DONE : PATH_CONTINUE;
}

// Park records (rt_device.h): the whole state of a pixel between two segments.
template <bool TOTAL_LDS>
DEV void park_store(const RenderArgs& a, uint32_t slot, const PixelState& s, uint32_t* ls, const Isect& I) {
    float4* q = a.q_out + (size_t)(slot >> 6) * (PARK_PLANES * 64u) + (slot & 63u);
    DIAG(22);
    auto u = [](uint32_t v) { return __uint_as_float(v); };
    f4 total = s.total;
    if constexpr (TOTAL_LDS)
        total = f4{__uint_as_float(ls[0]), __uint_as_float(ls[64]), __uint_as_float(ls[128]), __uint_as_float(ls[192])};
    // (the memo's 13 dwords as named values: an indexed local array kept its 52 bytes of scratch frame)
    float4 m0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), m1 = m0, m2 = m0;
    uint32_t m12 = 0u;
    const bool memo_kept = a.pixel_cache != 0u && a.primary_complete == 0u;  // (else the table holds it: RenderArgs::primary_complete)
    if (memo_kept)
        with_memo(a, ls, [&](auto pc) {
            m0 = make_float4(u(pc[0 * 64]), u(pc[1 * 64]), u(pc[2 * 64]), u(pc[3 * 64]));
            m1 = make_float4(u(pc[4 * 64]), u(pc[5 * 64]), u(pc[6 * 64]), u(pc[7 * 64]));
            m2 = make_float4(u(pc[8 * 64]), u(pc[9 * 64]), u(pc[10 * 64]), u(pc[11 * 64]));
            m12 = pc[12 * 64];
        });
    q[0 * 64] = make_float4(u(s.x), u(s.out_row), u(s.rng), u((uint32_t)s.j));
    q[1 * 64] = make_float4(u((uint32_t)s.seg), u(s.fresh ? 1u : 0u), u(s.meta), u(m12));
    q[2 * 64] = make_float4(s.ro.x, s.ro.y, s.ro.z, s.rd.x);
    q[3 * 64] = make_float4(s.rd.y, s.rd.z, s.T.x, s.T.y);
    q[4 * 64] = make_float4(s.T.z, s.T.w, s.light.x, s.light.y);
    q[5 * 64] = make_float4(s.light.z, s.light.w, I.win_point.y, I.win_point.z);
    q[6 * 64] = make_float4(total.x, total.y, total.z, total.w);
    if (memo_kept) {
        q[7 * 64] = m0;
        q[8 * 64] = m1;
        q[9 * 64] = m2;
    }
    q[10 * 64] = make_float4(I.closest, u((uint32_t)I.object), u((I.any ? 1u : 0u) | (I.s_inside ? 2u : 0u)), I.s_dst);
    q[11 * 64] = make_float4(I.win_u, I.win_v, u(I.win_tri), I.win_point.x);
}
// the closest-hit record of the parked segment and the walk's result
DEV void park_load_hit(const RenderArgs& a, uint32_t slot, Isect& I, CompactHit& walked) {
    const float4* q = a.q_in + (size_t)(slot >> 6) * (PARK_PLANES * 64u) + (slot & 63u);
    const float4 p10 = q[10 * 64], p11 = q[11 * 64], p5 = q[5 * 64], p13 = q[13 * 64];
    I.closest = p10.x;
    I.object = (int)fbits(p10.y);
    I.any = (fbits(p10.z) & 1u) != 0u;
    I.s_inside = (fbits(p10.z) & 2u) != 0u;
    I.s_dst = p10.w;
    I.win_u = p11.x;
    I.win_v = p11.y;
    I.win_tri = fbits(p11.z);
    I.win_point = f3{p11.w, p5.z, p5.w};
    walked = CompactHit{p13.x, p13.y, p13.z, fbits(p13.w)};
}
template <bool TOTAL_LDS>
DEV void park_load(const RenderArgs& a, uint32_t slot, PixelState& s, uint32_t* ls) {
    const float4* q = a.q_in + (size_t)(slot >> 6) * (PARK_PLANES * 64u) + (slot & 63u);
    DIAG(23);
    const float4 p0 = q[0 * 64], p1 = q[1 * 64], p2 = q[2 * 64], p3 = q[3 * 64], p4 = q[4 * 64], p5 = q[5 * 64], p6 = q[6 * 64];
    s.x = fbits(p0.x); s.out_row = fbits(p0.y); s.rng = fbits(p0.z); s.j = (int32_t)fbits(p0.w);
    s.seg = (int32_t)fbits(p1.x); s.fresh = fbits(p1.y) != 0u; s.meta = fbits(p1.z);
    s.ro = f3{p2.x, p2.y, p2.z};
    s.rd = f3{p2.w, p3.x, p3.y};
    s.T = f4{p3.z, p3.w, p4.x, p4.y};
    s.light = f4{p4.z, p4.w, p5.x, p5.y};
    s.total = f4{p6.x, p6.y, p6.z, p6.w};
    if constexpr (TOTAL_LDS) {
        ls[0] = fbits(p6.x); ls[64] = fbits(p6.y); ls[128] = fbits(p6.z); ls[192] = fbits(p6.w);
    }
    if (a.pixel_cache == 4u) {
        // (the memo is the pixel's table entry, read in place)
    } else if (a.pixel_cache != 0u && a.primary_complete != 0u) {
        const ColdArgs& ca = cold_args();
        memo_from_table(a, ca, s, frame_row_of(ca, s.out_row), ls);
    } else if (a.pixel_cache != 0u) {
        const float4 p7 = q[7 * 64], p8 = q[8 * 64], p9 = q[9 * 64];
        with_memo(a, ls, [&](auto pc) {
            pc[0 * 64] = fbits(p7.x); pc[1 * 64] = fbits(p7.y); pc[2 * 64] = fbits(p7.z); pc[3 * 64] = fbits(p7.w);
            pc[4 * 64] = fbits(p8.x); pc[5 * 64] = fbits(p8.y); pc[6 * 64] = fbits(p8.z); pc[7 * 64] = fbits(p8.w);
            pc[8 * 64] = fbits(p9.x); pc[9 * 64] = fbits(p9.y); pc[10 * 64] = fbits(p9.z); pc[11 * 64] = fbits(p9.w);
            pc[12 * 64] = fbits(p1.w);
        });
    }
}

// wgsl:498 + 154-161
template <bool LDS, class A>
DEV void pixel_finish(const A& a, const PixelState& s, const uint32_t* ls) {
    float n = (float)a.params.rays_per_pixel;
    f4 total = s.total;
    if constexpr (LDS)
        total = f4{__uint_as_float(ls[0]), __uint_as_float(ls[64]), __uint_as_float(ls[128]), __uint_as_float(ls[192])};
    // wgsl:498: total / f32(rays_per_pixel).  When the count is a power of two its reciprocal is
    // exact, and x * (1/n) and x / n are the same real number rounded once: the same float.
    const float r = a.spp_reciprocal;  // 0: not a power of two
    const uint32_t bf = s.meta >> 19;
    if (r != 0.0f) store_texel(a, s.x, s.out_row, bf, f4{total.x * r, total.y * r, total.z * r, total.w * r});
    else store_texel(a, s.x, s.out_row, bf, f4{total.x / n, total.y / n, total.z / n, total.w / n});
}

// Tile-cost feedback (rays per 8x8 tile, read by the next frame's scheduler).  A persistent
// wave renders every pixel of the tiles it pulls, so it sums a tile's rays in a small LDS
// table (8 tiles in flight, 3 dwords each: tile + 1, count << 24 | rays, pixel count) and
// issues ONE global add per tile; only a tile evicted from the table while still in flight
// falls back to per-pixel adds.  All updates are adds, so the total is exact either way.
constexpr uint32_t COST_SLOTS = 8;
template <class A>
DEV void tile_cost_pull(const A& a, uint32_t* tbl, uint32_t slot, uint32_t tile) {
    const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const uint32_t y0 = (ty * a.strip_world + a.strip_rank) * 8u;
    const uint32_t vx = a.params.width - tx * 8u < 8u ? a.params.width - tx * 8u : 8u;
    const uint32_t vy = y0 >= a.params.height ? 0u : (a.params.height - y0 < 8u ? a.params.height - y0 : 8u);
    const uint32_t old = tbl[slot * 3];
    if (old != 0u) atomicAdd(&a.tile_cost[old - 1u], tbl[slot * 3 + 1] & 0xffffffu);  // evict a straggler
    tbl[slot * 3] = vx * vy ? tile + 1u : 0u;
    tbl[slot * 3 + 1] = 0u;
    tbl[slot * 3 + 2] = vx * vy;
}
DEV void tile_cost_add(const RenderArgs& a, uint32_t* tbl, const PixelState& s) {
    const uint32_t tile = (s.out_row >> 3) * a.tiles_x + (s.x >> 3);
    const uint32_t rays = s.meta & 0xffffu, slot = (s.meta >> 16) & (COST_SLOTS - 1u);
    if (tbl[slot * 3] == tile + 1u) {
        const uint32_t add = (1u << 24) | rays;
        const uint32_t now = atomicAdd(&tbl[slot * 3 + 1], add) + add;
        if ((now >> 24) == tbl[slot * 3 + 2]) {  // last pixel of the tile
            atomicAdd(&a.tile_cost[tile], now & 0xffffffu);
            tbl[slot * 3] = 0u;
        }
    } else {
        atomicAdd(&a.tile_cost[tile], rays);
    }
}

template <bool STATS>
DEV void flush_counters(const RenderArgs& a, uint32_t n_segments, uint32_t n_reused_wave, int node_tests, int tri_tests,
                        uint32_t more_reused = 0u) {
    if (a.counters && __ballot(n_segments != 0u || (STATS && (node_tests | tri_tests) != 0)) != 0ull) {
        atomicAdd(&a.counters->segments, (unsigned long long)n_segments);
        if (more_reused != 0u) atomicAdd(&a.counters->reused, (unsigned long long)more_reused);  // (per lane)
        // (n_reused_wave is the same in every lane: accumulated from ballots taken with the whole wave present)
        if (n_reused_wave != 0u && (threadIdx.x & 63u) == 0u) atomicAdd(&a.counters->reused, (unsigned long long)n_reused_wave);
        if (STATS) {
            atomicAdd(&a.counters->node_tests, (unsigned long long)node_tests);
            atomicAdd(&a.counters->triangle_tests, (unsigned long long)tri_tests);
        }
    }
}

}  // namespace

// Variant 1: one wave per 8x8 tile (the reference's dispatch shape), four
// tiles per workgroup.
template <bool LDS, bool STATS, bool TLAS, bool SIMPLE>
__global__ void __launch_bounds__(BLOCK_THREADS, RT_MIN_WAVES) rt_render_tiles_kernel(const RenderArgs a) {
    uint32_t* ls = block_prologue<LDS>(a);
    const CameraConsts cam = camera_consts(a);
    const uint32_t slot = xcd_remap(blockIdx.x, gridDim.x) * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const bool tile_ok = slot < a.tiles_x * a.tiles_y;
    const uint32_t tile = tile_ok ? (a.tile_order ? a.tile_order[slot] : slot) : 0u;
    const PixelCoord px = pixel_of(a, tile, threadIdx.x & 63u);
    const bool valid = tile_ok && px.valid;
    PixelState s;
    pixel_begin<total_in_lds(LDS)>(a, cam, s, ls, px.x, px.y, px.out_row);
    pixel_cache_begin<false, !LDS>(a, a, cam, s, ls);
    uint32_t starve = 0;
    bool active = valid && a.params.rays_per_pixel > 0;
    uint32_t n_segments = 0, n_reused_wave = 0, more_reused = 0;  // (more_reused: per lane, path_end's FAST_MISS)
    int node_tests = 0, tri_tests = 0;
    while (__ballot(active) != 0ull) {
        bool reused = false, reused_pre = false;
        if (active && path_step<LDS, STATS, TLAS, false, SIMPLE>(a, s, ls, starve, n_segments, reused, reused_pre, more_reused, node_tests, tri_tests) == PATH_PIXEL_DONE)
            active = false;
        n_reused_wave += (uint32_t)__popcll(__ballot(reused)) + (uint32_t)__popcll(__ballot(reused_pre));  // (wave-uniform: every lane is here)
    }
    if (valid) pixel_finish<total_in_lds(LDS)>(a, s, ls);
    if (a.tile_cost && tile_ok) {  // one store per wave: the tile's rays
        uint32_t sum = n_segments;
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
        if ((threadIdx.x & 63u) == 0u) a.tile_cost[tile] = sum;
    }
    flush_counters<STATS>(a, n_segments, n_reused_wave, node_tests, tri_tests, more_reused);
}

// Variant 0 (default): persistent waves with active-lane refill.  Each wave
// pulls 8x8 tiles from a global counter; whenever lanes have finished their
// pixel, a ballot + prefix count (mbcnt) hands them the next pixels of the
// wave's current tile, so the wave stays full until the frame runs out.  The
// per-pixel RNG stream depends only on the pixel's coordinates, so the image
// does not depend on which lane rendered which pixel.
template <bool LDS, bool STATS, bool TLAS, bool PARK, bool SIMPLE, bool HYB = false>
__global__ void __launch_bounds__(BLOCK_THREADS, RT_MIN_WAVES) rt_render_persistent_kernel(const RenderArgs a) {
#if defined(RT_DIAG) || defined(RT_WAVE_TIMES)
    const unsigned long long t_wave_start = __builtin_amdgcn_s_memrealtime();
#endif
#if defined(RT_DIAGT)
    if ((threadIdx.x & 63u) < 24u) g_tacc[threadIdx.x >> 6][threadIdx.x & 63u] = 0ull;
#endif
    uint32_t* ls = block_prologue<LDS>(a);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_tiles = a.tiles_x * a.tiles_y;
    const bool have_samples = a.params.rays_per_pixel > 0;

    uint32_t pool_base = 0, pool_left = 0;  // wave-uniform: pixels left in the current tile
    uint32_t pool_frame = 0;                // wave-uniform: the current tile's frame inside a frame batch
    uint32_t pool_txy = 0;                  // wave-uniform: the current tile's column | (local) row << 16, divided out once per pull
    // work items: (frame, tile) pairs, or -- a later launch of the sorting rounds -- blocks of 64 park records
    const bool resuming = PARK && a.q_in != nullptr;  // (PARK: the instantiations the deferred-walk sequences launch)
    const uint32_t n_parked = resuming ? *a.q_in_count : 0u;
    const uint32_t n_items = resuming ? (n_parked + 63u) >> 6 : n_tiles * (a.batch_frames ? a.batch_frames : 1u);
    // (a launch with fewer items than waves -- the later rounds of a deferred-walk sequence -- would otherwise consist
    // of thousands of pulls queueing up on one address: the waves beyond the items never pull)
    bool exhausted = PARK && blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6) >= n_items;
    PixelState s;
    pixel_begin<total_in_lds(LDS)>(a, camera_consts(a), s, ls, 0, 0, 0);
    bool active = false;
    uint32_t n_segments = 0, n_reused_wave = 0, more_reused = 0;  // (more_reused: per lane, path_end's FAST_MISS)
    int node_tests = 0, tri_tests = 0;
    // this wave's tile-cost table sits behind the workgroup's traversal stacks
    uint32_t* cost_tbl = cost_table_of_wave<LDS>(a);
    if (lane < COST_SLOTS * 3u) cost_tbl[lane] = 0u;
    uint32_t starve = 0;
    uint32_t pull_seq = 0;
    uint32_t resume_slot = 0xffffffffu;  // (PARK) the park record this lane's pixel was just resumed from

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        TIC(t14);
        if (idle != 0ull && !exhausted) {
            if (pool_left == 0) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(a.work_counter, 1u);
                t = __builtin_amdgcn_readfirstlane(t);
                pull_seq += 1;
                if (t >= n_items) {
                    exhausted = true;
                } else if (resuming) {
                    pool_base = t * 64u;
                    pool_left = n_parked - pool_base < 64u ? n_parked - pool_base : 64u;
                } else {
                    pool_frame = 0u;
                    if (a.batch_frames != 0u) {
                        if (a.batch_tile_major != 0u) {  // (tile, frame) order: a tile's frames back to back
                            const uint32_t tt = t / a.batch_frames;
                            pool_frame = t - tt * a.batch_frames;
                            t = tt;
                        } else {
                            pool_frame = t / n_tiles;
                            t -= pool_frame * n_tiles;
                        }
                    }
                    if (a.tile_order) t = a.tile_order[t];  // heaviest tiles first (an earlier frame's cost)
                    pool_base = t * 64u;
                    pool_left = 64u;
                    {
                        const uint32_t ty = t / a.tiles_x;
                        pool_txy = (t - ty * a.tiles_x) | (ty << 16);
                    }
                    // (a batch records the costs of its first frame)
                    if (a.tile_cost && lane == 0 && pool_frame == 0u) tile_cost_pull(a, cost_tbl, pull_seq & (COST_SLOTS - 1u), t);
                }
            }
            if (pool_left != 0) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (!active && rank < pool_left && resuming) {
                    if constexpr (PARK) {  // resume a parked pixel (path_step finishes its segment)
                        TIC(t21);
                        resume_slot = pool_base + rank;
                        park_load<total_in_lds(LDS)>(a, resume_slot, s, ls);
                        active = true;
                        TOC(t21, 21);
                    }
                } else if (!active && rank < pool_left) {
                    const uint32_t q = pool_base + rank;
                    ColdArgs& ca = cold_args();
                    const PixelCoord px = pixel_at(ca, pool_txy & 0xffffu, pool_txy >> 16, q & 63u);  // (q >> 6 is the pulled tile)
                    if (px.valid) {
                        DIAG(15);
                        const CameraConsts cam = camera_consts(ca);
                        pixel_begin<total_in_lds(LDS)>(ca, cam, s, ls, px.x, px.y, px.out_row, pool_frame);
                        pixel_cache_begin<false, !LDS>(a, ca, cam, s, ls);
                        s.meta = ((pull_seq & (COST_SLOTS - 1u)) << 16) | (pool_frame << 19);
                        if (have_samples) {
                            active = true;
                        } else {
                            pixel_finish<total_in_lds(LDS)>(ca, s, ls);  // 0 / 0 = NaN, as the shader would store
                            if (a.tile_cost && pool_frame == 0u) tile_cost_add(a, cost_tbl, s);
                        }
                    }
                }
                const uint32_t n_idle = (uint32_t)__popcll(idle);
                const uint32_t n = n_idle < pool_left ? n_idle : pool_left;
                pool_base += n;
                pool_left -= n;
            }
        }
        TOC(t14, 14);
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;
        }
        TIC(t15);
        uint32_t step = PATH_CONTINUE;
        bool reused = false, reused_pre = false;
        if (active) {
            step = path_step<LDS, STATS, TLAS, PARK, SIMPLE, HYB>(a, s, ls, starve, n_segments, reused, reused_pre, more_reused, node_tests, tri_tests, resume_slot);
            resume_slot = 0xffffffffu;
            if (step == PATH_PIXEL_DONE) {
                DIAG(16);
                pixel_finish<total_in_lds(LDS)>(cold_args(), s, ls);
                if (a.tile_cost && (s.meta >> 19) == 0u) tile_cost_add(a, cost_tbl, s);
                active = false;
            }
        }
        if (step == PATH_PARK) active = false;  // (parked by path_step)
        n_reused_wave += (uint32_t)__popcll(__ballot(reused)) + (uint32_t)__popcll(__ballot(reused_pre));  // (wave-uniform: every lane is here)
        TOC(t15, 15);
    }
    flush_counters<STATS>(a, n_segments, n_reused_wave, node_tests, tri_tests, more_reused);
#if defined(RT_DIAGT)
    if (lane < 24u) atomicAdd(&g_diag[40 + lane], g_tacc[threadIdx.x >> 6][lane]);
#endif
#if defined(RT_DIAG) || defined(RT_WAVE_TIMES)
    if (lane == 0) {
        const uint32_t w = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
        if (w < 8192) {
            g_wave_times[2 * w] = t_wave_start;
            g_wave_times[2 * w + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// Deferred walks (RenderArgs::park): the big mesh's walk for every parked ray.  A record's ray is taken to the
// mesh's local space with the operations of ITEM_NEW_XFORM and walked exactly as traverse_mesh walks it (far child
// pushed, near child next, entries popped without re-testing, a leaf's triangles in order); the closest triangle hit
// goes back into the record (plane 13).  A lane takes its next ray as soon as its walk ends, so the node-visit and
// triangle passes stay populated whatever the lengths of the individual walks.
// ---------------------------------------------------------------------------
template <bool STATS>
__global__ void __launch_bounds__(BLOCK_THREADS, 8) rt_walk_kernel(const RenderArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* stack = reinterpret_cast<uint32_t*>(lds_mem) + (threadIdx.x >> 6) * stack_dwords(a) + lane;
    const LaneStack st{stack, a.stack_wide != 0u};
    const uint32_t n = *a.q_in_count;
    const uint32_t xo = a.lay.mesh_off + a.defer_xform * MESH_REC_BYTES;
    const float4 c0 = ld4<false>(a, xo), c1 = ld4<false>(a, xo + 16), c2 = ld4<false>(a, xo + 32), c3 = ld4<false>(a, xo + 48);
    const float4 hdr = ld4<false>(a, a.lay.mesh_off + a.defer_mesh * MESH_REC_BYTES + 128);
    const uint32_t root = fbits(hdr.y);  // (an internal root: the host defers no other mesh)
    const bool cull = (fbits(hdr.x) & DMESH_GLASS) == 0u;
    const bool deep = (fbits(hdr.x) & DMESH_DEEP) != 0u;  // BVH of height >= 32: the shader's literal, clamped stack
    const uint32_t tri0 = a.lay.tri_off;
    uint32_t pool_base = 0, pool_left = 0;  // wave-uniform: records reserved by this wave
    bool exhausted = (blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 256u >= n;  // (waves beyond the work never pull)
    uint32_t slot = 0, cur = 0, cur_count = 0, sp = 0;
    bool have = false;
    f3 lo{0, 0, 0}, ld{0, 0, 0}, inv{0, 0, 0};
    MeshBest b;
    b.t = INF;
    b.tri = 0xffffffffu;
    b.u = b.v = 0.0f;
    int node_tests = 0, tri_tests = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        if (idle != 0ull && !exhausted) {
            if (pool_left == 0) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(a.work_counter, 256u);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= n) {
                    exhausted = true;
                } else {
                    pool_base = t;
                    pool_left = n - t < 256u ? n - t : 256u;
                }
            }
            if (pool_left != 0) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (!have && rank < pool_left) {
                    slot = pool_base + rank;
                    DIAG(43);
                    const float4* q = a.q_in + (size_t)(slot >> 6) * (PARK_PLANES * 64u) + (slot & 63u);
                    const float4 p2 = q[2 * 64], p3 = q[3 * 64];
                    lo = mat_cols_xyz(c0, c1, c2, c3, f3{p2.x, p2.y, p2.z}, 1.0f);
                    ld = normalize3<true>(mat_cols_xyz(c0, c1, c2, c3, f3{p2.w, p3.x, p3.y}, 0.0f));
                    inv = f3{rcp_(ld.x), rcp_(ld.y), rcp_(ld.z)};
                    cur = root;
                    cur_count = 0;
                    sp = 0;
                    b.t = INF;
                    b.tri = 0xffffffffu;
                    b.u = b.v = 0.0f;
                    have = true;
                }
                const uint32_t n_idle = (uint32_t)__popcll(idle);
                const uint32_t taken = n_idle < pool_left ? n_idle : pool_left;
                pool_base += taken;
                pool_left -= taken;
            }
        }
        if (__ballot(have) == 0ull) {
            if (exhausted) break;
            continue;
        }
        const bool walking = have;
        if (deep) {
            // traverse_mesh's literal walk (wgsl:297-333 with naga's index clamping), one stack entry per trip;
            // here sp is the shader's stack_index and the entry in hand is always the popped one
            auto slot_of = [](uint32_t i) { return i < RT_BVH_STACK ? i : RT_BVH_STACK - 1u; };
            if (have) {
                if (cur_count > 0) {
                    if (STATS) tri_tests += (int)cur_count;
                    for (uint32_t j = 0; j < cur_count; ++j) {
                        const uint32_t t = tri0 + (cur + j) * TRI_ISECT_BYTES;
                        tri_test<40>(lo, ld, ld4<false>(a, t), ld4<false>(a, t + 16), ld4<false>(a, t + 32), cull, cur + j, b);
                    }
                } else {
                    DIAG(42);
                    float4 q0, q1, q2, q3;
                    load_wide<false>(a, cur, q0, q1, q2, q3);
                    const float da = aabb_dist(lo, inv, q0, q1, b.t);
                    const float db = aabb_dist(lo, inv, q2, q3, b.t);
                    if (STATS) node_tests += 2;
                    const bool left_closer = da < db;
                    const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
                    if (far_d < b.t) {
                        stack_put(st, slot_of(sp), fbits(left_closer ? q3.z : q1.z), fbits(left_closer ? q3.w : q1.w));
                        sp += 1;
                    }
                    if (near_d < b.t) {
                        stack_put(st, slot_of(sp), fbits(left_closer ? q1.z : q3.z), fbits(left_closer ? q1.w : q3.w));
                        sp += 1;
                    }
                }
                if (sp == 0) {
                    have = false;
                } else {
                    sp -= 1;
                    stack_get(st, slot_of(sp), cur, cur_count);
                }
            }
        } else {
#if RT_WALK2
        // Experiment (rt_device.h RT_WALK2): one fetch serves TWO levels of the near-first descent.  The record of `cur`
        // holds its children's boxes (the old wide record) and a copy of each child's own wide record; the second step
        // below is the very step the next trip of the loop would make after fetching the near child's record -- the same
        // tests against the same b.t (nothing happens in between), the same pushes in the same order.
        while (have && cur_count == 0) {
            DIAG(42);
            const float4* r = a.walk2 + (size_t)(cur - a.walk2_base) * 12u;
            const float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
            const float4 ca0 = r[4], ca1 = r[5], ca2 = r[6], ca3 = r[7], cb0 = r[8], cb1 = r[9], cb2 = r[10], cb3 = r[11];
            const float da = aabb_dist(lo, inv, q0, q1, b.t);
            const float db = aabb_dist(lo, inv, q2, q3, b.t);
            if (STATS) node_tests += 2;
            const bool left_closer = da < db;
            const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
            const uint32_t near_i = fbits(left_closer ? q1.z : q3.z), near_c = fbits(left_closer ? q1.w : q3.w);
            const uint32_t far_i = fbits(left_closer ? q3.z : q1.z), far_c = fbits(left_closer ? q3.w : q1.w);
            if (far_d < b.t) {
                stack_put(st, sp, far_i, far_c);
                ++sp;
            }
            if (near_d < b.t) {
                cur = near_i;
                cur_count = near_c;
                if (cur_count == 0) {   // the near child is internal: its step, from the copy of its record
                    const float4 n0 = left_closer ? ca0 : cb0, n1 = left_closer ? ca1 : cb1, n2 = left_closer ? ca2 : cb2, n3 = left_closer ? ca3 : cb3;
                    const float ea = aabb_dist(lo, inv, n0, n1, b.t);
                    const float eb = aabb_dist(lo, inv, n2, n3, b.t);
                    if (STATS) node_tests += 2;
                    const bool l2 = ea < eb;
                    const float near2 = l2 ? ea : eb, far2 = l2 ? eb : ea;
                    const uint32_t near2_i = fbits(l2 ? n1.z : n3.z), near2_c = fbits(l2 ? n1.w : n3.w);
                    const uint32_t far2_i = fbits(l2 ? n3.z : n1.z), far2_c = fbits(l2 ? n3.w : n1.w);
                    if (far2 < b.t) {
                        stack_put(st, sp, far2_i, far2_c);
                        ++sp;
                    }
                    if (near2 < b.t) {
                        cur = near2_i;
                        cur_count = near2_c;
                    } else if (sp == 0) {
                        have = false;
                    } else {
                        --sp;
                        stack_get(st, sp, cur, cur_count);
                    }
                }
            } else if (sp == 0) {
                have = false;
            } else {
                --sp;
                stack_get(st, sp, cur, cur_count);
            }
        }
#else
        while (have && cur_count == 0) {  // descend to the next leaf (traverse_mesh's step)
            DIAG(42);
            float4 q0, q1, q2, q3;
            load_wide<false>(a, cur, q0, q1, q2, q3);
            const float da = aabb_dist(lo, inv, q0, q1, b.t);
            const float db = aabb_dist(lo, inv, q2, q3, b.t);
            if (STATS) node_tests += 2;
            const bool left_closer = da < db;
            const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
            const uint32_t near_i = fbits(left_closer ? q1.z : q3.z), near_c = fbits(left_closer ? q1.w : q3.w);
            const uint32_t far_i = fbits(left_closer ? q3.z : q1.z), far_c = fbits(left_closer ? q3.w : q1.w);
            if (far_d < b.t) {
                stack_put(st, sp, far_i, far_c);
                ++sp;
            }
            if (near_d < b.t) {
                cur = near_i;
                cur_count = near_c;
            } else if (sp == 0) {
                have = false;
            } else {
                --sp;
                stack_get(st, sp, cur, cur_count);
            }
        }
#endif
        if (have) {  // a leaf
            if (STATS) tri_tests += (int)cur_count;
            for (uint32_t j = 0; j < cur_count; ++j) {
                const uint32_t t = tri0 + (cur + j) * TRI_ISECT_BYTES;
                tri_test<40>(lo, ld, ld4<false>(a, t), ld4<false>(a, t + 16), ld4<false>(a, t + 32), cull, cur + j, b);
            }
            if (sp == 0) {
                have = false;
            } else {
                --sp;
                stack_get(st, sp, cur, cur_count);
            }
        }
        }
        if (walking && !have) {  // the walk ended: its result goes back into the record
            float4* q = a.q_in + (size_t)(slot >> 6) * (PARK_PLANES * 64u) + (slot & 63u);
            q[13 * 64] = make_float4(b.t, b.u, b.v, __uint_as_float(b.tri));
        }
    }
    if (STATS && a.counters && __ballot((node_tests | tri_tests) != 0) != 0ull) {
        atomicAdd(&a.counters->node_tests, (unsigned long long)node_tests);
        atomicAdd(&a.counters->triangle_tests, (unsigned long long)tri_tests);
    }
}

#if RT_EXPERIMENTS
#include "experiments/rt_wavefront.inl"  // rt_wf_shade_kernel, rt_wf_walk_kernel (option "wavefront")
#endif


// ---------------------------------------------------------------------------
// wgsl debug_trace (wgsl:502-573): one primary ray, no RNG.
// ---------------------------------------------------------------------------
template <bool LDS, bool TLAS>
__global__ void __launch_bounds__(BLOCK_THREADS) rt_debug_kernel(const RenderArgs a) {
    uint32_t* stack = stack_of<total_in_lds(LDS)>(block_prologue<LDS>(a));
    const uint32_t tile = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (tile >= a.tiles_x * a.tiles_y) return;
    const PixelCoord px = pixel_of(a, tile, threadIdx.x & 63u);
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
    Isect isect;
    Hit hit = intersect_scene<LDS, true, TLAS>(a, cam_origin, rd, stack, s0, s1, isect);
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
                const int flag = ldi<LDS>(a, hit.mat_off + M_FLAG);
                const int normal_index = ldi<LDS>(a, hit.mat_off + M_NORMAL_IDX);
                if (flag == RT_MATERIAL_TEXTURE && normal_index != -1) {
                    f4 x = sample_texture(a, normal_index, hit.u, hit.v);
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
    store_texel(a, px.x, px.out_row, 0u, out);
    if (a.counters) {
        atomicAdd(&a.counters->segments, 1ull);
        atomicAdd(&a.counters->node_tests, (unsigned long long)s0);
        atomicAdd(&a.counters->triangle_tests, (unsigned long long)s1);
    }
}

// Primary table: the memo of every pixel of the launch's tiles (the rank's strips) -- the pixel's constant primary ray
// and, when `with_hits`, that ray's hit: what path_begin / memo_hit_store would put into the lane's memo the first time
// a sample of the pixel is traced, computed once per (camera, frame size, scene) instead of once per pixel and frame.
// The ray comes from memo_ray_of (the operations of wgsl:479-495 with the zero-strength jitter folded, see path_begin),
// the hit from intersect_scene with the origin every memoised ray shares (RenderArgs::memo_ro).  One wave per 8x8 tile,
// launched with the render launch's arguments and LDS size.
template <bool LDS, bool TLAS>
__global__ void __launch_bounds__(BLOCK_THREADS) rt_primary_kernel(const RenderArgs a, uint4* __restrict__ table, uint32_t with_hits) {
    uint32_t* stack = stack_of<total_in_lds(LDS)>(block_prologue<LDS>(a));
    const uint32_t tile = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (tile >= a.tiles_x * a.tiles_y) return;
    const PixelCoord px = pixel_of(a, tile, threadIdx.x & 63u);
    if (!px.valid) return;
    const CameraConsts c = camera_consts(a);
    bool constant_ray;
    const f3 rd = memo_ray_of(a, c, px.x, px.y, constant_ray);
    uint32_t word = constant_ray ? MEMO_RAY : 0u;
    Hit hit;
    hit.hit = false;
    hit.backface = false;
    hit.dst = 0.0f;
    hit.point = hit.normal = f3{0, 0, 0};
    hit.u = hit.v = 0.0f;
    hit.mat_off = 0u;
    if (constant_ray && with_hits != 0u) {
        int s0 = 0, s1 = 0;
        Isect unused;
        hit = intersect_scene<LDS, false, TLAS>(a, f3{a.memo_ro[0], a.memo_ro[1], a.memo_ro[2]}, rd, stack, s0, s1, unused);
        word = (hit.hit ? (hit.mat_off & ~15u) | MEMO_HIT : 0u) | (hit.backface ? MEMO_BACKFACE : 0u) | MEMO_RAY | MEMO_HIT_VALID;
    }
    uint4* e = table + primary_index(a.params.width, px.x, px.y) * 4u;
    e[0] = make_uint4(__float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z), __float_as_uint(hit.dst));
    e[1] = make_uint4(__float_as_uint(hit.point.x), __float_as_uint(hit.point.y), __float_as_uint(hit.point.z), __float_as_uint(hit.normal.x));
    e[2] = make_uint4(__float_as_uint(hit.normal.y), __float_as_uint(hit.normal.z), __float_as_uint(hit.u), __float_as_uint(hit.v));
    e[3] = make_uint4(word, 0u, 0u, 0u);
}

// wgsl:154-161 for the frames of a batch, in frame order: the only dependency between frames is
// this per-texel blend, so rt_render_frames samples all frames of a batch in ONE launch (their
// samples land in scratch images) and this dense pass applies `prev * (1 - w) + cur * w` -- the same
// two IEEE operations per component, weights from the host as in a one-frame launch -- frame by frame.
__global__ void __launch_bounds__(256) rt_blend_frames_kernel(const BlendArgs b) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= b.texels) return;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool have = false;
    for (uint32_t k = 0; k < b.n; ++k) {
        const float4 cur = b.scratch[(unsigned long long)k * b.stride + i];
        const int32_t frames = (int32_t)((uint32_t)b.frames0 + k);
        if (frames >= 1) {
            if (!have) acc = b.image[i];  // the image an earlier launch left
            const float w = b.weight[k], om = b.rest[k];
            acc = make_float4(acc.x * om + cur.x * w, acc.y * om + cur.y * w, acc.z * om + cur.z * w, acc.w * om + cur.w * w);
        } else {
            acc = cur;
        }
        have = true;
    }
    if (have) b.image[i] = acc;
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

// Tile schedule for the next frame: counting sort of the tiles by the rays
// they took in the last frame, heaviest first, so that the frame ends on cheap
// uniform tiles instead of a few long paths (the image never depends on it).
// One workgroup; n_tiles is a few 10^4.
constexpr uint32_t ORDER_BINS = 2048;
__global__ void __launch_bounds__(1024) rt_tile_order_kernel(const uint32_t* __restrict__ cost, uint32_t n_tiles,
                                                               uint32_t max_cost, uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[ORDER_BINS];
    __shared__ uint32_t chunk_sum[1024 / 64];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < ORDER_BINS; i += 1024) hist[i] = 0;
    __syncthreads();
    const unsigned long long scale = max_cost ? max_cost : 1u;
    auto bin_of = [&](uint32_t c) {
        unsigned long long b = (unsigned long long)(c < max_cost ? c : max_cost) * (ORDER_BINS - 1) / scale;
        return (ORDER_BINS - 1) - (uint32_t)b;  // heavy tiles -> low bins -> first
    };
    for (uint32_t t = tid; t < n_tiles; t += 1024) atomicAdd(&hist[bin_of(cost[t])], 1u);
    __syncthreads();
    // exclusive prefix over the bins: each thread owns 2 consecutive bins
    uint32_t a0 = hist[2 * tid], a1 = hist[2 * tid + 1];
    uint32_t mine = a0 + a1, scan = mine;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        uint32_t v = __shfl_up(scan, d);
        if ((tid & 63u) >= d) scan += v;
    }
    if ((tid & 63u) == 63u) chunk_sum[tid >> 6] = scan;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) base += chunk_sum[w];
    const uint32_t excl = base + scan - mine;
    __syncthreads();
    hist[2 * tid] = excl;
    hist[2 * tid + 1] = excl + a0;
    __syncthreads();
    for (uint32_t t = tid; t < n_tiles; t += 1024) order[atomicAdd(&hist[bin_of(cost[t])], 1u)] = t;
}

hipError_t launch_tile_order(const uint32_t* cost, uint32_t n_tiles, uint32_t max_cost, uint32_t* order,
                             hipStream_t stream) {
    hipLaunchKernelGGL(rt_tile_order_kernel, dim3(1), dim3(1024), 0, stream, cost, n_tiles, max_cost, order);
    return hipGetLastError();
}

#if defined(RT_DIAG) || defined(RT_WAVE_TIMES)
hipError_t diag_wave_times(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), sizeof(unsigned long long) * 2 * 8192);
}
#endif
#if defined(RT_DIAG) || defined(RT_DIAGT)
hipError_t diag_read(unsigned long long* out, bool reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag), sizeof(unsigned long long) * 128);
    if (e != hipSuccess) return e;
    if (reset) {
        unsigned long long z[128] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof(z));
    }
    return e;
}
#endif

// Launchers (called from rt_api.hip)
size_t render_lds_bytes(const RenderArgs& a) {
    size_t stacks = ((size_t)(a.stack_entries ? a.stack_entries : 1u) * (a.stack_wide ? 128u : 64u) + (size_t)a.tlas_entries * 64u) *
                    sizeof(uint32_t) * WAVES_PER_BLOCK;
    size_t cost_tables = 8u * 3u * sizeof(uint32_t) * WAVES_PER_BLOCK;
    size_t lane_state = total_in_lds(a.lds_scene != 0u) ? (size_t)LANE_STATE_DWORDS * 64u * sizeof(uint32_t) * WAVES_PER_BLOCK : 0u;
    size_t cache = a.pixel_cache == 1u ? (size_t)PIXEL_MEMO_DWORDS * 64u * sizeof(uint32_t) * WAVES_PER_BLOCK : 0u;
    // (= 4 x wave_region_dwords + the cost tables, see the LDS map)
    size_t scene = a.lds_scene ? a.lay.bytes : 0u;
#if RT_EXPERIMENTS
    if (!a.lds_scene) scene = ((size_t)a.top_count + a.tlas_lds) * WIDE_REC_BYTES;
#endif
    return stacks + cost_tables + lane_state + cache + scene;
}

// Dynamic LDS above 64 KiB (deep-BVH stacks) has to be opted into per kernel.
template <typename K>
static void launch_k(K kernel, uint32_t blocks, size_t lds, hipStream_t stream, const RenderArgs& a) {
    if (lds > 64u * 1024u) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(BLOCK_THREADS), lds, stream, a);
}

// Which instantiations exist (the product library; every one of them is what some BASELINE config or test runs):
//   persistent <LDS | global> x <few-mesh: general, SIMPLE, counters | many-mesh: general, counters>   = 10
//   + the deferred-walk launches, global-memory few-mesh scenes only: general, SIMPLE, counters            =  3
//   one wave per tile (small launches) <LDS | global> x <few-mesh: general, SIMPLE | many-mesh: general> =  6
// Counter launches always take the persistent kernel; a scene that fits the LDS has no mesh worth deferring (the host
// never asks: rt_api.hip render_impl).  -DRT_EXPERIMENTS=1 adds the hybrid launches and the wavefront kernels.
template <bool LDS, bool TLAS>
static void launch_variant(const RenderArgs& a, uint32_t ntiles, size_t lds, hipStream_t stream) {
    const uint32_t tile_blocks = (ntiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    // the SIMPLE instantiations exist for the product kernels (no counters) of the few-mesh scenes
    const bool simple = !TLAS && a.simple != 0u && a.count_tests == 0u;
    if (a.params.debug_flag != 0) {
        launch_k(rt_debug_kernel<LDS, TLAS>, tile_blocks, lds, stream, a);
    } else if (a.kernel_variant == 1 && a.count_tests == 0u) {
        if (simple) launch_k(rt_render_tiles_kernel<LDS, false, TLAS, !TLAS>, tile_blocks, lds, stream, a);
        else launch_k(rt_render_tiles_kernel<LDS, false, TLAS, false>, tile_blocks, lds, stream, a);
    } else {
        uint32_t blocks = a.persistent_blocks < tile_blocks ? a.persistent_blocks : tile_blocks;
        if (blocks == 0) blocks = 1;
        // a launch of a deferred-walk sequence (PARK): few-mesh scenes read from global memory
        constexpr bool CAN_PARK = !TLAS && !LDS;
        const bool park = CAN_PARK && (a.park != 0u || a.q_in != nullptr);
#if RT_EXPERIMENTS
        if (LDS && !TLAS && (a.park != 0u || a.q_in != nullptr)) {  // (host: hybrid launches only -- the small blob is staged)
            if (simple) launch_k(rt_render_persistent_kernel<LDS, false, TLAS, !TLAS, !TLAS, LDS && !TLAS>, blocks, lds, stream, a);
            else launch_k(rt_render_persistent_kernel<LDS, false, TLAS, !TLAS, false, LDS && !TLAS>, blocks, lds, stream, a);
            return;
        }
#endif
        if (park) {
            if (a.count_tests) launch_k(rt_render_persistent_kernel<LDS, true, TLAS, CAN_PARK, false>, blocks, lds, stream, a);
            else if (simple) launch_k(rt_render_persistent_kernel<LDS, false, TLAS, CAN_PARK, CAN_PARK>, blocks, lds, stream, a);
            else launch_k(rt_render_persistent_kernel<LDS, false, TLAS, CAN_PARK, false>, blocks, lds, stream, a);
        } else {
            if (a.count_tests) launch_k(rt_render_persistent_kernel<LDS, true, TLAS, false, false>, blocks, lds, stream, a);
            else if (simple) launch_k(rt_render_persistent_kernel<LDS, false, TLAS, false, !TLAS>, blocks, lds, stream, a);
            else launch_k(rt_render_persistent_kernel<LDS, false, TLAS, false, false>, blocks, lds, stream, a);
        }
    }
}

hipError_t launch_render(const RenderArgs& a, hipStream_t stream) {
    uint32_t ntiles = a.tiles_x * a.tiles_y;
    if (ntiles == 0) return hipSuccess;
    size_t lds = render_lds_bytes(a);
    // the many-mesh code (top-level trees, root-box culling) is only compiled into the kernels
    // that need it (register budget)
    const bool tlas = a.many_mesh != 0;
    if (a.lds_scene) {
        if (tlas) launch_variant<true, true>(a, ntiles, lds, stream);
        else launch_variant<true, false>(a, ntiles, lds, stream);
    } else {
        if (tlas) launch_variant<false, true>(a, ntiles, lds, stream);
        else launch_variant<false, false>(a, ntiles, lds, stream);
    }
    return hipGetLastError();
}

// the deferred mesh's walk for the records of a.q_in (persistent grid: 8 workgroups per CU)
hipError_t launch_walk(const RenderArgs& a, uint32_t compute_units, hipStream_t stream) {
    const size_t lds = (size_t)(a.stack_entries ? a.stack_entries : 1u) * (a.stack_wide ? 128u : 64u) * sizeof(uint32_t) * WAVES_PER_BLOCK;
    const uint32_t blocks = compute_units * 8u;
    if (a.count_tests) {
        if (lds > 64u * 1024u) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rt_walk_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(rt_walk_kernel<true>, dim3(blocks), dim3(BLOCK_THREADS), lds, stream, a);
    } else {
        if (lds > 64u * 1024u) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rt_walk_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(rt_walk_kernel<false>, dim3(blocks), dim3(BLOCK_THREADS), lds, stream, a);
    }
    return hipGetLastError();
}

#if RT_EXPERIMENTS
#include "experiments/rt_wavefront_launch.inl"
#endif

#if !RT_EXPERIMENTS
// (the host never gets here in the product build: option "wavefront" cannot be switched on)
size_t wf_walk_lds_bytes(const RenderArgs&) { return 0; }
hipError_t launch_wf_shade(const RenderArgs&, uint32_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_wf_walk(const RenderArgs&, uint32_t, hipStream_t) { return hipErrorNotSupported; }
#endif

hipError_t launch_blend_frames(const BlendArgs& b, hipStream_t stream) {
    if (b.texels == 0 || b.n == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_blend_frames_kernel, dim3((uint32_t)((b.texels + 255u) / 256u)), dim3(256), 0, stream, b);
    return hipGetLastError();
}

#ifndef RT_TEST_ENTRIES
#define RT_TEST_ENTRIES 0
#endif
#if RT_TEST_ENTRIES   // (the test library only: ray_tracer_2_amd/librt2_mi355x_test.so, include/rt_test_abi.h)
// ---------------------------------------------------------------------------
// Test-only: the device's evaluation of the implementation-defined builtins (rt_transc.h), IEEE
// division / sqrt, the RNG and the texture filter, one element per thread, so that
// tests/test_gpu_device_units.py can compare them bit for bit with the host compile of the same
// headers (the oracle) -- including the edge values whole-image parity tests almost never reach
// (rand() == 0 -> log(0), rand() == 1, trig sign bits at exact zeros, subnormals, inf, NaN).
// fn: 0 log, 1 cos, 2 sin, 3 exp, 4 exp2, 5 log2, 6 pow(x, y), 7 acos, 8 atan2(x, y), 9 sqrt, 10 x / y
// (the oracle's numbering), 11 rand() of RNG state bits x -> float, 12 the generator's u32 output for
// state x, 13 trig_signbits(x), 14 rand_normal_dist() of state x, 15 f32(u32 x) * 2^-32 (rand()'s
// conversion for a raw generator output), 16 normalize(x, y, x*y).x (division by a sqrt), 17 rcp_(x), 18 sqrt_dev(x)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rt_units_kernel(int fn, const float* __restrict__ x, const float* __restrict__ y,
                                                       float* __restrict__ out, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = y[i];
    float r = 0.0f;
    switch (fn) {
        case 0: r = rtm::log_(a); break;
        case 1: r = rtm::cos_(a); break;
        case 2: r = rtm::sin_(a); break;
        case 3: r = rtm::exp_(a); break;
        case 4: r = rtm::exp2_(a); break;
        case 5: r = rtm::log2_(a); break;
        case 6: r = rtm::pow_(a, b); break;
        case 7: r = rtm::acos_(a); break;
        case 8: r = rtm::atan2_(a, b); break;
        case 9: r = rtm::sqrt_(a); break;
        case 10: r = a / b; break;
        case 11: { uint32_t s = __float_as_uint(a); r = rand_(s); break; }
        case 12: { uint32_t s = __float_as_uint(a); r = __uint_as_float(next_random_number(s)); break; }
        case 13: r = __uint_as_float(rtm::trig_signbits(a)); break;
        case 14: { uint32_t s = __float_as_uint(a); r = rand_normal_dist(s); break; }
        case 15: r = (float)__float_as_uint(a) * 0x1p-32f; break;
        case 16: r = normalize3(f3{a, b, a * b}).x; break;
        case 17: r = rcp_(a); break;
        case 18: r = sqrt_dev(a); break;
        default: break;
    }
    out[i] = r;
}

__global__ void __launch_bounds__(256) rt_units_texture_kernel(const uint8_t* rgba8, uint32_t width, uint32_t height,
                                                               const float* srgb_lut, const float* __restrict__ uv,
                                                               float* __restrict__ out, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    typedef const __attribute__((address_space(1))) uint32_t* GWords;
    typedef const __attribute__((address_space(1))) float* GFloats;
    float o[4];
    rtm::sample_bilinear_words((GWords)(const void*)rgba8, width, height, (GFloats)(const void*)srgb_lut, uv[2 * i], uv[2 * i + 1], o);
    out[4 * i] = o[0]; out[4 * i + 1] = o[1]; out[4 * i + 2] = o[2]; out[4 * i + 3] = o[3];
}

// Every float x the short forms serve: rcp_core(x) against the compiler's IEEE 1.0f / x (which = 0), sqrt_core(x)
// against its sqrt (which = 1), and the sky's shortcuts against their literal forms -- sky_gradient_t (2) and
// ground_to_sky_t (3) for every float in [-1.5, 1.5] plus NaN and the infinities, sun_term (4) for every float in
// [0, 1.5] (max(0, .) never hands it anything negative) -- bit for bit.  out[0] = floats checked, out[1] = mismatches,
// out[2] = a mismatching bit pattern.
__global__ void __launch_bounds__(256) rt_sweep_kernel(int which, unsigned long long* out) {
    unsigned long long checked = 0, bad = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * 256u + threadIdx.x; b < (1ull << 32); b += (unsigned long long)gridDim.x * 256u) {
        const float x = __uint_as_float((uint32_t)b);
        bool in_range;
        if (which == 0) in_range = rcp_in_range(x);
        else if (which == 1) in_range = sqrt_in_range(x);
        else if (which == 4) in_range = (x >= 0.0f && x <= 1.5f) || x != x;
        else in_range = rtm::abs_(x) <= 1.5f || x != x || rtm::abs_(x) == __uint_as_float(0x7f800000u);
        if (!in_range) continue;
        float q = which == 0 ? 1.0f / x : which == 1 ? rtm::sqrt_(x) : which == 2 ? sky_gradient_t_literal(x)
                                                     : which == 3 ? ground_to_sky_t_literal(x) : sun_literal(x);
        asm volatile("" : "+v"(q));
        const float f = which == 0 ? rcp_core(x) : which == 1 ? sqrt_core(x) : which == 2 ? sky_gradient_t(x)
                                                 : which == 3 ? ground_to_sky_t(x) : sun_term(x);
        checked += 1;
        const bool both_nan = f != f && q != q;  // (NaN sign and payload are outside the arithmetic contract)
        if (!both_nan && __float_as_uint(f) != __float_as_uint(q)) {
            bad += 1;
            out[2] = b;
        }
    }
    atomicAdd(&out[0], checked);
    if (bad) atomicAdd(&out[1], bad);
}
hipError_t launch_sweep(int which, unsigned long long* out, hipStream_t stream) {
    hipLaunchKernelGGL(rt_sweep_kernel, dim3(4096), dim3(256), 0, stream, which, out);
    return hipGetLastError();
}

hipError_t launch_units(int fn, const float* x, const float* y, float* out, unsigned long long n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_units_kernel, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, stream, fn, x, y, out, n);
    return hipGetLastError();
}

hipError_t launch_units_texture(const uint8_t* rgba8, uint32_t width, uint32_t height, const float* srgb_lut, const float* uv,
                                float* out, unsigned long long n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_units_texture_kernel, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, stream, rgba8, width, height,
                       srgb_lut, uv, out, n);
    return hipGetLastError();
}
#endif  // RT_TEST_ENTRIES

hipError_t launch_primary(const RenderArgs& a, void* table, bool with_hits, hipStream_t stream) {
    const uint32_t ntiles = a.tiles_x * a.tiles_y;
    if (ntiles == 0) return hipSuccess;
    const size_t lds = render_lds_bytes(a);
    const uint32_t blocks = (ntiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    auto go = [&](auto kernel) {
        if (lds > 64u * 1024u) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(BLOCK_THREADS), lds, stream, a, (uint4*)table, with_hits ? 1u : 0u);
    };
    const bool tlas = a.many_mesh != 0;
    if (a.lds_scene) {
        if (tlas) go(rt_primary_kernel<true, true>);
        else go(rt_primary_kernel<true, false>);
    } else {
        if (tlas) go(rt_primary_kernel<false, true>);
        else go(rt_primary_kernel<false, false>);
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
