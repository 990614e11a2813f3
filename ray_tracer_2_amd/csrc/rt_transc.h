// rt_transc.h -- the canonical f32 transcendental functions of this build.
//
// WGSL leaves the precision of log, cos, sin, exp, exp2, log2, pow, acos and
// atan2 to the driver (naga 25 / Vulkan; SURVEY.md 8c "parity unpinned"), so
// a path tracer cannot be compared across math libraries: a 1-ULP change in a
// bounce direction occasionally flips a hit or a Russian-roulette decision.
// This header therefore DEFINES those functions once, in IEEE binary32
// arithmetic only (+, -, *, /, sqrt, fma, integer bit operations -- all of
// them correctly rounded on both x86-64 and gfx950), and both the HIP kernel
// (rt_kernel.hip) and the CPU oracle (oracle/shader_oracle.cpp) evaluate
// exactly these bodies.  Results are bit-identical on host and device when
// compiled with -ffp-contract=off (every fused operation below is an explicit
// fma).  tests/test_transc.py bounds their error against double-precision
// libm (<= 2 ULP on the ranges the shader uses).
//
// Algorithms: argument reduction + polynomial kernels in the style of the
// classic float libm routines (log2: msun e_logf's kernel; log, sin/cos: own
// near-minimax polynomials (tools/micro/fit_transc.py) behind msun's argument
// reductions -- a 3-term Cody-Waite reduction for the angles; exp2/exp: own
// minimax polynomials; atan/asin: Cephes atanf/asinf).  msun notice: "Developed at SunPro, a Sun
// Microsystems, Inc. business. Permission to use, copy, modify, and
// distribute this software is freely granted, provided that this notice is
// preserved."
#ifndef RT_TRANSC_H
#define RT_TRANSC_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtm {

RT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
RT_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RT_HD float sqrt_(float x) { return __builtin_sqrtf(x); }  // IEEE correctly rounded
RT_HD float rint_(float x) { return __builtin_rintf(x); }  // round to nearest even
RT_HD float abs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }

// 2^k * p for finite p, any int k (two-step so that the intermediate factors
// stay normal; a subnormal result is rounded once more, deterministically).
RT_HD float scale2(float p, int k) {
    if (k > 127) {
        p = p * 0x1p127f;
        k -= 127;
        if (k > 127) {
            p = p * 0x1p127f;
            k -= 127;
            if (k > 127) k = 127;
        }
    } else if (k < -126) {
        p = p * 0x1p-102f;  // 2^-126 * 2^24
        k += 102;
        if (k < -126) {
            p = p * 0x1p-102f;
            k += 102;
            if (k < -126) k = -126;
        }
    }
    return p * u2f((uint32_t)(k + 127) << 23);
}

// ---------------------------------------------------------------- log ----
// Returns k and f with x = 2^k * (1 + f), sqrt(1/2) <= 1 + f < sqrt(2),
// for positive finite x.
struct LogRed {
    float f;
    int k;
};

RT_HD LogRed log_reduce(float x) {
    int32_t ix = (int32_t)f2u(x);
    int k = 0;
    if (ix < 0x00800000) {  // subnormal: scale up
        x = x * 0x1p25f;
        ix = (int32_t)f2u(x);
        k = -25;
    }
    k += (ix >> 23) - 127;
    ix &= 0x007fffff;
    int32_t i = (ix + (0x95f64 << 3)) & 0x800000;
    float m = u2f((uint32_t)(ix | (i ^ 0x3f800000)));  // m or m/2
    k += (i >> 23);
    LogRed r;
    r.f = m - 1.0f;
    r.k = k;
    return r;
}

// log(x) = k ln2 + log(1 + f), log(1 + f) = f - f^2/2 + f^3 P(f) with P of degree 7 (near-minimax for the relative
// error on [sqrt(1/2) - 1, sqrt(2) - 1], fit error 0.1 ULP: tools/micro/fit_transc.py), Horner with explicit fma.
// Round 3: replaces the msun form above (a correctly rounded division plus 11 unfused operations) -- Box-Muller
// (wgsl:181-185) takes a logarithm per normal deviate, three per bounce.  <= 1 ULP on rand()'s outputs
// (tests/test_transc.py).
RT_HD float log_(float x) {
    uint32_t ux = f2u(x);
    if ((ux & 0x7fffffffu) == 0) return u2f(0xff800000u);  // log(+-0) = -inf
    if (ux >= 0x7f800000u) {
        if (ux == 0x7f800000u) return x;    // +inf
        if (ux > 0xff800000u || (ux > 0x7f800000u && ux < 0x80000000u))
            return x + x;                   // NaN
        return u2f(0x7fc00000u);            // negative -> NaN
    }
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;  // (ln2_hi has 9 trailing zero bits: k * ln2_hi is exact)
    const float P0 = 0.3333333134651184f, P1 = -0.250008225440979f, P2 = 0.20001231133937836f,
                P3 = -0.16623249650001526f, P4 = 0.14201568067073822f, P5 = -0.13161104917526245f,
                P6 = 0.12763474881649017f, P7 = -0.07634374499320984f;
    LogRed r = log_reduce(x);
    const float f = r.f, z = f * f, dk = (float)r.k;
    float p = fma_(P7, f, P6);
    p = fma_(p, f, P5);
    p = fma_(p, f, P4);
    p = fma_(p, f, P3);
    p = fma_(p, f, P2);
    p = fma_(p, f, P1);
    p = fma_(p, f, P0);
    float y = (f * z) * p;
    y = fma_(dk, ln2_lo, y);
    y = fma_(-0.5f, z, y);
    return fma_(dk, ln2_hi, y + f);
}

// log2(x) = k + log(1 + f) / ln2 with log_'s polynomial: log(1 + f) = f + t, t = f^3 P(f) - f^2/2; the product
// f / ln2 is formed exactly (a + e with one fma) and 1/ln2 carried as hi + lo, so the sum keeps ~30 bits before the
// final rounding -- pow(x, 500) in the sky's sun term multiplies this error by 500.  Round 3: replaces the msun
// split-precision form (a correctly rounded division and 25 unfused operations) within the same bound: <= 1 ULP
// (tests/test_transc.py).
RT_HD float log2_(float x) {
    uint32_t ux = f2u(x);
    if ((ux & 0x7fffffffu) == 0) return u2f(0xff800000u);
    if (ux >= 0x7f800000u) {
        if (ux == 0x7f800000u) return x;
        if (ux > 0xff800000u || (ux > 0x7f800000u && ux < 0x80000000u)) return x + x;
        return u2f(0x7fc00000u);
    }
    const float ivln2 = 1.4426950216293335f, ivln2_lo = 1.925963033500011e-08f;
    const float P0 = 0.3333333134651184f, P1 = -0.250008225440979f, P2 = 0.20001231133937836f,
                P3 = -0.16623249650001526f, P4 = 0.14201568067073822f, P5 = -0.13161104917526245f,
                P6 = 0.12763474881649017f, P7 = -0.07634374499320984f;
    LogRed r = log_reduce(x);
    const float f = r.f, z = f * f, dk = (float)r.k;
    float p = fma_(P7, f, P6);
    p = fma_(p, f, P5);
    p = fma_(p, f, P4);
    p = fma_(p, f, P3);
    p = fma_(p, f, P2);
    p = fma_(p, f, P1);
    p = fma_(p, f, P0);
    float t = (f * z) * p;
    t = fma_(-0.5f, z, t);              // log(1 + f) - f
    const float a = f * ivln2;
    const float e = fma_(f, ivln2, -a);  // f * ivln2 == a + e exactly
    const float lo = fma_(t, ivln2, fma_(f, ivln2_lo, e));
    const float hi = dk + a;
    const float he = a - (hi - dk);      // (exact: |dk| >= 1 > |a| or dk == 0)
    return hi + (lo + he);
}

// --------------------------------------------------------------- exp2 ----
// 2^r for |r| <= 0.5: degree-7 polynomial in r (Taylor/minimax of exp(r ln2)).
RT_HD float exp2_kernel(float r) {
    const float c1 = 6.9314718056e-01f, c2 = 2.4022650696e-01f, c3 = 5.5504108665e-02f,
                c4 = 9.6181291076e-03f, c5 = 1.3333558146e-03f, c6 = 1.5403530393e-04f,
                c7 = 1.5252733804e-05f;
    float p = fma_(c7, r, c6);
    p = fma_(p, r, c5);
    p = fma_(p, r, c4);
    p = fma_(p, r, c3);
    p = fma_(p, r, c2);
    p = fma_(p, r, c1);
    return fma_(p, r, 1.0f);
}

RT_HD float exp2_(float x) {
    uint32_t ax = f2u(x) & 0x7fffffffu;
    if (ax > 0x7f800000u) return x + x;                  // NaN
    if (x >= 128.0f) return u2f(0x7f800000u);            // overflow (incl. +inf)
    if (x < -150.0f) return 0.0f;                        // underflow (incl. -inf)
    float kf = rint_(x);
    float r = x - kf;  // exact
    return scale2(exp2_kernel(r), (int)kf);
}

RT_HD float exp_(float x) {
    uint32_t ax = f2u(x) & 0x7fffffffu;
    if (ax > 0x7f800000u) return x + x;
    if (x > 88.72284f) return u2f(0x7f800000u);
    if (x < -104.0f) return 0.0f;
    // x = k ln2 + t, |t| <= ln2/2; e^t = 2^(t/ln2) evaluated as exp2_kernel
    const float invln2 = 1.4426950216e+00f, ln2HI = 6.9314575195e-01f, ln2LO = 1.4286067653e-06f;
    float kf = rint_(x * invln2);
    float t = fma_(-kf, ln2HI, x);
    t = fma_(-kf, ln2LO, t);
    // e^t via its own polynomial (degree 6)
    const float d2 = 0.5f, d3 = 1.6666667163e-01f, d4 = 4.1666667908e-02f, d5 = 8.3333337680e-03f,
                d6 = 1.3888889225e-03f, d7 = 1.9841270114e-04f;
    float p = fma_(d7, t, d6);
    p = fma_(p, t, d5);
    p = fma_(p, t, d4);
    p = fma_(p, t, d3);
    p = fma_(p, t, d2);
    p = fma_(p, t * t, t);
    p = p + 1.0f;
    return scale2(p, (int)kf);
}

// WGSL: pow(x, y) = exp2(y * log2(x))  
RT_HD float pow_(float x, float y) { return exp2_(y * log2_(x)); }

// ------------------------------------------------------------ sin/cos ----
// x = n*(pi/2) + r, |r| <= ~pi/4.  Three-term Cody-Waite with fma; accurate
// for |x| < ~1e5 (the shader only passes angles in [0, 2*pi]).  Callers keep |x| < 2^31 so that the
// float -> int conversion of n is defined (it saturates on gfx950 and yields INT_MIN on x86: found by
// tests/test_gpu_device_units.py); beyond that, and for inf / NaN, sin_ and cos_ return the canonical
// quiet NaN 0x7fc00000 on both sides (an invalid operation's NaN would be negative on x86, positive on gfx950).
struct TrigRed {
    float r;
    int q;
};

RT_HD TrigRed trig_reduce(float x) {
    const float invpio2 = 6.3661974669e-01f;
    const float P1 = 1.5707963705e+00f;   // fl(pi/2)
    const float P2 = -4.3711388287e-08f;  // fl(pi/2 - P1)
    const float P3 = -1.7151245100e-15f;  // fl(pi/2 - P1 - P2)
    float n = rint_(x * invpio2);
    float r = fma_(-n, P1, x);
    r = fma_(-n, P2, r);
    r = fma_(-n, P3, r);
    TrigRed t;
    t.r = r;
    t.q = (int)n & 3;
    return t;
}

// sin(r) or cos(r) on the reduced interval as ONE polynomial with per-lane coefficients (round 3; the msun pair
// above evaluated both kernels and selected, 55 instructions on gfx950 -- Box-Muller takes a cosine per deviate):
//   sin(r) = r + (r z)(S1 + S2 z + S3 z^2 + S4 z^3),   cos(r) = 1 + z (C0 + C1 z + C2 z^2 + C3 z^3),   z = r^2
// -- the same shape b + a * p(z) with (a, b) = (r z, r) or (z, 1).  Near-minimax coefficients (fit error < 0.001 ULP:
// tools/micro/fit_transc.py), Horner with explicit fma.  Properties trig_signbits relies on: the cosine form is
// positive on the reduced interval; the sine form has the sign of r, and is +0 for r = +-0.
RT_HD float trig_poly(float r, bool sine) {
    const float S1 = -0.1666666716337204f, S2 = 0.008333329111337662f, S3 = -0.00019839330343529582f,
                S4 = 2.7182745725440327e-06f;
    const float C0 = -0.5f, C1 = 0.04166662320494652f, C2 = -0.0013886759988963604f, C3 = 2.4390043108724058e-05f;
    const float z = r * r;
    float p = fma_(z, sine ? S4 : C3, sine ? S3 : C2);
    p = fma_(z, p, sine ? S2 : C1);
    p = fma_(z, p, sine ? S1 : C0);
    return fma_(sine ? r * z : z, p, sine ? r : 1.0f);
}

RT_HD float sin_(float x) {
    uint32_t ax = f2u(x) & 0x7fffffffu;
    if (ax >= 0x4f000000u) return u2f(0x7fc00000u);  // |x| >= 2^31, inf, NaN -> NaN
    // (no shortcut for |x| < pi/4: there n = +-0, r = fma(-+0, P1, x) = x and q = 0, so the general path evaluates
    // the sine form at x itself, and a wave whose lanes straddle pi/4 does not execute two paths)
    TrigRed t = trig_reduce(x);
    float v = trig_poly(t.r, (t.q & 1) == 0);
    return (t.q & 2) ? -v : v;
}

RT_HD float cos_(float x) {
    uint32_t ax = f2u(x) & 0x7fffffffu;
    if (ax >= 0x4f000000u) return u2f(0x7fc00000u);
    TrigRed t = trig_reduce(x);  // (|x| < pi/4: n = 0, r = x, the result is the cosine form at x; see sin_)
    float v = trig_poly(t.r, (t.q & 1) != 0);
    return ((t.q + 1) & 2) ? -v : v;
}

// Sign bits of cos_(x) and sin_(x) for finite x, without evaluating the
// polynomials: trig_poly's cosine form is positive on the reduced interval and its sine form
// has the sign of r (and is +0 for r = +-0), so the signs follow from the
// quadrant and from r alone.  Used where only the sign of a product with an
// exact zero survives (rt_kernel.hip, zero-strength camera jitter).
RT_HD uint32_t trig_signbits(float x) {  // bit 0: cos_(x) negative, bit 1: sin_(x) negative
    uint32_t ax = f2u(x) & 0x7fffffffu;
    if (ax < 0x3f490fdau) return x < 0.0f ? 2u : 0u;
    if (ax >= 0x4f000000u) return 0u;  // sin_ / cos_ return the positive canonical NaN there
    TrigRed t = trig_reduce(x);
    const uint32_t s_neg = t.r < 0.0f ? 1u : 0u;
    const uint32_t q = (uint32_t)t.q;
    const uint32_t sin_neg = ((q & 1u) ? 0u : s_neg) ^ ((q >> 1) & 1u);
    const uint32_t cos_neg = ((q & 1u) ? s_neg : 0u) ^ (((q + 1u) >> 1) & 1u);
    return cos_neg | (sin_neg << 1);
}

// ---------------------------------------------------------- atan/acos ----
RT_HD float atan_pos(float x) {  // x >= 0 finite or +inf
    float y;
    if (x > 2.414213562373095f) {
        y = 1.5707963705e+00f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = 7.8539818525e-01f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z -
               3.33329491539e-1f) * z * x + x;
    return y + p;
}

RT_HD float atan_(float x) {
    uint32_t ux = f2u(x);
    if ((ux & 0x7fffffffu) > 0x7f800000u) return x + x;
    float r = atan_pos(abs_(x));
    return (ux >> 31) ? -r : r;
}

RT_HD float atan2_(float y, float x) {
    const float PI = 3.14159274101f, PIO2 = 1.57079637051f;
    uint32_t ux = f2u(x), uy = f2u(y);
    uint32_t ax = ux & 0x7fffffffu, ay = uy & 0x7fffffffu;
    if (ax > 0x7f800000u || ay > 0x7f800000u) return x + y;  // NaN
    bool xneg = (ux >> 31) != 0, yneg = (uy >> 31) != 0;
    if (ay == 0) {  // y = +-0
        float r = xneg ? PI : 0.0f;
        return yneg ? -r : r;
    }
    if (ax == 0) return yneg ? -PIO2 : PIO2;
    if (ax == 0x7f800000u) {
        float r;
        if (ay == 0x7f800000u) r = xneg ? 3.0f * 7.8539818525e-01f : 7.8539818525e-01f;
        else r = xneg ? PI : 0.0f;
        return yneg ? -r : r;
    }
    if (ay == 0x7f800000u) return yneg ? -PIO2 : PIO2;
    float a = atan_pos(abs_(y) / abs_(x));  // in [0, pi/2]
    float r = xneg ? (PI - a) : a;
    return yneg ? -r : r;
}

RT_HD float asin_core(float a) {  // 0 <= a <= 0.5 : asin(a)
    float z = a * a;
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z +
                7.4953002686e-2f) * z + 1.6666752422e-1f) * z * a + a;
    return p;
}

RT_HD float acos_(float x) {
    const float PI = 3.14159274101f, PIO2 = 1.57079637051f;
    uint32_t ax = f2u(x) & 0x7fffffffu;
    if (ax > 0x7f800000u) return x + x;
    if (ax > 0x3f800000u) return u2f(0x7fc00000u);  // |x| > 1
    if (x < -0.5f) return PI - 2.0f * asin_core(sqrt_(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_core(sqrt_(0.5f * (1.0f - x)));
    float a = abs_(x);
    float s = asin_core(a);
    return (f2u(x) >> 31) ? (PIO2 + s) : (PIO2 - s);
}

}  // namespace rtm

#endif  // RT_TRANSC_H
