// Microbenchmark + exhaustive check of cheaper correctly-rounded f32
// division / reciprocal / sqrt sequences against hipcc's own IEEE expansions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define DEV __device__ __forceinline__

DEV float rcp_core(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
DEV float div_core(float n, float d, float r1) {
    float q = n * r1;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r1, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r1, q);
}
DEV bool in_range(float x) {
    float ax = __builtin_fabsf(x);
    return ax >= 0x1p-60f && ax <= 0x1p60f;
}
// fast division: core + div_fixup, valid when no scaling would be needed
DEV float fdiv(float n, float d) {
    float q = __builtin_amdgcn_div_fixupf(div_core(n, d, rcp_core(d)), d, n);
    bool ok = in_range(d) && (in_range(n) || n == 0.0f);
    if (__ballot(!ok) != 0ull) {  // wave-uniform: the slow path is only issued when some lane needs it
        float slow = n / d;
        q = ok ? q : slow;
    }
    return q;
}
DEV float frcp(float d) {  // 1/d
    float q = div_core(1.0f, d, rcp_core(d));
    bool ok = in_range(d);
    if (__ballot(!ok) != 0ull) {
        float slow = 1.0f / d;
        q = ok ? q : slow;
    }
    return q;
}
// correctly rounded sqrt for x in [2^-60, 2^60]: hardware estimate, then pick
// among s-1ulp, s, s+1ulp by the sign of the exact residuals (same scheme as
// the compiler's expansion, without scaling / class handling)
DEV float fsqrt(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    float sd = __uint_as_float(__float_as_uint(s) - 1u);
    float su = __uint_as_float(__float_as_uint(s) + 1u);
    float rd = __builtin_fmaf(-sd, s, x);
    float ru = __builtin_fmaf(-su, s, x);
    float r = rd <= 0.0f ? sd : s;
    r = ru > 0.0f ? su : r;
    bool ok = x >= 0x1p-60f && x <= 0x1p60f;
    if (__ballot(!ok) != 0ull) {
        float slow = __builtin_sqrtf(x);
        r = ok ? r : slow;
    }
    return r;
}

template <int KIND>
__global__ void __launch_bounds__(64) bench(float* out, int iters, float seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.37f + i;
    const float b = seed * 0.5f + 1.37f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) a[u] = a[u] / b + 1.0f;
            if (KIND == 1) a[u] = fdiv(a[u], b) + 1.0f;
            if (KIND == 2) a[u] = 1.0f / a[u] + 1.5f;
            if (KIND == 3) a[u] = frcp(a[u]) + 1.5f;
            if (KIND == 4) a[u] = __builtin_sqrtf(a[u]) + 1.5f;
            if (KIND == 5) a[u] = fsqrt(a[u]) + 1.5f;
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// exhaustive: all 2^32 bit patterns of x for rcp and sqrt; division on a
// pseudo-random companion operand.
__global__ void check(unsigned long long* bad) {
    unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
    unsigned nbad_r = 0, nbad_s = 0, nbad_d = 0;
    for (unsigned k = 0; k < 256; ++k) {
        uint32_t u = (uint32_t)(i0 + k);
        float x = __uint_as_float(u);
        float r0 = 1.0f / x, r1 = frcp(x);
        float s0 = __builtin_sqrtf(x), s1 = fsqrt(x);
        uint32_t h = u * 2654435761u + 12345u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        float y = __uint_as_float(h);
        float d0 = x / y, d1 = fdiv(x, y);
        float y2 = __uint_as_float((h & 0x007fffffu) | 0x3f000000u);  // companion near 1: exercises in-range path
        float e0 = x / y2, e1 = fdiv(x, y2);
        auto same = [](float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); };
        nbad_r += !same(r0, r1);
        nbad_s += !same(s0, s1);
        nbad_d += !same(d0, d1) + !same(e0, e1);
    }
    if (nbad_r) atomicAdd(&bad[0], nbad_r);
    if (nbad_s) atomicAdd(&bad[1], nbad_s);
    if (nbad_d) atomicAdd(&bad[2], nbad_d);
}

template <int KIND>
void run(const char* name) {
    float* out;
    (void)hipMalloc(&out, 1 << 24);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int simds = p.multiProcessorCount * 4, wps = 4, iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(simds * wps), dim3(64), 0, 0, out, 10, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(simds * wps), dim3(64), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s %.2f cycles per op (+1 add) per SIMD at 4 waves/SIMD\n", name, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps));
    (void)hipFree(out);
}

int main() {
    run<0>("x/y");
    run<1>("fdiv(x,y)");
    run<2>("1/x");
    run<3>("frcp(x)");
    run<4>("sqrtf");
    run<5>("fsqrt");
    unsigned long long* bad;
    (void)hipMalloc(&bad, 24);
    (void)hipMemset(bad, 0, 24);
    hipLaunchKernelGGL(check, dim3(65536), dim3(256), 0, 0, bad);
    unsigned long long h[3];
    (void)hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost);
    printf("exhaustive over 2^32 x: rcp mismatches %llu, sqrt mismatches %llu, div mismatches %llu (of 2 x 2^32 pairs)\n", h[0], h[1], h[2]);
    return 0;
}
