// Microbenchmark: VALU issue rate on gfx950 for the instruction mix this path
// tracer uses (non-fused f32 add/mul, min/max, compare+select, fma), at 1..8
// waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = seed * 0.5f + 1.0f, c = seed + 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { a0 = a0 * b; a1 = a1 * b; a2 = a2 * b; a3 = a3 * b; a4 = a4 * b; a5 = a5 * b; a6 = a6 * b; a7 = a7 * b; }
            if (KIND == 1) { a0 = a0 + b; a1 = a1 + b; a2 = a2 + b; a3 = a3 + b; a4 = a4 + b; a5 = a5 + b; a6 = a6 + b; a7 = a7 + b; }
            if (KIND == 2) { a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c); a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c); }
            if (KIND == 3) { a0 = __builtin_fminf(a0, b) + c; a1 = __builtin_fminf(a1, b) + c; a2 = __builtin_fminf(a2, b) + c; a3 = __builtin_fminf(a3, b) + c; a4 = __builtin_fminf(a4, b) + c; a5 = __builtin_fminf(a5, b) + c; a6 = __builtin_fminf(a6, b) + c; a7 = __builtin_fminf(a7, b) + c; }
            if (KIND == 4) { a0 = a0 / b; a1 = a1 / b; a2 = a2 / b; a3 = a3 / b; a4 = a4 / b; a5 = a5 / b; a6 = a6 / b; a7 = a7 / b; }
            if (KIND == 5) { a0 = __builtin_sqrtf(a0) + c; a1 = __builtin_sqrtf(a1) + c; a2 = __builtin_sqrtf(a2) + c; a3 = __builtin_sqrtf(a3) + c; a4 = __builtin_sqrtf(a4) + c; a5 = __builtin_sqrtf(a5) + c; a6 = __builtin_sqrtf(a6) + c; a7 = __builtin_sqrtf(a7) + c; }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, int instr_per_unit) {
    float* out;
    hipMalloc(&out, 1 << 24);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int simds = p.multiProcessorCount * 4;
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = simds * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double ops_per_simd = (double)iters * 64 * wps;  // source-level ops per SIMD
        double cyc = ms * 1e-3 * 2.4e9;
        printf("%-10s waves/SIMD %d: %.3f ms, %.2f cycles per source op per SIMD (at 2.4 GHz; ~%d instr/op)\n", name, wps, ms,
               cyc / ops_per_simd, instr_per_unit);
    }
    hipFree(out);
}

int main() {
    run<0>("mul", 1);
    run<1>("add", 1);
    run<2>("fma", 1);
    run<3>("min+add", 2);
    run<4>("div", 11);
    run<5>("sqrt+add", 12);
    return 0;
}
