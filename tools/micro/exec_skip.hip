// Microbenchmark: does a VALU instruction cost less when whole 16-lane groups of the wave64
// are masked off?  Runs a dependent f32 multiply-add chain (non-packed) under different EXEC masks.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(64) k(float* out, int iters, float seed, unsigned long long mask) {
    const unsigned lane = threadIdx.x & 63u;
    float a0 = seed + lane, a1 = a0 * 1.5f;
    const float b = seed * 0.5f + 1.0f, c = seed + 0.25f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = __builtin_fminf(a0 * b, c);   // mul + min: not packable into v_pk_*
                a1 = __builtin_fmaxf(a1 * c, b);
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1;
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 24);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4, wps = 4, iters = 2000;
    struct { const char* name; unsigned long long mask; } cases[] = {
        {"64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lanes 0-7", 0xffull},
        {"lane 0", 1ull}, {"lanes 0,16,32,48", 0x0001000100010001ull}, {"lanes 0-15 + 32-47", 0x0000ffff0000ffffull},
        {"even lanes", 0x5555555555555555ull}, {"lanes 16-31", 0xffff0000ull}, {"lanes 48-63", 0xffff000000000000ull}};
    for (auto& cs : cases) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(simds * wps), dim3(64), 0, 0, out, 10, 1.0f, cs.mask);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(simds * wps), dim3(64), 0, 0, out, iters, 1.0f, cs.mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 16 * 4 * wps;  // 4 VALU per unrolled step
        printf("%-22s %.3f ms  %.2f cycles per VALU instruction per SIMD (at 2.4 GHz)\n", cs.name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    return 0;
}
