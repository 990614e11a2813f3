// Microbenchmark: issue cost of single gfx950 instructions (inline asm, so the compiler cannot
// pack or fold them), as cycles per instruction per SIMD at 1 and 4 waves per SIMD, for
// (a) 8 independent chains per wave (throughput) and (b) one dependent chain (latency).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

template <int KIND, bool DEP>
__global__ void __launch_bounds__(64) k(float* out, int iters, float seed, unsigned long long* cyc) {
    const unsigned long long t_begin = __builtin_readcyclecounter();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f + 1.0f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7}, pb{b, b};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_mul_f32
            if (DEP) { REP8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b));) }
            else { asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        } else if (KIND == 1) {  // v_fma_f32
            if (DEP) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(b));) }
            else { asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        } else if (KIND == 2) {  // v_pk_mul_f32 (two multiplies per instruction)
            if (DEP) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(pb));) }
            else { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
        } else if (KIND == 3) {  // v_pk_fma_f32
            if (DEP) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb));) }
            else { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
        } else if (KIND == 4) {  // v_rcp_f32
            if (DEP) { REP8(asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));) }
            else { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        } else if (KIND == 5) {  // v_sqrt_f32
            if (DEP) { REP8(asm volatile("v_sqrt_f32 %0, %0" : "+v"(a0));) }
            else { asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        } else if (KIND == 6) {  // v_mul_lo_u32
            if (DEP) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(b));) }
            else { asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        } else if (KIND == 7) {  // v_min_f32
            if (DEP) { REP8(asm volatile("v_min_f32 %0, %0, %1" : "+v"(a0) : "v"(b));) }
            else { asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        } else if (KIND == 8) {  // v_div_scale_f32 (writes VCC)
            if (DEP) { REP8(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a0) : "v"(b) : "vcc");) }
            else { asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_scale_f32 %3, vcc, %3, %8, %3\n v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_scale_f32 %5, vcc, %5, %8, %5\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_scale_f32 %7, vcc, %7, %8, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); }
        } else if (KIND == 9) {  // ds_bpermute_b32
            if (DEP) { REP8(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(b));) }
            else { asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        }
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t_begin;
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
void run(const char* name, int per_iter_indep) {
    float* out;
    hipMalloc(&out, 1 << 24);
    unsigned long long* cyc;
    hipMalloc(&cyc, 8 * 8192);
    static unsigned long long hc[8192];
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4, iters = 50000;
    for (int dep = 0; dep < 2; ++dep)
        for (int wps : {4, 8}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int it) {
                if (dep) hipLaunchKernelGGL((k<KIND, true>), dim3(simds * wps), dim3(64), 0, 0, out, it, 1.0f, cyc);
                else hipLaunchKernelGGL((k<KIND, false>), dim3(simds * wps), dim3(64), 0, 0, out, it, 1.0f, cyc);
            };
            launch(2000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch(iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)iters * (dep ? 8 : per_iter_indep) * wps;
            hipMemcpy(hc, cyc, 8 * simds * wps, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int q = 0; q < simds * wps; ++q) avg += (double)hc[q];
            avg /= simds * wps;
            printf("%-16s %-11s waves/SIMD %d: %6.2f shader cycles per instruction per SIMD (in-kernel counter); %.2f ms => clock %.2f GHz\n", name,
                   dep ? "dependent" : "independent", wps, avg / n, ms, avg / (ms * 1e-3) / 1e9);
        }
    hipFree(out);
}

int main() {
    run<0>("v_mul_f32", 8);
    run<1>("v_fma_f32", 8);
    run<2>("v_pk_mul_f32", 32);
    run<3>("v_pk_fma_f32", 32);
    run<4>("v_rcp_f32", 8);
    run<5>("v_sqrt_f32", 8);
    run<6>("v_mul_lo_u32", 8);
    run<7>("v_min_f32", 8);
    run<8>("v_div_scale_f32", 8);
    run<9>("ds_bpermute_b32", 8);
    return 0;
}
