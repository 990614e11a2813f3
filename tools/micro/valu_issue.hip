// Microbenchmark: sustained VALU issue rate of one SIMD-32 on gfx950, to reconcile
// tools/micro/issue_cost.hip (one wave64 instruction per ~3 cycles per SIMD measured in round 1) with
// MI355X_MICROARCH.md (2 cycles: 32 lanes per cycle).  Differences from issue_cost.hip:
//   * workgroups of 256 threads, so the hardware puts exactly one wave of each workgroup on each of
//     the CU's 4 SIMDs (64-thread workgroups are not guaranteed an even spread); k workgroups per CU
//     = k waves per SIMD; the placement is verified by reading HW_ID in the kernel;
//   * 16 independent accumulators per wave (a 16-instruction dependency distance);
//   * cycles from s_memtime inside the kernel (per wave) and from the wall clock.
// Prints, per instruction and waves/SIMD, shader cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

#define I16(OP) \
    OP("0") OP("1") OP("2") OP("3") OP("4") OP("5") OP("6") OP("7") OP("8") OP("9") OP("10") OP("11") OP("12") OP("13") OP("14") OP("15")
#define OPS_IN , "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])

template <int KIND>
__global__ void __launch_bounds__(256, 8) k(float* out, int iters, float seed, unsigned long long* cyc, unsigned* hwid,
                                            unsigned long long* rt) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x + i;
    float b = seed * 0.5f + 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#define OP(n) "v_mul_f32 %" n ", %" n ", %16\n"
            asm volatile(I16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b));
#undef OP
        } else if (KIND == 1) {
#define OP(n) "v_fma_f32 %" n ", %" n ", %16, %16\n"
            asm volatile(I16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b));
#undef OP
        } else if (KIND == 2) {
#define OP(n) "v_add_f32 %" n ", %" n ", %16\n"
            asm volatile(I16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b));
#undef OP
        } else if (KIND == 3) {
#define OP(n) "v_min_f32 %" n ", %" n ", %16\n"
            asm volatile(I16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b));
#undef OP
        } else if (KIND == 4) {  // VOP3 encoding with two SGPR-free sources and a literal-free select
#define OP(n) "v_cndmask_b32 %" n ", %" n ", %16, vcc\n"
            asm volatile(I16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b) : "vcc");
#undef OP
        } else if (KIND == 5) {
#define OP(n) "v_mul_lo_u32 %" n ", %" n ", %16\n"
            asm volatile(I16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b));
#undef OP
        } else if (KIND == 6) {  // the path tracer's mix: sub, mul, min, max, cmp + cndmask (slab test shape)
            asm volatile(
                "v_sub_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_min_f32 %2, %2, %16\n v_max_f32 %3, %3, %16\n"
                "v_sub_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_min_f32 %6, %6, %16\n v_max_f32 %7, %7, %16\n"
                "v_cmp_lt_f32 vcc, %8, %16\n v_cndmask_b32 %9, %9, %16, vcc\n v_add_f32 %10, %10, %16\n v_mul_f32 %11, %11, %16\n"
                "v_xor_b32 %12, %12, %16\n v_lshrrev_b32 %13, 3, %13\n v_add_u32 %14, %14, %16\n v_and_b32 %15, %15, %16\n"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) {
        cyc[wave] = t1 - t0;
        rt[2 * wave] = r0;
        rt[2 * wave + 1] = __builtin_amdgcn_s_memrealtime();
        unsigned id;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hwid[wave] = (id & 0xff30u) | ((xcc & 0xfu) << 16);  // simd [5:4], cu [11:8], sh [12], se [15:13], xcc
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    float* out;
    unsigned long long* cyc;
    unsigned* hwid;
    unsigned long long* rt;
    hipMalloc(&rt, (size_t)cus * 8 * 4 * 16);
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipMalloc(&cyc, (size_t)cus * 8 * 4 * 8);
    hipMalloc(&hwid, (size_t)cus * 8 * 4 * 4);
    for (int wps : {1, 2, 4, 5, 8}) {
        const int blocks = cus * wps, waves = blocks * 4;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 200, 1.0f, cyc, hwid, rt);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, cyc, hwid, rt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hc(waves);
        std::vector<unsigned> hid(waves);
        hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hid.data(), hwid, waves * 4, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> hr(2 * waves);
        hipMemcpy(hr.data(), rt, waves * 16, hipMemcpyDeviceToHost);
        // concurrency: summed wave lifetimes over the span from the first start to the last end (100 MHz ticks),
        // per SIMD = waves that were resident together on average; shader clock = s_memtime ticks per real time
        unsigned long long first = ~0ull, last = 0;
        double life = 0, clock = 0;
        for (int w = 0; w < waves; ++w) {
            first = hr[2 * w] < first ? hr[2 * w] : first;
            last = hr[2 * w + 1] > last ? hr[2 * w + 1] : last;
            life += (double)(hr[2 * w + 1] - hr[2 * w]);
            clock += (double)hc[w] / ((double)(hr[2 * w + 1] - hr[2 * w]) * 10e-9) / 1e9;
        }
        const double together = life / (double)(last - first) / (cus * 4);
        clock /= waves;
        double avg = 0, mx = 0;
        std::map<unsigned, int> per_simd;  // (xcc | se | sh | cu | simd) -> resident waves of this launch
        for (int w = 0; w < waves; ++w) {
            avg += (double)hc[w];
            if ((double)hc[w] > mx) mx = (double)hc[w];
            per_simd[hid[w]]++;
        }
        avg /= waves;
        int lo = 1 << 30, hi = 0;
        for (auto& kv : per_simd) { lo = kv.second < lo ? kv.second : lo; hi = kv.second > hi ? kv.second : hi; }
        const double n = (double)iters * 16 * wps;  // wave-instructions per SIMD if every SIMD holds wps waves
        (void)mx;
        // SIMD rate from real time: instructions of one SIMD / (span x measured clock)
        const double span_s = (double)(last - first) * 10e-9;
        printf("%-14s waves/SIMD %d: %5.2f shader cycles per wave64 instruction per SIMD (span %.3f ms x %.2f GHz measured clock / %.0f "
               "instructions); one wave sees %5.2f cycles per instruction; %.2f waves per SIMD resident together; placement %zu SIMDs, %d..%d waves each\n",
               name, wps, span_s * clock * 1e9 / n, span_s * 1e3, clock, n, avg / ((double)iters * 16), together, per_simd.size(), lo, hi);
    }
    hipFree(out);
    hipFree(cyc);
    hipFree(hwid);
}

int main() {
    run<0>("v_mul_f32");
    run<1>("v_fma_f32");
    run<2>("v_add_f32");
    run<3>("v_min_f32");
    run<4>("v_cndmask_b32");
    run<5>("v_mul_lo_u32");
    run<6>("tracer mix");
    return 0;
}
