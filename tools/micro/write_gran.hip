// Microbenchmark: what does WRITE_SIZE (rocprofv3 --pmc) count for stores that do not fill their 32-byte sector / 128-byte
// line?  Each kernel writes `bytes` bytes per lane at a per-lane stride into a buffer far larger than the L2 + Infinity
// Cache (2 GiB), so nothing is merged after the fact.  Run as
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- ./write_gran
// and read the per-kernel counter values (KiB) beside the useful bytes each kernel prints.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int DW>
__global__ void __launch_bounds__(256) store_k(uint32_t* p, size_t stride_dw, uint32_t v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t* q = p + i * stride_dw;
#pragma unroll
    for (int k = 0; k < DW; ++k) q[k] = v + k;
}
// one dword per lane, lane-interleaved like the old global-memory memo: only `active` lanes of each wave store
__global__ void __launch_bounds__(256) store_masked(uint32_t* p, unsigned long long mask, uint32_t v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((mask >> (threadIdx.x & 63u)) & 1ull) p[i] = v;
}

int main() {
    const size_t bytes = 2ull << 30;
    uint32_t* p;
    if (hipMalloc(&p, bytes) != hipSuccess) return 1;
    hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    const int blocks = 4096;  // 1 Mi lanes
    const size_t lanes = (size_t)blocks * 256;
#define RUN(DW, STRIDE)                                                                                   \
    hipLaunchKernelGGL(store_k<DW>, dim3(blocks), dim3(256), 0, 0, p, (size_t)(STRIDE) / 4, 7u);          \
    hipDeviceSynchronize();                                                                               \
    printf("store_k<%d> stride %4d B: useful %.1f MB\n", DW, STRIDE, lanes * DW * 4 / 1e6);
    RUN(4, 16)    // dense 16-byte stores
    RUN(4, 32)    // 16 bytes in every 32-byte sector
    RUN(4, 64)
    RUN(4, 128)
    RUN(4, 256)
    RUN(1, 32)    // one dword per 32-byte sector
    RUN(1, 128)
    RUN(1, 256)
    const unsigned long long masks[] = {~0ull, 0x1ull, 0x0101010101010101ull, 0x00000000000000ffull, 0x5555555555555555ull};
    for (unsigned long long m : masks) {
        hipLaunchKernelGGL(store_masked, dim3(blocks), dim3(256), 0, 0, p, m, 9u);
        hipDeviceSynchronize();
        printf("store_masked mask %016llx: useful %.1f MB\n", m, lanes / 64 * __builtin_popcountll(m) * 4 / 1e6);
    }
    hipFree(p);
    return 0;
}
