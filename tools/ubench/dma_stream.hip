// Microbenchmark: GEMM-like LDS-DMA streaming from HBM (footprint >> L2/MALL): every workgroup walks its own
// 512-row x 8 KiB panel along k in 128-B columns.  ROT=0: all rows read the same column index each round
// (power-of-two row stride: partition camping?); ROT=1: row r reads column (round + r) % 64.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ROT, int KEEP>
__global__ void __launch_bounds__(512) k(const char* __restrict__ src, int tiles, int kp, int npanels, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int slot = 0;
    for (int t = 0; t < tiles; ++t) {
        const char* base = src + (size_t)((blockIdx.x * tiles + t) % npanels) * 512 * kp;
        for (int r = 0; r < 64; ++r) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int p = d * 512 + tid;
                const int row = p >> 3;
                const int col = ROT ? ((r + row) & 63) : r;
                const char* g = base + (size_t)row * kp + (size_t)col * 128 + (p & 7) * 16;
                char* dst = smem + ((slot * 512 + wave * 64) * 16) % 131072;
                slot = (slot + 1) & 15;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            if (KEEP == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[tid] == 77 && smem[tid + 1] == 78) sink[0] = 1;
}
template <int ROT, int KEEP> void run(const char* src, int kp, int npanels, int* sink) {
    hipFuncSetAttribute((const void*)k<ROT, KEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int tiles = 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<ROT, KEEP><<<256, 512, 131072>>>(src, tiles, kp, npanels, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<ROT, KEEP><<<256, 512, 131072>>>(src, tiles, kp, npanels, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * tiles * 64 * 65536;
    std::printf("rot %d keep %d: %7.3f ms %6.2f TB/s %6.1f GB/s/CU\n", ROT, KEEP, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
}
int main() {
    const int kp = 8192, npanels = 1024;  // 1024 x 4 MiB = 4 GiB
    char* src;
    hipMalloc(&src, (size_t)npanels * 512 * kp);
    hipMemset(src, 1, (size_t)npanels * 512 * kp);
    int* sink;
    hipMalloc(&sink, 64);
    run<0, 8>(src, kp, npanels, sink);
    run<1, 8>(src, kp, npanels, sink);
    run<0, 0>(src, kp, npanels, sink);
    run<1, 0>(src, kp, npanels, sink);
    return 0;
}
