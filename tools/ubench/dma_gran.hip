// Microbenchmark: LDS-DMA throughput per CU vs contiguous segment size per row (GEMM-like strided rows).
// SEG = bytes contiguous per row per pass (128: 8 lanes/row, 64: 4 lanes/row, 32: 2 lanes/row); rows kp apart.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SEG, int DEPTH, bool SECOND_HALF_LATER>
__global__ void __launch_bounds__(512) k(const char* __restrict__ src, int ksteps, int kp, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LPR = SEG / 16;                 // lanes per row
    const char* base = src + (size_t)(blockIdx.x % 32) * 256 * kp;  // 32 row panels of 256 rows
    for (int kt = 0; kt < ksteps; ++kt) {
        // per K-step fetch 256 rows x 128 B for each of 2 operands = 64 KB, in passes of 8 KB
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int p = d * 512 + tid;           // 16-B slot index within this batch
            const int row = (p / LPR) % 256;
            const int seg = (p / LPR) / 256;       // which SEG-wide column block of the 128-B row
            const int c = p % LPR;
            const char* g = base + (size_t)row * kp + (size_t)kt * 128 + seg * SEG + c * 16;
            char* dst = smem + ((d * 512 + wave * 64) * 16) % 65536;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (smem[tid] == 77 && smem[tid + 1] == 78) sink[0] = 1;
}
template <int SEG, int DEPTH> void run(const char* src, int kp, int* sink) {
    hipFuncSetAttribute((const void*)k<SEG, DEPTH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int ksteps = kp / 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<SEG, DEPTH, false><<<256 * 4, 512, 65536>>>(src, ksteps, kp, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SEG, DEPTH, false><<<256 * 4, 512, 65536>>>(src, ksteps, kp, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 1024.0 * ksteps * DEPTH * 512 * 16;
    std::printf("seg %3d B  depth %d: %7.3f ms %6.2f TB/s %6.1f GB/s/CU\n", SEG, DEPTH, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
}
int main() {
    const int kp = 8192;
    char* src;
    hipMalloc(&src, (size_t)32 * 256 * kp + (4 << 20));
    hipMemset(src, 1, (size_t)32 * 256 * kp + (4 << 20));
    int* sink;
    hipMalloc(&sink, 64);
    run<128, 8>(src, kp, sink);
    run<64, 8>(src, kp, sink);
    run<32, 8>(src, kp, sink);
    run<128, 4>(src, kp, sink);
    run<64, 4>(src, kp, sink);
    return 0;
}
