// Microbenchmark: LDS-DMA ring streaming, 1 workgroup per CU (LDS-limited), GEMM-like strided rows.
//   PER  = DMA passes issued per wave between barriers
//   KEEP = passes allowed to stay in flight (counted vmcnt) -- 0 means drain every round
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}
template <int PER, int KEEP, int NT, int SEG, int SWZ>
__global__ void __launch_bounds__(NT) k(const char* __restrict__ src, int rounds, int kp, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LPR = SEG / 16;
    const char* base = src + (size_t)(blockIdx.x % 32) * 256 * kp;
    int slot = 0;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int d = 0; d < PER; ++d) {
            const int p = d * NT + tid;
            const int row = (p / LPR) % 256;
            int cc = p % LPR;
            if (SWZ == 1) cc ^= (row >> 1) & (LPR - 1);
            if (SWZ == 2) cc = (cc + (row >> 1)) & (LPR - 1);
            const int col = (SWZ >= 10) ? ((SWZ == 11 ? (r + row) : r) % 56) : ((r * PER + d) % 56);
            const char* g = base + (size_t)row * kp + (size_t)col * 128 + cc * 16;
            char* dst = smem + ((slot * NT + wave * 64) * 16) % 131072;
            slot = (slot + 1) & 15;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        wait_vm<KEEP>();
        __builtin_amdgcn_s_barrier();
    }
    wait_vm<0>();
    if (smem[tid] == 77 && smem[tid + 1] == 78) sink[0] = 1;
}
template <int PER, int KEEP, int NT, int SEG, int SWZ = 0> void run(const char* src, int kp, int* sink, int lds) {
    hipFuncSetAttribute((const void*)k<PER, KEEP, NT, SEG, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int rounds = 8192 / PER;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<PER, KEEP, NT, SEG, SWZ><<<256, NT, lds>>>(src, rounds, kp, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<PER, KEEP, NT, SEG, SWZ><<<256, NT, lds>>>(src, rounds, kp, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * rounds * PER * NT * 16;
    std::printf("swz %d per %d keep %2d threads %4d seg %3d lds %3dK: %7.3f ms %6.2f TB/s %6.1f GB/s/CU\n", SWZ, PER, KEEP, NT, SEG, lds / 1024, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
}
int main() {
    const int kp = 8192;
    char* src;
    hipMalloc(&src, (size_t)32 * 256 * kp + (4 << 20));
    hipMemset(src, 1, (size_t)32 * 256 * kp + (4 << 20));
    int* sink;
    hipMalloc(&sink, 64);
    run<8, 8, 512, 128, 0>(src, kp, sink, 131072);    // columns vary per pass
    run<8, 8, 512, 128, 10>(src, kp, sink, 131072);   // GEMM-like: all 256 rows at the SAME column per round
    run<8, 8, 512, 128, 11>(src, kp, sink, 131072);   // rotated: row r reads column (round + r)
    run<8, 0, 512, 128, 10>(src, kp, sink, 131072);
    run<8, 0, 512, 128, 11>(src, kp, sink, 131072);
    return 0;
}
