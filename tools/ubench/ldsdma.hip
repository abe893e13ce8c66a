// Microbenchmark: per-CU throughput of global->LDS paths on gfx950 from an L2-resident source.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), DEPTH passes in flight per wave, vmcnt(0) + barrier per round
//   mode 1: global_load_dwordx4 to VGPRs (no LDS write), DEPTH loads in flight
//   mode 2: global_load_dwordx4 to VGPRs + ds_write_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH, int NT>
__global__ void __launch_bounds__(NT) k(const char* __restrict__ src, int rounds, size_t span, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)(blockIdx.x % 64) * span;  // a few MB total: L2/MALL resident
    v4i acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        const size_t off0 = ((size_t)r * DEPTH * NT * 16) % (span - DEPTH * NT * 16);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* g = base + off0 + (size_t)(d * NT + tid) * 16;
            if constexpr (MODE == 0) {
                char* dst = smem + ((d * NT + wave * 64) * 16) % 65536;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else {
                v4i v = *(const v4i*)g;
                if constexpr (MODE == 2) *(v4i*)(smem + ((d * NT + tid) * 16) % 65536) = v;
                else acc += v;
            }
        }
        if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (MODE == 0) acc[0] = smem[tid * 4];
    if (acc[0] == 0x12345678) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int MODE, int DEPTH, int NT> void run(const char* src, size_t span, int* sink, const char* name) {
    const int rounds = 4096 / DEPTH;
    hipFuncSetAttribute((const void*)k<MODE, DEPTH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<MODE, DEPTH, NT><<<256, NT, 65536>>>(src, rounds, span, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, DEPTH, NT><<<256, NT, 65536>>>(src, rounds, span, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * rounds * DEPTH * NT * 16;
    std::printf("%-28s depth %2d threads %4d: %7.3f ms  %6.2f TB/s  %5.1f GB/s/CU\n", name, DEPTH, NT, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
}

int main() {
    const size_t span = 1 << 20;  // 1 MiB per block-slot, 64 slots = 64 MiB
    char* src;
    hipMalloc(&src, span * 64);
    hipMemset(src, 1, span * 64);
    int* sink;
    hipMalloc(&sink, 64);
    run<0, 1, 512>(src, span, sink, "lds-dma");
    run<0, 2, 512>(src, span, sink, "lds-dma");
    run<0, 4, 512>(src, span, sink, "lds-dma");
    run<0, 8, 512>(src, span, sink, "lds-dma");
    run<0, 8, 1024>(src, span, sink, "lds-dma");
    run<0, 4, 256>(src, span, sink, "lds-dma");
    run<1, 4, 512>(src, span, sink, "global_load->vgpr");
    run<1, 8, 512>(src, span, sink, "global_load->vgpr");
    run<1, 8, 1024>(src, span, sink, "global_load->vgpr");
    run<2, 4, 512>(src, span, sink, "global_load+ds_write");
    run<2, 8, 512>(src, span, sink, "global_load+ds_write");
    return 0;
}
