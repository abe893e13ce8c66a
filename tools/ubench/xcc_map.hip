// Which XCD does workgroup b of a launch run on?  The GEMM kernels' tile maps (csrc/oz2_gemm_common.hpp) assume the round-robin dispatch "b -> XCD b % 8" (up to a
// rotation).  Reads HW_REG_XCC_ID per workgroup for several grids launched back to back and prints, per launch, the XCC id of workgroups 0..15 and whether
// xcc(b) == (xcc(0) + b) % 8 holds for the whole grid.  usage: ./xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v & 0xF;
    for (int i = 0; i < spin; ++i) asm volatile("s_sleep 8");   // keep the workgroups resident so that the whole grid is placed at once
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4096 * 4);
    const int grids[] = {256, 256, 264, 100, 256, 1024, 255, 256};
    for (int g : grids) {
        hipMemset(d, 0xFF, 4096 * 4);
        hipLaunchKernelGGL(probe, dim3(g), dim3(768), 160 * 1024 - 1024, 0, d, 200);
        hipDeviceSynchronize();
        std::vector<unsigned> h(g);
        hipMemcpy(h.data(), d, g * 4, hipMemcpyDeviceToHost);
        int bad = 0, cnt[16] = {};
        for (int b = 0; b < g; ++b) {
            bad += h[b] != (h[0] + b) % 8;
            cnt[h[b] & 15]++;
        }
        printf("grid %4d: xcc of b = 0..15:", g);
        for (int b = 0; b < 16 && b < g; ++b) printf(" %u", h[b]);
        printf("  | workgroups off the round-robin: %d | per XCC:", bad);
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("\n");
    }
    return 0;
}
