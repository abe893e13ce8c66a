// Sustained INT8 / FP8 MFMA rate of the whole chip with NO memory traffic: every wave keeps 8 independent 32x32 accumulator
// tiles busy from registers holding pseudo-random operands.  Together with tools/clk_probe.sh-style power sampling this gives
// the board's power-limited matrix ceiling (the number the GEMM kernels can at best approach), as opposed to the nominal
// 5 POP/s at 2.4 GHz.   build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ;  run: ./mfma_peak [seconds] [i8|f8] [zero]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <bool F8, bool AG = false, bool ROT = false> __global__ void __launch_bounds__(512) spin(int iters, int zero, int* sink) {
    unsigned s = zero ? 0u : (threadIdx.x * 2654435761u + blockIdx.x * 40503u) | 1u;
    auto rnd = [&]() {
        s ^= s << 13, s ^= s >> 17, s ^= s << 5;
        return (int)(zero ? 0u : (F8 ? (s & 0x3F3F3F3Fu) : s));   // FP8: keep the e4m3 bytes finite and small
    };
    if constexpr (!F8) {
        // ROT: four operand sets used in turn (like the four sub-steps of a K-step): the operand buses toggle as in a real GEMM
        constexpr int SETS = ROT ? 4 : 1;
        v4i as[SETS][4], bs[SETS][2];
        for (auto& st : as)
            for (auto& x : st) x = v4i{rnd(), rnd(), rnd(), rnd()};
        for (auto& st : bs)
            for (auto& x : st) x = v4i{rnd(), rnd(), rnd(), rnd()};
        v16i acc[4][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const v4i(&a)[4] = as[ROT ? u : 0];
                const v4i(&b)[2] = bs[ROT ? u : 0];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (AG) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));  // accumulators in AccVGPRs
                        else acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        int t = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) t += acc[i][j][0];
        if (t == 0x7fffffff) sink[0] = t;
    } else {
        v8i a[4], b[2];
        for (auto& x : a) x = v8i{rnd(), rnd(), rnd(), rnd(), rnd(), rnd(), rnd(), rnd()};
        for (auto& x : b) x = v8i{rnd(), rnd(), rnd(), rnd(), rnd(), rnd(), rnd(), rnd()};
        v16f acc[4][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
        float t = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) t += acc[i][j][0];
        if (t == 12345.678f) sink[0] = 1;
    }
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const bool f8 = argc > 2 && !strcmp(argv[2], "f8");
    const int zero = argc > 3 && !strcmp(argv[3], "zero");
    const bool rot = argc > 3 && !strcmp(argv[3], "rot");     // i8 only: four rotating operand sets
    const bool agpr = argc > 3 && !strcmp(argv[3], "agpr");   // i8 only: accumulators in AccVGPRs (random operands)
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int* sink;
    hipMalloc(&sink, 4);
    const int iters = 20000;                                  // per launch: 8 waves x iters x 32 MFMA (i8) per CU
    const double ops_per_launch = (double)p.multiProcessorCount * 8 * iters * (f8 ? 16.0 * 2 * 32 * 32 * 64 : 32.0 * 2 * 32 * 32 * 32);
    auto launch = [&]() {
        if (f8) hipLaunchKernelGGL(spin<true>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, zero, sink);
        else if (agpr) hipLaunchKernelGGL((spin<false, true>), dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, zero, sink);
        else if (rot) hipLaunchKernelGGL((spin<false, false, true>), dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, zero, sink);
        else hipLaunchKernelGGL(spin<false>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, zero, sink);
    };
    launch();
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 4; ++i) launch();
        hipDeviceSynchronize();
        n += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("%s MFMA, %s operands, %d CUs x 8 waves: %.0f TOP/s sustained over %.1f s\n", f8 ? "FP8 32x32x64 (scaled)" : "INT8 32x32x32",
           zero ? "all-zero" : "pseudo-random", p.multiProcessorCount, ops_per_launch * n / el * 1e-12, el);
    return 0;
}
