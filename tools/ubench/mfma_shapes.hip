// Power-limited INT8 MFMA ceiling by INSTRUCTION SHAPE and by OPERAND DISTRIBUTION, no memory traffic (registers only).
// Round-2 question: does v_mfma_i32_16x16x64_i8 sustain more than the 3.43-3.64 POP/s that
// v_mfma_i32_32x32x32_i8 holds on random data, and how much does the operand distribution move the ceiling?
//   build: hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip -o mfma_shapes
//   run  : ./mfma_shapes <seconds> <shape: 32|16|16w> <data: rand|res|small|zero|sparse>
//          32  = 32x32x32, 8 accumulator tiles per wave (the shipped consumer-wave blocking 4x2)
//          16  = 16x16x64, 32 accumulator tiles per wave (8x4: the same 128x64 wave tile)
//          16w = 16x16x64, 16 accumulator tiles (4x4: a 64x64 wave tile)
//          f32 = FP8 v_mfma_scale_f32_32x32x64_f8f6f4 (8 tiles), f16 = FP8 v_mfma_scale_f32_16x16x128_f8f6f4 (32 tiles);
//                data for these: rand = random finite e4m3 bytes, res = integers uniform in [-16, 16], zero
//          s32 / s16 = the same two instructions on FP6 (e2m3) operands (cbsz = blgp = 2; 6 registers per operand): round 5
//          data: rand = random bytes; res = symmetric residues mod 251 (uniform in [-125,125], what the GEMM really multiplies);
//                small = |x| <= 7; zero; sparse = random bytes with 3 of 4 zeroed
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// FP8 (e4m3) operand dword: mode 0 = random finite bytes, 1 = integers uniform in [-16, 16] (what the FP8 backend multiplies), 3 = zero
__device__ inline int gen8(unsigned& s, int mode) {
    if (mode == 3) return 0;
    unsigned w = 0;
    for (int h = 0; h < 2; ++h) {
        s ^= s << 13, s ^= s >> 17, s ^= s << 5;
        if (mode == 0) {
            w |= (s & 0x3F3Fu) << (16 * h);
        } else {
            const float a = (float)((int)((s >> 8) % 33u) - 16), b = (float)((int)((s >> 16) % 33u) - 16);
            w |= ((unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xFFFFu) << (16 * h);
        }
    }
    return (int)w;
}
// FP6 (e2m3) operand: 32 codes of 6 bits in 6 dwords; an integer v in [-16, 16] scaled by 2^-3 is the code sign << 5 | |v| (subnormal /
// normal e2m3 happen to be the plain binary magnitude up to 16); mode 0 = random codes, 1 = integers uniform in [-16, 16], 3 = zero
// (the builtin takes 8-register operands for every format; with cbsz / blgp = 2 the instruction reads the low six)
__device__ inline v8i gen6(unsigned& s, int mode) {
    unsigned w[6] = {0, 0, 0, 0, 0, 0};
    if (mode != 3)
        for (int i = 0; i < 32; ++i) {
            s ^= s << 13, s ^= s >> 17, s ^= s << 5;
            unsigned c;
            if (mode == 0) c = (s >> 9) & 63u;
            else {
                const int v = (int)((s >> 8) % 33u) - 16;
                c = (v < 0 ? 32u : 0u) | (unsigned)(v < 0 ? -v : v);
            }
            const int bit = 6 * i;
            w[bit >> 5] |= c << (bit & 31);
            if ((bit & 31) > 26) w[(bit >> 5) + 1] |= c >> (32 - (bit & 31));
        }
    return v8i{(int)w[0], (int)w[1], (int)w[2], (int)w[3], (int)w[4], (int)w[5], 0, 0};
}
// FP6 x FP6 (cbsz = blgp = 2), unit scales: the same loop bodies as spin8
template <int SHAPE> __global__ void __launch_bounds__(512) spin6(int iters, int mode, int* sink) {
    unsigned s = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) | 1u;
    if constexpr (SHAPE == 32) {
        v8i a[4], b[2];
        for (auto& x : a) x = gen6(s, mode);
        for (auto& x : b) x = gen6(s, mode);
        v16f acc[4][2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        float t = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) t += acc[i][j][0];
        if (t == 12345.678f) sink[0] = 1;
    } else {
        v8i a[4], b[4];
        for (auto& x : a) x = gen6(s, mode);
        for (auto& x : b) x = gen6(s, mode);
        v4f acc[8][4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 3], b[j], acc[i][j], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        float t = 0;
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 4; ++j) t += acc[i][j][0];
        if (t == 12345.678f) sink[0] = 1;
    }
}
template <int SHAPE> __global__ void __launch_bounds__(512) spin8(int iters, int mode, int* sink) {
    unsigned s = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) | 1u;
    auto g8 = [&]() { return v8i{gen8(s, mode), gen8(s, mode), gen8(s, mode), gen8(s, mode), gen8(s, mode), gen8(s, mode), gen8(s, mode), gen8(s, mode)}; };
    if constexpr (SHAPE == 32) {  // 32x32x64, 4 x 2 accumulator tiles
        v8i a[4], b[2];
        for (auto& x : a) x = g8();
        for (auto& x : b) x = g8();
        v16f acc[4][2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        float t = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) t += acc[i][j][0];
        if (t == 12345.678f) sink[0] = 1;
    } else {  // 16x16x128, 8 x 4 accumulator tiles (A fragments 8 x 8 regs would not fit: 4 x 4 tiles re-used twice)
        v8i a[4], b[4];
        for (auto& x : a) x = g8();
        for (auto& x : b) x = g8();
        v4f acc[8][4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 3], b[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        float t = 0;
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 4; ++j) t += acc[i][j][0];
        if (t == 12345.678f) sink[0] = 1;
    }
}

__device__ inline int gen(unsigned& s, int mode) {
    unsigned w = 0;
    for (int b = 0; b < 4; ++b) {
        s ^= s << 13, s ^= s >> 17, s ^= s << 5;
        int v;
        switch (mode) {
        case 0: v = (int)(s & 255u); break;                                  // rand
        case 1: v = (int)((s >> 8) % 251u) - 125; break;                     // residues mod 251
        case 2: v = (int)((s >> 8) % 15u) - 7; break;                        // small
        case 3: v = 0; break;                                                // zero
        default: v = ((s >> 20) & 3u) ? 0 : (int)(s & 255u); break;          // sparse
        }
        w |= ((unsigned)v & 255u) << (8 * b);
    }
    return (int)w;
}

template <int SHAPE> __global__ void __launch_bounds__(512) spin(int iters, int mode, int* sink) {
    unsigned s = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) | 1u;
    if constexpr (SHAPE == 32) {
        v4i a[4], b[2];
        for (auto& x : a) x = v4i{gen(s, mode), gen(s, mode), gen(s, mode), gen(s, mode)};
        for (auto& x : b) x = v4i{gen(s, mode), gen(s, mode), gen(s, mode), gen(s, mode)};
        v16i acc[4][2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        int t = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) t += acc[i][j][0];
        if (t == 0x7fffffff) sink[0] = t;
    } else {
        constexpr int NI = SHAPE == 16 ? 8 : 4, NJ = 4;
        v4i a[NI], b[NJ];
        for (auto& x : a) x = v4i{gen(s, mode), gen(s, mode), gen(s, mode), gen(s, mode)};
        for (auto& x : b) x = v4i{gen(s, mode), gen(s, mode), gen(s, mode), gen(s, mode)};
        v4i acc[NI][NJ] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < (SHAPE == 16 ? 2 : 4); ++u)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        int t = 0;
        for (int i = 0; i < NI; ++i)
            for (int j = 0; j < NJ; ++j) t += acc[i][j][0];
        if (t == 0x7fffffff) sink[0] = t;
    }
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const char* shape = argc > 2 ? argv[2] : "32";
    const char* data = argc > 3 ? argv[3] : "rand";
    const int mode = !strcmp(data, "rand") ? 0 : !strcmp(data, "res") ? 1 : !strcmp(data, "small") ? 2 : !strcmp(data, "zero") ? 3 : 4;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int* sink;
    hipMalloc(&sink, 4);
    const int iters = 20000;
    // every variant issues 64 x (2 * 16384 ops) = 2^21 ops per wave and iteration: 32: 4*8 MFMAs of 65536; 16: 2*32 of 32768; 16w: 4*16
    const double ops_per_launch = (double)p.multiProcessorCount * 8 * iters * 32.0 * 65536.0;
    auto launch = [&]() {
        if (!strcmp(shape, "s32")) hipLaunchKernelGGL(spin6<32>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, mode, sink);
        else if (!strcmp(shape, "s16")) hipLaunchKernelGGL(spin6<16>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, mode, sink);
        else if (!strcmp(shape, "f32")) hipLaunchKernelGGL(spin8<32>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, mode, sink);
        else if (!strcmp(shape, "f16")) hipLaunchKernelGGL(spin8<16>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, mode, sink);
        else if (!strcmp(shape, "32")) hipLaunchKernelGGL(spin<32>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, mode, sink);
        else if (!strcmp(shape, "16")) hipLaunchKernelGGL(spin<16>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, mode, sink);
        else hipLaunchKernelGGL(spin<17>, dim3(p.multiProcessorCount), dim3(512), 0, 0, iters, mode, sink);
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
    printf("MFMA shape %s, data %s, %d CUs x 8 waves: %.0f TOP/s sustained over %.1f s\n", shape, data, p.multiProcessorCount,
           ops_per_launch * n / el * 1e-12, el);
    return 0;
}
