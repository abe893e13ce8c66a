// Small-tile INT8 bound GEMM for gfx950: the accurate mode's ONE-plane product with the row / column maxima epilogue
// (GEMMul8/src/scaling_accu_real.hpp:142-226,415-432, scaling_accu_complex.hpp:132-224,441-460) when its 256 x 256 tiles would
// leave most of the chip idle.
//
// The persistent kernel of oz2_gemm_i8.hip owns a whole CU per 256 x 256 tile; a single plane of a 1024^2 / 2048^2 / 2048 x 4096
// product is 16 / 64 / 128 such tiles on 256 CUs -- the bound GEMM ran latency-bound on a fraction of the chip (27.7 us at 1024^3 against
// 21.7 us for the 224 tiles of the 14 residue planes).  Here: 128 x 128 tiles, one 256-thread workgroup each
// (2 x 2 waves, wave tile 64 x 64 = 4 x 4 v_mfma_i32_16x16x64_i8 tiles, 64 accumulator registers), two workgroups per CU, 4x the
// workgroups.  Same operands (K-contiguous "TN", up to 3 concatenated K-segments), same maxima: exact integers, max-combined with
// atomicMax -- the result is identical to the big kernel's whatever the tiling (tests/test_gpu_parity.py runs both).
//
// Structure: BK = 128 bytes per K-step; the two 128 x 128-byte operand panels of a K-step go global -> registers (4 + 4 16-byte loads
// per thread, full 128-byte lines per 8 lanes) -> LDS (two stages of 32 KiB), the loads of K-step g + 1 are issued before the MFMAs of
// K-step g, one workgroup barrier per K-step.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 as in
// the big kernel: the 16 rows of a fragment read hit 16 different bank slots.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "oz2_gemm_common.hpp"
#include "oz2_kernels.h"

namespace oz2 {

namespace {
constexpr int SBM = 128, SBN = 128;
struct SmallArgs {
    const int8_t* A[3];
    const int8_t* B[3];
    int nseg;
    int kp;
    int m, n;
    int tiles_m, tiles_n;
    int* rowmax;
    int* colmax;
    int ks_mid;      // > 0: the maxima are also taken after this many K-steps (partial sums of a K-concatenation), see launch_gemm_i8_max
    size_t bstride;  // bytes between the workspaces of consecutive batch items (gridDim.z)
};

__device__ __forceinline__ unsigned lds_off(unsigned row, unsigned chunk) { return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4); }

__global__ void __launch_bounds__(256, 2) gemm_i8_max_small_kernel(const SmallArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][SBM * BK];  // [stage][A | B]
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned wm = wave >> 1, wn = wave & 1u;
    const int tm = (int)(blockIdx.x % (unsigned)a.tiles_m), tn = (int)(blockIdx.x / (unsigned)a.tiles_m);
    const int row0 = tm * SBM, col0 = tn * SBN;
    const size_t boff = (size_t)blockIdx.z * a.bstride;

    // global -> register staging: piece i of a thread = row (tid + 256 i) / 8, 16-byte chunk (tid + 256 i) % 8 of the panel
    size_t goffA[4], goffB[4];
    unsigned loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned idx = tid + 256u * (unsigned)i, r = idx >> 3, c = idx & 7u;
        // A planes are padded to 256 rows (rows m.. hold whatever the extract left: masked in the epilogue); B has exactly n rows
        const int rb = col0 + (int)r < a.n ? col0 + (int)r : a.n - 1;
        goffA[i] = boff + (size_t)(row0 + (int)r) * (size_t)a.kp + c * 16u;
        goffB[i] = boff + (size_t)rb * (size_t)a.kp + c * 16u;
        loff[i] = lds_off(r, c);
    }
    const int kper = a.kp / BK;
    const int ksteps = a.nseg * kper;
    // (macros, not lambdas: arrays captured by reference stayed in memory -- scratch and a promoted LDS copy)
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define OZ2_GLOAD(ks_)                                                                                                  \
    {                                                                                                                   \
        const int seg_ = (ks_) / kper;                                                                                  \
        const size_t kk_ = (size_t)((ks_) - seg_ * kper) * BK;                                                          \
        /* selects, not a.A[seg]: a run-time index into the by-value argument block would move it to scratch */        \
        const int8_t* pa_ = (seg_ == 0 ? a.A[0] : seg_ == 1 ? a.A[1] : a.A[2]) + kk_;                                   \
        const int8_t* pb_ = (seg_ == 0 ? a.B[0] : seg_ == 1 ? a.B[1] : a.B[2]) + kk_;                                   \
        ra0 = *(const uint4*)(pa_ + goffA[0]), ra1 = *(const uint4*)(pa_ + goffA[1]);                                   \
        ra2 = *(const uint4*)(pa_ + goffA[2]), ra3 = *(const uint4*)(pa_ + goffA[3]);                                   \
        rb0 = *(const uint4*)(pb_ + goffB[0]), rb1 = *(const uint4*)(pb_ + goffB[1]);                                   \
        rb2 = *(const uint4*)(pb_ + goffB[2]), rb3 = *(const uint4*)(pb_ + goffB[3]);                                   \
    }
#define OZ2_LSTORE(stage_)                                                                                              \
    {                                                                                                                   \
        *(uint4*)(&lds[stage_][0][loff[0]]) = ra0, *(uint4*)(&lds[stage_][0][loff[1]]) = ra1;                           \
        *(uint4*)(&lds[stage_][0][loff[2]]) = ra2, *(uint4*)(&lds[stage_][0][loff[3]]) = ra3;                           \
        *(uint4*)(&lds[stage_][1][loff[0]]) = rb0, *(uint4*)(&lds[stage_][1][loff[1]]) = rb1;                           \
        *(uint4*)(&lds[stage_][1][loff[2]]) = rb2, *(uint4*)(&lds[stage_][1][loff[3]]) = rb3;                           \
    }

    v4i acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    OZ2_GLOAD(0)
    OZ2_LSTORE(0)
    __syncthreads();
    const unsigned fr = lane & 15u, fq = lane >> 4;
    int ks = 0;
    const int nph = a.ks_mid > 0 ? 2 : 1;
    for (int ph = 0; ph < nph; ++ph) {
    const int ks_end = ph + 1 < nph ? a.ks_mid : ksteps;
    for (; ks < ks_end; ++ks) {
        const int st = ks & 1;
        if (ks + 1 < ksteps) OZ2_GLOAD(ks + 1)
#pragma unroll
        for (unsigned kh = 0; kh < 2; ++kh) {
            v4i af[4], bf[4];
#pragma unroll
            for (unsigned t = 0; t < 4; ++t) {
                af[t] = *(const v4i*)(&lds[st][0][lds_off(wm * 64u + t * 16u + fr, 4u * kh + fq)]);
                bf[t] = *(const v4i*)(&lds[st][1][lds_off(wn * 64u + t * 16u + fr, 4u * kh + fq)]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < ksteps) OZ2_LSTORE(st ^ 1)
        __syncthreads();
    }

    // maxima of the wave's 64 x 64 block (accumulator map: col = lane & 15, row = 4 (lane >> 4) + reg)
    const int i0 = row0 + (int)wm * 64, j0 = col0 + (int)wn * 64;
    const int c16 = (int)fr, q = (int)fq;
    int* const rowmax_ = (int*)((char*)a.rowmax + boff);
    int* const colmax_ = (int*)((char*)a.colmax + boff);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        int cm = 0;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + ti * 16 + 4 * q + r;
                const int v = (row < a.m) ? acc[ti][tj][r] : 0;
                cm = v > cm ? v : cm;
            }
        int other = __shfl_xor(cm, 16);
        cm = other > cm ? other : cm;
        other = __shfl_xor(cm, 32);
        cm = other > cm ? other : cm;
        const int col = j0 + tj * 16 + c16;
        if (q == 0 && col < a.n && cm > 0) atomicMax(colmax_ + col, cm);
    }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        int w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int v = 0;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const int col = j0 + tj * 16 + c16;
                const int x = (col < a.n) ? acc[ti][tj][r] : 0;
                v = x > v ? x : v;
            }
            w[r] = v;
        }
        tile_rowmax_atomic16(w, rowmax_, i0 + ti * 16, a.m, (int)lane);
    }
    }  // phase
#undef OZ2_GLOAD
#undef OZ2_LSTORE
}
}  // namespace

hipError_t launch_gemm_i8_max_small(hipStream_t stream, int nseg, const int8_t* const* A, const int8_t* const* B, size_t kp, size_t m, size_t n,
                                    int* rowmax, int* colmax, int mid_seg) {
    if (m == 0 || n == 0) return hipSuccess;
    SmallArgs a{};
    for (int s = 0; s < nseg; ++s) a.A[s] = A[s], a.B[s] = B[s];
    a.nseg = nseg;
    a.kp = (int)kp;
    a.m = (int)m;
    a.n = (int)n;
    a.tiles_m = (int)((m + SBM - 1) / SBM);
    a.tiles_n = (int)((n + SBN - 1) / SBN);
    a.rowmax = rowmax;
    a.colmax = colmax;
    a.ks_mid = mid_seg > 0 && mid_seg < nseg ? mid_seg * (int)(kp / BK) : 0;
    a.bstride = g_batch.ws;
    const size_t tiles = (size_t)a.tiles_m * a.tiles_n;
    if (tiles > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(gemm_i8_max_small_kernel, dim3((unsigned)tiles, 1, g_batch.batch), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace oz2
