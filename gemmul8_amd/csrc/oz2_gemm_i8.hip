// INT8 x INT8 -> INT32 "TN" GEMM on gfx950 MFMA with fused epilogues (Ozaki-II hot loop).
//
// Replaces the vendor-BLAS call sites of the reference (GEMMul8/src/matmult.hpp:120-175 i8x1,
// :213-302 i8x3) AND the separate requantise pass (src/conv_hi2mid_real.hpp:9-25,
// src/conv_hi2mid_complex.hpp:9-127) / the bound-matrix max passes
// (src/scaling_accu_real.hpp:142-226): the INT32 accumulators never leave registers.
//
//   C[t](i,j) = sum_kk A_lo[t](i,kk) * B_lo[t](j,kk)      both operands K-contiguous ("TN")
//   EPI_MOD : C_mid[t](i,j) = int8( symmetric residue of C[t](i,j) mod p_t )
//   EPI_MAX : rowmax[i] = max_j C(i,j), colmax[j] = max_i C(i,j)      (accurate-mode bound GEMM)
//
// Tiling (CDNA4): 256x256 output tile per 512-thread workgroup (8 waves = 2(M) x 4(N), wave tile
// 128x64 = 4x2 MFMA_I32_32x32x32_I8 accumulators = 128 AGPR/VGPR per lane), BK = 128 bytes per
// K-step, LDS double buffer 2 x (32 KiB A + 32 KiB B) = 128 KiB filled by global_load_lds_dwordx4
// (LDS-DMA, no VGPR round trip).  LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with
// (row>>1)&7 on the SOURCE address (the DMA destination must stay lane-linear) and on the
// ds_read_b128 address, which makes every 16-lane read group hit 16 distinct 16-B bank slots.
// Workgroup -> tile mapping is XCD-aware (contiguous tile range per XCD, 8-row tile groups) so the
// 32 CUs of one XCD share A/B panels through their private L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "oz2_kernels.h"

namespace oz2 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 128;
constexpr int NTHREADS = 512;
constexpr int TILE_BYTES = BM * BK;            // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // double buffer = 128 KiB

enum { EPI_MOD = 0, EPI_MAX = 1 };

struct GemmArgs {
    const int8_t* A;       // plane 0 of A_lo: [rowsA(pad 256)][kp]
    const int8_t* B;       // plane 0 of B_lo: [n][kp]
    size_t strideA;        // bytes between consecutive planes
    size_t strideB;
    int kp;                // padded K (multiple of 256) = row pitch in bytes
    int m;                 // valid rows of C
    int n;                 // valid cols of C
    int tiles_m, tiles_n;
    int t_begin;           // first modulus index handled (plane p <-> modulus t_begin + p)
    int8_t* Cmid;          // plane of modulus t at Cmid + t*strideC : [n][ldc]
    size_t ldc;
    size_t strideC;
    int* rowmax;           // EPI_MAX
    int* colmax;
    int moduli[20];
    int pinv32[20];
};

__device__ __forceinline__ void issue_tile_loads(const int8_t* __restrict__ gA, const int8_t* __restrict__ gB, int kp, int nB_valid_rows,
                                                 char* lds_stage, int kt, int tid, int wave) {
    // 2048 16-byte slots per operand tile; slot p <-> (row = p>>3, physical chunk = p&7)
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int p = pass * NTHREADS + tid;
        const int row = p >> 3;
        const int c = (p & 7) ^ ((row >> 1) & 7);
        const int8_t* src = gA + (size_t)row * kp + (size_t)kt * BK + c * 16;
        char* dst = lds_stage + (pass * NTHREADS + wave * 64) * 16;  // wave-uniform; HW adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int p = pass * NTHREADS + tid;
        int row = p >> 3;
        const int c = (p & 7) ^ ((row >> 1) & 7);
        row = row < nB_valid_rows ? row : nB_valid_rows - 1;  // clamp: B_lo has exactly n rows
        const int8_t* src = gB + (size_t)row * kp + (size_t)kt * BK + c * 16;
        char* dst = lds_stage + TILE_BYTES + (pass * NTHREADS + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

template <int EPI>
__global__ void __launch_bounds__(NTHREADS, 1) gemm_i8_kernel(const GemmArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous tile range
    const int tiles_per_plane = args.tiles_m * args.tiles_n;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int plane = bid / tiles_per_plane;
    int rem = bid - plane * tiles_per_plane;
    // grouped ordering: 8 tile-rows per group, tile-row fastest inside a group
    constexpr int GM = 8;
    const int group_sz = GM * args.tiles_n;
    const int g = rem / group_sz;
    const int first_m = g * GM;
    const int gm = (args.tiles_m - first_m) < GM ? (args.tiles_m - first_m) : GM;
    rem -= g * group_sz;
    const int tm = first_m + rem % gm;
    const int tn = rem / gm;

    const int8_t* gA = args.A + (size_t)plane * args.strideA + (size_t)tm * BM * args.kp;
    const int8_t* gB = args.B + (size_t)plane * args.strideB + (size_t)tn * BN * args.kp;
    const int nB_valid = (args.n - tn * BN) < BN ? (args.n - tn * BN) : BN;
    const int KT = args.kp / BK;

    v16i acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // per-lane fragment addressing
    const int frow = lane & 31;
    const int khalf = lane >> 5;
    const int sw = (frow >> 1) & 7;
    const int a_base = (wm * 128 + frow) * BK;
    const int b_base = TILE_BYTES + (wn * 64 + frow) * BK;

    issue_tile_loads(gA, gB, args.kp, nB_valid, smem, 0, tid, wave);

    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE_BYTES;
        if (kt + 1 < KT) {
            issue_tile_loads(gA, gB, args.kp, nB_valid, smem + ((kt + 1) & 1) * STAGE_BYTES, kt + 1, tid, wave);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the 8 loads of tile kt have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();

#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = (((ks << 1) | khalf) ^ sw) << 4;
            v4i af[4], bf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const v4i*)(cur + a_base + i * 32 * BK + coff);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const v4i*)(cur + b_base + j * 32 * BK + coff);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        // every wave must be done reading `cur` before the next iteration's DMA overwrites it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    const int i0 = tm * BM + wm * 128;
    const int j0 = tn * BN + wn * 64;

    if constexpr (EPI == EPI_MOD) {
        const int t = args.t_begin + plane;
        const int p = args.moduli[t];
        const int pinv = args.pinv32[t];
        int8_t* Cp = args.Cmid + (size_t)t * args.strideC;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j0 + j * 32 + frow;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned d[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned w = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int r = mod_i32_sym(acc[i][j][q * 4 + b], p, pinv);
                        w |= ((unsigned)r & 0xFFu) << (8 * b);
                    }
                    d[q] = w;
                }
                // rows held: lane-half h owns rows 8q+4h..8q+4h+3.  Exchange so that h=0 owns rows 0..15, h=1 rows 16..31.
                auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                uint4 out = make_uint4(s0[0], s0[1], s1[0], s1[1]);
                if (col < args.n) {
                    *(uint4*)(Cp + (size_t)col * args.ldc + i0 + i * 32 + khalf * 16) = out;
                }
            }
        }
    } else {
        // column max: over this lane's 64 rows (masked to valid rows), then across the two lane halves
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int cm = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const int v = (row < args.m) ? acc[i][j][r] : 0;
                    cm = v > cm ? v : cm;
                }
            const int other = __shfl_xor(cm, 32);
            cm = other > cm ? other : cm;
            const int col = j0 + j * 32 + frow;
            if (khalf == 0 && col < args.n && cm > 0) atomicMax(args.colmax + col, cm);
        }
        // row max: across the 32 lanes (columns) of each half, for each of the 64 rows this lane touches
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int v = 0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = j0 + j * 32 + frow;
                    const int a = (col < args.n) ? acc[i][j][r] : 0;
                    v = a > v ? a : v;
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {
                    const int o = __shfl_xor(v, off);
                    v = o > v ? o : v;
                }
                const int row = i0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (frow == 0 && row < args.m && v > 0) atomicMax(args.rowmax + row, v);
            }
    }
}

static void fill_moduli(GemmArgs& a, int backend) {
    for (int t = 0; t < 20; ++t) {
        const int p = backend == kINT8 ? GEMMUL8_MODULI_INT8[t] : GEMMUL8_MODULI_FP8[t];
        a.moduli[t] = p;
        a.pinv32[t] = (int)(4294967296ull / (unsigned long long)p);
    }
}

template <int EPI>
static hipError_t launch(hipStream_t stream, const GemmArgs& a, int planes) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = planes * a.tiles_m * a.tiles_n;
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL(gemm_i8_kernel<EPI>, dim3(grid), dim3(NTHREADS), LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t launch_gemm_i8_mod(hipStream_t stream, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                              size_t n, int t_begin, int t_end, int8_t* Cmid, size_t ldc, size_t strideC) {
    GemmArgs a{};
    a.A = A + (size_t)t_begin * strideA;
    a.B = B + (size_t)t_begin * strideB;
    a.strideA = strideA;
    a.strideB = strideB;
    a.kp = (int)kp;
    a.m = (int)m;
    a.n = (int)n;
    a.tiles_m = (int)((m + BM - 1) / BM);
    a.tiles_n = (int)((n + BN - 1) / BN);
    a.t_begin = t_begin;
    a.Cmid = Cmid;
    a.ldc = ldc;
    a.strideC = strideC;
    fill_moduli(a, kINT8);
    return launch<EPI_MOD>(stream, a, t_end - t_begin);
}

hipError_t launch_gemm_i8_max(hipStream_t stream, const int8_t* A, const int8_t* B, size_t kp, size_t m, size_t n, int* rowmax,
                              int* colmax) {
    GemmArgs a{};
    a.A = A;
    a.B = B;
    a.kp = (int)kp;
    a.m = (int)m;
    a.n = (int)n;
    a.tiles_m = (int)((m + BM - 1) / BM);
    a.tiles_n = (int)((n + BN - 1) / BN);
    a.rowmax = rowmax;
    a.colmax = colmax;
    fill_moduli(a, kINT8);
    return launch<EPI_MAX>(stream, a, 1);
}

}  // namespace oz2
