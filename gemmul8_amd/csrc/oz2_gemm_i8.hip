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
#ifndef OZ2_ISSUE_STEPS
#define OZ2_ISSUE_STEPS 2
#endif
#ifndef OZ2_SCHED
#define OZ2_SCHED 3
#endif
#ifndef OZ2_SUB
#define OZ2_SUB 1
#endif
constexpr int ISSUE_STEPS = OZ2_ISSUE_STEPS;   // k-substeps over which the next tile's 8 DMA passes are spread

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

// One DMA pass = 512 lanes x 16 B = 8 KiB of one operand tile (2048 16-byte slots per tile: slot p <->
// row = p>>3, physical chunk = p&7; logical chunk = physical ^ ((row>>1)&7)).
template <bool IS_B>
__device__ __forceinline__ void issue_pass(const int8_t* __restrict__ g, int kp, int valid_rows, char* lds_tile, int kt, int pass, int tid,
                                           int wave) {
    const int p = pass * NTHREADS + tid;
    int row = p >> 3;
    const int c = (p & 7) ^ ((row >> 1) & 7);
    if (IS_B) row = row < valid_rows ? row : valid_rows - 1;  // B_lo has exactly n rows: clamp instead of padding
    const int8_t* src = g + (size_t)row * kp + (size_t)kt * BK + c * 16;
    char* dst = lds_tile + (pass * NTHREADS + wave * 64) * 16;  // wave-uniform; HW adds lane*16
#ifdef OZ2_ABL_NODMA
    if (kt > 0) return;  // ablation: only the first tile is fetched
#endif
#ifdef OZ2_ABL_SAMETILE
    kt = 0;              // ablation: every step re-fetches tile 0 (always L2-resident)
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0,
                                     0);
}
__device__ __forceinline__ void issue_tile_loads(const int8_t* __restrict__ gA, const int8_t* __restrict__ gB, int kp, int nB_valid_rows,
                                                 char* lds_stage, int kt, int tid, int wave) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) issue_pass<false>(gA, kp, 0, lds_stage, kt, pass, tid, wave);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) issue_pass<true>(gB, kp, nB_valid_rows, lds_stage + TILE_BYTES, kt, pass, tid, wave);
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

#if OZ2_SCHED == 0
    // ---- lock-step schedule: two barriers per K-step, all 8 waves in the same phase
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        const int ktn = kt + 1 < KT ? kt + 1 : kt;        // last step re-fetches its own tile into the idle buffer (no branch)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of tile kt have landed
        __builtin_amdgcn_s_barrier();                     // ... and everybody else's
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = (((ks << 1) | khalf) ^ sw) << 4;
            v4i af[4], bf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const v4i*)(cur + a_base + i * 32 * BK + coff);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const v4i*)(cur + b_base + j * 32 * BK + coff);
            if (ks < ISSUE_STEPS) {
#pragma unroll
                for (int q = 0; q < 4 / ISSUE_STEPS; ++q) {
                    issue_pass<false>(gA, args.kp, 0, nxt, ktn, ks * (4 / ISSUE_STEPS) + q, tid, wave);
                    issue_pass<true>(gB, args.kp, nB_valid, nxt + TILE_BYTES, ktn, ks * (4 / ISSUE_STEPS) + q, tid, wave);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#elif OZ2_SCHED == 1
    // ---- ping-pong schedule.  Every wave alternates a LOAD segment (ds_read of SUB k-substeps of
    // fragments + its share of the next tile's DMA) and an MFMA segment (8*SUB MFMAs from registers),
    // one s_barrier after each.  The wm=1 half of the workgroup runs ONE segment behind the wm=0 half
    // (one extra barrier up front, one extra for wm=0 at the end), so on every SIMD one wave feeds the
    // matrix pipe while its partner reads LDS / issues DMA.  Hazards (slot = barrier interval):
    //   RAW  tile kt+1 is DMA'd in the first LOAD segments of step kt; every wave drains vmcnt(0) at the
    //        end of its LAST LOAD segment of step kt, one barrier before the leading half reads it;
    //   WAR  the stage being refilled was last read in the trailing half's last LOAD segment of step
    //        kt-1, which ends (lgkmcnt(0) + barrier) before the leading half's first DMA of step kt.
    constexpr int SUB = OZ2_SUB;          // k-substeps per segment (1 or 2)
    constexpr int NSEG = 4 / SUB;         // LOAD/MFMA segment pairs per K-step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        const int ktn = kt + 1 < KT ? kt + 1 : kt;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            v4i af[SUB][4], bf[SUB][2];
#pragma unroll
            for (int u = 0; u < SUB; ++u) {
                const int ks = sg * SUB + u;
                const int coff = (((ks << 1) | khalf) ^ sw) << 4;
#ifdef OZ2_ABL_NOLDS
#pragma unroll
                for (int i = 0; i < 4; ++i) af[u][i] = v4i{coff, i, kt, lane};
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[u][j] = v4i{coff, j, kt, lane};
#else
#pragma unroll
                for (int i = 0; i < 4; ++i) af[u][i] = *(const v4i*)(cur + a_base + i * 32 * BK + coff);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[u][j] = *(const v4i*)(cur + b_base + j * 32 * BK + coff);
#endif
            }
            if (sg == NSEG - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            // The next tile's DMA (8 passes per wave per K-step) goes out in the MFMA segments of the FIRST
            // half of the step, one pass behind every other MFMA: its 4 issue slots fit in the 32-cycle
            // shadow of the MFMA just issued instead of lengthening a LOAD segment.
            constexpr int DMA_SEGS = NSEG >= 2 ? NSEG / 2 : 1;
            constexpr int PER_SEG = 8 / DMA_SEGS;          // DMA passes per MFMA segment (A and B counted separately)
            constexpr int NMFMA = 8 * SUB;
            constexpr int EVERY = NMFMA / PER_SEG > 0 ? NMFMA / PER_SEG : 1;
#pragma unroll
            for (int u = 0; u < SUB; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#ifdef OZ2_ABL_NOMFMA
                        acc[i][j][0] += af[u][i][0] + bf[u][j][1];
#else
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[u][i], bf[u][j], acc[i][j], 0, 0, 0);
#endif
                        const int idx = (u * 4 + i) * 2 + j;
                        if (sg < DMA_SEGS && idx % EVERY == 0 && idx / EVERY < PER_SEG) {
                            const int d = sg * PER_SEG + idx / EVERY;  // 0..7: even -> A pass d/2, odd -> B pass d/2
                            __builtin_amdgcn_sched_barrier(0);
                            if (d & 1) issue_pass<true>(gB, args.kp, nB_valid, nxt + TILE_BYTES, ktn, d >> 1, tid, wave);
                            else issue_pass<false>(gA, args.kp, 0, nxt, ktn, d >> 1, tid, wave);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#else
    // (SCHED 2 is implemented by gemm_i8_ring_kernel below)
#endif

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


// =====================================================================================================
// Ring-pipelined ping-pong kernel (default).  BK = 64 bytes per K-step, NSTAGE-deep LDS ring of
// 32 KiB stages (A 256x64 | B 256x64): tile kt+NSTAGE-1 is DMA'd while tile kt is consumed, so
// ~64-96 KiB of LDS-DMA stay in flight per CU at all times -- the LDS-DMA path has ~600 cycles of
// latency and streams ~46 B/clk/CU only when it is kept full (tools/ubench/ldsdma.hip); a 2-stage
// BK=128 pipeline leaves it idle between "drain" and "re-issue" and caps the kernel at ~45 % of peak.
// Waits are COUNTED (s_waitcnt vmcnt(4*(NSTAGE-2))), never 0, in the steady state.
// Schedule: every wave alternates LOAD (6 ds_read_b128 + 2 DMA passes) and MFMA (8 MFMAs) segments
// with one s_barrier after each; the wm=1 half runs one segment behind the wm=0 half so each SIMD always
// has one wave on the matrix pipe while its partner is on LDS/VMEM.
//   slot(G0: L_j(kt)) = 4kt+2j, M_j -> +1; G1 one slot later.
//   WAR: stage (kt-1)%NSTAGE is last read in G1's L1(kt-1) (slot 4kt-1); first refilled by G0's L0(kt) (slot 4kt).
//   RAW: every wave passes vmcnt(<= tiles kt+2..) at the end of its L1(kt) (slots 4kt+2 / 4kt+3), i.e.
//        before the barrier that opens slot 4kt+4 where G0's L0(kt+1) reads tile kt+1.
// =====================================================================================================
constexpr int RBK = 64;
constexpr int RTILE = BM * RBK;        // 16 KiB per operand per stage
constexpr int RSTAGE = 2 * RTILE;      // 32 KiB
#ifndef OZ2_NSTAGE
#define OZ2_NSTAGE 4
#endif
constexpr int NSTAGE = OZ2_NSTAGE;
constexpr int RING_LDS_BYTES = NSTAGE * RSTAGE;

// one DMA pass = 512 lanes x 16 B = 8 KiB = 128 rows x 64 B; slot p <-> row = p>>2, physical chunk = p&3,
// logical chunk = physical ^ ((row>>2)&3)  (16 rows of a ds_read_b128 lane group -> 16 distinct 16-B slots)
template <bool IS_B>
__device__ __forceinline__ void ring_issue(const int8_t* __restrict__ g, int kp, int valid_rows, char* lds_tile, int kt, int pass, int tid,
                                           int wave) {
    const int p = pass * NTHREADS + tid;
    int row = p >> 2;
    const int c = (p & 3) ^ ((row >> 2) & 3);
    if (IS_B) row = row < valid_rows ? row : valid_rows - 1;
    const int8_t* src = g + (size_t)row * kp + (size_t)kt * RBK + c * 16;
    char* dst = lds_tile + (pass * NTHREADS + wave * 64) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0,
                                     0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "unsupported vmcnt");
}

template <int EPI>
__global__ void __launch_bounds__(NTHREADS, 1) gemm_i8_ring_kernel(const GemmArgs args) {
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

    v16i acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int frow = lane & 31;
    const int khalf = lane >> 5;
    const int KT = args.kp / RBK;
    const int sw = (frow >> 2) & 3;
    const int a_base = (wm * 128 + frow) * RBK;
    const int b_base = RTILE + (wn * 64 + frow) * RBK;

    // prologue: tiles 0 .. NSTAGE-2 in flight
#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t) {
        const int kt0 = t < KT ? t : KT - 1;
        char* st = smem + t * RSTAGE;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            ring_issue<false>(gA, args.kp, 0, st, kt0, pass, tid, wave);
            ring_issue<true>(gB, args.kp, nB_valid, st + RTILE, kt0, pass, tid, wave);
        }
    }
    wait_vmcnt<4 * (NSTAGE - 2)>();  // tile 0 landed
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();

    int stage = 0;  // kt % NSTAGE
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + stage * RSTAGE;
        int fill = stage + NSTAGE - 1;
        if (fill >= NSTAGE) fill -= NSTAGE;
        char* nxt = smem + fill * RSTAGE;  // stage of tile kt+NSTAGE-1 == stage of tile kt-1 (free)
        const int ktn = kt + NSTAGE - 1 < KT ? kt + NSTAGE - 1 : KT - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = (((ks << 1) | khalf) ^ sw) << 4;
            v4i af[4], bf[2];
#ifdef OZ2_ABL_NOLDS
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = v4i{coff, i, kt, lane};
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = v4i{coff, j, kt, lane};
#else
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const v4i*)(cur + a_base + i * 32 * RBK + coff);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const v4i*)(cur + b_base + j * 32 * RBK + coff);
#endif
#ifndef OZ2_ABL_NODMA
            ring_issue<false>(gA, args.kp, 0, nxt, ktn, ks, tid, wave);
            ring_issue<true>(gB, args.kp, nB_valid, nxt + RTILE, ktn, ks, tid, wave);
#endif
            if (ks == 1) wait_vmcnt<4 * (NSTAGE - 2)>();  // everything up to tile kt+1 has landed (this wave's pieces)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#ifdef OZ2_ABL_NOMFMA
                    acc[i][j][0] += af[i][0] + bf[j][1];
#else
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
#endif
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();  // the clamped tail re-fetches must not outlive the workgroup's LDS

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



// =====================================================================================================
// Wave-specialised kernel (default, OZ2_SCHED == 3): 8 consumer waves (ds_read + MFMA, ping-pong as
// above) + 4 producer waves (one per SIMD) that do nothing but issue the LDS-DMA of the next K-tile.
// Why: the global->LDS path streams 64 B/clk/CU only for 128-byte row segments and only while its
// queue is kept full (tools/ubench/dma_ring.hip: 127 GB/s/CU ringed vs 61 drained, 68 for 64-B
// segments); issuing the 64 DMA instructions of a 64 KiB K-tile from the MFMA waves costs them
// ~1000 cycles of VMEM issue per K-step (ablation: DMA alone = 5.4 ms of the 6.8 ms kernel).  A
// producer wave blocks on the VMEM queue instead of the matrix pipe's feeders.
// BK = 128 (128-B segments), 2 LDS stages; tile kt+1 is issued from slot 8kt (the barrier that retires
// the last reader of its stage) and drained (vmcnt(0)) by its issuing wave in slot 8kt+7.
// =====================================================================================================
constexpr int WS_THREADS = 768;   // 8 consumers + 4 producers
#ifndef OZ2_PSLOTS
#define OZ2_PSLOTS 4
#endif
constexpr int PSLOTS = OZ2_PSLOTS;
  // slots (of 8 per K-step) over which a producer spreads its 16 DMA instructions

template <int EPI>
__global__ void __launch_bounds__(WS_THREADS) gemm_i8_ws_kernel(const GemmArgs args) {
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

    v16i acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int frow = lane & 31;
    const int khalf = lane >> 5;
    const int KT = args.kp / BK;
    const int sw = (frow >> 1) & 7;
    const int a_base = (wm * 128 + frow) * BK;
    const int b_base = TILE_BYTES + (wn * 64 + frow) * BK;

    if (wave >= 8) {
        // ------------------------------ producer wave pw = 0..3: DMA instructions q = pw*16 .. pw*16+15 of each tile
        const int pw = wave - 8;
        auto issue = [&](int kt, int q, char* stage) {
            const int p = (pw * 16 + q) * 64 + lane;   // 0..4095: first 2048 slots = A tile, next 2048 = B tile
            const bool isB = p >= 2048;
            const int pp = p & 2047;
            int row = pp >> 3;
            const int c = (pp & 7) ^ ((row >> 1) & 7);
            if (isB) row = row < nB_valid ? row : nB_valid - 1;
#ifdef OZ2_ABL_SAMETILE
            kt &= 3;  // ablation: a workgroup re-reads its first 4 K-tiles (L2-resident)
#endif
            const int8_t* src = (isB ? gB : gA) + (size_t)row * args.kp + (size_t)kt * BK + c * 16;
            char* dst = stage + ((pw * 16 + q) * 64) * 16;  // wave-uniform; HW adds lane*16 (B tile follows A tile linearly)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst,
                                             16, 0, 0);
        };
#pragma unroll
        for (int q = 0; q < 16; ++q) issue(0, q, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < KT; ++kt) {
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            const bool more = kt + 1 < KT;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                if (sl < PSLOTS && more) {
#pragma unroll
                    for (int q = 0; q < 16 / PSLOTS; ++q) issue(kt + 1, sl * (16 / PSLOTS) + q, nxt);
                }
                if (sl == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ------------------------------ consumer waves
    __builtin_amdgcn_s_barrier();               // tile 0 published by the producers
    if (wm == 1) __builtin_amdgcn_s_barrier();  // trailing half: one segment behind
    for (int kt = 0; kt < KT; ++kt) {
        char* cur = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = (((ks << 1) | khalf) ^ sw) << 4;
            v4i af[4], bf[2];
#ifdef OZ2_ABL_NOLDS
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = v4i{coff, i, kt, lane};
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = v4i{coff, j, kt, lane};
#else
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const v4i*)(cur + a_base + i * 32 * BK + coff);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const v4i*)(cur + b_base + j * 32 * BK + coff);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#ifdef OZ2_ABL_NOMFMA
                    acc[i][j][0] += af[i][0] + bf[j][1];
#else
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
#endif
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();

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
#if OZ2_SCHED == 3
    constexpr int lds = LDS_BYTES;
    constexpr int nthreads = WS_THREADS;
    auto kern = gemm_i8_ws_kernel<EPI>;
#elif OZ2_SCHED == 2
    constexpr int lds = RING_LDS_BYTES;
    constexpr int nthreads = NTHREADS;
    auto kern = gemm_i8_ring_kernel<EPI>;
#else
    constexpr int nthreads = NTHREADS;
    constexpr int lds = LDS_BYTES;
    auto kern = gemm_i8_kernel<EPI>;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = planes * a.tiles_m * a.tiles_n;
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), lds, stream, a);
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
