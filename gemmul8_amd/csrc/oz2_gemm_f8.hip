// FP8 (OCP e4m3) x FP8 -> FP32 "TN" GEMM on gfx950 MFMA with fused epilogues (FP8 backend of Ozaki-II).
//
// Replaces gemm_low_prec_f8x1 / f8x3 (GEMMul8/src/matmult.hpp:180-208,307-350) and the FP8 requantise
// pass (src/conv_hi2mid_real.hpp:28-46, src/mod.hpp:106-130).  All values the main GEMMs multiply are
// integers of magnitude <= 16 held as e4m3 (src/mod.hpp:159-189), so products and FP32 sums are exact
// for k <= 65536; the bound GEMM (values up to 256 with 3-bit mantissas) is inexact and is inflated by
// (k+1)*2^-24 exactly like the reference (src/find_max.hpp:82-96).
//
// Per modulus three GEMMs (src/gemmul8_real.hpp:159-181):
//   square moduli (t < 6, p = s^2, a = s*hi + lo):  C0 = Ahi*Blo, C1 = Alo*Bhi, C2 = Alo*Blo,  value = s*(C0+C1) + C2
//   Karatsuba   (t >= 6,        a = 16*hi + lo):    C0 = Ahi*Bhi, C1 = Alo*Blo, C2 = (Ahi+Alo)(Bhi+Blo),
//                                                   value = 256*C0 + 16*(C2-C0-C1) + C1
//   EPI_PART  : out = int16 residue of one of C0 / C1 (scratch planes)
//   EPI_FINAL : C = C2; combines with the residues of C0, C1 and stores C_mid[t] = int16(value mod p_t)
//   EPI_MAX   : row/col maxima of fma_ru(ku, C, C) as float bit patterns (atomicMax on non-negative floats)
//
// Same tiling as oz2_gemm_i8.hip (256x256 tile, BK = 128 bytes, swizzled 128-B LDS rows fed by LDS-DMA, ping-pong
// LOAD/MFMA segments with the two wave halves one slot apart); the matrix instruction is the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales (E8M0 0x7F) -- the scaled forms are the only full-rate FP8 MFMAs on CDNA4
// (the unscaled fp8 forms run at the BF16 rate), and at the board's power cap the 16x16x128 shape sustains 4.28 POP/s on the
// integers in [-16, 16] this backend multiplies where 32x32x64 holds 3.97 (tools/ubench/mfma_shapes.hip,
// profiles/archive/r02_mfma_shapes.txt).  One instruction consumes a whole 128-byte K-step of 16 rows: lane l holds row l & 15 and the
// 16-byte chunks q and q + 4 (q = l >> 4) of it -- any assignment of K positions works as long as A and B agree, and this one
// keeps the ds_read_b128 pattern of the INT8 kernel (conflict-free with the row-XOR swizzle).  A K-step is two segments (row
// halves) of 16 MFMAs: A fragments of 64 rows (32 registers) per segment, the B fragments of the wave's 64 columns (32
// registers) loaded in the first and kept for the second.  128 accumulators + 64 operand registers do not fit the 168-VGPR budget
// of the 12-wave INT8 layout, so this kernel runs 8 waves (2 per SIMD, 256 VGPRs) and the consumer waves issue the LDS-DMA
// themselves in their LOAD segments (8 instructions per wave and operand panel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "oz2_gemm_common.hpp"
#include "oz2_kernels.h"

#include "oz2_gemm_f8_epi.hpp"

#ifndef OZ2_F8_KBAR
#define OZ2_F8_KBAR 1  // 1 (round 5): one workgroup barrier per K-step with the two wave halves in anti-phase -- residue GEMMs neutral (+-0.6 %), the accurate mode's single-plane bound GEMM +6 % (scaling phase of config 3 5.41 -> 5.23 ms), bit-identical: profiles/r05z_f8_kbar_ab.txt; 0: a barrier after every LOAD / MFMA segment (rounds 2-4)
#endif
namespace oz2 {

// Persistent: one workgroup per CU loops over tiles vb = blockIdx.x, blockIdx.x + gridDim.x, ...; the two-stage K pipeline
// runs straight through tile boundaries (the first K-tile of the next tile is fetched during the last K-step of the current
// one, the epilogue's stores drain behind the next tile's MFMAs, no workgroup launch between tiles).
template <int EPI>
__global__ void __launch_bounds__(F8_THREADS) gemm_f8_kernel(const F8Args args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT1 = args.kp / BK;     // K-steps per segment
    const int KT = KT1 * args.nseg;  // K-steps per tile
    const int total = args.total_tiles;
    const int G = gridDim.x;

    // LDS-DMA: a wave issues 8 instructions per operand panel (1 KiB = 8 rows x 128 B each; rows 64 (wave & 3) .. +63 of the
    // panel; slot layout inside a panel as in oz2_gemm_i8.hip).  Per tile: a wave-uniform base pointer and 8 per-lane byte
    // offsets (B rows clamped to the rows that exist), so the K loop issues global_load_lds with SGPR base + VGPR offset.
    // The leading half (waves 0-3) fetches B, whose panels are needed one K-step after they are issued: it can drain them as late
    // as the end of its last MFMA segment (one barrier before its own first LOAD of the next K-step); the trailing half
    // (waves 4-7) fetches A two K-steps ahead.
    const bool isB = wave < 4;
    unsigned doff[8];
    const int8_t* gsrc;
    long long gdelta = 0;  // nseg == 2: byte offset from a first-segment panel to the second-segment panel of the same rows and K-step
    auto uniform = [](const int8_t* ptr) {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const int8_t*)(((unsigned long long)hi << 32) | lo);
    };
#define F8_SET_TILE(vb_)                                                                                                     \
    do {                                                                                                                     \
        const TileMap tmap_ = map_tile((vb_), total, args.map);                                            \
        const F8Plane pl_ = f8_plane(args, tmap_.plane);                                                                     \
        gsrc = uniform(isB ? args.B + pl_.boff + (size_t)args.planeB[pl_.tt] * args.strideB + (size_t)tmap_.tn * BN * args.kp \
                           : args.A + pl_.boff + (size_t)args.planeA[pl_.tt] * args.strideA + (size_t)tmap_.tm * BM * args.kp); \
        gdelta = isB ? ((long long)args.planeB2[pl_.tt] - args.planeB[pl_.tt]) * (long long)args.strideB - (long long)KT1 * BK       \
                     : ((long long)args.planeA2[pl_.tt] - args.planeA[pl_.tt]) * (long long)args.strideA - (long long)KT1 * BK;      \
        const int nvalid_ = isB ? ((args.n - tmap_.tn * BN) < BN ? (args.n - tmap_.tn * BN) : BN) : BM;                       \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                                      \
            const int pp_ = (((wave & 3) * 8 + q) * 64 + lane);                                                              \
            int row_ = pp_ >> 3;                                                                                             \
            const int c_ = (pp_ & 7) ^ ((row_ >> 1) & 7);                                                                    \
            row_ = row_ < nvalid_ ? row_ : nvalid_ - 1;                                                                      \
            doff[q] = (unsigned)row_ * (unsigned)args.kp + c_ * 16;                                                          \
        }                                                                                                                    \
    } while (0)
#define F8_DMA(src_, q_, stage_)                                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((src_) + doff[q_]),                     \
                                     (__attribute__((address_space(3))) void*)((stage_) + ((wave & 3) * 8 + (q_)) * 1024), 16, 0, 0)

    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int a_base = (wm * 128 + r16) * BK;
    const int b_base = (wn * 64 + r16) * BK;
    const int clo = (q ^ sw) << 4, chi = ((q | 4) ^ sw) << 4;  // this lane's two 16-byte chunks (q, q + 4) of a 128-byte row, swizzled

    auto frag = [&](const char* base) {  // 32 bytes of this lane's row
        const v4i lo = *(const v4i*)(base + clo);
        const v4i hi = *(const v4i*)(base + chi);
        return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    constexpr int UNIT = 0x7F7F7F7F;  // E8M0 scale 2^0 for every 32-element block

    // the whole persistent loop is instantiated twice (A-fetching waves 0-3, B-fetching waves 4-7) so that the fetch schedule is
    // branch-free inside the LOAD segments
    auto run = [&]<bool ISB>() {
        // Measured on config 3 (18 GEMMs, interleaved builds, profiles/archive/r02_f8_ab.txt): this kernel 54.9 ms, the round-1 32x32x64 kernel
        // (four segments of 4 MFMAs) 55.5 ms, a four-segment (row x column halves) form of this one 55.8 ms.  The matrix pipes stay
        // ~25 % idle, and the cost sits in the L2 -> LDS operand path itself: with the DMA issued in the first tile only (real data
        // in LDS; probe build -DOZ2_PROBE=4) the 18 GEMMs take 44.2 instead of 54.2 ms (-18.5 %; the INT8 kernel: -13 %), of which
        // L2-miss latency explains 6 % (round 1, L2-resident operands).  WHERE the DMA is issued from does not matter: a 12-wave
        // variant with dedicated producer waves (the INT8 layout; 32x32x64 so that 128 accumulators + 32 operand registers fit 168
        // VGPRs, address set folded into one register by XOR variants) was built, is bit-exact and runs 58.9 ms, 46.5 without
        // DMA -- tools/experiments/gemm_f8_ws_experiment.inc.
        // 2.5-stage ring of operand panels as in oz2_gemm_i8.hip: panel h = 2 g + isB in slot h % 5 of five 32 KiB panels.  The A
        // waves fetch A(g+2) during K-step g, four instructions in each of their two LOAD segments; the B waves fetch B(g+1), all eight
        // instructions in their first LOAD segment (B must land within the K-step).  Against the two-stage pipeline (32x32x64 kernel of
        // round 1): 63.4 -> 56.2 ms on the 18 FP8 GEMMs of config 3.
        // Fetch state = the panel fetched LAST; every K-step first advances it (at the TOP of the K-step: a branch after the
        // LOAD/MFMA segments lets the compiler sink all MFMAs of the K-step behind it, which destroys the ping-pong) and then
        // issues that panel.  Past the last panel the state stops advancing and the fetch repeats the last valid source into
        // slots nobody will read, so the segments themselves stay branch-free; the wave drains everything before it exits.
        int vb_next = blockIdx.x, kt_next = 0;
        bool more = true;
        int hs = ISB ? 1 : 0;  // slot of the panel to fetch
        const int8_t* fsrc;
        char* fdst;
        F8_SET_TILE(vb_next);
#define F8_FETCH_ADVANCE()                                                                                                   \
    do {                                                                                                                     \
        hs = hs + 2 >= 5 ? hs - 3 : hs + 2;                                                                                  \
        if (more && ++kt_next == KT) {                                                                                       \
            kt_next = 0;                                                                                                     \
            vb_next += G;                                                                                                    \
            more = vb_next < total;                                                                                          \
            if (more) F8_SET_TILE(vb_next);                                                                                  \
            else kt_next = KT - 1;                                                                                           \
        }                                                                                                                    \
    } while (0)
#define F8_FETCH_BEGIN()                                                                                                     \
    do {                                                                                                                     \
        fsrc = gsrc + (size_t)OZ2_HOOK_KSTEP(kt_next) * BK + (kt_next >= KT1 ? gdelta : 0);                                  \
        fdst = smem + hs * TILE_BYTES;                                                                                       \
    } while (0)
        F8_FETCH_BEGIN();
#pragma unroll
        for (int q = 0; q < 8; ++q) F8_DMA(fsrc, q, fdst);
        if constexpr (!ISB) {
            F8_FETCH_ADVANCE();
            F8_FETCH_BEGIN();
#pragma unroll
            for (int q = 0; q < 8; ++q) F8_DMA(fsrc, q, fdst);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (!OZ2_F8_KBAR && wm == 1) __builtin_amdgcn_s_barrier();
        // Hazards: a panel's slot was last read two (A) / one (B) K-steps before the refill is issued, and every wave finishes a
        // K-step's LOAD segments (lgkmcnt(0) + barrier) before the leading half enters slot 0 of the next one; each wave drains the
        // DMA the NEXT K-step needs in its last LOAD segment (A waves: everything but the 8 instructions just issued), one barrier
        // before anyone reads the new panels.
        int sA = 0;  // slot of A(g); B(g) sits in the next slot (mod 5)
        for (int vb = blockIdx.x; vb < total; vb += G) {
            v4f acc[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

            for (int kt = 0; kt < KT; ++kt) {
                const char* curA = smem + sA * TILE_BYTES + a_base;
                const char* curB = smem + (sA == 4 ? 0 : sA + 1) * TILE_BYTES + b_base;
                sA = sA + 2 >= 5 ? sA - 3 : sA + 2;
                F8_FETCH_ADVANCE();
                F8_FETCH_BEGIN();
                v8i bf[4];
#pragma unroll
                for (int ah = 0; ah < 2; ++ah) {  // LOAD segment ah of this K-step
                    v8i af[4];
                    if (OZ2_HOOK_DMA_ON(vb == (int)blockIdx.x)) {  // (laboratory hook: always true in the product)
                        if constexpr (ISB) {
                            if (ah == 0) {
#pragma unroll
                                for (int q8 = 0; q8 < 8; ++q8) F8_DMA(fsrc, q8, fdst);
                            }
                        } else {
#pragma unroll
                            for (int q8 = 0; q8 < 4; ++q8) F8_DMA(fsrc, ah * 4 + q8, fdst);
                        }
                    }
                    if (ah == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bf[j] = frag(curB + j * 16 * BK);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = frag(curA + (ah * 4 + i) * 16 * BK);
                    // As in oz2_gemm_i8.hip: only the K-step's LAST load segment completes its LDS reads (and, on the A waves, the DMA the next K-step
                    // needs) before the barrier -- the one the ring's hazards count on; the first segment arrives at the barrier with its reads issued
                    // and waits behind it (config 3 whole call +1.25 %, SGEMM 8192^2 x 2048 / 8192 +1.2 / +0.7 %: profiles/archive/r04e_f8_late_wait_ab.txt)
#if OZ2_F8_KBAR
                    // K-step-barrier schedule (as oz2_gemm_i8.hip below k = 5120): ONE workgroup barrier per K-step.  The A-fetching half (wm = 0) runs
                    // L0 M0 L1 M1 | barrier, the B-fetching half (wm = 1) L0 M0 L1 | barrier | M1 -- its last fragments cross the barrier in registers --
                    // so on every SIMD one wave's LDS reads sit beside the other's MFMAs without a barrier per segment.  RAW: every wave's pieces of
                    // the panels of K-step g + 1 have landed (vmcnt) before it arrives; WAR: every read of K-step g is complete (lgkmcnt(0) / consumed
                    // by issued MFMAs) before it arrives, the refills of those slots are issued behind the barrier.
                    if (ISB && ah == 1) {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#else
                    if (ah == 1) {
                        if (!ISB) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (ah == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#endif
                    // serpentine order over the 4 x 4 fragment pairs, as in oz2_gemm_i8.hip (consecutive MFMAs share an operand register across the row change)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = (i & 1) ? 3 - jj : jj;
                            acc[ah * 4 + i][j] =
                                __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bf[j], acc[ah * 4 + i][j], 0, 0, 0, UNIT, 0, UNIT);
                        }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
#if OZ2_F8_KBAR
                    if (!ISB && ah == 1) {
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
#else
                    if (ah == 1 && ISB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
            const TileMap tmap = map_tile(vb, total, args.map);
            const int i0 = tmap.tm * BM + wm * 128, j0 = tmap.tn * BN + wn * 64;
            const F8Plane pl = f8_plane(args, tmap.plane);
            if constexpr (EPI == EPI_PART || EPI == EPI_FINAL || EPI == EPI_FINAL_CPLX) {
                f8_epilogue_mod<EPI>(acc, args, pl, i0, j0, lane);
            } else {
                f8_epilogue_bound<EPI>(acc, args, pl, i0, j0, lane);
            }
        }
    };
    if (isB) run.template operator()<true>();
    else run.template operator()<false>();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
    if (!OZ2_F8_KBAR && wm == 0) __builtin_amdgcn_s_barrier();
#undef F8_SET_TILE
#undef F8_DMA
#undef F8_FETCH_BEGIN
#undef F8_FETCH_ADVANCE
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 256;
        n = prop.multiProcessorCount;
    }
    return n;
}

template <int EPI> static hipError_t launch(hipStream_t stream, F8Args& a, int planes) {
    // the attribute belongs to the function on ONE device; setting it is idempotent, so concurrent first calls from several host
    // threads only need the flag itself to be race-free
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 0;
    if (!attr_set_dev[dev_].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_f8_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * TILE_BYTES);
        if (e != hipSuccess) return e;
        attr_set_dev[dev_].store(true, std::memory_order_release);
    }
    // the items of a batched call fold into the plane sequence: plane p = item p / ppi (oz2_gemm_i8.hip does the same)
    a.ppi = planes;
    a.m_ppi = map_magic((unsigned)planes);
    a.bstride = g_batch.ws;
    planes *= (int)g_batch.batch;
    a.total_tiles = planes * a.tiles_m * a.tiles_n;
    if (a.total_tiles <= 0) return hipSuccess;
    if (a.nseg < 1) a.nseg = 1;
    if (a.nres < 1) a.nres = 2;
    a.colblock = map_colblock((size_t)a.tiles_n, (size_t)a.kp * (size_t)a.nseg);
    a.map = make_tile_map(a.tiles_m, a.tiles_n, a.colblock);
    int grid = num_cus() & ~7;  // persistent: one workgroup per CU (see oz2_gemm_i8.hip)
    if (grid <= 0) grid = 8;
    if (a.total_tiles < grid) grid = a.total_tiles;
    hipLaunchKernelGGL(gemm_f8_kernel<EPI>, dim3(grid), dim3(F8_THREADS), 5 * TILE_BYTES, stream, a);
    return hipGetLastError();
}

// which = 0,1: partial products C0 / C1 -> int16 residue scratch; which = 2: C2 with the final combine -> out (C_mid plane, or
// a scratch plane holding the residue of a complex part); which = 3: C2 of the third complex part with the complex combine
// (rx, ry = residues of X and Y, same stride as r0/r1) -> interleaved (Cr, Ci) int16 pairs in out; 4 / 5 / 6: the K-concatenated
// forms (f8_fill_planes, oz2_gemm_f8_epi.hpp).
hipError_t launch_gemm_f8(hipStream_t stream, int which, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                          size_t n, int t_begin, int t_end, int16_t* out, size_t ldo, size_t strideO, const int16_t* r0, const int16_t* r1,
                          size_t strideR, const int16_t* rx, const int16_t* ry) {
    F8Args a{};
    if (f8_fill_planes(a, which, A, B, strideA, strideB, kp, m, n, t_begin, t_end, out, ldo, strideO, r0, r1, strideR, rx, ry) != 0) return hipErrorInvalidValue;
    const int planes = t_end - t_begin;
    if (which == 3 || which == 6) return launch<EPI_FINAL_CPLX>(stream, a, planes);
    return (which == 2 || which == 5) ? launch<EPI_FINAL>(stream, a, planes) : launch<EPI_PART>(stream, a, planes);
}

// Inflation of the accurate-mode bound sums.  The reference uses ku = (k+1) * 2^-24, a bound on IEEE FP32 summation
// (find_max.hpp:82-96).  v_mfma_scale_f32_16x16x128_f8f6f4 does not sum like that (tools/ubench/f8_accum.hip,
// profiles/archive/r02_f8_mfma_accumulation.txt): inside a group of 8 products everything is aligned to the group's largest product and
// bits below 2^-13 of it are TRUNCATED -- up to 7 * 2^-13 of a non-negative group sum is lost -- and the group sums / accumulator
// are added with ~21-22 bits below the largest addend, again truncating (<= 16 * 2^-20 per instruction, k/128 instructions:
// <= 2 (k+1) * 2^-24 overall).  A bound that comes out LOW can push the shift up by one and break accurate mode's no-wrap guarantee
// |A'B'| < P/2, so the default here covers the engine: ku = 7 * 2^-13 + 4 (k+1) * 2^-24 (a factor 2 of margin on the accumulator
// term; costs < 0.015 bit of shift at k = 65536).  Mode 1 restores the reference's formula (gemmul8_set_fp8_bound_mode).
// Complex types (round 4): the bound of |Re C| is T = C0 + C1 with C0 = sum (|Ar|-|Ai|)(|Br|-|Bi|) -- terms of both signs -- and
// C1 = sum |Ar||Bi| + |Ai||Br|.  With eps the engine's relative loss on a sum of magnitudes, computed C0 >= C0 - eps (T + C1) (the
// magnitudes of C0's terms sum to at most T + C1), so T <= (C0_computed + (1 + eps) C1) / (1 - eps): the inflation of C0 must scale with
// |C0| + 2 C1, not with C0 -- where the large terms of C0 cancel (T ~ C1) the reference's form fma_ru(ku, C0, C0) + s12 leaves T low by up
// to 2 eps - ku (tests/test_gpu_fp8_bound.py::test_fp8_bound_adversarial_complex builds such matrices: with mode 2 every element of C
// comes back wrong).  Mode 0 uses s0 = fma_ru(ku, |C0| + 2 s12, C0) + s12 (all rounded up); mode 2 = mode 0's ku with the reference's
// combination (the round-3 default, kept for that test); mode 1 = the reference throughout.
static std::atomic<int> g_f8_bound_mode{0};
void set_f8_bound_mode(int mode) { g_f8_bound_mode.store(mode == 1 || mode == 2 ? mode : 0); }
int get_f8_bound_mode() { return g_f8_bound_mode.load(); }
// Absolute part (round 4, mode 0 only).  A 12000-seed fuzz sweep found real-type bound sums 4.7e-5 / 2.2e-6 BELOW the exact sum with the
// relative inflation alone (sums below 1 in bound-plane units: one column, wide exponent range); tools/ubench/f8_accum2.hip shows why: the
// alignment exponent of a group of 8 products is the largest SUM OF THE OPERANDS' EXPONENT FIELDS, and an e4m3 subnormal (or zero) carries
// the field of 2^-6 whatever its value -- a product 2^-9 x 2^7 is aligned as if it were 2^1: beside it only 10 bits below the TRUE largest
// product survive (13 when both operands of the reference product are normal).  A subnormal operand is < 2^-6 and its partner < 2^9, so such a
// reference exponent is at most 2: grid <= 2^-11, at most 7 products of a group lose less than that each -- an ABSOLUTE loss of at most
// 7 * 2^-11 per group of 8, 7 kp 2^-14 per sum, independent of the sum's size.  kabs adds exactly that (negligible against any sum that
// matters: a bound-plane row / column has its maximum in [2^7, 2^8); it only moves the shift of rows whose products nearly all vanish).
static float bound_kabs(size_t kp) { return g_f8_bound_mode.load() == 0 ? 7.0f * (float)kp * 0x1.0p-14f : 0.0f; }
static float bound_ku(size_t k) {
    const float ieee = (float)(k + 1) * 0x1.0p-24f;
    return g_f8_bound_mode.load() == 1 ? ieee : 0x1.cp-11f + 4.0f * ieee;
}

hipError_t launch_gemm_f8_max(hipStream_t stream, const int8_t* A, const int8_t* B, size_t kp, size_t k, size_t m, size_t n, int* rowmax,
                              int* colmax) {
    F8Args a{};
    a.A = A;
    a.B = B;
    a.rowmax = rowmax;
    a.colmax = colmax;
    a.ku = bound_ku(k);
    a.kabs = bound_kabs(kp);
    f8_fill_common(a, kp, m, n);
    return launch<EPI_FMAX>(stream, a, 1);
}

// complex accurate-mode bound, stage 1..3 (see EPI_FB*): fbuf = float scratch [n][ldf]
hipError_t launch_gemm_f8_bound_cplx(hipStream_t stream, int stage, const int8_t* A, const int8_t* B, size_t kp, size_t k, size_t m, size_t n,
                                     float* fbuf, size_t ldf, int* rowmax, int* colmax) {
    F8Args a{};
    a.A = A;
    a.B = B;
    a.rowmax = rowmax;
    a.colmax = colmax;
    a.fbuf = fbuf;
    a.ldo = ldf;
    a.ku = bound_ku(k);
    a.kabs = bound_kabs(kp);
    a.cplx_rule = g_f8_bound_mode.load() == 0 ? 1 : 0;
    f8_fill_common(a, kp, m, n);
    if (stage == 1) return launch<EPI_FB1>(stream, a, 1);
    if (stage == 2) return launch<EPI_FB2>(stream, a, 1);
    return launch<EPI_FB3>(stream, a, 1);
}

}  // namespace oz2
