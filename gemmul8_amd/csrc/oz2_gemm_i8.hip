// INT8 x INT8 -> INT32 "TN" GEMM on gfx950 MFMA with fused epilogues (Ozaki-II hot loop).
//
// Replaces the vendor-BLAS call sites of the reference (GEMMul8/src/matmult.hpp:120-175 i8x1,
// :213-302 i8x3) AND the separate requantise pass (src/conv_hi2mid_real.hpp:9-25,
// src/conv_hi2mid_complex.hpp:9-127) / the bound-matrix max passes
// (src/scaling_accu_real.hpp:142-226, src/scaling_accu_complex.hpp:132-224): the INT32 accumulators
// never leave registers.
//
//   C(i,j) = sum over K-segments s, kk:  A_s(i,kk) * B_s(j,kk)        operands K-contiguous ("TN")
//   EPI_MOD  : out(i,j)  = int8( symmetric residue of C mod p_t )                      (real C_mid, complex X/Y partials)
//   EPI_CPLX : C = Z = (Ar+Ai)(Br+Bi); reads the residues rx, ry of X = ArBr, Y = AiBi written by two
//              EPI_MOD launches and stores interleaved (Cr, Ci) = (X-Y, Z-X-Y) mod p_t   (conv_hi2mid_complex.hpp:9-26)
//   EPI_MAX  : rowmax[i] = max_j C(i,j), colmax[j] = max_i C(i,j) (atomicMax)           (accurate-mode bound GEMM)
// Up to 3 K-segments are concatenated (virtual K = nseg*kp): the complex bound matrices
// ArBi+AiBr and ArBi+AiBr+(Ar-Ai)(Br-Bi) are single GEMMs this way.
//
// Kernel structure (CDNA4), see DESIGN.md 3.1 for the measurements behind each choice:
//  * PERSISTENT: one workgroup per CU loops over 256x256 output tiles.  12 waves: 8 CONSUMER waves (2(M) x 4(N), wave tile
//    128x64 = 8x4 v_mfma_i32_16x16x64_i8 accumulator tiles = 128 registers/lane) + 4 PRODUCER waves (one per SIMD) that only
//    issue the LDS-DMA (global_load_lds_dwordx4, SGPR base + one VGPR offset per instruction), so the matrix pipe's feeders
//    never block on the VMEM queue and spend no VALU cycles on addresses.  168 VGPRs -> 3 waves/SIMD.
//  * BK = 128 bytes (every DMA row segment is a full 128-B line: the global->LDS path does 127 GB/s/CU with 128-B segments,
//    68 with 64-B ones).  LDS = a 2.5-stage ring of five 32 KiB operand panels: panel h = 2g + isB of K-step g in slot h % 5;
//    during K-step g the A producers fetch A(g+2), 2 instructions in every slot, the B producers B(g+1), 4 instructions in
//    slots 0-3 (smooth issue matters as much as depth: see the producer branch).
//  * LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the DMA SOURCE address
//    (the destination must stay lane-linear) and on the ds_read_b128 address: every 16-lane read group hits
//    16 distinct 16-B bank slots (SQ_LDS_BANK_CONFLICT = 0).
//  * Ping-pong schedule (k > 4096; K-step-barrier schedule below that, see launch<EPI>): each consumer alternates a LOAD segment
//    (4-8 ds_read_b128) and an MFMA segment (16 MFMAs, s_setprio 1), four of each per K-step, one s_barrier after each; the wm=1 half runs one segment behind the wm=0 half
//    so on every SIMD one wave feeds the matrix pipe while its partner reads LDS.  With g counting K-steps across tiles:
//      slot(wm=0: L_j(g)) = 8g+2j, M_j -> +1 (j = 0..3); wm=1 one slot later; the producers drain what K-step g+1 needs in slot 8g+7.
//      WAR: the slots refilled during K-step g held panels of K-steps g-1 (-> B(g+1)) and g-2/g-1 (-> A(g+2)), last read in
//           wm=1's L3(g-1) (slot 8g-1) < the first DMA of K-step g (slot 8g).
//      RAW: the producers' vmcnt wait + barrier closes slot 8g+7; wm=0's L0(g+1) opens slot 8g+8.
//  * Workgroup -> tile mapping is XCD-aware and chunked (oz2_gemm_common.hpp): the 32 CUs of an XCD share 8+4 operand
//    panels through their L2 (measured TCC hit rate 81 %) and all XCDs work on one plane, so misses land in the Infinity Cache.
//  * Cost split measured with real-data probes (tools/experiments/probes; 16x16x64 kernel, k = 8192): the board's
//    power-limited ceiling for this instruction on residue data is 3.97 POP/s, MFMA + barriers alone reach 93 % of it; the
//    L2 -> LDS operand path costs 13 %, the per-segment barriers 7 %, the epilogue 6 %, the LDS reads 5 %.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "oz2_crt_common.hpp"
#include "oz2_gemm_common.hpp"
#include "oz2_gemm_i8_epi.hpp"
#include "oz2_kernels.h"

namespace oz2 {

#ifndef OZ2_PB
#define OZ2_PB 4
#endif
#ifndef OZ2_KBAR_MAX_KP
#define OZ2_KBAR_MAX_KP 5120  // padded k up to which the K-step-barrier schedule is used (see launch<EPI>); 0 = never, 1 << 30 = always.  Round 4 (after the epilogue / tile-prologue work): +1.0 / +1.3 % at k = 4608 / 5120 on 8192^2 x 14 planes, +0.5 % at 5120 on 16384^2 x 6; at 6144 +0.9 % / -1.1 %, at 7168 0 / -1.3 %, at 8192 -1.3 % (profiles/archive/r04_gemm_ab_kbar_threshold.txt)
#endif
#ifndef OZ2_KBAR_PEEL_FIRST
#define OZ2_KBAR_PEEL_FIRST 0
#endif
#ifndef OZ2_MOD256
#define OZ2_MOD256 0  // 1: K <= 256 launches carry their accumulators as float patterns (EPI_MOD256 / RED_MAGIC: two instead of three instructions per accumulator in the
                      // residue epilogue).  Built and bit-identical in round 6, measured NEUTRAL (8192^2 x 128 / 256: -1 / -2 %, 16384^2 x 256: +1 %,
                      // profiles/r06_short_k_epilogue_ab.txt): the instruction count is not what bounds that epilogue.  Not instantiated in the shipped library.
#endif
#ifndef OZ2_SLEEP_A
#define OZ2_SLEEP_A 4  // s_sleep units (64 clocks) between the A producers' 8 groups of 2 LDS-DMA instructions
#endif
#ifndef OZ2_SLEEP_B
#define OZ2_SLEEP_B 4  // ... between the B producers' 4 groups of 4
#endif
constexpr int RING_LDS_BYTES = 5 * TILE_BYTES;  // five 32 KiB operand panels = the whole 160 KiB of LDS

// Persistent, wave-specialised kernel: one workgroup per CU (all 160 KiB of LDS) loops over tiles vb = blockIdx.x,
// blockIdx.x + gridDim.x, ...; 8 consumer waves (2 x 4, each 128 x 64 of the 256 x 256 tile) run MFMA + epilogue, 4
// producer waves only issue LDS-DMA into a 2.5-stage ring of operand panels (see the producer branch): A panels are fetched
// two K-steps ahead with their 16 instructions per wave spread evenly over the K-step, B panels one K-step ahead -- smooth
// issue matters as much as depth (all 16 in one slot: +7 % kernel time; the ring with even issue: -6..9 % against the
// two-stage pipeline at every k).  The K pipeline runs straight through tile boundaries: while the consumers
// are in the last K-step and the epilogue of a tile the producers already fetch the first K-tile of the next one, the
// epilogue's stores drain behind the next tile's MFMAs, and there is no workgroup launch / LDS re-allocation between
// tiles -- a non-persistent version of this kernel lost ~11 us of a ~120 us tile to those three.
// FUSE: always 0 in libgemmul8.so.  The laboratory build of tools/experiments/fused_crt instantiates FUSE = 1 | 2 (double | float output): the
// TILE-STATIONARY order -- a workgroup runs all args.planes residue planes of one output tile back to back and accumulates the CRT of the
// tile inside the kernel (bit-exact, measured 10-28 % slower than the two-launch path: DESIGN.md 3.4).
struct NoCrt {
    int unused;
};
#ifdef OZ2_LAB_FUSED_CRT
#include OZ2_LAB_FUSED_CRT  // i8_crt_tail, i8_producer_crt
#endif
// Re-initialisation of the accumulators by the (idle) matrix pipe, sub-block by sub-block BEHIND the epilogue's reads (round 6): as soon as the residue
// epilogue has reduced the four accumulator tiles of a sub-block (hook phase 0), four MFMAs on all-zero fragments with C = the start value rewrite them --
// 32 matrix instructions per wave and tile on a pipe that has nothing else to do, instead of 128 v_mov_b32 at the head of the next tile on the vector
// ALU that the two waves' epilogues (~650 instructions each) are bound by.
#ifndef OZ2_KBAR_MFMA_REINIT
#define OZ2_KBAR_MFMA_REINIT 1
#endif
template <bool ZERO> struct MfmaReinitHook {  // ZERO: the start value is 0 (short-K launches): the C operand is the inline constant
    v4i (*acc)[4];
    int acc0;
    __device__ __forceinline__ void operator()(int sb, int phase) const {
        if (phase != 0) return;
        const int tj = sb >> 1, tg = sb & 1;
        const v4i z = {0, 0, 0, 0};
        const v4i c = ZERO ? v4i{0, 0, 0, 0} : v4i{acc0, acc0, acc0, acc0};
        __builtin_amdgcn_sched_barrier(0);  // not above the reduction's last read of these registers (a hoisted definition needs a second register set)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)  // B = the dead accumulator tile itself (times an all-zero A): four DISTINCT instructions -- on identical operands the
            acc[tg * 4 + ti][tj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(z, acc[tg * 4 + ti][tj], c, 0, 0, 0);  // compiler keeps one and copies its result three times
    }
};
// SMALLK: the launch has K <= 512 (accumulators start at 0, three-instruction residue); the epilogue form is a compile-time property of the kernel
template <int EPI, bool KBAR, int FUSE, bool SMALLK = false>
__global__ void __launch_bounds__(WS_THREADS) gemm_i8_kernel(const GemmArgs args, const std::conditional_t<FUSE != 0, CrtArgs, NoCrt> crt) {
#ifndef OZ2_LAB_FUSED_CRT
    static_assert(FUSE == 0, "the in-kernel CRT forms are laboratory code: tools/experiments/fused_crt");
#endif
    static_assert(FUSE == 0 || EPI == EPI_MOD, "the CRT tail follows the real requantise epilogue");
    static_assert(EPI != EPI_MOD256 || (KBAR && SMALLK && FUSE == 0), "the K <= 256 form is an instantiation of the short-K K-step-barrier kernel");
    static_assert(offsetof(CrtArgs, Cmid) == 0 && alignof(CrtArgs) == 8, "i8_crt_tail locates the block in the kernel-argument segment");
    (void)crt;
    const int planes_per_tile = FUSE ? args.planes : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT1 = args.kp / BK;    // K-steps per segment
    const int KT = KT1 * args.nseg;  // K-steps per tile
    const int total = args.total_tiles;
    const int G = gridDim.x;

    if (wave >= 8) {  // ------------------------------ producer waves: LDS-DMA only
#ifdef OZ2_LAB_FUSED_CRT
        if constexpr (KBAR && FUSE != 0) {
            i8_producer_crt<FUSE>((__attribute__((address_space(3))) char*)smem,  // CRT on the producer waves: out of line (see there)
                                  (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr());
            return;
        }
#endif
#include "oz2_gemm_i8_producer.inc"
        if constexpr (KBAR) {
        // ONE workgroup barrier per K-step (see the consumer branch).  Without per-segment barriers to pace them the producers space
        // their instructions with s_sleep (64 clocks per unit): bursts of LDS-DMA cost (all 16 at once: +7 % kernel time).
        for (int vb = blockIdx.x; vb < total; vb += G) {
            for (int kt = 0; kt < KT * planes_per_tile; ++kt) {
                const bool issued = more && OZ2_HOOK_DMA_ON(vb == (int)blockIdx.x);
                if (issued) {
                    PRODUCER_BEGIN();
                    if (isB) {  // needed next K-step: 4 groups of 4 in the first part of the K-step
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) PRODUCER_DMA(fsrc, gq * 4 + q, fdst);
                            __builtin_amdgcn_s_sleep(OZ2_SLEEP_B);
                        }
                    } else {    // needed in two K-steps: 8 groups of 2 over the whole K-step
#pragma unroll
                        for (int gq = 0; gq < 8; ++gq) {
#pragma unroll
                            for (int q = 0; q < 2; ++q) PRODUCER_DMA(fsrc, gq * 2 + q, fdst);
                            __builtin_amdgcn_s_sleep(OZ2_SLEEP_A);
                        }
                    }
                    PRODUCER_ADVANCE();
                }
                // the panel needed NEXT K-step must have landed: for the A producers everything except the 16 instructions just issued
                if (!isB && issued) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        } else {
        for (int vb = blockIdx.x; vb < total; vb += G) {
            for (int kt = 0; kt < KT * planes_per_tile; ++kt) {
                // A producers: A(g+2); B producers: B(g+1); afterwards the panel needed NEXT K-step must have landed, which
                // for the A producers means everything except the 16 instructions just issued
                const bool issued = more && OZ2_HOOK_DMA_ON(vb == (int)blockIdx.x);
                if (issued) PRODUCER_BEGIN();
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    // B (needed next K-step): 4 instructions in each of slots 0-3; A (needed in two K-steps): 2 in every slot.
                    // Bursts cost: all 16 in slot 0 is 7 % slower than the spread issue.
                    if (issued) {
                        if (isB) {
                            constexpr int PB = OZ2_PB;  // slots over which B's 16 instructions are spread
#pragma unroll
                            for (int q = 0; q < 16; ++q)
                                if (q * PB / 16 == sl) PRODUCER_DMA(fsrc, q, fdst);
                        } else {
#pragma unroll
                            for (int q = 0; q < 2; ++q) PRODUCER_DMA(fsrc, sl * 2 + q, fdst);
                        }
                    }
                    if (sl == 7 && issued) PRODUCER_ADVANCE();
                    if (sl == 7) {
                        if (!isB && issued) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        }
#include "oz2_gemm_i8_producer_undef.inc"
        return;
    }

    // ------------------------------ consumer waves
    // Matrix instruction: v_mfma_i32_16x16x64_i8 (A / B operand: lane l = row l & 15, K bytes 16 (l >> 4) .. + 15 of a 64-byte K
    // slice; result: column l & 15, rows 4 (l >> 4) + r).  At the board's power cap it sustains 3.95-3.98 POP/s on uniformly
    // distributed residues where v_mfma_i32_32x32x32_i8 holds 3.45 (tools/ubench/mfma_shapes.hip, profiles/archive/r02_mfma_shapes.txt):
    // the same MACs with a quarter of the accumulator registers read and written per instruction.  Wave tile 128 x 64 = 8 x 4
    // accumulator tiles (128 registers); a K-step (128 bytes) is four segments (K half ks2) x (row half ah) of 16 MFMAs: the B
    // fragments of a K half are loaded in its first segment and kept for the second, the A fragments of 64 rows per segment.
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int a_base = (wm * 128 + r16) * BK;
    const int b_base = (wn * 64 + r16) * BK;

    if constexpr (KBAR) {
    // K-step-barrier schedule: ONE workgroup barrier per K-step -- the only one the LDS ring needs (RAW: the producers' vmcnt wait for
    // panel g + 1 precedes it; WAR: every read of K-step g precedes it, every refill of those slots follows it).  The two waves of a
    // SIMD stay in anti-phase by construction instead of by per-segment barriers: the wm = 0 half runs L0 M0 L1 M1 L2 M2 L3 M3 inside
    // a K-step, the wm = 1 half M3' L0 M0 L1 M1 L2 M2 L3 (M3' = the last MFMA segment of the PREVIOUS K-step, whose fragments it
    // keeps in registers across the barrier), so one wave's LDS reads always sit beside the other's MFMAs, and when both have MFMAs
    // ready they simply share the pipe.  The per-segment barriers of the ping-pong version cost ~7 % (every hand-over idles the
    // matrix pipe for the barrier latency: mfma-only probe 3.70 vs 3.97 POP/s free-running).
    __builtin_amdgcn_s_barrier();  // K-tile 0 published by the producers
    // the whole persistent loop is instantiated once per half (WM1 = lagging half) so that each gets its own register allocation
    auto run = [&]<bool WM1>() {
        int sA = 0;  // slot of A(g); B(g) sits in the next slot (mod 5)
        // MFMA_REINIT: the accumulators live across the tile loop; the v_mov initialisation runs once, every later tile finds them rewritten by
        // the matrix pipe behind the previous epilogue (MfmaReinitHook)
        constexpr bool MFMA_REINIT = OZ2_KBAR_MFMA_REINIT && SMALLK && (EPI == EPI_MOD || EPI == EPI_MOD256) && FUSE == 0;  // SMALLK only: with a register C operand (start value -2^31) the form measured -0.6 ... -1.6 % at k = 1024 / 2048 (profiles/r06_short_k_epilogue_ab.txt)
        v4i acc[8][4];
        if constexpr (MFMA_REINIT) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = v4i{args.acc0, args.acc0, args.acc0, args.acc0};
        }
        for (int vb = blockIdx.x; vb < total; vb += G) {
            const TileMap tmap = map_tile(vb, total, args.map);
            const PlaneRef pref = (FUSE || EPI == EPI_MAX) ? PlaneRef{0, 0} : plane_ref(args, tmap.plane);  // (the bound GEMM looks its plane up at the epilogue: one value less across its K loop)
            const PlaneConsts pcon = (FUSE || EPI == EPI_MAX) ? PlaneConsts{} : plane_consts(args, pref);  // fetched here: the latency passes behind the K loop
            for (int pl = 0; pl < planes_per_tile; ++pl) {
            v4i af[4], bf[4];
            // PEEL_FIRST (round 6, NOT enabled: -DOZ2_KBAR_PEEL_FIRST=1): SMALLK (k <= 512: accumulators start at 0) without the 128 v_mov_b32 per wave and
            // tile -- the first K-step's MFMAs take the inline constant 0 as their C operand.  The compiler then moves seven accumulator quads through
            // scratch INSIDE the peeled K-step (140 bytes; the peeled copy's results and the loop's registers do not coalesce), as with round 4's attempt.
            constexpr bool PEEL_FIRST = OZ2_KBAR_PEEL_FIRST && SMALLK && EPI == EPI_MOD && FUSE == 0;
            if constexpr (!PEEL_FIRST && !MFMA_REINIT) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = v4i{EPI == EPI_MAX ? 0 : args.acc0, EPI == EPI_MAX ? 0 : args.acc0, EPI == EPI_MAX ? 0 : args.acc0, EPI == EPI_MAX ? 0 : args.acc0};
            }
// The fragment reads of a segment are issued in the order the MFMAs use them (pinned: the scheduler otherwise issues the first-used fragment last) and the
// waits in front of the MFMAs are the compiler's own per-fragment lgkmcnt(3..0): the first four MFMAs of a segment start when THEIR fragment has landed.  No
// barrier covers the read latency in this schedule (in the ping-pong schedule one does: there the same change is neutral).  8192^2 x 14 planes, interleaved,
// 25 rounds: k = 256 / 512 / 1024: +0.4 / +0.8 / +1.5 %, neutral from 2048 (profiles/r04f_kbar_partial_wait_ab.txt)
#define OZ2_KBAR_PIN() __builtin_amdgcn_sched_barrier(0)
#define OZ2_KBAR_WAIT() do {} while (0)
#define OZ2_LOAD_SEG(seg_)                                                                                                   \
    do {                                                                                                                     \
        const int coff_ = (((((seg_) >> 1) << 2) | q) ^ sw) << 4;                                                            \
        if (((seg_) & 1) == 0) {                                                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[j] = *(const v4i*)(curB + j * 16 * BK + coff_);                 \
        }                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { af[i] = *(const v4i*)(curA + (((seg_) & 1) * 4 + i) * 16 * BK + coff_); OZ2_KBAR_PIN(); } \
    } while (0)
#define OZ2_MMA_SEG(seg_, first_)                                                                                            \
    do {                                                                                                                     \
        OZ2_KBAR_WAIT();                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                     \
            const int j = (i & 1) ? 3 - jj : jj; /* serpentine: see the ping-pong branch */                                   \
            if ((first_) && (seg_) < 2) /* the K-step's first touch of these accumulators: C = 0 */                          \
                acc[((seg_) & 1) * 4 + i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], v4i{0, 0, 0, 0}, 0, 0, 0); \
            else                                                                                                             \
                acc[((seg_) & 1) * 4 + i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[((seg_) & 1) * 4 + i][j], 0, 0, 0); \
        }                                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    } while (0)
#define OZ2_SET_PANELS()                                                                                                     \
    const char* curA = smem + sA * TILE_BYTES + a_base;                                                                      \
    const char* curB = smem + (sA == 4 ? 0 : sA + 1) * TILE_BYTES + b_base;                                                  \
    sA = sA + 2 >= 5 ? sA - 3 : sA + 2
            // EPI_MAX with kt_mid: two phases per tile (K-steps [0, kt_mid) and [kt_mid, KT)), the maxima epilogue after each; the
            // accumulators run through.  Every other instantiation: one phase.
            int kt = 0;
            const int nph = (EPI == EPI_MAX && args.kt_mid > 0) ? 2 : 1;
            for (int ph = 0; ph < nph; ++ph) {
            const int kt_end = (EPI == EPI_MAX && ph + 1 < nph) ? args.kt_mid : KT;
            auto kstep = [&]<bool FIRST>() {
                OZ2_SET_PANELS();
#pragma unroll
                for (int seg = 0; seg < 4; ++seg) {
                    OZ2_LOAD_SEG(seg);
                    // the lagging half crosses the K-step barrier between its last LOAD and its last MFMA segment (all its LDS reads of the
                    // K-step precede the barrier; the fragments cross it in registers): same instruction stream, shifted by one segment
                    if (WM1 && seg == 3) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    OZ2_MMA_SEG(seg, FIRST);
                }
                if (!WM1) {
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (PEEL_FIRST) {  // (one phase, kt == 0 here, KT >= 2)
                kstep.template operator()<true>();
                ++kt;
            }
            for (; kt < kt_end; ++kt) kstep.template operator()<false>();
#undef OZ2_SET_PANELS
#undef OZ2_LOAD_SEG
#undef OZ2_MMA_SEG
#undef OZ2_KBAR_PIN
#undef OZ2_KBAR_WAIT
#if OZ2_HOOK_SKIP_EPILOGUE
            (void)tmap;  // laboratory probe: no epilogue; the accumulators stay live
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
#ifndef OZ2_KBAR_EPI_LANE_LIVE
#define OZ2_KBAR_EPI_LANE_LIVE 0
#endif
            int lane_e = lane;
            if constexpr (!OZ2_KBAR_EPI_LANE_LIVE && (EPI == EPI_MOD || EPI == EPI_MOD256)) {  // (the complex combine keeps the live lane id: recomputed there, 116 bytes of accumulator spills appear in its K loop)
            // The lane id of the epilogue is RECOMPUTED here (two v_mbcnt behind an opaque asm, so that it is not hoisted): kept live across the K loop it was
            // spilled (12 bytes of scratch), and the reload's vmcnt(0) made every tile wait for the PREVIOUS tile's residue stores to be acknowledged --
            // free behind a long K loop, a stall of the order of the store latency at k <= 1024 where the K loop is 2-8 us (round 6).  The consumer waves
            // issue no other VMEM loads, so no vmcnt wait is left on their path at all.
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
            if constexpr (FUSE != 0) i8_epilogue<EPI, NoHook, -1>(acc, args, PlaneRef{0, pl}, tmap.tm * BM + (WM1 ? 128 : 0), tmap.tn * BN + wn * 64, lane_e);
            else if constexpr (EPI == EPI_MAX) i8_epilogue<EPI, NoHook, 0>(acc, args, plane_ref(args, tmap.plane), PlaneConsts{}, tmap.tm * BM + (WM1 ? 128 : 0), tmap.tn * BN + wn * 64, lane_e);
            else if constexpr (MFMA_REINIT) i8_epilogue<EPI, MfmaReinitHook<EPI != EPI_MOD256>, (int)SMALLK>(acc, args, pref, pcon, tmap.tm * BM + (WM1 ? 128 : 0), tmap.tn * BN + wn * 64, lane_e, MfmaReinitHook<EPI != EPI_MOD256>{acc, args.acc0});
            else i8_epilogue<EPI, NoHook, (int)SMALLK>(acc, args, pref, pcon, tmap.tm * BM + (WM1 ? 128 : 0), tmap.tn * BN + wn * 64, lane_e);
#endif
            }  // phase
            }
            // FUSE: the producer waves accumulate the CRT of this tile during the next one; they read the residue planes one K-step
            // after the barrier that follows this wait
            if constexpr (FUSE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (FUSE != 0) __builtin_amdgcn_s_barrier();  // the last tile's residues are complete: the producers drain
    };
    if (wm == 0) run.template operator()<false>();
    else run.template operator()<true>();
    } else {
    __builtin_amdgcn_s_barrier();               // K-tile 0 published by the producers
    if (wm == 1) __builtin_amdgcn_s_barrier();  // trailing half: one segment behind
    int sA = 0;                                  // slot of A(g); B(g) sits in the next slot (mod 5)
    for (int vb = blockIdx.x; vb < total; vb += G) {
        const TileMap tmap = map_tile(vb, total, args.map);
        const PlaneRef pref = (FUSE || EPI == EPI_MAX) ? PlaneRef{0, 0} : plane_ref(args, tmap.plane);  // (the bound GEMM looks its plane up at the epilogue: one value less across its K loop)
        const PlaneConsts pcon = (FUSE || EPI == EPI_MAX) ? PlaneConsts{} : plane_consts(args, pref);  // fetched here: the latency passes behind the K loop
        for (int pl = 0; pl < planes_per_tile; ++pl) {
        v4i acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = v4i{EPI == EPI_MAX ? 0 : args.acc0, EPI == EPI_MAX ? 0 : args.acc0, EPI == EPI_MAX ? 0 : args.acc0, EPI == EPI_MAX ? 0 : args.acc0};

        int kt = 0;  // phases: see the K-step-barrier branch
        const int nph = (EPI == EPI_MAX && args.kt_mid > 0) ? 2 : 1;
        for (int ph = 0; ph < nph; ++ph) {
        const int kt_end = (EPI == EPI_MAX && ph + 1 < nph) ? args.kt_mid : KT;
        for (; kt < kt_end; ++kt) {
            const char* curA = smem + sA * TILE_BYTES + a_base;
            const char* curB = smem + (sA == 4 ? 0 : sA + 1) * TILE_BYTES + b_base;
            sA = sA + 2 >= 5 ? sA - 3 : sA + 2;
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const int coff = (((ks2 << 2) | q) ^ sw) << 4;
                v4i bf[4];
#pragma unroll
                for (int ah = 0; ah < 2; ++ah) {
                    v4i af[4];
                    if (ah == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            bf[j] = *(const v4i*)(curB + j * 16 * BK + coff);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        af[i] = *(const v4i*)(curA + (ah * 4 + i) * 16 * BK + coff);
                    // Only the K-step's LAST load segment completes its reads BEFORE the barrier: that is the barrier the ring's WAR rule counts on
                    // (LDS operations of a wave complete in order, so its wait covers every read of the K-step).  The other three segments arrive at
                    // the barrier as soon as their reads are ISSUED and wait behind it: a wave's LDS latency no longer delays the hand-over of all
                    // twelve (+0.4 / +0.5 % at k = 6144 / 8192, planes bit-identical: profiles/archive/r04e_late_wait_ab.txt)
                    if (ks2 == 1 && ah == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(ks2 == 1 && ah == 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
                    // serpentine order over the 4 x 4 fragment pairs: consecutive MFMAs share an operand register also across the row change
                    // (B fragment 3, 3, 0, 0 ...): +0.5 ... 0.65 % at k >= 8192, interleaved (profiles/archive/r04_gemm_ab_serpentine_kbar.txt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = (i & 1) ? 3 - jj : jj;
                            acc[ah * 4 + i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[ah * 4 + i][j], 0, 0, 0);
                        }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#if OZ2_HOOK_SKIP_EPILOGUE
        (void)tmap;  // laboratory probe: no epilogue; the accumulators stay live
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
        // every reload still pending on some path is waited for HERE (the previous tile's stores retired a whole K loop ago: free), so that the waitcnt
        // pass has no reason to put a vmcnt wait into the K loop (it did, in the K-step-barrier kernels until round 4 and in the bound GEMM after a
        // register-allocation change: tests/test_kernel_resources.py)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if constexpr (FUSE != 0) i8_epilogue<EPI, NoHook, -1>(acc, args, PlaneRef{0, pl}, tmap.tm * BM + wm * 128, tmap.tn * BN + wn * 64, lane);
        else if constexpr (EPI == EPI_MAX) i8_epilogue<EPI, NoHook, 0>(acc, args, plane_ref(args, tmap.plane), PlaneConsts{}, tmap.tm * BM + wm * 128, tmap.tn * BN + wn * 64, lane);
        else i8_epilogue<EPI, NoHook, (int)SMALLK>(acc, args, pref, pcon, tmap.tm * BM + wm * 128, tmap.tn * BN + wn * 64, lane);
#endif
        }  // phase
        }
#ifdef OZ2_LAB_FUSED_CRT
        if constexpr (FUSE != 0) i8_crt_tail<std::conditional_t<FUSE == 2, float, double>>(args, tmap.tm * BM + wm * 128, tmap.tn * BN + wn * 64, lane);
#endif
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    }
}

#ifdef OZ2_LAB_SELFPIPE  // laboratory build only (tools/build_probes.sh sp="-DOZ2_LAB_SELFPIPE=1"): the self-pipelined 8-wave form of the residue GEMM
#include "../../tools/experiments/gemm_i8_selfpipe.inc"
#endif
#ifdef OZ2_LAB_W4  // laboratory build only (tools/build_probes.sh w4="-DOZ2_LAB_W4=1"): four waves x 512 registers, wave tile 128 x 128
#include "../../tools/experiments/gemm_i8_w4.inc"
#endif

static void fill_common(GemmArgs& a, size_t kp, size_t m, size_t n) {
    a.kp = (int)kp;
    a.m = (int)m;
    a.n = (int)n;
    a.tiles_m = (int)((m + BM - 1) / BM);
    a.tiles_n = (int)((n + BN - 1) / BN);
    for (int t = 0; t < 20; ++t) {
        const int p = GEMMUL8_MODULI_INT8[t];
        a.moduli[t] = p;
        a.pinv32[t] = (int)(4294967296ull / (unsigned long long)p);
        a.dotw[t] = 1u | ((256u % p) << 8) | ((65536u % p) << 16) | ((16777216u % p) << 24);
        a.dotc[t] = (unsigned)((p - (int)(2147483648u % (unsigned)p)) % p);
    }
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

template <int EPI, bool KBAR, int FUSE = 0, bool SMALLK = false>
static hipError_t launch_sched(hipStream_t stream, GemmArgs& a, const std::conditional_t<FUSE != 0, CrtArgs, NoCrt>& crt = {}) {
    // the attribute belongs to the function on ONE device; setting it is idempotent, so concurrent first calls from several host
    // threads only need the flag itself to be race-free
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 0;
    if (!attr_set_dev[dev_].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_kernel<EPI, KBAR, FUSE, SMALLK>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set_dev[dev_].store(true, std::memory_order_release);
    }
    // persistent: one workgroup per CU; a multiple of 8 keeps "workgroup b runs on XCD b % 8" aligned with map_tile.
    // Measured interleaved against one-workgroup-per-tile launches of the same kernel (tools/gemm_ab.py, 8192 x 8192 x k,
    // 14 planes): 13 % faster at k = 256, 7 % at k = 2048, 1 % at k = 8192.
    int grid = num_cus() & ~7;
    if (const int want = knobs().gemm_cus; want > 0 && want < grid) grid = want;  // measurement switch (oz2_knobs.hpp): same results
    if (grid <= 0) grid = 8;
    if (a.total_tiles < grid) grid = a.total_tiles;
    hipLaunchKernelGGL((gemm_i8_kernel<EPI, KBAR, FUSE, SMALLK>), dim3(grid), dim3(WS_THREADS), RING_LDS_BYTES, stream, a, crt);
    return hipGetLastError();
}

// Two schedules of the same tile loop (bit-identical results): with a workgroup barrier after every LOAD / MFMA segment (ping-pong)
// or with one barrier per K-step.  Interleaved on one box (tools/gemm_ab.py, 8192^2 x k, 14 planes, profiles/archive/r02_sched_ab.txt): the
// K-step-barrier form is faster by 10 / 7.5 / 4.7 / 2.4 % at k = 512 / 1024 / 2048 / 4096 (tile boundaries -- epilogue beside the
// other half's first K-steps -- overlap better) and 2.4 % slower at k = 8192, where the board is power-bound and removing stalls
// buys nothing while the s_sleep-paced LDS-DMA is a little less smooth than the barrier-paced one.
template <int EPI> static hipError_t launch(hipStream_t stream, GemmArgs& a, int planes) {
    // batched call: the items' planes are one long plane sequence (item-major), the persistent tile loop runs over all of them
    a.ppi = planes;
    a.m_ppi = map_magic((unsigned)planes);
    a.bstride = g_batch.ws;
    planes *= (int)g_batch.batch;
    a.total_tiles = planes * a.tiles_m * a.tiles_n;
    if (a.total_tiles <= 0) return hipSuccess;
    a.colblock = map_colblock((size_t)a.tiles_n, (size_t)a.kp * (size_t)a.nseg);
    a.map = make_tile_map(a.tiles_m, a.tiles_n, a.colblock);
    a.acc0 = (size_t)a.kp * (size_t)a.nseg <= 512 ? 0 : (int)0x80000000u;  // RED_ODD_SMALL needs |sum| < 2^23
    // (the bound GEMM keeps the ping-pong schedule at every k.  In round 2 its K-step-barrier instantiation spilled accumulators INSIDE
    // the MFMA loop; with the round-3 source it no longer does, but the single-plane launch still runs slower with it: bounds phase
    // 88.4 -> 92.4 us at 3072^3, 130.9 -> 134.6 at 4096^3, equal at 2048^3 and 8192^3.  round-3 A/B: profiles/archive/r03_bound_ab.txt)
#ifdef OZ2_LAB_SHORTK  // laboratory build only (tools/experiments/shortk): the half-tile ping-pong kernel for padded k <= OZ2_LAB_SHORTK
    if constexpr (EPI != EPI_MAX) {
        if (a.nseg == 1 && a.kp <= OZ2_LAB_SHORTK) return launch_gemm_i8_shortk(stream, a, EPI);
    }
#endif
#ifdef OZ2_LAB_SELFPIPE
    if constexpr (EPI != EPI_MAX) {
        if (a.nseg == 1) {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)gemm_i8_selfpipe_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_i8_selfpipe_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES);
                attr_set = true;
            }
            int grid = num_cus() & ~7;
            if (a.total_tiles < grid) grid = a.total_tiles;
            if (a.acc0 == 0) hipLaunchKernelGGL((gemm_i8_selfpipe_kernel<EPI, true>), dim3(grid), dim3(512), RING_LDS_BYTES, stream, a);
            else hipLaunchKernelGGL((gemm_i8_selfpipe_kernel<EPI, false>), dim3(grid), dim3(512), RING_LDS_BYTES, stream, a);
            return hipGetLastError();
        }
    }
#endif
#ifdef OZ2_LAB_W4
    if constexpr (EPI != EPI_MAX) {
        if (a.nseg == 1 && a.kp >= 4 * BK) {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)gemm_i8_w4_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_i8_w4_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES);
                attr_set = true;
            }
            int grid = num_cus() & ~7;
            if (a.total_tiles < grid) grid = a.total_tiles;
            if (a.acc0 == 0) hipLaunchKernelGGL((gemm_i8_w4_kernel<EPI, true>), dim3(grid), dim3(256), RING_LDS_BYTES, stream, a);
            else hipLaunchKernelGGL((gemm_i8_w4_kernel<EPI, false>), dim3(grid), dim3(256), RING_LDS_BYTES, stream, a);
            return hipGetLastError();
        }
    }
#endif
    if constexpr (EPI == EPI_MOD && OZ2_MOD256) {
        // K <= 256: accumulators carried as float patterns (RED_MAGIC, oz2_gemm_i8_epi.hpp): 8192^2 x 256, 14 planes: see profiles/r06_short_k_epilogue_ab.txt
        if ((size_t)a.kp * (size_t)a.nseg <= 256) {
            a.acc0 = 0x4B400000;
            return launch_sched<EPI_MOD256, true, 0, true>(stream, a);
        }
    }
    if constexpr (EPI != EPI_MAX) {
        if (a.acc0 == 0) return launch_sched<EPI, true, 0, true>(stream, a);  // K <= 512
        if (a.kp * a.nseg <= OZ2_KBAR_MAX_KP) return launch_sched<EPI, true>(stream, a);
    }
    return launch_sched<EPI, false>(stream, a);
}

// Non-temporal residue stores.  The residue planes are written once and read once, by the CRT pass, N planes later; written with the
// default policy they are allocated in the 256 MiB Infinity Cache on their way to HBM and push out the operand planes A_lo / B_lo,
// which the quantise kernels have just left there and which every plane's 32 x 32 tiles re-read through eight L2s.  When ALL operand
// planes of the launch fit the Infinity Cache, keeping them there is worth 8-14 % of the WHOLE call (8192^2: k = 256 / 512 / 1024
// 0.917 -> 0.815 / 1.059 -> 0.940 / 1.470 -> 1.285 ms; 16384^2: k = 256 / 512 3.22 -> 2.88 / 3.88 -> 3.59 ms); when they do not fit
// there is nothing to protect and the CRT pass loses the tail of C_mid it would have found in the cache (-0.5 ... -2 % from k = 1536
// at 8192^2, k = 1024 at 16384^2); with small outputs (4096^2 and below) it is a wash.  profiles/archive/r03_epi_nt_grid.txt
static int nt_residue_planes(const GemmArgs& a, int planes, bool stream_out, bool automatic = true) {
    if (const int force = knobs().epi_nt; force >= 0) return force == 1 && stream_out ? planes : 0;  // testing switch: one policy for all planes (same results)
    if (!stream_out || !automatic) return 0;
    const size_t all = (size_t)planes * g_batch.batch;
    const size_t operands = all * (a.strideA + a.strideB), residues = all * a.strideO;
    // (keeping the default policy for the last 2-6 planes, so that the CRT finds them in the cache, non-temporal stores for the
    // leading planes of launches whose operands do NOT fit, and walking the planes last to first -- the order the quantise kernels
    // left them in the cache -- were all measured: no consistent gain, profiles/archive/r03_epi_nt_keep.txt; between 240 and ~450 MiB of
    // operand planes the sign of the effect differs from box to box, -2 ... +4 %)
    return residues >= ((size_t)256 << 20) && operands <= ((size_t)240 << 20) ? planes : 0;
}

hipError_t launch_gemm_i8_mod(hipStream_t stream, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                              size_t n, int t_begin, int t_end, int8_t* out, size_t ldo, size_t strideO, bool stream_out) {
    GemmArgs a{};
    a.A[0] = A;
    a.B[0] = B;
    a.nseg = 1;
    a.strideA = strideA;
    a.strideB = strideB;
    a.t_begin = t_begin;
    a.out = out;
    a.ldo = ldo;
    a.strideO = strideO;
    fill_common(a, kp, m, n);
    a.nt_planes = nt_residue_planes(a, t_end - t_begin, stream_out);
    return launch<EPI_MOD>(stream, a, t_end - t_begin);
}

hipError_t launch_gemm_i8_cplx(hipStream_t stream, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                               size_t n, int t_begin, int t_end, const int8_t* rx, const int8_t* ry, size_t strideR, int8_t* out,
                               size_t ldo, size_t strideO) {
    GemmArgs a{};
    a.A[0] = A;
    a.B[0] = B;
    a.nseg = 1;
    a.strideA = strideA;
    a.strideB = strideB;
    a.t_begin = t_begin;
    a.out = out;
    a.ldo = ldo;
    a.strideO = strideO;
    a.rx = rx;
    a.ry = ry;
    a.strideR = strideR;
    fill_common(a, kp, m, n);
    // (the size rule of nt_residue_planes LOSES 1-3 % of the whole call on this launch -- ZGEMM 8192^2 x 512 ... 8192, 14 moduli; CGEMM x 768 ...
    // 2048, 7 moduli: the operand planes of the three parts never fit the Infinity Cache together, and the CRT finds more of the
    // interleaved plane there with the default policy -- so the combine launch keeps the default policy unless GEMMUL8_EPI_NT forces it)
    a.nt_planes = nt_residue_planes(a, t_end - t_begin, true, false);
    return launch<EPI_CPLX>(stream, a, t_end - t_begin);
}

#ifndef OZ2_MAX_SMALL_TILES
#define OZ2_MAX_SMALL_TILES 128  // the bound GEMM takes the 128 x 128-tile kernel when (batch x) its 256 x 256 tiles number at most this (of 256 CUs): bounds phase 33 -> 28 / 41 -> 32 / 66 -> 55 us at 512^3 / 1024^3 / 2048^3, but 98 -> 111 us at 3072^3 (144 tiles), profiles/archive/r03_bound_ab.txt
#endif
// mid_seg > 0: the row / column maxima are taken twice per tile -- of the partial sums after the first mid_seg K-segments and of the
// full sums.  The complex bound (max over the elements of C1 = ArBi + AiBr and of C1 + C0, C0 = (Ar-Ai)(Br-Bi)) is then ONE launch over
// three segments with mid_seg = 2 (three real GEMMs of work) instead of a 2-segment and a 3-segment launch (five).
hipError_t launch_gemm_i8_max(hipStream_t stream, int nseg, const int8_t* const* A, const int8_t* const* B, size_t kp, size_t m, size_t n,
                              int* rowmax, int* colmax, int mid_seg) {
    {
        // GEMMUL8_BOUND_TILE = 128 | 256 forces one kernel (tests run every case through both; the maxima are identical)
        const int force = knobs().bound_tile;
        const size_t tiles = ((m + BM - 1) / BM) * ((n + BN - 1) / BN) * g_batch.batch;
        const bool small = force == 128 ? true : force == 256 ? false : tiles <= (size_t)OZ2_MAX_SMALL_TILES;
        if (small) return launch_gemm_i8_max_small(stream, nseg, A, B, kp, m, n, rowmax, colmax, mid_seg);
    }
    GemmArgs a{};
    for (int s = 0; s < nseg; ++s) a.A[s] = A[s], a.B[s] = B[s];
    a.nseg = nseg;
    a.rowmax = rowmax;
    a.colmax = colmax;
    a.kt_mid = mid_seg > 0 && mid_seg < nseg ? mid_seg * (int)(kp / BK) : 0;
    fill_common(a, kp, m, n);
    return launch<EPI_MAX>(stream, a, 1);
}

}  // namespace oz2
