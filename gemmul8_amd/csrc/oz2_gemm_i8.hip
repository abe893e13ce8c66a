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
//  * Cost split measured with real-data probes (OZ2_PROBE_LDS; 16x16x64 kernel, k = 8192): the board's
//    power-limited ceiling for this instruction on residue data is 3.97 POP/s, MFMA + barriers alone reach 93 % of it; the
//    L2 -> LDS operand path costs 13 %, the per-segment barriers 7 %, the epilogue 6 %, the LDS reads 5 %.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "oz2_crt_common.hpp"
#include "oz2_gemm_common.hpp"
#include "oz2_kernels.h"

namespace oz2 {

enum { EPI_MOD = 0, EPI_MAX = 1, EPI_CPLX = 2 };

struct GemmArgs {
    const int8_t* A[3];    // K-segment s of plane 0: A[s] + plane*strideA : [rows(pad 256)][kp]
    const int8_t* B[3];    //                          B[s] + plane*strideB : [n][kp]
    int nseg;
    size_t strideA;        // bytes between consecutive planes (moduli)
    size_t strideB;
    int kp;                // padded K (multiple of 256) = row pitch in bytes
    int m, n;              // valid rows / cols of C
    int tiles_m, tiles_n;
    int colblock;          // tile-columns per column block of the tile walk (map_colblock; 0 = full width)
    int t_begin;           // plane p <-> modulus t_begin + p
    int8_t* out;           // EPI_MOD: plane p at out + p*strideO, [n][ldo] int8; EPI_CPLX: [n][ldo] char2
    size_t ldo;
    size_t strideO;
    const int8_t* rx;      // EPI_CPLX: residues of X and Y, plane p at rx/ry + p*strideR, [n][ldo] int8
    const int8_t* ry;
    size_t strideR;
    int* rowmax;           // EPI_MAX
    int* colmax;
    int kt_mid;            // EPI_MAX: > 0 = the maxima are ALSO taken after this many K-steps of every tile (partial sums of a K-concatenation)
    int total_tiles;       // planes * tiles_m * tiles_n (tile-stationary order, FUSE != 0: tiles_m * tiles_n)
    int planes;            // FUSE != 0: residue planes every workgroup runs through per output tile
    int ppi;               // planes per batch item (plane p = item p / ppi, modulus-relative plane p % ppi); = all planes for one GEMM
    size_t bstride;        // bytes between the workspaces of consecutive batch items (every pointer above lives in the workspace)
    int moduli[20];
    int pinv32[20];
    int nt_planes;         // EPI_MOD: planes tt < nt_planes (of each batch item) leave with non-temporal stores (launch_gemm_i8_mod decides)
    int acc0;              // EPI_MOD / EPI_CPLX: initial accumulator value: -2^31 (RED_ODD reads the register as x + 2^31), or 0 when K <= 512 (RED_ODD_SMALL)
    unsigned dotw[20];     // RED_ODD: bytes (256^j mod p), j = 0..3 (byte 0 = 1)
    unsigned dotc[20];     //          (-2^31) mod p
};

// Epilogues on a wave's 128 x 64 accumulator block (first row i0, first column j0) = 8 x 4 tiles of v_mfma_i32_16x16x64_i8, whose
// accumulator map is col = lane & 15, row = 4 * (lane >> 4) + reg; MFMA rows <-> C rows i (A_lo rows), MFMA cols <-> C cols j.
#ifndef OZ2_DMA_AUX
#define OZ2_DMA_AUX 0  // cache-policy bits of the LDS-DMA loads (1 = sc0, 2 = nt, 16 = sc1): sc0/sc1 measured neutral, nt 10 % slower
#endif
#ifndef OZ2_PROBE_LDS
#define OZ2_PROBE_LDS 0  // timing probes on REAL data (wrong results; tools/README.md): bit 0/1 B/A fragments re-read at ks == 0 only, bit 2 DMA in the
                        // first tile only, bit 3 no epilogue, bit 4 operands from the first 8 K-steps only (L2 hits); 0 in every shipped build
#endif
#ifndef OZ2_EPI_NT
#define OZ2_EPI_NT 0  // compile-time residue-store policy for A/B builds: 1 non-temporal always, 2 sc0, 3 sc1, 4 sc0 sc1 (sc bits: 2-6 % slower
                     // everywhere, profiles/r03_epi_store_policy.txt).  The shipped build chooses non-temporal stores per launch and plane: args.nt_planes
#endif
#ifndef OZ2_CPLX_NT
#define OZ2_CPLX_NT 0  // 1: the size rule of nt_residue_planes also for the complex combine launch.  Forced (GEMMUL8_EPI_NT=1) it LOSES 1-3 % of the
                      // whole call (ZGEMM 8192^2 x 512 ... 8192, 14 moduli; CGEMM x 768 ... 2048, 7 moduli): the operand planes of the three parts
                      // never fit the Infinity Cache together, and the CRT finds more of the interleaved plane there with the default policy
#endif
#ifndef OZ2_RED_SMALL
#define OZ2_RED_SMALL 1  // K <= 512: three-instruction residue straight from the (unbiased) accumulator, see RED_ODD_SMALL (8192^2 x 256 / 512,
                         // 14 planes: 0.440 -> 0.394 / 0.614 -> 0.565 ms; profiles/r03_red_small_ab.txt)
#endif
#ifndef OZ2_RED_DOT4
#define OZ2_RED_DOT4 1  // odd moduli: residue of an accumulator by byte dot product (4 full-rate 32-bit instructions) instead of the FP64 quotient (5):
                        // 14 planes 8192 x 8192, k = 1024 / 4096 / 8192: 1.011 -> 0.990 / 2.911 -> 2.891 / 5.356 -> 5.348 ms (profiles/r03_red_dot4_ab.txt)
#endif
#ifndef OZ2_CPLX_ABL
#define OZ2_CPLX_ABL 0  // timing ablations of the complex combine epilogue (wrong results): 1 no X / Y loads, 2 no stores, 4 half the stores.
                        // ZGEMM 8192^3, 20 moduli, low-precision phase: 25.24 ms shipped, 24.38 (1), 24.47 (2), 23.89 (3) against 23.3 ms
                        // for 60 plain residue planes -- the whole combine costs 8 %, its compute 2.5 %, loads and stores 2.5-3.5 % each; a
                        // producer-side touch of the X / Y lines ahead of the epilogue changed nothing (profiles/r03_cplx_abl.txt)
#endif
#ifndef OZ2_CPLX_PK16
#define OZ2_CPLX_PK16 0  // 1: complex combine epilogue with packed 16-bit arithmetic (two elements per instruction, half the VALU work; bit-identical).
                         // Measured NOT faster (ZGEMM 8192^3 x 20 moduli low-precision phase 25.85 vs 25.58 ms): the epilogue waits on its X / Y loads, not on VALU
#endif
#ifndef OZ2_ABL_EPI
#define OZ2_ABL_EPI 0  // timing ablations only: 1 no stores, 2 every plane takes the p = 256 path, 3 all stores of a plane land in one 1 MiB window.
                       // Epilogue of a 256 x 256 tile = 5.6 us (k = 1024: 17.5 us per tile, 11.9 without epilogue): residue arithmetic 1.8,
                       // stores 2.6-3.0, packing / transposes 1.2.  The store cost is NOT the instructions: into an L2-resident window (3) they
                       // cost 0.3 us.  It is the 64 KiB of fresh lines per tile on the memory side, which delays the LDS-DMA reads of the
                       // next tile.  Measured and not kept: lane order with 64 contiguous bytes per lane quad (+3 %), complete 128-byte
                       // lines per store instruction via v_mov_dpp row_ror:8 + ds_bpermute (+6 % at k = 1024, +0.5 % at 8192), workgroups
                       // started up to one tile apart (k = 1024: -1 % at a quarter tile, 0 beyond; k = 8192: +1.4 ... +4 %), non-temporal
                       // stores (round 2).  profiles/r03_epi_probe.txt, r03_epi_store_probe.txt, r03_epi_laneperm_ab.txt, r03_epi_fullline_ab.txt
#endif
// plane p of a (batched) launch: byte offset of its item's workspace and its plane index inside the item
struct PlaneRef {
    size_t boff;
    int tt;
};
template <typename Args> __device__ __forceinline__ PlaneRef plane_ref(const Args& args, int plane) {
    const int p = __builtin_amdgcn_readfirstlane(plane);
    const int b = p / args.ppi;
    return {(size_t)b * args.bstride, p - b * args.ppi};
}
enum { RED_GENERIC = 0, RED_ODD = 1, RED_256 = 2, RED_ODD_SMALL = 3 };
// RED selects how an accumulator is reduced (uniform per plane): RED_256: p = 256, the symmetric residue IS the low byte;
// RED_ODD: odd p, ONE exact FP64 quotient step for any int32 accumulator (v_cvt_f64_i32, v_mul_f64, v_rndne_f64, v_fma_f64,
// v_cvt_i32_f64: FP64 VALU runs at the FP32 rate on gfx950); the two-step fp32 form it replaced cost 10 instructions and 12 % of
// the kernel time at k = 1024.  (Reading the quotient from the low dword of fma(a, 1/p, 1.5 * 2^52) and finishing with
// v_mad_i32_i24 -- three instructions -- measured 10 % SLOWER at k = 1024: the dependent FP64 chains no longer overlap.)
// RED_GENERIC: 32-bit multiply-high (even p other than 256: no INT8 modulus, kept for completeness).
template <int EPI, int RED>
__device__ __forceinline__ void i8_epilogue_mod(const v4i (&acc)[8][4], const GemmArgs& args, PlaneRef pl, int i0, int j0, int lane) {
    const int c16 = lane & 15;
    const int q = lane >> 4;
    const int t = args.t_begin + pl.tt;
    const int p = args.moduli[t];
    const int pinv = args.pinv32[t];
    const float invp = 1.0f / (float)p;
    [[maybe_unused]] const double pd = (double)p, invpd = 1.0 / (double)p;
    [[maybe_unused]] const unsigned dotw = args.dotw[t], dotc = args.dotc[t];
    auto red = [&](int x) {
        if constexpr (RED == RED_256) return x;  // the bias 2^31 of OZ2_RED_DOT4 does not touch the low byte
        else if constexpr (RED == RED_ODD) {
#if OZ2_RED_DOT4
            // the accumulators start at -2^31 (acc init in the kernel): read as unsigned the register holds u = x + 2^31 for ANY int32 sum
            // x, and s = sum_j byte_j(u) (256^j mod p) + ((-2^31) mod p) == x (mod p), 0 <= s < 2^18: v_dot4_u32_u8.  One fp32 quotient
            // and the 24-bit multiply-add give the canonical residue (mod_small_sym_u, oz2_device.hpp).
            return mod_small_sym_u(__builtin_amdgcn_udot4((unsigned)x, dotw, dotc, false), p, invp);
#else
            return mod_i32_sym_odd_f64(x, pd, invpd);
#endif
        } else if constexpr (RED == RED_ODD_SMALL) {
            // short K (kp * nseg <= 512: |x| <= 512 * 127^2 < 2^23; the accumulators start at 0, GemmArgs.acc0): the quotient comes
            // straight from the accumulator -- v_cvt_f32_i32, one fma against 1.5 * 2^23 (its low 24 bits are 2^22 + q for either sign
            // of q), v_mad_i32_i24: the canonical residue minus p 2^22, i.e. the canonical LOW BYTE, which is all the epilogue stores.
            // Three instructions instead of four.  The bound is 2^23, not the 2^24 of fp32 exactness: |x| |RN(1/p) - 1/p| must stay
            // below the 1/(2p) that separates x / p from a rounding tie (exhaustive CPU model: first wrong byte at |x| = 8 454 907 for
            // p = 255, tests/test_residue_math.py; tests/test_gpu_parity.py::test_epilogue_reduction_on_extreme_accumulators).
            const float qf = fmaf((float)x, invp, 12582912.0f);
            int r;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(__float_as_int(qf)), "s"(-p), "v"(x));
            return r;
        } else return mod_i32_sym((int)((unsigned)x ^ ((OZ2_RED_DOT4 && args.acc0) ? 0x80000000u : 0u)), p, pinv);
    };
    auto red_small = [&](int x) {
        if constexpr (RED == RED_256) return x;
        else if constexpr (RED == RED_ODD || RED == RED_ODD_SMALL) return mod_small_sym_odd(x, p, invp);
        else return mod_i32_sym(x, p, pinv);
    };
    // After the 4 x 4 dword transpose below lane (q, c16) owns the 16 consecutive rows i0 + 64 tg + 16 q .. + 15 of column
    // j0 + 16 tj + c16: one 64-bit element offset per lane for the whole block, the (tg, tj) sub-blocks add 64 tg and 16 tj * ldo --
    // no per-store multiplies (v_mul_lo_u32 / v_mad_u64_u32 are quarter rate)
    const size_t e00 = (size_t)(j0 + c16) * args.ldo + i0 + q * 16;
    const size_t po = pl.boff + (size_t)pl.tt * args.strideO, pr = pl.boff + (size_t)pl.tt * args.strideR;  // wave-uniform: scalar multiplies
    const size_t ejs = (size_t)16 * args.ldo;
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        const int col = j0 + tj * 16 + c16;
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            unsigned d[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                int r[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) r[b] = red(acc[tg * 4 + ti][tj][b]);
                // low bytes of four residues -> one dword with two v_perm_b32 and an OR (selector bytes: 0-3 = second operand,
                // 4-7 = first operand, 0x0c = zero)
                d[ti] = __builtin_amdgcn_perm((unsigned)r[1], (unsigned)r[0], 0x0c0c0400u) |
                        __builtin_amdgcn_perm((unsigned)r[3], (unsigned)r[2], 0x04000c0cu);
            }
            // lane quad q holds rows 4 q .. 4 q + 3 of the four 16-row tiles ti.  4 x 4 transpose over the quads (lane bits 5, 4) so
            // that quad q holds all 16 rows of tile ti = q: bit 5 with v_permlane32_swap, bit 4 with v_permlane16_swap.
            const auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);  // [0]: tile 2 qh, rows of quad (0, ql); [1]: of quad (1, ql)
            const auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);  // the same for tile 2 qh + 1
            const auto w01 = __builtin_amdgcn_permlane16_swap(s0[0], s1[0], false, false);  // tile q: rows 0-3, rows 4-7
            const auto w23 = __builtin_amdgcn_permlane16_swap(s0[1], s1[1], false, false);  //         rows 8-11, rows 12-15
            const unsigned z[4] = {w01[0], w01[1], w23[0], w23[1]};
            if (col < args.n && !(OZ2_ABL_EPI == 1 && args.kp > 0)) {
                const size_t e = (e00 + tj * ejs + tg * 64) & (OZ2_ABL_EPI == 3 ? (size_t)0xFFFF0 : ~(size_t)0);  // first of 16 consecutive rows
                if constexpr (EPI == EPI_MOD) {
#if OZ2_EPI_NT == 1
                    {
                        typedef unsigned v4u __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(v4u{z[0], z[1], z[2], z[3]}, (v4u*)(args.out + po + e));
                    }
#elif OZ2_EPI_NT >= 2  // cache-policy bits of the residue stores: 2 = sc0, 3 = sc1, 4 = sc0 sc1
                    {
                        typedef unsigned v4u __attribute__((ext_vector_type(4)));
                        const v4u zv = {z[0], z[1], z[2], z[3]};
                        const int8_t* ptr = args.out + po + e;
                        if (OZ2_EPI_NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(ptr), "v"(zv) : "memory");
                        else if (OZ2_EPI_NT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(zv) : "memory");
                        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(zv) : "memory");
                    }
#else
                    if (pl.tt < args.nt_planes) {  // wave-uniform
                        typedef unsigned v4u __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(v4u{z[0], z[1], z[2], z[3]}, (v4u*)(args.out + po + e));
                    } else {
                        *(uint4*)(args.out + po + e) = make_uint4(z[0], z[1], z[2], z[3]);
                    }
#endif
                } else {
                    // eight rows at a time: 8 bytes of X and Y in, 16 bytes of (Cr, Ci) pairs out -- with all 16 rows in flight the epilogue
                    // needed 16 more registers than the 168-VGPR budget leaves beside the accumulators (51-62 spilled registers)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#if OZ2_CPLX_ABL & 1  // timing ablations of the complex epilogue (wrong results; -DOZ2_CPLX_ABL=bits): 1 = no X / Y loads
                        const uint2 x2 = make_uint2(z[2 * h] * 3, z[2 * h + 1] * 5), y2 = make_uint2(z[2 * h] ^ 0x55u, z[2 * h + 1] + 7u);
#else
                        const uint2 x2 = *(const uint2*)(args.rx + pr + e + 8 * h);
                        const uint2 y2 = *(const uint2*)(args.ry + pr + e + 8 * h);
#endif
                        const unsigned xs[2] = {x2.x, x2.y}, ys[2] = {y2.x, y2.y};
                        unsigned o[4];
#pragma unroll
                        for (int w2 = 0; w2 < 2; ++w2) {
                            unsigned lo = 0, hi = 0;
                            if constexpr ((RED == RED_ODD || RED == RED_ODD_SMALL) && OZ2_CPLX_PK16) {
                                // packed 16-bit form: |X|, |Y|, |Z| <= (p-1)/2, so X - Y lies in (-p, p) and Z - X - Y in (-1.5 p, 1.5 p):
                                // ONE wrap r = d + p ([d < -h] - [d > h]) gives the canonical residue, and v_pk_*_i16 does two elements per
                                // instruction (comparisons as arithmetic shifts of h -+ d).  Half the VALU work of the per-element fp32 steps.
                                typedef short v2s __attribute__((ext_vector_type(2)));
                                const v2s hv = {(short)((p - 1) >> 1), (short)((p - 1) >> 1)}, pv = {(short)p, (short)p};
                                auto ev = [](unsigned w) {  // bytes 0, 2 sign-extended into the 16-bit halves
                                    v2s t;
                                    __builtin_memcpy(&t, &w, 4);
                                    return (v2s)((t << 8) >> 8);
                                };
                                auto od = [](unsigned w) {  // bytes 1, 3
                                    v2s t;
                                    __builtin_memcpy(&t, &w, 4);
                                    return (v2s)(t >> 8);
                                };
                                auto wrap = [&](v2s d) { return (v2s)(d + (((hv - d) >> 15) - ((d + hv) >> 15)) * pv); };
                                auto bits = [](v2s v) {
                                    unsigned u;
                                    __builtin_memcpy(&u, &v, 4);
                                    return u;
                                };
                                const unsigned xw = xs[w2], yw = ys[w2], zw_ = z[2 * h + w2];
                                const v2s xe = ev(xw), ye = ev(yw), ze = ev(zw_), xo = od(xw), yo = od(yw), zo = od(zw_);
                                const v2s cre = wrap(xe - ye), cro = wrap(xo - yo);
                                const v2s cie = wrap(ze - xe - ye), cio = wrap(zo - xo - yo);
                                // t0 = (Cr, Ci) pairs of bytes 0 and 2, t1 = of bytes 1 and 3; lo = elements 0, 1, hi = elements 2, 3
                                const unsigned t0 = __builtin_amdgcn_perm(bits(cie), bits(cre), 0x06020400u);
                                const unsigned t1 = __builtin_amdgcn_perm(bits(cio), bits(cro), 0x06020400u);
                                lo = __builtin_amdgcn_perm(t1, t0, 0x05040100u);
                                hi = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
                            } else
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const int X = (int)(int8_t)(xs[w2] >> (8 * b)), Y = (int)(int8_t)(ys[w2] >> (8 * b)), Z = (int)(int8_t)(z[2 * h + w2] >> (8 * b));
                                const int cr = red_small(X - Y), ci = red_small(Z - X - Y);
                                const unsigned pair = ((unsigned)cr & 0xFFu) | (((unsigned)ci & 0xFFu) << 8);
                                if (b < 2) lo |= pair << (16 * b);
                                else hi |= pair << (16 * (b - 2));
                            }
                            o[2 * w2] = lo;
                            o[2 * w2 + 1] = hi;
                        }
#if OZ2_CPLX_ABL & 2  // timing ablation (wrong results): no stores
                        asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
#elif OZ2_CPLX_ABL & 4  // timing ablation (wrong results): half the stores
                        if (h == 0) *((uint4*)(args.out + po + 2 * e) + h) = make_uint4(o[0] ^ o[2], o[1] ^ o[3], o[2], o[3]);
#else
                        if (pl.tt < args.nt_planes) {  // wave-uniform
                            typedef unsigned v4u __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store(v4u{o[0], o[1], o[2], o[3]}, (v4u*)(args.out + po + 2 * e) + h);
                        } else {
                            *((uint4*)(args.out + po + 2 * e) + h) = make_uint4(o[0], o[1], o[2], o[3]);
                        }
#endif
                    }
                }
            }
        }
    }
}

template <int EPI>
__device__ __forceinline__ void i8_epilogue(const v4i (&acc)[8][4], const GemmArgs& args, PlaneRef pl, int i0, int j0, int lane) {
    const int c16 = lane & 15;
    const int q = lane >> 4;

    if constexpr (EPI == EPI_MOD || EPI == EPI_CPLX) {
        const int p = args.moduli[args.t_begin + pl.tt];
        if (p == 256 || OZ2_ABL_EPI == 2) i8_epilogue_mod<EPI, RED_256>(acc, args, pl, i0, j0, lane);
        else if ((p & 1) && OZ2_RED_DOT4 && args.acc0 == 0) i8_epilogue_mod<EPI, RED_ODD_SMALL>(acc, args, pl, i0, j0, lane);
        else if (p & 1) i8_epilogue_mod<EPI, RED_ODD>(acc, args, pl, i0, j0, lane);
        else i8_epilogue_mod<EPI, RED_GENERIC>(acc, args, pl, i0, j0, lane);
    } else {
        int* const rowmax_ = (int*)((char*)args.rowmax + pl.boff);
        int* const colmax_ = (int*)((char*)args.colmax + pl.boff);
        // column max over this lane's 32 rows (masked to valid rows), then across the four lane quads
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            int cm = 0;
#pragma unroll
            for (int ti = 0; ti < 8; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + ti * 16 + 4 * q + r;
                    const int v = (row < args.m) ? acc[ti][tj][r] : 0;
                    cm = v > cm ? v : cm;
                }
            int other = __shfl_xor(cm, 16);
            cm = other > cm ? other : cm;
            other = __shfl_xor(cm, 32);
            cm = other > cm ? other : cm;
            const int col = j0 + tj * 16 + c16;
            if (q == 0 && col < args.n && cm > 0) atomicMax(colmax_ + col, cm);
        }
        // row max across the 16 lanes (columns) of each quad, one 16-row tile row at a time
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
            int w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int v = 0;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    const int col = j0 + tj * 16 + c16;
                    const int a = (col < args.n) ? acc[ti][tj][r] : 0;
                    v = a > v ? a : v;
                }
                w[r] = v;
            }
            tile_rowmax_atomic16(w, rowmax_, i0 + ti * 16, args.m, lane);
        }
    }
}


// CRT tail of the tile-stationary order (FUSE != 0): after the epilogue of the LAST residue plane of an output tile
// every consumer lane re-reads the 16-byte residue vectors IT stored for the N planes (same addresses, same lane: ordered by its own
// vmcnt wait, no cross-wave visibility needed), accumulates the CRT sums exactly as crt_kernel does (oz2_crt.hip: same chains, same
// order t = 0..N-1, same reduction, scalbn and axpby forms) and writes 16 consecutive rows of one column of C.  The residue planes
// were written microseconds earlier: the re-read is served by the L2 / Infinity Cache, the stand-alone CRT pass over C_mid (N bytes
// per element from HBM, 5 % of the config-2 call, 20 % at k = 1024) and its launch disappear.
// The CRT block is read from the kernel-argument segment through a pointer the compiler cannot see through: taken by value it
// hoists the 60 table doubles into SGPRs at kernel entry and keeps them (spilled to VGPR lanes) across the K loop, which pushed
// accumulator spills INTO the MFMA loop.
#ifndef OZ2_PRIO_MODE
#define OZ2_PRIO_MODE 0  // wave priorities (experiment switch): 0 = s_setprio 1 around every MFMA segment (shipped), 1 = none, 2 = the lagging
                        // half at priority 1 for the whole kernel, no flips, 3 = as 0 with the producer waves at priority 3
#endif
#ifndef OZ2_MAX_KBAR
#define OZ2_MAX_KBAR 0
#endif
#ifndef OZ2_PCRT_ABL
#define OZ2_PCRT_ABL 0  // timing ablations of the producer-wave CRT (wrong results): 1 no CRT work at all (order + schedule only), 2 no residue re-reads, 4 no C stores
#endif
#ifndef OZ2_TAIL_ABL
#define OZ2_TAIL_ABL 0  // timing ablations of the CRT tail (wrong results): 1 no residue re-reads, 2 no C stores, 4 no CRT chains, 8 no tail
#endif
template <typename U>
__device__ __forceinline__ void i8_crt_tail(const GemmArgs& args, int i0, int j0, int lane) {
    if (OZ2_TAIL_ABL & 8) return;
    typedef const __attribute__((address_space(4))) CrtArgs* CrtPtr;
    CrtPtr cp = (CrtPtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(GemmArgs) + 7) & ~size_t(7)));
    asm volatile("" : "+s"(cp)::"memory");
    const auto& c = *cp;
    const double cPhi = c.Phi, cPlo = c.Plo, cinvP = c.invP;
    const bool use_dd = c.use_dd != 0;
    auto reduce = [&](double Sh, double Sl) {  // crt_reduce of oz2_crt_common.hpp
        const double q = rint(cinvP * Sh);
        if (!use_dd) return fma(cPhi, q, Sh);
        return fma(cPlo, q, fma(cPhi, q, Sh) + Sl);
    };
    const int c16 = lane & 15;
    const int q = lane >> 4;
    const size_t e00 = (size_t)(j0 + c16) * args.ldo + i0 + q * 16;
    const size_t ejs = (size_t)16 * args.ldo;
    U al = (U)c.alpha[0], be = (U)c.beta[0];
    int mode = c.mode;
    if (mode == 5) {
        al = *(const U*)c.alpha_dev;
        be = *(const U*)c.beta_dev;
        mode = 0;
    }
    const bool reads_c = mode == 0 || mode == 2 || mode == 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's residue stores of the last plane
#pragma unroll 1
    for (int sb = 0; sb < 8; ++sb) {
        const int tj = sb >> 1, tg = sb & 1;
        const int col = j0 + tj * 16 + c16;
        const int r0 = i0 + tg * 64 + q * 16;
        if (col >= args.n || r0 >= args.m) continue;
        const size_t e = e00 + tj * ejs + tg * 64;
        uint4 rv[20];
#pragma unroll
        for (unsigned t = 0; t < 20; ++t)
            if (t < c.N) {
                if (OZ2_TAIL_ABL & 1) rv[t] = make_uint4(lane + t, lane * 3 + t, lane * 5 + t, lane * 7 + t);
                else rv[t] = *(const uint4*)(args.out + (size_t)t * args.strideO + e);
            }
        const int sB = (int)c.sftB[col];
        const uint4 sa0 = *(const uint4*)(c.sftA + r0), sa1 = *(const uint4*)(c.sftA + r0 + 8);  // the shift vector is padded to 256 rows
        const unsigned saw[8] = {sa0.x, sa0.y, sa0.z, sa0.w, sa1.x, sa1.y, sa1.z, sa1.w};
        U* Cc = (U*)c.C + (size_t)col * c.ldc + r0;
        const bool aligned = (reinterpret_cast<uintptr_t>(Cc) & 15) == 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // 8 rows at a time
            double Sh[8], Sl[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) Sh[x] = 0.0, Sl[x] = 0.0;
#pragma unroll
            for (unsigned t = 0; t < 20; ++t) {
                if (t < c.N) {
                    const unsigned w0 = h == 0 ? rv[t].x : rv[t].z, w1 = h == 0 ? rv[t].y : rv[t].w;
                    if (OZ2_TAIL_ABL & 4) {
                        Sh[t & 7] += (double)(int)(w0 ^ w1);
                    } else if (use_dd) {
                        const double qh = c.qh[t], ql = c.ql[t];
#pragma unroll
                        for (int x = 0; x < 8; ++x) {
                            const double cd = (double)(int)(int8_t)((x < 4 ? w0 : w1) >> (8 * (x & 3)));
                            Sh[x] = fma(qh, cd, Sh[x]);
                            Sl[x] = fma(ql, cd, Sl[x]);
                        }
                    } else {
                        const double q1 = c.q1[t];
#pragma unroll
                        for (int x = 0; x < 8; ++x) Sh[x] = fma(q1, (double)(int)(int8_t)((x < 4 ? w0 : w1) >> (8 * (x & 3))), Sh[x]);
                    }
                }
            }
            const int rb = r0 + 8 * h;
            const bool full = rb + 8 <= args.m;
            U oldc[8], outv[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) oldc[x] = (U)0;
            if (reads_c) {
                if (full && aligned) {
                    __builtin_memcpy(oldc, Cc + 8 * h, sizeof(oldc));
                } else {
#pragma unroll
                    for (int x = 0; x < 8; ++x)
                        if (rb + x < args.m) oldc[x] = Cc[8 * h + x];
                }
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int sA = (int)(int16_t)(saw[4 * h + (x >> 1)] >> (16 * (x & 1)));
                const U AB = scalb<U>((U)reduce(Sh[x], Sl[x]), sA + sB);
                switch (mode) {
                case 1: outv[x] = AB; break;
                case 2: outv[x] = oldc[x] + AB; break;
                case 3: outv[x] = -AB; break;
                case 4: outv[x] = oldc[x] - AB; break;
                default: outv[x] = fmaU<U>(be, oldc[x], al * AB); break;
                }
            }
            if (OZ2_TAIL_ABL & 2) {
#pragma unroll
                for (int x = 0; x < 8; ++x) asm volatile("" ::"v"(outv[x]));
            } else if (full && aligned) {
                typedef U VecU __attribute__((ext_vector_type(16 / sizeof(U))));
                constexpr int PER = 16 / (int)sizeof(U);
#pragma unroll
                for (int v = 0; v < 8 / PER; ++v) {
                    VecU ov;
#pragma unroll
                    for (int x = 0; x < PER; ++x) ov[x] = outv[v * PER + x];
                    __builtin_nontemporal_store(ov, (VecU*)(Cc + 8 * h) + v);
                }
            } else {
#pragma unroll
                for (int x = 0; x < 8; ++x)
                    if (rb + x < args.m) Cc[8 * h + x] = outv[x];
            }
        }
    }
}

#ifndef OZ2_PB
#define OZ2_PB 4
#endif
#ifndef OZ2_KBAR_MAX_KP
#define OZ2_KBAR_MAX_KP 4096  // padded k up to which the K-step-barrier schedule is used (see launch<EPI>); 0 = never, 1 << 30 = always
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
// The producer waves of the tile-stationary kernel, OUT OF LINE: as a separate function they get their own register allocation.  Inlined,
// their scalar state (two tile cursors, the CRT block) spilled SGPRs into VGPR lanes, and the two VGPRs the compiler reserves for that
// in the WHOLE kernel pushed accumulator spills into the consumers' MFMA loop (which uses all 168 registers).
template <int FUSE>
__device__ __attribute__((noinline)) void i8_producer_crt(__attribute__((address_space(3))) char* smem3,
                                                          const __attribute__((address_space(4))) char* kernarg) {
    // the kernel-argument segment pointer is not available in a callable function (the builtin yields null there): the kernel passes
    // it; made wave-uniform again (arguments arrive in VGPRs) so that every field is a scalar load
    {
        const unsigned long long v = (unsigned long long)kernarg;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        kernarg = (const __attribute__((address_space(4))) char*)(((unsigned long long)hi << 32) | lo);
    }
    typedef const __attribute__((address_space(4))) GemmArgs* ArgsPtr;
    const auto& args = *(ArgsPtr)kernarg;
    using OutT = std::conditional_t<FUSE == 2, float, double>;
    char* smem = (char*)smem3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT1 = args.kp / BK;
    const int KT = KT1 * args.nseg;
    const int total = args.total_tiles;
    const int G = gridDim.x;
    const int planes_per_tile = args.planes;
#include "oz2_gemm_i8_producer.inc"
        // ---- CRT ON THE PRODUCER WAVES.  The consumers keep their accumulators and their registers; the four
        // producer waves -- one per SIMD, idle between LDS-DMA instructions -- accumulate the CRT of the PREVIOUS output tile beside the
        // MFMAs of the current one: the FP64 chains run on the vector ALUs while the matrix pipes are busy, the re-read of the residue
        // planes (written by this CU during the previous tile) comes from the L2 / Infinity Cache, and the stand-alone CRT pass and its
        // launch disappear.  Unit of work: one column of the 256 x 256 tile (4 rows per lane: 256 contiguous bytes per residue plane,
        // 2 KiB of C per wave); wave pw takes columns pw, pw + 4, ...  A three-stage pipeline moves one unit per K-step:
        //   K-step j, start : stores of unit j - 2; loads of unit j (N residue words, 4 shifts, old C) -- all memory instructions sit
        //                     BEFORE the K-step's LDS-DMA, so the end-of-K-step vmcnt wait (16 for the A waves: loads return in order)
        //                     covers them and never waits for the DMA just issued;
        //   K-step j, slots : the s_sleep pauses between the DMA groups become the FMA chains of unit j - 1 (3 planes per slot);
        //   K-step j, end   : mod-P reduction, scalbn, axpby of unit j - 1; after the barrier the loaded registers change stage.
        // The loads are inline asm (invisible to the compiler's waitcnt insertion, which would otherwise put a vmcnt(0) in front of the
        // first use and stall the DMA pipeline); their destination registers are not touched between issue and the end-of-K-step wait
        // (read-write operands: the value loaded and the value kept when no unit is loaded live in the SAME register, so no copy is
        // placed at the merge before the data has arrived).
        // Tile r's residues may be read one K-step after the consumers' vmcnt(0) + barrier that follow its last epilogue.  At small k the
        // CRT falls behind and finishes in the drain loop after the last K-step, as does the last tile's.
        typedef const __attribute__((address_space(4))) CrtArgs* CrtPtr;
        CrtPtr cp = (CrtPtr)(kernarg + ((sizeof(GemmArgs) + 7) & ~size_t(7)));
        asm volatile("" : "+s"(cp)::"memory");
#define c (*cp)
        using U = OutT;
        const unsigned cN = c.N;
        const bool use_dd = c.use_dd != 0;
        const double cPhi = c.Phi, cPlo = c.Plo, cinvP = c.invP;
        U al = (U)c.alpha[0], be = (U)c.beta[0];
        int mode = c.mode;
        if (mode == 5) {
            al = *(const U*)c.alpha_dev;
            be = *(const U*)c.beta_dev;
            mode = 0;
        }
        const bool reads_c = mode == 0 || mode == 2 || mode == 4;
        const int S = KT * planes_per_tile;  // K-steps per output tile
        int c_vb = blockIdx.x, c_u = 0;      // CRT cursor: tile and next unit (0..63) to load
        TileMap c_map = map_tile(c_vb, total, args.tiles_m, args.tiles_n, args.colblock);
        int g = 0, ready_at = S + 1;         // K-steps completed; K-step from which the cursor tile's residues may be read
        unsigned rvN[20], rvC[20];           // residue words (4 rows): stage "loaded" / stage "accumulating"
        unsigned long long saN = 0, saC = 0; // four int16 shifts of A
        int sBN = 0, sBC = 0;
        v4i ocN[2], ocC[2];                  // old C (4 x U; float uses ocN[0] only)
        U outv[4];
        U *CcN = nullptr, *CcC = nullptr, *CcS = nullptr;  // this lane's first element of C per stage
        int vrN = 0, vrC = 0, vrS = 0;       // its number of valid rows (0..4)
        bool nval = false, cval = false, sval = false;
        double Sh[4], Sl[4];
#pragma unroll
        for (int t = 0; t < 20; ++t) rvN[t] = rvC[t] = 0;
        ocN[0] = ocN[1] = ocC[0] = ocC[1] = v4i{0, 0, 0, 0};
#pragma unroll
        for (int x = 0; x < 4; ++x) outv[x] = (U)0;

#define PCRT_MEM_SLOT()                                                                                                      \
    do {                                                                                                                     \
        asm volatile("" : "+s"(cp));                                                                                         \
        if (sval) {                                                                                                          \
            if (OZ2_PCRT_ABL & 4) {                                                                                          \
                _Pragma("unroll") for (int x = 0; x < 4; ++x) asm volatile("" ::"v"(outv[x]));                               \
            } else if (vrS == 4) {                                                                                                  \
                typedef U VecU __attribute__((ext_vector_type(16 / sizeof(U))));                                             \
                constexpr int PER = 16 / (int)sizeof(U);                                                                     \
                _Pragma("unroll") for (int v = 0; v < 4 / PER; ++v) {                                                        \
                    VecU ov;                                                                                                 \
                    _Pragma("unroll") for (int x = 0; x < PER; ++x) ov[x] = outv[v * PER + x];                               \
                    typedef VecU __attribute__((aligned(sizeof(U)))) VecUU;                                                  \
                    __builtin_nontemporal_store(ov, (VecUU*)CcS + v);                                                        \
                }                                                                                                            \
            } else {                                                                                                         \
                _Pragma("unroll") for (int x = 0; x < 4; ++x) if (x < vrS) CcS[x] = outv[x];                                 \
            }                                                                                                                \
            sval = false;                                                                                                    \
        }                                                                                                                    \
        nval = false;                                                                                                        \
        if (c_vb < total && g >= ready_at) {                                                                                 \
            const int col_ = c_map.tn * BN + 4 * c_u + pw;                                                                   \
            const int i0_ = c_map.tm * BM;                                                                                   \
            if (col_ < args.n) {                                                                                             \
                const int8_t* ub_ = uniform(args.out + (size_t)col_ * args.ldo + i0_);                                       \
                const unsigned vo_ = (unsigned)lane * 4u;                                                                    \
                _Pragma("unroll") for (unsigned t = 0; t < 20; ++t) if (t < cN) {                                            \
                    const int8_t* pb_ = uniform(ub_ + (size_t)t * args.strideO);                                             \
                    if (OZ2_PCRT_ABL & 2) rvN[t] = vo_ * 2654435761u + t;                                                     \
                    else asm volatile("global_load_dword %0, %1, %2 sc0" : "+v"(rvN[t]) : "v"(vo_), "s"(pb_) : "memory");    \
                }                                                                                                            \
                const int8_t* sp_ = uniform((const int8_t*)(c.sftA + i0_));                                                  \
                const unsigned so_ = (unsigned)lane * 8u;                                                                    \
                asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(saN) : "v"(so_), "s"(sp_) : "memory");                  \
                int sw_;                                                                                                     \
                const int8_t* sq_ = uniform((const int8_t*)c.sftB + 4 * (col_ >> 1));                                        \
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sw_) : "s"(sq_) : "memory");          \
                sBN = (int)(int16_t)((col_ & 1) ? (sw_ >> 16) : sw_);                                                         \
                const int left_ = (int)args.m - (i0_ + 4 * lane);                                                            \
                vrN = left_ < 0 ? 0 : left_ > 4 ? 4 : left_;                                                                 \
                CcN = (U*)c.C + (size_t)col_ * c.ldc + (size_t)(i0_ + 4 * lane);                                             \
                if (reads_c) {                                                                                               \
                    if (vrN == 4) {                                                                                          \
                        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(ocN[0]) : "v"(CcN) : "memory");                \
                        if (sizeof(U) == 8) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "+v"(ocN[1]) : "v"(CcN) : "memory"); \
                    } else {                                                                                                 \
                        _Pragma("unroll") for (int x = 0; x < 4; ++x) if (x < vrN) {                                         \
                            if (sizeof(U) == 8) {                                                                            \
                                unsigned long long w_;                                                                       \
                                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(w_) : "v"(CcN + x) : "memory");        \
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* ragged last tile row only */             \
                                ocN[x >> 1][2 * (x & 1)] = (int)w_, ocN[x >> 1][2 * (x & 1) + 1] = (int)(w_ >> 32);           \
                            } else {                                                                                         \
                                unsigned w_;                                                                                 \
                                asm volatile("global_load_dword %0, %1, off" : "=v"(w_) : "v"(CcN + x) : "memory");          \
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
                                ocN[0][x] = (int)w_;                                                                         \
                            }                                                                                                \
                        }                                                                                                    \
                    }                                                                                                        \
                }                                                                                                            \
                nval = true;                                                                                                 \
            }                                                                                                                \
            if (++c_u == 64) {                                                                                               \
                c_u = 0;                                                                                                     \
                c_vb += G;                                                                                                   \
                ready_at += S;                                                                                               \
                if (c_vb < total) c_map = map_tile(c_vb, total, args.tiles_m, args.tiles_n, args.colblock);                                 \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#define PCRT_ACC(slot_)                                                                                                      \
    do {                                                                                                                     \
        asm volatile("" : "+s"(cp)); /* the table doubles of this slot only: hoisted, the 60 of them spill SGPRs into VGPR lanes */ \
        if ((slot_) == 0) {                                                                                                  \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) Sh[x] = 0.0, Sl[x] = 0.0;                                          \
        }                                                                                                                    \
        _Pragma("unroll") for (unsigned t = 3 * (slot_); t < 3 * (slot_) + 3 && t < 20; ++t) if (t < cN) {                   \
            const unsigned w_ = rvC[t];                                                                                      \
            if (use_dd) {                                                                                                    \
                const double qh_ = c.qh[t], ql_ = c.ql[t];                                                                   \
                _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                              \
                    const double cd_ = (double)(int)(int8_t)(w_ >> (8 * x));                                                 \
                    Sh[x] = fma(qh_, cd_, Sh[x]);                                                                            \
                    Sl[x] = fma(ql_, cd_, Sl[x]);                                                                            \
                }                                                                                                            \
            } else {                                                                                                         \
                const double q1_ = c.q1[t];                                                                                  \
                _Pragma("unroll") for (int x = 0; x < 4; ++x) Sh[x] = fma(q1_, (double)(int)(int8_t)(w_ >> (8 * x)), Sh[x]); \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#define PCRT_FIN()                                                                                                           \
    do {                                                                                                                     \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                                      \
            const double qq_ = rint(cinvP * Sh[x]);                                                                          \
            const double R_ = use_dd ? fma(cPlo, qq_, fma(cPhi, qq_, Sh[x]) + Sl[x]) : fma(cPhi, qq_, Sh[x]);                 \
            const int sA_ = (int)(int16_t)(saC >> (16 * x));                                                                 \
            const U AB_ = scalb<U>((U)R_, sA_ + sBC);                                                                        \
            U old_;                                                                                                          \
            if (sizeof(U) == 8) {                                                                                            \
                const unsigned long long b_ = (unsigned long long)(unsigned)ocC[x >> 1][2 * (x & 1)] |                       \
                                              ((unsigned long long)(unsigned)ocC[x >> 1][2 * (x & 1) + 1] << 32);            \
                __builtin_memcpy(&old_, &b_, sizeof(U) == 8 ? 8 : 4);                                                        \
            } else {                                                                                                         \
                const unsigned b_ = (unsigned)ocC[0][x];                                                                     \
                __builtin_memcpy(&old_, &b_, sizeof(U) == 8 ? 4 : 4);                                                        \
            }                                                                                                                \
            switch (mode) {                                                                                                  \
            case 1: outv[x] = AB_; break;                                                                                    \
            case 2: outv[x] = old_ + AB_; break;                                                                             \
            case 3: outv[x] = -AB_; break;                                                                                   \
            case 4: outv[x] = old_ - AB_; break;                                                                             \
            default: outv[x] = fmaU<U>(be, old_, al * AB_); break;                                                           \
            }                                                                                                                \
        }                                                                                                                    \
        CcS = CcC, vrS = vrC, sval = true;                                                                                   \
    } while (0)
#define PCRT_ROTATE()                                                                                                        \
    do {                                                                                                                     \
        if (nval) {                                                                                                          \
            _Pragma("unroll") for (int t = 0; t < 20; ++t) {                                                                 \
                asm volatile("" : "+v"(rvN[t]));                                                                             \
                rvC[t] = rvN[t];                                                                                             \
            }                                                                                                                \
            asm volatile("" : "+v"(saN), "+v"(ocN[0]), "+v"(ocN[1]));                                                        \
            saC = saN, sBC = sBN, ocC[0] = ocN[0], ocC[1] = ocN[1], CcC = CcN, vrC = vrN;                                    \
        }                                                                                                                    \
        cval = nval;                                                                                                         \
        ++g;                                                                                                                 \
    } while (0)

        for (int vb = blockIdx.x; vb < total; vb += G) {
            for (int kt = 0; kt < S; ++kt) {
                const bool issued = more;
                // K-steps without CRT work (most of them at large k: the CRT of a tile takes 64 + 2 of its planes * KT K-steps) run the
                // compact loop body of the plain kernel: the CRT body is ~50 KiB of straight-line code, and walking through it every
                // K-step evicted the consumers' MFMA loop from the instruction cache the CU pair shares (+10 % kernel time at every k)
                if ((OZ2_PCRT_ABL & 1) || !(cval || sval || (c_vb < total && g >= ready_at))) {
                    if (issued) {
                        PRODUCER_BEGIN();
                        if (isB) {
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) PRODUCER_DMA(fsrc, gq * 4 + q, fdst);
                                __builtin_amdgcn_s_sleep(OZ2_SLEEP_B);
                            }
                        } else {
#pragma unroll
                            for (int gq = 0; gq < 8; ++gq) {
#pragma unroll
                                for (int q = 0; q < 2; ++q) PRODUCER_DMA(fsrc, gq * 2 + q, fdst);
                                __builtin_amdgcn_s_sleep(OZ2_SLEEP_A);
                            }
                        }
                        PRODUCER_ADVANCE();
                    }
                    if (!isB && issued) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    ++g;
                    continue;
                }
                PCRT_MEM_SLOT();
                if (issued) PRODUCER_BEGIN();
#pragma unroll
                for (int gq = 0; gq < 8; ++gq) {
                    if (issued) {
                        if (isB) {
                            if (gq < 4) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) PRODUCER_DMA(fsrc, gq * 4 + q, fdst);
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < 2; ++q) PRODUCER_DMA(fsrc, gq * 2 + q, fdst);
                        }
                    }
                    if (cval) {
                        if (gq < 7) PCRT_ACC(gq);
                    } else if (isB ? gq < 4 : true) {
                        if (isB) __builtin_amdgcn_s_sleep(OZ2_SLEEP_B);
                        else __builtin_amdgcn_s_sleep(OZ2_SLEEP_A);
                    }
                }
                if (cval) PCRT_FIN();
                if (issued) PRODUCER_ADVANCE();
                if (!isB && issued) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                PCRT_ROTATE();
            }
        }
        // the consumers finish the last tile's epilogue, wait for their stores and meet the producers here; then drain the pipeline
        __builtin_amdgcn_s_barrier();
        g = 0x3fffffff;
        while (!(OZ2_PCRT_ABL & 1) && (c_vb < total || cval || sval)) {
            PCRT_MEM_SLOT();
            if (cval) {
#pragma unroll
                for (int gq = 0; gq < 7; ++gq) PCRT_ACC(gq);
                PCRT_FIN();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PCRT_ROTATE();
            g = 0x3fffffff;
        }
#undef PCRT_MEM_SLOT
#undef PCRT_ACC
#undef PCRT_FIN
#undef PCRT_ROTATE
#undef c
#include "oz2_gemm_i8_producer_undef.inc"
}

// FUSE != 0 (EPI_MOD only; 1: double, 2: float output): TILE-STATIONARY order -- a workgroup runs all args.planes residue planes of one
// output tile back to back (at any moment the 256 workgroups still work on the same few planes of a chunk of 256 tiles, so the L2 /
// Infinity Cache sharing of map_tile is unchanged) and then accumulates the CRT for that tile itself (i8_crt_tail).
struct NoCrt {
    int unused;
};
template <int EPI, bool KBAR, int FUSE>
__global__ void __launch_bounds__(WS_THREADS) gemm_i8_kernel(const GemmArgs args, const std::conditional_t<FUSE != 0, CrtArgs, NoCrt> crt) {
    static_assert(FUSE == 0 || EPI == EPI_MOD, "the CRT tail follows the real requantise epilogue");
    using OutT = std::conditional_t<FUSE == 2, float, double>;
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
        if constexpr (KBAR && FUSE != 0) {
            i8_producer_crt<FUSE>((__attribute__((address_space(3))) char*)smem,  // CRT on the producer waves: out of line (see there)
                                  (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr());
            return;
        }
        if (OZ2_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(3);
#include "oz2_gemm_i8_producer.inc"
        if constexpr (KBAR) {
        // ONE workgroup barrier per K-step (see the consumer branch).  Without per-segment barriers to pace them the producers space
        // their instructions with s_sleep (64 clocks per unit): bursts of LDS-DMA cost (all 16 at once: +7 % kernel time).
        for (int vb = blockIdx.x; vb < total; vb += G) {
            for (int kt = 0; kt < KT * planes_per_tile; ++kt) {
                const bool issued = more && (!(OZ2_PROBE_LDS & 4) || vb == (int)blockIdx.x);  // probe bit 2: DMA during the first tile only
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
                const bool issued = more && (!(OZ2_PROBE_LDS & 4) || vb == (int)blockIdx.x);  // probe bit 2: DMA during the first tile only
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
    // distributed residues where v_mfma_i32_32x32x32_i8 holds 3.45 (tools/ubench/mfma_shapes.hip, profiles/r02_mfma_shapes.txt):
    // the same MACs with a quarter of the accumulator registers read and written per instruction.  Wave tile 128 x 64 = 8 x 4
    // accumulator tiles (128 registers); a K-step (128 bytes) is four segments (K half ks2) x (row half ah) of 16 MFMAs: the B
    // fragments of a K half are loaded in its first segment and kept for the second, the A fragments of 64 rows per segment.
    const int wm = wave >> 2, wn = wave & 3;
    if (OZ2_PRIO_MODE == 2 && wm == 1) __builtin_amdgcn_s_setprio(1);
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
        for (int vb = blockIdx.x; vb < total; vb += G) {
            const TileMap tmap = map_tile(vb, total, args.tiles_m, args.tiles_n, args.colblock);
            for (int pl = 0; pl < planes_per_tile; ++pl) {
            v4i acc[8][4];
            v4i af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = (EPI == EPI_MAX || !OZ2_RED_DOT4) ? 0 : args.acc0;
#define OZ2_LOAD_SEG(seg_)                                                                                                   \
    do {                                                                                                                     \
        const int coff_ = (((((seg_) >> 1) << 2) | q) ^ sw) << 4;                                                            \
        if (((seg_) & 1) == 0) {                                                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[j] = *(const v4i*)(curB + j * 16 * BK + coff_);                 \
        }                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = *(const v4i*)(curA + (((seg_) & 1) * 4 + i) * 16 * BK + coff_); \
    } while (0)
#define OZ2_MMA_SEG(seg_)                                                                                                    \
    do {                                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        if (OZ2_PRIO_MODE == 0 || OZ2_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(1);                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                          \
            acc[((seg_) & 1) * 4 + i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[((seg_) & 1) * 4 + i][j], 0, 0, 0); \
        if (OZ2_PRIO_MODE == 0 || OZ2_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(0);                                         \
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
            for (; kt < kt_end; ++kt) {
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
                    OZ2_MMA_SEG(seg);
                }
                if (!WM1) {
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef OZ2_SET_PANELS
#undef OZ2_LOAD_SEG
#undef OZ2_MMA_SEG
#if OZ2_PROBE_LDS & 8
            (void)tmap;  // probe bit 3: no epilogue; the accumulators stay live
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
            i8_epilogue<EPI>(acc, args, FUSE ? PlaneRef{0, pl} : plane_ref(args, tmap.plane), tmap.tm * BM + (WM1 ? 128 : 0), tmap.tn * BN + wn * 64, lane);
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
        const TileMap tmap = map_tile(vb, total, args.tiles_m, args.tiles_n, args.colblock);
        for (int pl = 0; pl < planes_per_tile; ++pl) {
        v4i acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (EPI == EPI_MAX || !OZ2_RED_DOT4) ? 0 : args.acc0;

        int kt = 0;  // phases: see the K-step-barrier branch
        const int nph = (EPI == EPI_MAX && args.kt_mid > 0) ? 2 : 1;
        for (int ph = 0; ph < nph; ++ph) {
        const int kt_end = (EPI == EPI_MAX && ph + 1 < nph) ? args.kt_mid : KT;
        for (; kt < kt_end; ++kt) {
            const char* curA = smem + sA * TILE_BYTES + a_base;
            const char* curB = smem + (sA == 4 ? 0 : sA + 1) * TILE_BYTES + b_base;
            sA = sA + 2 >= 5 ? sA - 3 : sA + 2;
#if OZ2_PROBE_LDS
            v4i af[4], bf[4];  // timing probe only (wrong results): fragments re-read only in the first segment (bit 0: B, bit 1: A)
#endif
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const int coff = (((ks2 << 2) | q) ^ sw) << 4;
#if !OZ2_PROBE_LDS
                v4i bf[4];
#endif
#pragma unroll
                for (int ah = 0; ah < 2; ++ah) {
#if !OZ2_PROBE_LDS
                    v4i af[4];
#endif
                    if (ah == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (!(OZ2_PROBE_LDS & 1) || ks2 == 0) bf[j] = *(const v4i*)(curB + j * 16 * BK + coff);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (!(OZ2_PROBE_LDS & 2) || (ks2 == 0 && ah == 0)) af[i] = *(const v4i*)(curA + (ah * 4 + i) * 16 * BK + coff);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (OZ2_PRIO_MODE == 0 || OZ2_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[ah * 4 + i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[ah * 4 + i][j], 0, 0, 0);
                    if (OZ2_PRIO_MODE == 0 || OZ2_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#if OZ2_PROBE_LDS & 8
        (void)tmap;  // probe bit 3: no epilogue; the accumulators stay live
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
        i8_epilogue<EPI>(acc, args, FUSE ? PlaneRef{0, pl} : plane_ref(args, tmap.plane), tmap.tm * BM + wm * 128, tmap.tn * BN + wn * 64, lane);
#endif
        }  // phase
        }
        if constexpr (FUSE != 0) i8_crt_tail<OutT>(args, tmap.tm * BM + wm * 128, tmap.tn * BN + wn * 64, lane);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    }
}

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

template <int EPI, bool KBAR, int FUSE = 0>
static hipError_t launch_sched(hipStream_t stream, GemmArgs& a, const std::conditional_t<FUSE != 0, CrtArgs, NoCrt>& crt = {}) {
    // the attribute belongs to the function on ONE device; setting it is idempotent, so concurrent first calls from several host
    // threads only need the flag itself to be race-free
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 0;
    if (!attr_set_dev[dev_].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_kernel<EPI, KBAR, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set_dev[dev_].store(true, std::memory_order_release);
    }
    // persistent: one workgroup per CU; a multiple of 8 keeps "workgroup b runs on XCD b % 8" aligned with map_tile.
    // Measured interleaved against one-workgroup-per-tile launches of the same kernel (tools/gemm_ab.py, 8192 x 8192 x k,
    // 14 planes): 13 % faster at k = 256, 7 % at k = 2048, 1 % at k = 8192.
    int grid = num_cus() & ~7;
    if (grid <= 0) grid = 8;
    if (a.total_tiles < grid) grid = a.total_tiles;
    hipLaunchKernelGGL((gemm_i8_kernel<EPI, KBAR, FUSE>), dim3(grid), dim3(WS_THREADS), RING_LDS_BYTES, stream, a, crt);
    return hipGetLastError();
}

// Two schedules of the same tile loop (bit-identical results): with a workgroup barrier after every LOAD / MFMA segment (ping-pong)
// or with one barrier per K-step.  Interleaved on one box (tools/gemm_ab.py, 8192^2 x k, 14 planes, profiles/r02_sched_ab.txt): the
// K-step-barrier form is faster by 10 / 7.5 / 4.7 / 2.4 % at k = 512 / 1024 / 2048 / 4096 (tile boundaries -- epilogue beside the
// other half's first K-steps -- overlap better) and 2.4 % slower at k = 8192, where the board is power-bound and removing stalls
// buys nothing while the s_sleep-paced LDS-DMA is a little less smooth than the barrier-paced one.
template <int EPI> static hipError_t launch(hipStream_t stream, GemmArgs& a, int planes) {
    // batched call: the items' planes are one long plane sequence (item-major), the persistent tile loop runs over all of them
    a.ppi = planes;
    a.bstride = g_batch.ws;
    planes *= (int)g_batch.batch;
    a.total_tiles = planes * a.tiles_m * a.tiles_n;
    if (a.total_tiles <= 0) return hipSuccess;
    a.colblock = map_colblock((size_t)a.tiles_n, (size_t)a.kp * (size_t)a.nseg);
    a.acc0 = (OZ2_RED_SMALL && (size_t)a.kp * (size_t)a.nseg <= 512) ? 0 : (int)0x80000000u;
    // (the bound GEMM keeps the ping-pong schedule at every k.  In round 2 its K-step-barrier instantiation spilled accumulators INSIDE
    // the MFMA loop; with the round-3 source it no longer does, but the single-plane launch still runs slower with it: bounds phase
    // 88.4 -> 92.4 us at 3072^3, 130.9 -> 134.6 at 4096^3, equal at 2048^3 and 8192^3.  OZ2_MAX_KBAR=1 restores it for A/B runs)
    if ((EPI != EPI_MAX || OZ2_MAX_KBAR) && a.kp * a.nseg <= OZ2_KBAR_MAX_KP) return launch_sched<EPI, true>(stream, a);
    return launch_sched<EPI, false>(stream, a);
}

// Non-temporal residue stores.  The residue planes are written once and read once, by the CRT pass, N planes later; written with the
// default policy they are allocated in the 256 MiB Infinity Cache on their way to HBM and push out the operand planes A_lo / B_lo,
// which the quantise kernels have just left there and which every plane's 32 x 32 tiles re-read through eight L2s.  When ALL operand
// planes of the launch fit the Infinity Cache, keeping them there is worth 8-14 % of the WHOLE call (8192^2: k = 256 / 512 / 1024
// 0.917 -> 0.815 / 1.059 -> 0.940 / 1.470 -> 1.285 ms; 16384^2: k = 256 / 512 3.22 -> 2.88 / 3.88 -> 3.59 ms); when they do not fit
// there is nothing to protect and the CRT pass loses the tail of C_mid it would have found in the cache (-0.5 ... -2 % from k = 1536
// at 8192^2, k = 1024 at 16384^2); with small outputs (4096^2 and below) it is a wash.  profiles/r03_epi_nt_grid.txt
static int nt_residue_planes(const GemmArgs& a, int planes, bool stream_out, bool automatic = true) {
    const char* e = getenv("GEMMUL8_EPI_NT");  // testing switch, read per launch: 0 / 1 forces the policy for all planes (results are identical)
    if (e && (e[0] == '0' || e[0] == '1') && !e[1]) return e[0] == '1' && stream_out ? planes : 0;
    if (!stream_out || !automatic) return 0;
    const size_t all = (size_t)planes * g_batch.batch;
    const size_t operands = all * (a.strideA + a.strideB), residues = all * a.strideO;
    // (keeping the default policy for the last 2-6 planes, so that the CRT finds them in the cache, non-temporal stores for the
    // leading planes of launches whose operands do NOT fit, and walking the planes last to first -- the order the quantise kernels
    // left them in the cache -- were all measured: no consistent gain, profiles/r03_epi_nt_keep.txt; between 240 and ~450 MiB of
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

// Tile-stationary GEMM + requantise + CRT in one launch (real types, all N moduli).  Worth it when the tiles of ONE plane fill the
// chip about as well as the tiles of all planes do: the unit of work per workgroup is N times larger.
bool gemm_i8_crt_fusable(size_t m, size_t n, unsigned N) {
    const long tiles = (long)((m + BM - 1) / BM) * (long)((n + BN - 1) / BN);
    long grid = num_cus() & ~7;
    if (grid <= 0) grid = 8;
    const long rounds_fused = (tiles + grid - 1) / grid * (long)N;    // tile-times on the busiest workgroup
    const long rounds_plain = (tiles * (long)N + grid - 1) / grid;
    return tiles >= grid && rounds_fused * 100 <= rounds_plain * 104;
}

hipError_t launch_gemm_i8_mod_crt(hipStream_t stream, int dtype, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp,
                                  size_t m, size_t n, unsigned N, int8_t* out, size_t ldo, size_t strideO, const int16_t* sftA,
                                  const int16_t* sftB, const void* alpha, const void* beta, bool scalars_on_device, void* C, size_t ldc,
                                  int variant) {
    if (dtype != kF64 && dtype != kF32) return hipErrorInvalidValue;
    GemmArgs a{};
    a.A[0] = A;
    a.B[0] = B;
    a.nseg = 1;
    a.strideA = strideA;
    a.strideB = strideB;
    a.t_begin = 0;
    a.out = out;
    a.ldo = ldo;
    a.strideO = strideO;
    fill_common(a, kp, m, n);
    a.planes = (int)N;
    a.ppi = (int)N;
    a.total_tiles = a.tiles_m * a.tiles_n;
    if (a.total_tiles <= 0) return hipSuccess;
    a.colblock = map_colblock((size_t)a.tiles_n, (size_t)a.kp);
    a.acc0 = (OZ2_RED_SMALL && (size_t)a.kp <= 512) ? 0 : (int)0x80000000u;
    CrtArgs c{};
    c.m = m;
    c.n = n;
    c.sftA = sftA;
    c.sftB = sftB;
    c.C = C;
    c.ldc = ldc;
    fill_crt_tables(c, dtype, kINT8, N);
    fill_crt_scalars(c, dtype, alpha, beta, scalars_on_device);
    // variant 1: CRT on the producer waves beside the next tile's MFMAs (K-step-barrier schedule at every k); variant 2: CRT tail on
    // the consumer waves after each tile (ping-pong schedule; kept as the measured baseline of DESIGN.md 3.4)
    const bool kbar = variant != 2;
    if (dtype == kF64) return kbar ? launch_sched<EPI_MOD, true, 1>(stream, a, c) : launch_sched<EPI_MOD, false, 1>(stream, a, c);
    return kbar ? launch_sched<EPI_MOD, true, 2>(stream, a, c) : launch_sched<EPI_MOD, false, 2>(stream, a, c);
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
    a.nt_planes = nt_residue_planes(a, t_end - t_begin, true, OZ2_CPLX_NT);
    return launch<EPI_CPLX>(stream, a, t_end - t_begin);
}

#ifndef OZ2_MAX_SMALL_TILES
#define OZ2_MAX_SMALL_TILES 128  // the bound GEMM takes the 128 x 128-tile kernel when (batch x) its 256 x 256 tiles number at most this (of 256 CUs): bounds phase 33 -> 28 / 41 -> 32 / 66 -> 55 us at 512^3 / 1024^3 / 2048^3, but 98 -> 111 us at 3072^3 (144 tiles), profiles/r03_bound_ab.txt
#endif
// mid_seg > 0: the row / column maxima are taken twice per tile -- of the partial sums after the first mid_seg K-segments and of the
// full sums.  The complex bound (max over the elements of C1 = ArBi + AiBr and of C1 + C0, C0 = (Ar-Ai)(Br-Bi)) is then ONE launch over
// three segments with mid_seg = 2 (three real GEMMs of work) instead of a 2-segment and a 3-segment launch (five).
hipError_t launch_gemm_i8_max(hipStream_t stream, int nseg, const int8_t* const* A, const int8_t* const* B, size_t kp, size_t m, size_t n,
                              int* rowmax, int* colmax, int mid_seg) {
    {
        // GEMMUL8_BOUND_TILE = 128 | 256 forces one kernel (tests run every case through both; the maxima are identical)
        const char* force = getenv("GEMMUL8_BOUND_TILE");
        const size_t tiles = ((m + BM - 1) / BM) * ((n + BN - 1) / BN) * g_batch.batch;
        const bool small = force && force[0] == '1' ? true : force && force[0] == '2' ? false : tiles <= (size_t)OZ2_MAX_SMALL_TILES;
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
