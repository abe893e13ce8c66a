// INT8 GEMM kernels, shared pieces: the kernel-argument block and the fused epilogues on a wave's 128 x 64 accumulator block (requantise
// EPI_MOD / complex combine EPI_CPLX / bound maxima EPI_MAX).  Used by the persistent 256 x 256-tile kernel (oz2_gemm_i8.hip) and by the
// short-K kernel (oz2_gemm_i8_shortk.hip).  Replaces src/conv_hi2mid_real.hpp:9-25, src/conv_hi2mid_complex.hpp:9-127 and the maxima passes
// src/scaling_accu_real.hpp:142-226, src/scaling_accu_complex.hpp:132-224 of the reference: the INT32 accumulators never leave registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "oz2_gemm_common.hpp"
#include "oz2_kernels.h"

namespace oz2 {

// ---- Laboratory boundary.  The INT8 GEMM translation units (oz2_gemm_i8.hip, oz2_gemm_i8_shortk.hip) are the PRODUCT: they instantiate exactly the kernels gemmul8_gemm can reach and carry
// no timing ablation.  Laboratory builds (tools/experiments/: real-data timing probes, the in-kernel CRT forms) compile a second TU that
// defines OZ2_LAB_* and #includes this file; the shipped Makefile passes -DOZ2_PRODUCT_BUILD, which refuses every such macro, so no
// -D in EXTRA can turn libgemmul8.so into a library that computes something else.
#if defined(OZ2_PRODUCT_BUILD) && (defined(OZ2_LAB_HOOKS) || defined(OZ2_LAB_FUSED_CRT) || defined(OZ2_LAB_SHORTK))
#error "laboratory switches (OZ2_LAB_*) are not allowed in the product build of libgemmul8.so: use tools/experiments/"
#endif
#ifdef OZ2_LAB_HOOKS
#include OZ2_LAB_HOOKS  // tools/experiments/probes/lab_hooks.hpp: redefines the three hook points below (timing probes on real data)
#endif
#ifndef OZ2_HOOK_DMA_ON
#define OZ2_HOOK_DMA_ON(first_tile) true  // producers: issue the LDS-DMA of this K-step
#endif
#ifndef OZ2_HOOK_KSTEP
#define OZ2_HOOK_KSTEP(kin) (kin)         // producers: K-step of the segment whose panel is fetched
#endif
#ifndef OZ2_HOOK_SKIP_EPILOGUE
#define OZ2_HOOK_SKIP_EPILOGUE 0          // consumers: 1 = keep the accumulators live, no epilogue
#endif

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
// Hook: called as hook(s, 0) before and hook(s, 1) after the stores of sub-block s = 2 tj + tg (64 rows x 16 columns; 8 per wave tile); NoHook
// (nothing) in every product kernel -- the laboratory short-K kernel places its workgroup barriers there (tools/experiments/shortk).
struct NoHook {
    __device__ __forceinline__ void operator()(int, int) const {}
};
template <int EPI, int RED, typename Hook = NoHook>
__device__ __forceinline__ void i8_epilogue_mod(const v4i (&acc)[8][4], const GemmArgs& args, PlaneRef pl, int i0, int j0, int lane, Hook hook = {}) {
    const int c16 = lane & 15;
    const int q = lane >> 4;
    const int t = args.t_begin + pl.tt;
    const int p = args.moduli[t];
    const int pinv = args.pinv32[t];
    const float invp = 1.0f / (float)p;
    [[maybe_unused]] const unsigned dotw = args.dotw[t], dotc = args.dotc[t];
    auto red = [&](int x) {
        if constexpr (RED == RED_256) return x;  // the accumulator bias 2^31 (GemmArgs.acc0) does not touch the low byte
        else if constexpr (RED == RED_ODD) {
            // the accumulators start at -2^31 (acc init in the kernel): read as unsigned the register holds u = x + 2^31 for ANY int32 sum
            // x, and s = sum_j byte_j(u) (256^j mod p) + ((-2^31) mod p) == x (mod p), 0 <= s < 2^18: v_dot4_u32_u8.  One fp32 quotient
            // and the 24-bit multiply-add give the canonical residue (mod_small_sym_u, oz2_device.hpp).
            return mod_small_sym_u(__builtin_amdgcn_udot4((unsigned)x, dotw, dotc, false), p, invp);
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
        } else return mod_i32_sym((int)((unsigned)x ^ (args.acc0 ? 0x80000000u : 0u)), p, pinv);
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
            hook(2 * tj + tg, 0);
            if (col < args.n) {
                const size_t e = e00 + tj * ejs + tg * 64;  // first of 16 consecutive rows
                if constexpr (EPI == EPI_MOD) {
                    if (pl.tt < args.nt_planes) {  // wave-uniform
                        typedef unsigned v4u __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(v4u{z[0], z[1], z[2], z[3]}, (v4u*)(args.out + po + e));
                    } else {
                        *(uint4*)(args.out + po + e) = make_uint4(z[0], z[1], z[2], z[3]);
                    }
                } else {
                    // eight rows at a time: 8 bytes of X and Y in, 16 bytes of (Cr, Ci) pairs out -- with all 16 rows in flight the epilogue
                    // needed 16 more registers than the 168-VGPR budget leaves beside the accumulators (51-62 spilled registers)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint2 x2 = *(const uint2*)(args.rx + pr + e + 8 * h);
                        const uint2 y2 = *(const uint2*)(args.ry + pr + e + 8 * h);
                        const unsigned xs[2] = {x2.x, x2.y}, ys[2] = {y2.x, y2.y};
                        unsigned o[4];
#pragma unroll
                        for (int w2 = 0; w2 < 2; ++w2) {
                            unsigned lo = 0, hi = 0;
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
                        if (pl.tt < args.nt_planes) {  // wave-uniform
                            typedef unsigned v4u __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store(v4u{o[0], o[1], o[2], o[3]}, (v4u*)(args.out + po + 2 * e) + h);
                        } else {
                            *((uint4*)(args.out + po + 2 * e) + h) = make_uint4(o[0], o[1], o[2], o[3]);
                        }
                    }
                }
            }
            hook(2 * tj + tg, 1);
        }
    }
}

template <int EPI, typename Hook = NoHook>
__device__ __forceinline__ void i8_epilogue(const v4i (&acc)[8][4], const GemmArgs& args, PlaneRef pl, int i0, int j0, int lane, Hook hook = {}) {
    const int c16 = lane & 15;
    const int q = lane >> 4;

    if constexpr (EPI == EPI_MOD || EPI == EPI_CPLX) {
        const int p = args.moduli[args.t_begin + pl.tt];
        if (p == 256) i8_epilogue_mod<EPI, RED_256, Hook>(acc, args, pl, i0, j0, lane, hook);
        else if ((p & 1) && args.acc0 == 0) i8_epilogue_mod<EPI, RED_ODD_SMALL, Hook>(acc, args, pl, i0, j0, lane, hook);
        else if (p & 1) i8_epilogue_mod<EPI, RED_ODD, Hook>(acc, args, pl, i0, j0, lane, hook);
        else i8_epilogue_mod<EPI, RED_GENERIC, Hook>(acc, args, pl, i0, j0, lane, hook);
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


#ifdef OZ2_LAB_SHORTK  // laboratory kernel (tools/experiments/shortk/oz2_gemm_i8_shortk.hip); `a` complete as launch<EPI> of oz2_gemm_i8.hip leaves it
hipError_t launch_gemm_i8_shortk(hipStream_t stream, const GemmArgs& a, int epi);
#endif

}  // namespace oz2
