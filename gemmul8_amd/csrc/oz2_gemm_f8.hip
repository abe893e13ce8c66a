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
// profiles/r02_mfma_shapes.txt).  One instruction consumes a whole 128-byte K-step of 16 rows: lane l holds row l & 15 and the
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

namespace oz2 {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// EPI_FINAL_CPLX: like EPI_FINAL for the third complex part Z = (Ar+Ai)(Br+Bi), then (Cr, Ci) = (X - Y, Z - X - Y) mod p with
// the residues X, Y of the first two parts -> interleaved int16 pairs (conv_hi2mid_complex.hpp:28-41).
// EPI_FB1/2/3: the three bound GEMMs of the complex accurate mode (find_max.hpp:117-140,218-251, complex FP8): u = fma_ru(ku, c, c);
//   1: store u (ArBi)   2: store add_ru(stored, u) (+ AiBr = s12)   3: s0 from c = (|Ar|-|Ai|)(|Br|-|Bi|) and s12, maxima of max(s0, s12)
//   Stage 3, reference (args.cplx_rule = 0): s0 = add_ru(fma_ru(ku, c, c), s12).  Default here (cplx_rule = 1):
//   s0 = add_ru(fma_ru(ku, add_ru(|c|, 2 s12), c), s12) -- c is a sum of products of BOTH signs, so the engine's truncation error on it
//   scales with the sum of the MAGNITUDES of its terms (<= T + C1, T = the bound sought, C1 <= s12), not with |c|: see bound_ku below
enum { EPI_PART = 0, EPI_FINAL = 1, EPI_FMAX = 2, EPI_FINAL_CPLX = 3, EPI_FB1 = 4, EPI_FB2 = 5, EPI_FB3 = 6 };

struct F8Args {
    const int8_t* A;      // base of the A planes; plane of block b at A + planeA[b]*strideA
    const int8_t* B;
    size_t strideA, strideB;
    int planeA[20], planeB[20];
    int planeA2[20], planeB2[20];  // nseg == 2: operand planes of the second K segment (K-concatenation: C0 + C1 in ONE accumulator)
    int nseg;                      // 1, or 2: virtual K = 2 kp -- exact while 2 k * 256 <= 2^24 (launch_gemm_f8 checks)
    int nres;                      // EPI_FINAL / EPI_FINAL_CPLX: residue planes combined with the accumulator: 2 (r0, r1), or 1 (r0 = residue of C0 + C1)
    int kp, m, n, tiles_m, tiles_n;
    int colblock;  // tile-columns per column block of the tile walk (map_colblock; 0 = full width)
    TileMapArgs map;  // the same with the divisors' magic numbers (make_tile_map)
    int t_begin;          // block b <-> modulus t_begin + b
    int16_t* out;         // EPI_PART: scratch plane b at out + b*strideO; EPI_FINAL: C_mid plane (t_begin+b) likewise
    size_t ldo, strideO;
    const int16_t* r0;    // EPI_FINAL: residues of C0, C1 (plane b at r0/r1 + b*strideR)
    const int16_t* r1;
    size_t strideR;
    const int16_t* rx;    // EPI_FINAL_CPLX: residues of the complex parts X, Y (plane b at rx/ry + b*strideR)
    const int16_t* ry;
    float* fbuf;          // EPI_FB*: m x n float scratch, leading dimension ldo
    int* rowmax;          // EPI_FMAX (float bit patterns)
    int* colmax;
    float ku;             // bound inflation, see bound_ku (reference: (k+1) * 2^-24)
    float kabs;           // absolute part of the bound inflation (bound-plane units), see bound_kabs; 0 with the reference's formula
    int cplx_rule;        // EPI_FB3: 1 = inflate the mixed-sign product by ku (|c| + 2 s12) (default), 0 = by ku c as the reference does
    int total_tiles;      // planes * tiles_m * tiles_n
    int ppi;              // planes per batch item (plane p = item p / ppi, item-relative plane p % ppi); = all planes for one GEMM
    unsigned m_ppi;       // floor(2^32 / ppi) (map_magic)
    size_t bstride;       // bytes between the workspaces of consecutive batch items (every pointer above lives in the workspace)
    int moduli[20];
    int sqrtp[6];
};

// plane p of a (batched) launch: byte offset of its item's workspace and its plane index inside the item (as in oz2_gemm_i8.hip)
struct F8Plane {
    size_t boff;
    int tt;
};
__device__ __forceinline__ F8Plane f8_plane(const F8Args& args, int plane) {
    const int p = __builtin_amdgcn_readfirstlane(plane);
    unsigned b, tt;
    udivmod_magic((unsigned)p, (unsigned)args.ppi, args.m_ppi, b, tt);
    return {(size_t)b * args.bstride, (int)tt};
}

__device__ __forceinline__ unsigned pack16(int a, int b) { return ((unsigned)a & 0xFFFFu) | ((unsigned)b << 16); }

constexpr int F8_THREADS = 512;
// Laboratory hook points (neutral here; a probe build of tools/build_probes.sh defines them through tools/experiments/probes/lab_hooks.hpp; the
// product build -- -DOZ2_PRODUCT_BUILD -- refuses OZ2_LAB_HOOKS): see oz2_gemm_i8_epi.hpp
#if defined(OZ2_PRODUCT_BUILD) && defined(OZ2_LAB_HOOKS)
#error "laboratory switches (OZ2_LAB_*) are not allowed in the product build of libgemmul8.so: use tools/experiments/"
#endif
#ifdef OZ2_LAB_HOOKS
#include OZ2_LAB_HOOKS
#endif
#ifndef OZ2_HOOK_DMA_ON
#define OZ2_HOOK_DMA_ON(first_tile) true
#endif
#ifndef OZ2_HOOK_KSTEP
#define OZ2_HOOK_KSTEP(kin) (kin)
#endif

// int16 residue epilogues (EPI_PART / EPI_FINAL / EPI_FINAL_CPLX) of a wave's 128 x 64 accumulator block.  The accumulators are exact integers
// (|c| <= 2^24): one exact FP64 quotient step (five full-rate instructions); the combined value (|v| < 2^18) needs one fp32 step.
// ONE reduction form for every modulus (round 4): q = ceil(x / p - 1/2), r = x - q p, the representative in (-p/2, p/2].  For odd p that is the
// symmetric residue (x / p - 1/2 is never an integer; its distance from one is >= 1/(2p) = 4.6e-4, the evaluation errors are 1e-12 in FP64 and
// 4.5e-5 in fp32 for |v| < 2^18: CPU models in tests/test_residue_math.py), for p = 1024 -- the only even FP8 modulus, where the arithmetic is
// exact -- it keeps the reference's representative +512 of the tie.  Rounds 1-3 chose between an odd and an even form per tile at run time; the
// structurizer lays such a choice out as a straight line of predicated blocks, which keeps the accumulators live through the first form's
// whole epilogue (oz2_gemm_i8_epi.hpp, i8_epilogue).
// EPI_FINAL / EPI_FINAL_CPLX load the partial residues of earlier launches.  Vector-memory operations of a wave complete in issue order, so the
// loads are issued per sub-block right behind its reduction -- all of them in flight behind the residue arithmetic, none behind a store -- and
// waited for ONCE; then the combination and the stores.  (Sub-block by sub-block -- load, wait, combine, store -- every sub-block paid a full
// memory round trip: eight to sixteen serialised latencies per tile.)
template <int EPI>
__device__ __forceinline__ void f8_epilogue_mod(const v4f (&acc)[8][4], const F8Args& args, F8Plane pl, int i0, int j0, int lane) {
    const int c16 = lane & 15;
    const int q = lane >> 4;
    const int plane = pl.tt;
    const int t = args.t_begin + plane;
    // batch item: every plane pointer moves by the item's workspace offset (bytes)
    int16_t* const out_ = (int16_t*)((char*)args.out + pl.boff);
    const int16_t* const r0_ = (const int16_t*)((const char*)args.r0 + pl.boff);
    const int16_t* const r1_ = (const int16_t*)((const char*)args.r1 + pl.boff);
    const int16_t* const rx_ = (const int16_t*)((const char*)args.rx + pl.boff);
    const int16_t* const ry_ = (const int16_t*)((const char*)args.ry + pl.boff);
    const int p = args.moduli[t];
    // value = k0*R0 + k1*R1 + k2*R2:  square moduli s*(R0+R1) + R2;  Karatsuba 256*R0 + 16*(R2-R0-R1) + R1
    const int k0 = t < 6 ? args.sqrtp[t < 6 ? t : 0] : 240;
    const int k1 = t < 6 ? k0 : -15;
    const int k2 = t < 6 ? 1 : 16;
    const float pf = (float)p, invp = 1.0f / pf;
    const double pd = (double)p, invpd = 1.0 / pd;
    auto red_acc = [&](float c) -> int {
        const double x = (double)c;
        return (int)fma(-ceil(fma(x, invpd, -0.5)), pd, x);
    };
    auto red_small = [&](int v) -> int {
        const float vf = (float)v;
        return (int)fmaf(-ceilf(fmaf(vf, invp, -0.5f)), pf, vf);
    };
    // int16 residues of the 64 x 16 sub-block (tj, tg): z[0..7] = this lane's 16 consecutive rows (first row i0 + 64 tg + 16 q) of column j0 + 16 tj + c16
    auto reduce_block = [&](int tj, int tg, unsigned (&z)[8]) {
        unsigned d[4][2];  // tile ti of the group: this lane quad's rows 4 q .. 4 q + 3 as 4 x int16
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            int r[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) r[b] = red_acc(acc[tg * 4 + ti][tj][b]);
            d[ti][0] = pack16(r[0], r[1]);
            d[ti][1] = pack16(r[2], r[3]);
        }
        // 4 x 4 transpose over the lane quads (bits 5, 4) as in oz2_gemm_i8.hip: afterwards quad q holds the 16 consecutive rows
        // 64 tg + 16 q .. + 15 (tile ti = q): rows 4 s .. 4 s + 3 from source quad s in z[2 s], z[2 s + 1]
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(d[0][w], d[2][w], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(d[1][w], d[3][w], false, false);
            const auto w01 = __builtin_amdgcn_permlane16_swap(s0[0], s1[0], false, false);
            const auto w23 = __builtin_amdgcn_permlane16_swap(s0[1], s1[1], false, false);
            z[0 + w] = w01[0];  // rows 0-3
            z[2 + w] = w01[1];  // rows 4-7
            z[4 + w] = w23[0];  // rows 8-11
            z[6 + w] = w23[1];  // rows 12-15
        }
    };
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const size_t po = (size_t)plane * args.strideO, pr = (size_t)plane * args.strideR;
    if constexpr (EPI == EPI_PART || EPI == EPI_FINAL_CPLX) {
        // EPI_FINAL_CPLX keeps the sub-block-by-sub-block form: five plane pointers and 64 more registers of X / Y residues beside the kernel's DMA
        // state do not fit the two-pass form below (it was built: 290-750 bytes of scratch, reloads between the stores)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const int col = j0 + tj * 16 + c16;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                unsigned z[8];
                reduce_block(tj, tg, z);
                if (col < args.n) {
                    const size_t e = (size_t)col * args.ldo + i0 + tg * 64 + q * 16;
                    if constexpr (EPI == EPI_PART) {
                        v4u* dst = (v4u*)(out_ + po + e);
                        dst[0] = v4u{z[0], z[1], z[2], z[3]};
                        dst[1] = v4u{z[4], z[5], z[6], z[7]};
                    } else {
                        const v4u* p0 = (const v4u*)(r0_ + pr + e);
                        const v4u x0 = p0[0], x1 = p0[1];
                        v4u y0 = v4u{0, 0, 0, 0}, y1 = y0;
                        if (args.nres == 2) {  // wave-uniform; nres == 1: r0 holds the residue of C0 + C1, R1 = 0
                            const v4u* p1 = (const v4u*)(r1_ + pr + e);
                            y0 = p1[0], y1 = p1[1];
                        }
                        const unsigned xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                        const unsigned ys[8] = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            int o[2];
#pragma unroll
                            for (int hlf = 0; hlf < 2; ++hlf) {
                                const int a0 = (int)(int16_t)(xs[w] >> (16 * hlf)), a1 = (int)(int16_t)(ys[w] >> (16 * hlf)), a2 = (int)(int16_t)(z[w] >> (16 * hlf));
                                o[hlf] = red_small(k0 * a0 + k1 * a1 + k2 * a2);
                            }
                            z[w] = pack16(o[0], o[1]);
                        }
                        const v4u* px = (const v4u*)(rx_ + pr + e);
                        const v4u* py = (const v4u*)(ry_ + pr + e);
                        const v4u cx0 = px[0], cx1 = px[1], cy0 = py[0], cy1 = py[1];
                        const unsigned cxs[8] = {cx0[0], cx0[1], cx0[2], cx0[3], cx1[0], cx1[1], cx1[2], cx1[3]};
                        const unsigned cys[8] = {cy0[0], cy0[1], cy0[2], cy0[3], cy1[0], cy1[1], cy1[2], cy1[3]};
                        unsigned o[16];  // 16 rows x (Cr, Ci) int16 pairs
#pragma unroll
                        for (int w = 0; w < 8; ++w)
#pragma unroll
                            for (int hlf = 0; hlf < 2; ++hlf) {
                                const int xv = (int)(int16_t)(cxs[w] >> (16 * hlf)), yv = (int)(int16_t)(cys[w] >> (16 * hlf)), zv = (int)(int16_t)(z[w] >> (16 * hlf));
                                o[2 * w + hlf] = pack16(red_small(xv - yv), red_small(zv - xv - yv));
                            }
                        v4u* dc = (v4u*)(out_ + po + 2 * e);
#pragma unroll
                        for (int w = 0; w < 4; ++w) dc[w] = v4u{o[4 * w], o[4 * w + 1], o[4 * w + 2], o[4 * w + 3]};
                    }
                }
            }
        }
    } else {
        unsigned z[8][8];
        v4u R0[8][2], R1[8][2];
        // pass 1: reduce; the partial residues of a sub-block are requested as soon as its accumulators are dead.  Columns beyond n read the
        // last existing column instead (their results are never stored): no branch around the loads.
        auto request = [&](int tj, int tg, size_t ec) {
            const int sb = 2 * tj + tg;
            const v4u* p0 = (const v4u*)(r0_ + pr + ec + tg * 64);
            R0[sb][0] = p0[0], R0[sb][1] = p0[1];
            R1[sb][0] = R1[sb][1] = v4u{0, 0, 0, 0};
            if (args.nres == 2) {  // wave-uniform; nres == 1: r0 holds the residue of C0 + C1, R1 = 0
                const v4u* p1 = (const v4u*)(r1_ + pr + ec + tg * 64);
                R1[sb][0] = p1[0], R1[sb][1] = p1[1];
            }
        };
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const size_t ec = (size_t)min(j0 + tj * 16 + c16, args.n - 1) * args.ldo + i0 + q * 16;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                reduce_block(tj, tg, z[2 * tj + tg]);
                request(tj, tg, ec);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        // pass 2: residue of k0 R0 + k1 R1 + k2 R2 in place
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const int col = j0 + tj * 16 + c16;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                const int sb = 2 * tj + tg;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    int o[2];
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const int a0 = (int)(int16_t)(R0[sb][w >> 2][w & 3] >> (16 * hlf)), a1 = (int)(int16_t)(R1[sb][w >> 2][w & 3] >> (16 * hlf)),
                                  a2 = (int)(int16_t)(z[sb][w] >> (16 * hlf));
                        o[hlf] = red_small(k0 * a0 + k1 * a1 + k2 * a2);
                    }
                    z[sb][w] = pack16(o[0], o[1]);
                }
                if (col < args.n) {
                    v4u* dst = (v4u*)(out_ + po + (size_t)col * args.ldo + i0 + tg * 64 + q * 16);
                    dst[0] = v4u{z[sb][0], z[sb][1], z[sb][2], z[sb][3]};
                    dst[1] = v4u{z[sb][4], z[sb][5], z[sb][6], z[sb][7]};
                }
            }
        }
    }
}

// EPI_FMAX / EPI_FB*: u = fma_ru(ku, c, c) (find_max.hpp:82-96: the (k+1)*2^-24 inflation covers the FP32 accumulation
// error of the inexact bound products); FMAX and FB3 then reduce row / column maxima (atomicMax on the bit patterns of
// non-negative floats).
template <int EPI>
__device__ __forceinline__ void f8_epilogue_bound(v4f (&acc)[8][4], const F8Args& args, F8Plane pl, int i0, int j0, int lane) {
    const int c16 = lane & 15;
    const int q = lane >> 4;
    const float ku = args.ku, kabs = args.kabs;
    float* const fbuf_ = (float*)((char*)args.fbuf + pl.boff);
    int* const rowmax_ = (int*)((char*)args.rowmax + pl.boff);
    int* const colmax_ = (int*)((char*)args.colmax + pl.boff);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        const int col = j0 + tj * 16 + c16;
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
            float u[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) u[b] = __fadd_ru(__fmaf_ru(ku, acc[ti][tj][b], acc[ti][tj][b]), kabs);
            if constexpr (EPI != EPI_FMAX) {
                float4* fp = (float4*)(fbuf_ + (size_t)col * args.ldo + i0 + ti * 16 + 4 * q);  // this lane's 4 consecutive rows
                if (col < args.n) {
                    if constexpr (EPI == EPI_FB1) {
                        *fp = make_float4(u[0], u[1], u[2], u[3]);
                    } else {
                        const float4 w = *fp;
                        const float ws[4] = {w.x, w.y, w.z, w.w};
                        if constexpr (EPI == EPI_FB2) {
                            *fp = make_float4(__fadd_ru(ws[0], u[0]), __fadd_ru(ws[1], u[1]), __fadd_ru(ws[2], u[2]), __fadd_ru(ws[3], u[3]));
                        } else {
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const float c = acc[ti][tj][b];
                                const float up = args.cplx_rule ? __fadd_ru(__fmaf_ru(ku, __fadd_ru(fabsf(c), __fadd_ru(ws[b], ws[b])), c), kabs) : u[b];
                                const float s0 = __fadd_ru(up, ws[b]);
                                u[b] = s0 > ws[b] ? s0 : ws[b];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[ti][tj][b] = u[b];
        }
    }
    if constexpr (EPI == EPI_FMAX || EPI == EPI_FB3) {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            float cm = 0.0f;
#pragma unroll
            for (int ti = 0; ti < 8; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + ti * 16 + 4 * q + r;
                    cm = fmaxf(cm, (row < args.m) ? acc[ti][tj][r] : 0.0f);
                }
            cm = fmaxf(cm, __shfl_xor(cm, 16));
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            const int col = j0 + tj * 16 + c16;
            if (q == 0 && col < args.n && cm > 0.0f) atomicMax(colmax_ + col, __float_as_int(cm));
        }
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
            int w[4];  // bit patterns of non-negative floats order like ints
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = 0.0f;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    const int col = j0 + tj * 16 + c16;
                    v = fmaxf(v, (col < args.n) ? acc[ti][tj][r] : 0.0f);
                }
                w[r] = __float_as_int(v);
            }
            tile_rowmax_atomic16(w, rowmax_, i0 + ti * 16, args.m, lane);
        }
    }
}

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
        // Measured on config 3 (18 GEMMs, interleaved builds, profiles/r02_f8_ab.txt): this kernel 54.9 ms, the round-1 32x32x64 kernel
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
        if (wm == 1) __builtin_amdgcn_s_barrier();
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
                    // and waits behind it (config 3 whole call +1.25 %, SGEMM 8192^2 x 2048 / 8192 +1.2 / +0.7 %: profiles/r04e_f8_late_wait_ab.txt)
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
                    if (ah == 1 && ISB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
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
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef F8_SET_TILE
#undef F8_DMA
#undef F8_FETCH_BEGIN
#undef F8_FETCH_ADVANCE
}

static void fill_common(F8Args& a, size_t kp, size_t m, size_t n) {
    a.kp = (int)kp;
    a.m = (int)m;
    a.n = (int)n;
    a.tiles_m = (int)((m + BM - 1) / BM);
    a.tiles_n = (int)((n + BN - 1) / BN);
    for (int t = 0; t < 20; ++t) {
        const int p = GEMMUL8_MODULI_FP8[t];
        a.moduli[t] = p;
    }
    for (int t = 0; t < 6; ++t) a.sqrtp[t] = GEMMUL8_SQRT_MODULI_FP8[t];
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

// first low-precision plane of modulus t: 2 planes for t < 6 (hi, lo), 3 afterwards (hi, lo, hi+lo)  (table.hpp:69-75)
static int first_plane(int t) { return t < 6 ? 2 * t : 12 + 3 * (t - 6); }

// which = 0,1: partial products C0 / C1 -> int16 residue scratch; which = 2: C2 with the final combine -> out (C_mid plane, or
// a scratch plane holding the residue of a complex part); which = 3: C2 of the third complex part with the complex combine
// (rx, ry = residues of X and Y, same stride as r0/r1) -> interleaved (Cr, Ci) int16 pairs in out.
hipError_t launch_gemm_f8(hipStream_t stream, int which, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                          size_t n, int t_begin, int t_end, int16_t* out, size_t ldo, size_t strideO, const int16_t* r0, const int16_t* r1,
                          size_t strideR, const int16_t* rx, const int16_t* ry) {
    F8Args a{};
    a.A = A;
    a.B = B;
    a.strideA = strideA;
    a.strideB = strideB;
    a.t_begin = t_begin;
    a.out = out;
    a.ldo = ldo;
    a.strideO = strideO;
    a.r0 = r0;
    a.r1 = r1;
    a.strideR = strideR;
    a.rx = rx;
    a.ry = ry;
    // which = 4: C0 + C1 of the square moduli (t < 6) as ONE GEMM over the K-concatenation [Ahi | Alo] x [Blo ; Bhi] -> one int16 residue
    //            plane and one epilogue instead of two (exact in the FP32 accumulators while 2 k * 16 * 16 <= 2^24: the caller checks k);
    // which = 5 / 6: the final GEMM C2 = Alo * Blo combined with that single residue, value = s R0 + C2 (5), and the same with the
    //            complex combine of which = 3 behind it (6).  Bit-identical to the three-GEMM form (s (R0' + R1') + R2 == s R0 + R2 mod p).
    const bool concat = which == 4, single = which == 5 || which == 6;
    const int wh = (which == 3 || single) ? 2 : which;
    a.nseg = concat ? 2 : 1;
    a.nres = single ? 1 : 2;
    for (int t = t_begin; t < t_end; ++t) {
        const int q = first_plane(t), b = t - t_begin;
        if (concat) {  // (t < 6 only) segment 1: C0 = Ahi * Blo, segment 2: C1 = Alo * Bhi
            if (t >= 6) return hipErrorInvalidValue;
            a.planeA[b] = q, a.planeB[b] = q + 1;
            a.planeA2[b] = q + 1, a.planeB2[b] = q;
        } else if (t < 6) {  // hi = q, lo = q+1 :  C0 = Ahi*Blo, C1 = Alo*Bhi, C2 = Alo*Blo
            a.planeA[b] = wh == 0 ? q : q + 1;
            a.planeB[b] = wh == 1 ? q : q + 1;
        } else {      // C0 = hi*hi, C1 = lo*lo, C2 = (hi+lo)*(hi+lo)
            if (single) return hipErrorInvalidValue;
            a.planeA[b] = q + wh;
            a.planeB[b] = q + wh;
        }
        if (!concat) a.planeA2[b] = a.planeA[b], a.planeB2[b] = a.planeB[b];
    }
    fill_common(a, kp, m, n);
    const int planes = t_end - t_begin;
    if (which == 3 || which == 6) return launch<EPI_FINAL_CPLX>(stream, a, planes);
    return (which == 2 || which == 5) ? launch<EPI_FINAL>(stream, a, planes) : launch<EPI_PART>(stream, a, planes);
}

// Inflation of the accurate-mode bound sums.  The reference uses ku = (k+1) * 2^-24, a bound on IEEE FP32 summation
// (find_max.hpp:82-96).  v_mfma_scale_f32_16x16x128_f8f6f4 does not sum like that (tools/ubench/f8_accum.hip,
// profiles/r02_f8_mfma_accumulation.txt): inside a group of 8 products everything is aligned to the group's largest product and
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
    fill_common(a, kp, m, n);
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
    fill_common(a, kp, m, n);
    if (stage == 1) return launch<EPI_FB1>(stream, a, 1);
    if (stage == 2) return launch<EPI_FB2>(stream, a, 1);
    return launch<EPI_FB3>(stream, a, 1);
}

}  // namespace oz2
