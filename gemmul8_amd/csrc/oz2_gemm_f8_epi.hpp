// Argument block and epilogues of the FP8-backend MFMA GEMM kernels, shared by the e4m3 kernel (oz2_gemm_f8.hip) and the FP6 kernel
// (oz2_gemm_f6.hip: the same integers of magnitude <= 16 as e2m3 codes at twice the matrix rate).  See oz2_gemm_f8.hip for the algebra.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
// EPI_FUSED / EPI_FUSED_CPLX (round 5, FP6 kernel): the three products of a modulus as ONE tile loop over three K segments -- between the segments the
//   accumulators are replaced by m x (their loose residue), so that the tile ends with  gam x acc == value (mod p): no partial-residue planes at all.
enum { EPI_PART = 0, EPI_FINAL = 1, EPI_FMAX = 2, EPI_FINAL_CPLX = 3, EPI_FB1 = 4, EPI_FB2 = 5, EPI_FB3 = 6, EPI_FUSED = 7, EPI_FUSED_CPLX = 8 };

struct F8Args {
    const int8_t* A;      // base of the A planes; plane of block b at A + planeA[b]*strideA
    const int8_t* B;
    size_t strideA, strideB;
    int planeA[20], planeB[20];
    int planeA2[20], planeB2[20];  // nseg >= 2: operand planes of the second K segment (K-concatenation: C0 + C1 in ONE accumulator)
    int planeA3[20], planeB3[20];  // nseg == 3 (EPI_FUSED*): operand planes of the third K segment
    float m1[20], m2[20];          // EPI_FUSED*: accumulator <- m x loose residue behind segment 1 / 2 (f8_fill_planes)
    int gam[20];                   //             value == gam x final accumulator (mod p)
    int nseg;                      // 1; 2: virtual K = 2 kp -- exact while 2 k * 256 <= 2^24 (launch_gemm_f8 checks); 3: EPI_FUSED*
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
#ifndef OZ2_HOOK_SKIP_AH
#define OZ2_HOOK_SKIP_AH 0  // FP6 kernel: 1 = a third of the fragment reads (rows 64-127 of A) not issued (timing probe; stale registers)
#endif
#ifndef OZ2_HOOK_SKIP_EPILOGUE
#define OZ2_HOOK_SKIP_EPILOGUE 0
#endif
#if defined(OZ2_PRODUCT_BUILD) && (OZ2_HOOK_SKIP_AH || OZ2_HOOK_SKIP_EPILOGUE)
#error "timing probes (OZ2_HOOK_SKIP_*) compute something else: not allowed in the product build of libgemmul8.so"
#endif

// int16 residue epilogues (EPI_PART / EPI_FINAL / EPI_FINAL_CPLX) of a wave's 128 x 64 accumulator block.  The accumulators are exact integers
// (|c| <= 2^24): a loose three-instruction fp32 residue (red_acc below); the combined value (|v| < 2^18) needs one fp32 step for the canonical one.
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
    // Accumulators (exact integers, |c| <= 2^24): a LOOSE residue in three fp32 instructions (round 5; rounds 1-4: one FP64 quotient step, five
    // instructions at a quarter of the rate).  q = rint(RN32(c RN32(1/p))) is within 2^-7 of c / p, r = fma(-q, p, c) is formed EXACTLY (q p < 2^26
    // inside the fma, the result is a small integer): r == c (mod p), |r| <= (1/2 + 2^-7) p.  A loose residue is all these uses need -- the int16
    // scratch planes of the partial products and the input of the combination k0 R0 + k1 R1 + k2 R2 (|.| <= 271 * 0.508 * 511 < 2^17), whose own
    // reduction red_small yields the canonical representative; C_mid is unchanged bit for bit (CPU model: tests/test_residue_math.py).
    auto red_acc = [&](float c) -> int { return (int)fmaf(-rintf(c * invp), pf, c); };
    auto red_small = [&](int v) -> int {
        const float vf = (float)v;
        return (int)fmaf(-ceilf(fmaf(vf, invp, -0.5f)), pf, vf);
    };
    constexpr bool FUSED = EPI == EPI_FUSED || EPI == EPI_FUSED_CPLX;
    [[maybe_unused]] const int gam = args.gam[t];  // FUSED: the canonical residue is that of gam x (loose residue of the accumulator), |.| <= 16 * 0.508 p
    // int16 residues of the 64 x 16 sub-block (tj, tg): z[0..7] = this lane's 16 consecutive rows (first row i0 + 64 tg + 16 q) of column j0 + 16 tj + c16
    auto reduce_block = [&](int tj, int tg, unsigned (&z)[8]) {
        unsigned d[4][2];  // tile ti of the group: this lane quad's rows 4 q .. 4 q + 3 as 4 x int16
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            int r[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) r[b] = FUSED ? red_small(gam * red_acc(acc[tg * 4 + ti][tj][b])) : red_acc(acc[tg * 4 + ti][tj][b]);
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
    if constexpr (EPI == EPI_PART || EPI == EPI_FINAL_CPLX || FUSED) {
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
                    if constexpr (EPI == EPI_PART || EPI == EPI_FUSED) {
                        v4u* dst = (v4u*)(out_ + po + e);
                        dst[0] = v4u{z[0], z[1], z[2], z[3]};
                        dst[1] = v4u{z[4], z[5], z[6], z[7]};
                    } else {
                        if constexpr (EPI == EPI_FINAL_CPLX) {  // (EPI_FUSED_CPLX: z already holds the canonical residue of the part Z)
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


// ---- host side: argument block of a residue-GEMM launch, shared by the e4m3 and the FP6 kernels
inline void f8_fill_common(F8Args& a, size_t kp, size_t m, size_t n) {
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


// first low-precision plane of modulus t: 2 planes for t < 6 (hi, lo), 3 afterwards (hi, lo, hi+lo)  (table.hpp:69-75)
inline int f8_first_plane(int t) { return t < 6 ? 2 * t : 12 + 3 * (t - 6); }

// Operand planes and residue inputs of launch `which` over moduli [t_begin, t_end); returns non-zero for a combination that does not exist.
//   which = 0, 1: partial products C0 / C1;  2: C2 with the final combine;  3: C2 of the third complex part with the complex combine;
//   which = 4: C0 + C1 of the square moduli (t < 6) as ONE GEMM over the K-concatenation [Ahi | Alo] x [Blo ; Bhi] -> one int16 residue
//              plane and one epilogue instead of two (exact in the FP32 accumulators while 2 k * 16 * 16 <= 2^24: the caller checks k);
//   which = 5 / 6: the final GEMM C2 = Alo * Blo combined with that single residue, value = s R0 + C2 (5), and the same with the
//              complex combine of which = 3 behind it (6).  Bit-identical to the three-GEMM form (s (R0' + R1') + R2 == s R0 + R2 mod p).
inline int f8_fill_planes(F8Args& a, int which, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m, size_t n,
                          int t_begin, int t_end, int16_t* out, size_t ldo, size_t strideO, const int16_t* r0, const int16_t* r1, size_t strideR,
                          const int16_t* rx, const int16_t* ry) {
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
    const bool concat = which == 4, single = which == 5 || which == 6, fused = which == 7 || which == 8;
    const int wh = (which == 3 || single) ? 2 : which;
    a.nseg = fused ? 3 : concat ? 2 : 1;
    a.nres = single ? 1 : 2;
    for (int t = t_begin; t < t_end; ++t) {
        const int q = f8_first_plane(t), b = t - t_begin;
        if (fused) {
            // which = 7 / 8 (FP6 kernel): all three products of a modulus in ONE tile loop over three K segments (no partial-residue planes): behind
            // segment 1 the accumulators become m1 x (their loose residue), behind segment 2 m2 x ..., and value == gam x acc (mod p) at the end:
            //   squares (t < 6):  value = s (C0 + C1) + C2:   segments Ahi Blo | Alo Bhi | Alo Blo,      m1 = 1, m2 = s, gam = 1
            //   Karatsuba:        value = 240 C0 - 15 C1 + 16 C2 = 16 (m2 (-16 C0 + C1) + C2), m2 == -15 / 16 (mod p):
            //                     segments hi hi | lo lo | (hi+lo)(hi+lo),                               m1 = -16, gam = 16
            // Every accumulator stays an exact integer below 2^24 while kp * 256 + 255 * 0.508 p <= 2^24 (kp <= 65024: the caller checks).
            const int p = GEMMUL8_MODULI_FP8[t];
            if (t < 6) {
                a.planeA[b] = q, a.planeB[b] = q + 1;
                a.planeA2[b] = q + 1, a.planeB2[b] = q;
                a.planeA3[b] = q + 1, a.planeB3[b] = q + 1;
                a.m1[b] = 1.0f, a.m2[b] = (float)GEMMUL8_SQRT_MODULI_FP8[t], a.gam[t] = 1;
            } else {
                a.planeA[b] = q, a.planeB[b] = q;
                a.planeA2[b] = q + 1, a.planeB2[b] = q + 1;
                a.planeA3[b] = q + 2, a.planeB3[b] = q + 2;
                int inv16 = 1;
                while ((16 * inv16) % p != 1) ++inv16;
                int m2 = (int)(((long long)(p - 15) * inv16) % p);  // -15 / 16 mod p
                if (m2 > p / 2) m2 -= p;
                a.m1[b] = -16.0f, a.m2[b] = (float)m2, a.gam[t] = 16;
            }
            continue;
        }
        if (concat) {  // (t < 6 only) segment 1: C0 = Ahi * Blo, segment 2: C1 = Alo * Bhi
            if (t >= 6) return -1;
            a.planeA[b] = q, a.planeB[b] = q + 1;
            a.planeA2[b] = q + 1, a.planeB2[b] = q;
        } else if (t < 6) {  // hi = q, lo = q+1 :  C0 = Ahi*Blo, C1 = Alo*Bhi, C2 = Alo*Blo
            a.planeA[b] = wh == 0 ? q : q + 1;
            a.planeB[b] = wh == 1 ? q : q + 1;
        } else {      // C0 = hi*hi, C1 = lo*lo, C2 = (hi+lo)*(hi+lo)
            if (single) return -1;
            a.planeA[b] = q + wh;
            a.planeB[b] = q + wh;
        }
        if (!concat) a.planeA2[b] = a.planeA[b], a.planeB2[b] = a.planeB[b];
        a.planeA3[b] = a.planeA2[b], a.planeB3[b] = a.planeB2[b];
    }
    f8_fill_common(a, kp, m, n);
    return 0;
}

}  // namespace oz2
