// Scale / quantise kernels of the Ozaki-II emulation for gfx950 (HBM-bound byte/integer work).
//
// Replaces (paths relative to /root/reference/GEMMul8/src):
//   extract (accurate mode, 7-bit bounds) scaling_accu_real.hpp:23-136, scaling_accu_complex.hpp:6-126
//   accurate-mode shift                   scaling_accu_real.hpp:6-18,142-226
//   fast-mode shift                       scaling_fast_real.hpp:6-49, find_max.hpp:258-341
//   quantise + all-moduli residues        scaling_fast_real.hpp:54-164, scaling_fast_complex.hpp:9-133,
//                                         scaling.hpp:99-280, mod.hpp:8-98,194-355,638-877
//
// Layout in HBM: plane = [rows padded to 256][kp] bytes, row r contiguous in k, zero-filled for
// k <= kk < kp.  A row-strided operand (A with op N, B with op T/C: element (r,kk) at X[kk*ld+r]) is
// read with lanes along r (coalesced), staged RAW through LDS and re-read with lanes along k, so
// that every store is 256 contiguous bytes per wave; a K-major operand needs no staging.  Each
// thread owns 4 consecutive k and emits one dword per plane (the reference's char4 granularity),
// num_moduli is a run-time loop bound (no per-N template instantiation).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "oz2_kernels.h"

namespace oz2 {

ModTable make_mod_table(int backend) {
    ModTable T;
    for (int t = 0; t < 20; ++t) {
        const int p = backend == kINT8 ? GEMMUL8_MODULI_INT8[t] : GEMMUL8_MODULI_FP8[t];
        T.mc[t].p = p;
        T.mc[t].invp = 1.0f / (float)p;
        unsigned c[16];
        long long pw = 1;
        for (int i = 0; i < 16; ++i) {
            c[i] = i < 15 ? (unsigned)(pw % p) : 0u;
            pw = (pw % p) * 256;
        }
        for (int i = 0; i < 4; ++i) {
            unsigned lo = 0, hi = 0;
            for (int j = 0; j < 4; ++j) {
                const unsigned cj = c[4 * i + j];
                lo |= (backend == kINT8 ? cj : (cj & 31u)) << (8 * j);
                hi |= (backend == kINT8 ? 0u : (cj >> 5)) << (8 * j);
            }
            T.mc[t].cb[i] = lo;
            T.mc[t].cbh[i] = hi;
        }
        auto pow2 = [p](int e) {  // 2^e mod p
            long long r = 1;
            for (int i = 0; i < e; ++i) r = (r * 2) % p;
            return (int)r;
        };
        T.mc[t].k56 = (unsigned)((p - pow2(56)) % p);
        T.mc[t].k120 = (unsigned)((p - pow2(120)) % p);
    }
    return T;
}

// ------------------------------------------------------------------ element traits
template <typename T> struct ET;
template <> struct ET<float> {
    using U = float;
    static constexpr bool cplx = false;
    __device__ static double re(float v) { return (double)v; }
    __device__ static double im(float) { return 0.0; }
    __device__ static float zero() { return 0.f; }
};
template <> struct ET<double> {
    using U = double;
    static constexpr bool cplx = false;
    __device__ static double re(double v) { return v; }
    __device__ static double im(double) { return 0.0; }
    __device__ static double zero() { return 0.0; }
};
template <> struct ET<float2> {
    using U = float;
    static constexpr bool cplx = true;
    __device__ static double re(float2 v) { return (double)v.x; }
    __device__ static double im(float2 v) { return (double)v.y; }
    __device__ static float2 zero() { return make_float2(0.f, 0.f); }
};
template <> struct ET<double2> {
    using U = double;
    static constexpr bool cplx = true;
    __device__ static double re(double2 v) { return v.x; }
    __device__ static double im(double2 v) { return v.y; }
    __device__ static double2 zero() { return make_double2(0.0, 0.0); }
};

template <typename U> __device__ __forceinline__ U wave_max(U v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const U o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// ------------------------------------------------------------------ stage kernel (extract / quantise)
enum { MODE_BOUND = 0, MODE_MOD = 1 };

#ifndef OZ2_STAGE_V16
#define OZ2_STAGE_V16 0  // 1: INT8 residue planes leave as 16-byte stores (4 x 4 dword transpose over the lane quads) instead of one dword per
                         // lane and plane.  Measured SLOWER (quantise pair 8192^2 x 14 planes: 702 vs 672 us, profiles/archive/r03_hbm_ab.txt): the kernel is
                         // bound by VALU issue, not by its store pattern, and the transposes add 16 operations per 4 planes -- kept for reference
#endif
#ifndef OZ2_STAGE_VLOAD
#define OZ2_STAGE_VLOAD 1  // K-major operands: a thread's 4 consecutive elements as 16-byte non-temporal loads (672 -> 660 us); 0: element-wise loads
#endif

// 4 x 4 transpose of dwords over a lane quad (lanes 4i .. 4i+3): on return w[j] holds what lane j of the quad had in w[q], q = own lane
// & 3.  Two butterfly stages of quad_perm DPP moves (full rate, no LDS): 16 VALU operations.  The four lanes of a quad own 16
// consecutive k of one row and each produced one dword (4 residues) per plane: after the transpose lane q owns the 16 residues of plane
// t0 + q -- ONE 16-byte store per lane instead of four dword stores.
__device__ __forceinline__ void quad_transpose4(unsigned (&w)[4], unsigned q) {
    const bool o1 = q & 1u, o2 = q & 2u;
#pragma unroll
    for (int p = 0; p < 4; p += 2) {  // exchange with lane ^ 1: pairs (w0, w1), (w2, w3)
        const unsigned snd = o1 ? w[p] : w[p + 1];
        const unsigned rcv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
        w[p] = o1 ? rcv : w[p];
        w[p + 1] = o1 ? w[p + 1] : rcv;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {     // exchange with lane ^ 2: pairs (w0, w2), (w1, w3)
        const unsigned snd = o2 ? w[p] : w[p + 2];
        const unsigned rcv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
        w[p] = o2 ? rcv : w[p];
        w[p + 2] = o2 ? w[p + 2] : rcv;
    }
}

#ifndef OZ2_LOAD_NT
#define OZ2_LOAD_NT 1  // 0: no non-temporal operand loads anywhere.  Measured: no gain where the operands would fit the Infinity Cache (8192^2 x 256 ... 2048,
                       // 4096^2), and -5 % of the whole call at 8192^2 x 1024, where they evict the operand planes the GEMMs are about to read
#endif
// v[0..3] = x[k0 .. k0+3], zero beyond k: 16-byte loads when the four elements exist and start on a 16-byte boundary
// NT: non-temporal (the data is read once); false where the same workgroup re-reads the row right away (two-pass bound extract: with nt
// loads in the maxima pass the second pass went back to HBM -- ZGEMM 8192^3 bounds phase 2.58 -> 2.81 ms)
template <typename T, bool NT = true> __device__ __forceinline__ void load4(const T* x, size_t k0, size_t k, T (&v)[4]) {
    const T* p = x + k0;
    if (OZ2_STAGE_VLOAD && k0 + 4 <= k && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        typedef unsigned V4 __attribute__((ext_vector_type(4)));
        constexpr int NQ = (int)(4 * sizeof(T) / 16);
        V4 r[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) r[i] = (NT && OZ2_LOAD_NT) ? __builtin_nontemporal_load((const V4*)p + i) : ((const V4*)p)[i];
        __builtin_memcpy(v, r, sizeof(r));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k0 + e < k) ? p[e] : ET<T>::zero();
    }
}

struct StageArgs {
    const void* X;
    size_t ld;
    size_t rows, k, kp;
    int8_t* lo;
    size_t plane_stride;  // bytes between moduli planes (MODE_MOD)
    size_t part_stride;   // bytes between Re / Im / Re+Im plane sets
    const int16_t* sft;   // MODE_MOD: negated final shifts
    int16_t* sft0;        // MODE_BOUND: written (maxUFP - ilogb(amax))
    unsigned* zero_p;     // MODE_BOUND (first operand of extract_pair_kernel): the launch also zero-fills zero_words 32-bit words from here (the maxima arrays
    unsigned zero_words;  //   of the bounds phase: the bound GEMM's atomicMax targets) -- instead of a launch of its own
    int16_t* sft0_keep;   // MODE_BOUND: a second copy of sft0 in the scratch region (never overwritten by the final shifts: what the fused finalize reads)
    // MODE_MOD, accurate mode: the shift finalize (scaling_accu_real.hpp:6-18) folded into the quantise launch -- fin_max != nullptr: every workgroup
    // derives its rows' final shifts from (fin_sft0, fin_max) itself, the workgroup of a row's first k chunk stores the negated value to fin_out (= the
    // workspace's sft array, which `sft` then does not have to hold yet).  One launch less per accurate-mode call (4-5 us at launch-bound sizes).
    const int16_t* fin_sft0;
    const int* fin_max;
    int16_t* fin_out;
    float fin_log2P;
    int fin_float_max;
    const void* amax;     // MODE_BOUND, strided: per-row amax bit patterns (U-sized unsigned), amax_parts partial arrays amax_pstride elements apart
    unsigned amax_parts;  //   (round 6: amax_pair_kernel writes one array per k split, no atomics and no zero-fill; the extract takes their maximum)
    size_t amax_pstride;
    int backend;
    int conj;
    int t_begin, t_end;
    int sqrtp[6];
    ModTable mt;
    double pairP[10], pairInvP[10];  // MODE_MOD, float-domain residues: product of the moduli pair (t_begin + 2j, t_begin + 2j + 1) and RN(1 / it)
    size_t bx, bw;        // batched launch (gridDim.z items): bytes between the items' operands X / between their workspaces
    int f6;               // MODE_MOD, FP8 backend: 1 = the planes are FP6 panel images (oz2_gemm_f6.hip), 0 = e4m3 bytes, K-major rows
    unsigned f6_last;     // index of the plane's last 256-row block ...
    unsigned f6_rp_last;  // ... and its rows in the images (A: 256, B: its rows rounded up to 16)
};
// item blockIdx.z of a batched launch: every workspace pointer moves by bw, the operand by bx (both 0 for a single GEMM).  The offsets
// are applied at the few points of use: a modified COPY of the argument block lands in scratch memory (the quantise kernels ran 6x
// slower that way).
#define OZ2_ZW ((size_t)blockIdx.z * a.bw)
#define OZ2_ZX ((size_t)blockIdx.z * a.bx)
__device__ __forceinline__ int fused_final_shift(const StageArgs& a, size_t row, bool writer, size_t zw) {
    const int s0 = ((const int16_t*)((const char*)a.fin_sft0 + zw))[row];
    const int amax = ((const int*)((const char*)a.fin_max + zw))[row];  // INT8: int32 maximum; FP8: bit pattern of a non-negative float
    int f = 0;
    if (amax > 0) {
        const float l = __log2f(a.fin_float_max ? __int_as_float(amax) : __int2float_rn(amax));
        f = __float2int_rd(__fmaf_rd(-0x1.000006p-1f, l, a.fin_log2P));
    }
    const int16_t neg = (int16_t)(-(s0 + f));
    if (writer) ((int16_t*)((char*)a.fin_out + zw))[row] = neg;
    return -(int)neg;
}

// MODE_MOD: the (positive) shift of a row.  Plain form: the workspace's negated final shifts.  Fused finalize (a.fin_max): the same arithmetic as
// shift_finalize_kernel below on (sft0, bound maximum); `writer` (exactly one thread per row over the whole grid) publishes the negated final shift.
#define OZ2_ROW_SHIFT(a_, row_, writer_)                                                                                        \
    ((a_).fin_max == nullptr ? -(int)((const int16_t*)((const char*)(a_).sft + OZ2_ZW))[(row_)] : fused_final_shift((a_), (row_), (writer_), OZ2_ZW))

#ifndef OZ2_BOUND_FLOAT
#define OZ2_BOUND_FLOAT 1
#endif
#ifndef OZ2_STAGE_PAIRLOAD
#define OZ2_STAGE_PAIRLOAD 1  // row-strided kernels, 8-byte elements: two rows per lane and 16-byte load
#endif
#ifndef OZ2_AMAX_PREFETCH
#define OZ2_AMAX_PREFETCH 1  // row-maximum loops keep several loads in flight per thread (experiment switch)
#endif
#ifndef OZ2_STAGE_KCHUNK
#define OZ2_STAGE_KCHUNK 1  // K-major quantise: grid over (row, 1024-wide k chunk) instead of one workgroup looping over a whole row
#endif
#ifndef OZ2_STAGE_RTFAST
#define OZ2_STAGE_RTFAST 1  // row-strided kernels: the row-tile index is the fast grid dimension: the workgroups in flight read whole columns (one sequential
                           // window of the operand) and write 4 adjacent 128-byte runs per row and plane; 0 = k-tile index fastest (quantise A: 314 vs 283 us)
#endif
#if defined(OZ2_PRODUCT_BUILD) && defined(OZ2_PROBE_F6_NOSTORE)
#error "OZ2_PROBE_F6_NOSTORE is a timing probe (the planes are never written): not allowed in the product build of libgemmul8.so"
#endif
#ifndef OZ2_F6_FLOAT_CHAIN
#define OZ2_F6_FLOAT_CHAIN 1  // FP6 panel images: residue, split, codes and packing in fp32 (put_f6_planes_f); 0 = the integer chain of the e4m3 writer
#endif
#ifndef OZ2_STAGE_FLOATRES
#define OZ2_STAGE_FLOATRES 1  // 1: residues from a two-level FLOATING-POINT reduction (below); 0: the byte-wise integer path (v_dot4_u32_u8)
#endif

// FP8 residue planes: residues up to +-544 are split into 2-3 e4m3 planes of integers <= 16 (mod.hpp:159-189, 361-410)
__device__ __forceinline__ void put_fp8_planes(const StageArgs& a, int8_t* o, int t, const int (&rr)[4]) {
    if (t < 6) {
        const int sq = a.sqrtp[t];
        const float inv = 1.0f / (float)sq;
        float hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) fp8_split_sq(rr[e], sq, inv, hi[e], lo[e]);
        *(unsigned*)o = fp8x2_from_floats(hi[0], hi[1]) | (fp8x2_from_floats(hi[2], hi[3]) << 16);
        *(unsigned*)(o + a.plane_stride) = fp8x2_from_floats(lo[0], lo[1]) | (fp8x2_from_floats(lo[2], lo[3]) << 16);
    } else {
        int hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) fp8_split_kara(rr[e], hi[e], lo[e]);
        *(unsigned*)o = fp8x2_from_ints(hi[0], hi[1]) | (fp8x2_from_ints(hi[2], hi[3]) << 16);
        *(unsigned*)(o + a.plane_stride) = fp8x2_from_ints(lo[0], lo[1]) | (fp8x2_from_ints(lo[2], lo[3]) << 16);
        *(unsigned*)(o + 2 * a.plane_stride) =
            fp8x2_from_ints(hi[0] + lo[0], hi[1] + lo[1]) | (fp8x2_from_ints(hi[2] + lo[2], hi[3] + lo[3]) << 16);
    }
}

// FP6 panel images (oz2_gemm_f6.hip): the same integers as e2m3 codes, sign << 5 | |v|, four codes = 24 bits per lane.  A lane quad holds 16
// consecutive k of a row = 96 bits = three dwords of the 24-byte fragment of (row, K group): lane j < 3 of the quad assembles dword j from its
// own 24 bits and its right neighbour's (one quad-permute DPP move) and stores it at its own address o (f6_lane_addr below); lane 3 stores nothing.
__device__ __forceinline__ unsigned f6_code(int v) { return (unsigned)(v < 0 ? 32 - v : v); }
__device__ __forceinline__ void put_f6_word(int8_t* o, const int (&v)[4]) {
    const unsigned w = f6_code(v[0]) | (f6_code(v[1]) << 6) | (f6_code(v[2]) << 12) | (f6_code(v[3]) << 18);
    const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xF9, 0xF, 0xF, true);  // quad_perm [1, 2, 3, 3]
    const unsigned j = threadIdx.x & 3u;
    const unsigned d = (w >> (8u * j)) | (nb << (24u - 8u * j));
    if (j < 3u) *(unsigned*)o = d;
}
__device__ __forceinline__ void put_f6_planes(const StageArgs& a, int8_t* o, int t, const int (&rr)[4]) {
    int hi[4], lo[4];
    if (t < 6) {
        const int sq = a.sqrtp[t];
        const float inv = 1.0f / (float)sq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float h, l;
            fp8_split_sq(rr[e], sq, inv, h, l);
            hi[e] = (int)h, lo[e] = (int)l;
        }
        put_f6_word(o, hi);
        put_f6_word(o + a.plane_stride, lo);
    } else {
        int sm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) fp8_split_kara(rr[e], hi[e], lo[e]), sm[e] = hi[e] + lo[e];
        put_f6_word(o, hi);
        put_f6_word(o + a.plane_stride, lo);
        put_f6_word(o + 2 * a.plane_stride, sm);
    }
}
// byte offset, inside a plane, of the dword that THIS lane stores for the 16 k starting at k0 & ~15 of `row` (lane j = threadIdx.x & 3 holds
// k0 = (k0 & ~15) + 4 j): dword 3 h + j of the fragment (h = the half of the K group), see the image layout in oz2_gemm_f6.hip
__device__ __forceinline__ size_t f6_lane_offset(const StageArgs& a, size_t row, size_t k0) {
    const unsigned tb = (unsigned)(row >> 8), r = (unsigned)row & 255u;
    const unsigned kt = (unsigned)(k0 >> 7), qf = ((unsigned)k0 >> 5) & 3u, h = ((unsigned)k0 >> 4) & 1u;
    const unsigned d = 3u * h + (threadIdx.x & 3u);
    const unsigned rp = tb == a.f6_last ? a.f6_rp_last : 256u;
    const size_t panel = (size_t)tb * 256u * (a.kp / 4 * 3) + (size_t)kt * rp * 96u;
    const unsigned inner = d < 4u ? (qf * rp + r) * 16u + 4u * d : 64u * rp + ((qf >> 1) * 2u * rp + 2u * r + (qf & 1u)) * 8u + 4u * (d - 4u);
    return panel + inner;
}

// The same planes from residues kept as FLOATS (round 5).  The FP6 writer is VALU-bound (config 3: 3.0 ms against ~1.3 ms of memory time) and the integer
// form spends ~20 operations per value and modulus: residue (fma, mad, add), int -> float, split (mul, rint, fma), float -> int twice, two sign-magnitude
// codes (sub, cmp, select each), shifts and ors.  Every quantity here is an integer below 2^24, so the whole chain runs in fp32 with the SAME quotients:
//   residue  q = fma(R, RN(1/p), 1.5 * 2^23) - 1.5 * 2^23 (the very rounding of residue_from_small), r = fma(-q, p, R)              3
//   split    squares: hi = rint(r / s), lo = fma(-s, hi, r) (fp8_split_sq);  Karatsuba: hi = copysign(ceil(|r| / 16), r), lo = fma(-16, hi, r)   3
//   code     c = v < 0 ? 32 - v : v as a float (a negative zero stays a zero)                                                           3 per piece
//   word     ((c3 * 64 + c2) * 64 + c1) * 64 + c0 by three fma (exact: < 2^24), ONE float -> int conversion per four codes               1 per piece
// ~14 operations.  Bit-identical planes (the GPU parity tests and the fuzz sweep compare every code with the oracle's integers).
__device__ __forceinline__ float residue_wide_f(float Rf, const ModConst& mc, float pf) {
    const float q = fmaf(Rf, mc.invp, 12582912.0f) - 12582912.0f;
    float r = fmaf(-q, pf, Rf);
    if (!(mc.p & 1)) r = (r == -0.5f * pf) ? 0.5f * pf : r;  // (uniform) even p: a tie takes +p/2, as residue_from_small
    return r;
}
__device__ __forceinline__ float wrapping_f(float a, float pf, float hf) { return a > hf ? a - pf : (a < -hf ? a + pf : a); }
__device__ __forceinline__ void put_f6_word_f(int8_t* o, const float (&v)[4]) {
    float c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = v[e] < 0.0f ? 32.0f - v[e] : v[e];
    const unsigned w = (unsigned)fmaf(fmaf(fmaf(c[3], 64.0f, c[2]), 64.0f, c[1]), 64.0f, c[0]);
    const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xF9, 0xF, 0xF, true);  // quad_perm [1, 2, 3, 3]
    const unsigned j = threadIdx.x & 3u;
    const unsigned d = (w >> (8u * j)) | (nb << (24u - 8u * j));
#ifdef OZ2_PROBE_F6_NOSTORE  // timing probe: everything but the store
    asm volatile("" ::"v"(d), "v"(o));
#else
    if (j < 3u) *(unsigned*)o = d;
#endif
}
__device__ __forceinline__ void put_f6_planes_f(const StageArgs& a, int8_t* o, int t, const float (&r)[4]) {
    float hi[4], lo[4];
    if (t < 6) {
        const float sq = (float)a.sqrtp[t], inv = 1.0f / sq;
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[e] = rintf(r[e] * inv), lo[e] = fmaf(-sq, hi[e], r[e]);
        put_f6_word_f(o, hi);
        put_f6_word_f(o + a.plane_stride, lo);
    } else {
        float sm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = copysignf(ceilf(fabsf(r[e]) * 0.0625f), r[e]);
            lo[e] = fmaf(-16.0f, hi[e], r[e]);
            sm[e] = hi[e] + lo[e];
        }
        put_f6_word_f(o, hi);
        put_f6_word_f(o + a.plane_stride, lo);
        put_f6_word_f(o + 2 * a.plane_stride, sm);
    }
}

// Quantise + all residues of four consecutive k in the FLOATING-POINT domain.  The quantise kernels are bound by VALU issue, not by HBM
// (profiles/archive/r03_hbm_ab.txt, r03_valu_rates.txt): the integer path spends ~37 operations per element on trunc(x * 2^s) = +-M * 2^E and
// ~7.75 per residue (two v_dot4_u32_u8 over the bytes of M, sign correction, quotient step, packing).  Here:
//   xs = trunc(ldexp(x, s))                       2 FP64 operations per element; exact (a power-of-two scaling, then v_trunc_f64)
//   level 1, per PAIR of moduli, P = p_t * p_t+1:  R = fma(-rint(xs * RN(1/P)), P, xs)      3 FP64 operations per pair
//        q = rint(..) need not be the nearest integer: R = xs - q P is formed EXACTLY by the fma (an integer below 2^53) and
//        R == xs (mod P); |R| <= P/2 + 2 for |xs| < 2^53.  |xs| >= 2^53 (num_moduli > 15 only; wave-uniform test): |R| <= |xs| 2^-52
//        < 2^40, and one more step of the same form brings it to |R| <= P/2 + 1.
//   level 2, per modulus: the single-fma quotient of finish_residue on the SIGNED R (|R| < 2^20.2: exact in fp32):
//        qf = fma(float(R), RN(1/p), 1.5 * 2^23) rounds to 1.5 * 2^23 + q, q = rint(R / p) exactly (|R| * |RN(1/p) - 1/p| < 2^-4.9 / p
//        stays clear of the 1/(2p) tie distance of an odd p; p = 256 / 1024: a tie gives +-p/2, the same byte / fixed below);
//        its low 24 bits are 2^22 + q, so v_mad_i32_i24(bits, -p, R) = (R - q p) - p 2^22: the canonical residue in the low byte,
//        the full value after adding p << 22 (mod 2^32).
// ~10 operations per residue and 2 per element instead of ~14.7 and ~37: the kernels become HBM-bound.  Bit-identical planes
// (tests/test_gpu_parity.py and the fuzz sweep compare every byte with the oracle).
template <bool WIDE> __device__ __forceinline__ int residue_from_small(int Ri, float Rf, const ModConst& mc) {
    const float qf = fmaf(Rf, mc.invp, 12582912.0f);
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(__float_as_int(qf)), "s"(-mc.p), "v"(Ri));
    if constexpr (WIDE) {
        r += (int)((unsigned)mc.p << 22);
        if (!(mc.p & 1)) r = (r == -(mc.p >> 1)) ? (mc.p >> 1) : r;
    }
    return r;  // !WIDE: only the low byte is meaningful (the value is off by p * 2^22)
}

template <typename T> __device__ __forceinline__ void emit4_mod_float(const StageArgs& a, int8_t* out, const T (&v)[4], int s) {
    using E = ET<T>;
    double xr[4], xi[4];
    bool big = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xr[e] = trunc(ldexp(E::re(v[e]), s));
        big |= fabs(xr[e]) >= 0x1.0p53;
        if constexpr (E::cplx) {
            xi[e] = trunc(ldexp(E::im(v[e]), s));
            if (a.conj) xi[e] = -xi[e];
            big |= fabs(xi[e]) >= 0x1.0p53;
        } else {
            xi[e] = 0.0;
        }
    }
    auto run = [&]<bool WIDE, bool BIG>() {
        if constexpr (!WIDE && OZ2_STAGE_V16) {
            // experiment (see OZ2_STAGE_V16): four planes per trip, quad-transposed so that lane q stores the 16 bytes of plane t + q
            typedef unsigned V4 __attribute__((ext_vector_type(4)));
            const unsigned q = threadIdx.x & 3u;
            int8_t* oq = out - 4 * q;
            auto pack = [](const int (&r)[4]) {
                return ((unsigned)r[0] & 0xFFu) | (((unsigned)r[1] & 0xFFu) << 8) | (((unsigned)r[2] & 0xFFu) << 16) | ((unsigned)r[3] << 24);
            };
            for (int t = a.t_begin; t < a.t_end; t += 4) {
                unsigned wr[4] = {0u, 0u, 0u, 0u}, wi[4] = {0u, 0u, 0u, 0u}, ws[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tp = t + 2 * h;
                    if (tp >= a.t_end) break;
                    const double P = a.pairP[(tp - a.t_begin) >> 1], invP = a.pairInvP[(tp - a.t_begin) >> 1];
                    int Rr[4], Ri[4];
                    float Fr[4], Fi[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        double R = fma(-rint(xr[e] * invP), P, xr[e]);
                        if constexpr (BIG) R = fma(-rint(R * invP), P, R);
                        Rr[e] = (int)R, Fr[e] = (float)R;
                        if constexpr (E::cplx) {
                            double I = fma(-rint(xi[e] * invP), P, xi[e]);
                            if constexpr (BIG) I = fma(-rint(I * invP), P, I);
                            Ri[e] = (int)I, Fi[e] = (float)I;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int tt = tp + u;
                        if (tt >= a.t_end) break;
                        const ModConst mc = a.mt.mc[tt];
                        int rr[4], ri[4], rs[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            rr[e] = residue_from_small<false>(Rr[e], Fr[e], mc);
                            if constexpr (E::cplx) {
                                ri[e] = residue_from_small<false>(Ri[e], Fi[e], mc);
                                rs[e] = wrapping((int)(int8_t)rr[e] + (int)(int8_t)ri[e], mc.p);
                            }
                        }
                        wr[2 * h + u] = pack(rr);
                        if constexpr (E::cplx) wi[2 * h + u] = pack(ri), ws[2 * h + u] = pack(rs);
                    }
                }
                quad_transpose4(wr, q);
                if constexpr (E::cplx) quad_transpose4(wi, q), quad_transpose4(ws, q);
                if (t + (int)q < a.t_end) {
                    int8_t* o = oq + (size_t)(t + (int)q) * a.plane_stride;
                    *(V4*)o = V4{wr[0], wr[1], wr[2], wr[3]};
                    if constexpr (E::cplx) {
                        *(V4*)(o + a.part_stride) = V4{wi[0], wi[1], wi[2], wi[3]};
                        *(V4*)(o + 2 * a.part_stride) = V4{ws[0], ws[1], ws[2], ws[3]};
                    }
                }
            }
            return;
        }
        for (int t = a.t_begin; t < a.t_end; t += 2) {
            const double P = a.pairP[(t - a.t_begin) >> 1], invP = a.pairInvP[(t - a.t_begin) >> 1];
            int Rr[4], Ri[4];
            float Fr[4], Fi[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double R = fma(-rint(xr[e] * invP), P, xr[e]);
                if constexpr (BIG) R = fma(-rint(R * invP), P, R);
                Rr[e] = (int)R, Fr[e] = (float)R;
                if constexpr (E::cplx) {
                    double I = fma(-rint(xi[e] * invP), P, xi[e]);
                    if constexpr (BIG) I = fma(-rint(I * invP), P, I);
                    Ri[e] = (int)I, Fi[e] = (float)I;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tt = t + u;
                if (tt >= a.t_end) break;
                const ModConst mc = a.mt.mc[tt];
                if constexpr (WIDE && OZ2_F6_FLOAT_CHAIN) {
                    if (a.f6) {  // (uniform) FP6 panel images from float residues: see put_f6_planes_f
                        const float pf = (float)mc.p, hf = (float)(mc.p >> 1);
                        float fr[4], fi[4], fs[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            fr[e] = residue_wide_f(Fr[e], mc, pf);
                            if constexpr (E::cplx) fi[e] = residue_wide_f(Fi[e], mc, pf), fs[e] = wrapping_f(fr[e] + fi[e], pf, hf);
                        }
                        int8_t* o = out + (size_t)(tt < 6 ? 2 * tt : 12 + 3 * (tt - 6)) * a.plane_stride;
                        put_f6_planes_f(a, o, tt, fr);
                        if constexpr (E::cplx) {
                            put_f6_planes_f(a, o + a.part_stride, tt, fi);
                            put_f6_planes_f(a, o + 2 * a.part_stride, tt, fs);
                        }
                        continue;
                    }
                }
                int rr[4], ri[4], rs[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rr[e] = residue_from_small<WIDE>(Rr[e], Fr[e], mc);
                    if constexpr (E::cplx) {
                        ri[e] = residue_from_small<WIDE>(Ri[e], Fi[e], mc);
                        rs[e] = WIDE ? wrapping(rr[e] + ri[e], mc.p) : wrapping((int)(int8_t)rr[e] + (int)(int8_t)ri[e], mc.p);
                    }
                }
                if constexpr (WIDE) {
                    int8_t* o = out + (size_t)(tt < 6 ? 2 * tt : 12 + 3 * (tt - 6)) * a.plane_stride;
                    if (a.f6) {  // (uniform) FP6 panel images: `out` is this lane's dword address inside plane 0
                        put_f6_planes(a, o, tt, rr);
                        if constexpr (E::cplx) {
                            put_f6_planes(a, o + a.part_stride, tt, ri);
                            put_f6_planes(a, o + 2 * a.part_stride, tt, rs);
                        }
                    } else {
                        put_fp8_planes(a, o, tt, rr);
                        if constexpr (E::cplx) {
                            put_fp8_planes(a, o + a.part_stride, tt, ri);
                            put_fp8_planes(a, o + 2 * a.part_stride, tt, rs);
                        }
                    }
                } else {
                    auto pack = [](const int (&r)[4]) {
                        return ((unsigned)r[0] & 0xFFu) | (((unsigned)r[1] & 0xFFu) << 8) | (((unsigned)r[2] & 0xFFu) << 16) | ((unsigned)r[3] << 24);
                    };
                    int8_t* o = out + (size_t)tt * a.plane_stride;
                    *(unsigned*)o = pack(rr);
                    if constexpr (E::cplx) {
                        *(unsigned*)(o + a.part_stride) = pack(ri);
                        *(unsigned*)(o + 2 * a.part_stride) = pack(rs);
                    }
                }
            }
        }
    };
    const bool anybig = __any(big);
    if (a.backend == kFP8) {
        if (anybig) run.template operator()<true, true>();
        else run.template operator()<true, false>();
    } else {
        if (anybig) run.template operator()<false, true>();
        else run.template operator()<false, false>();
    }
}

template <typename T, int MODE>
__device__ __forceinline__ void emit4(const StageArgs& a, size_t row, size_t k0, const T (&v)[4], int s) {
    using E = ET<T>;
    int8_t* out = a.lo + OZ2_ZW + ((MODE == MODE_MOD && a.f6) ? f6_lane_offset(a, row, k0) : row * a.kp + k0);
    if constexpr (MODE == MODE_BOUND) {
        if (a.backend == kFP8) {
            // e4m3 round-up of |x|*2^s (< 2^8), computed in the input precision (scaling.hpp:77-82); complex: planes
            // |Re|, |Im| and fp8_e4m3_ru(|Re| - |Im|) of the two rounded values (sub_ru_8bit, scaling_accu_complex.hpp:7-10:
            // the difference of two e4m3 numbers is exact in fp16/fp32; the helper's "+1 encoding step" is applied
            // literally, also for negative differences)
            using U = typename E::U;
            unsigned wr = 0, wi = 0, wd = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const U x = (U)E::re(v[e]);
                const U sc = sizeof(U) == 8 ? (U)scalbn(fabs((double)x), s) : (U)scalbnf(fabsf((float)x), s);
                const unsigned br = fp8_round_up<U>(sc);
                wr |= br << (8 * e);
                if constexpr (E::cplx) {
                    const U y = (U)E::im(v[e]);
                    const U sd = sizeof(U) == 8 ? (U)scalbn(fabs((double)y), s) : (U)scalbnf(fabsf((float)y), s);
                    const unsigned bi = fp8_round_up<U>(sd);
                    wi |= bi << (8 * e);
                    wd |= (fp8_round_up<float>(fp8_to_float(br) - fp8_to_float(bi)) & 0xFFu) << (8 * e);
                }
            }
            *(unsigned*)out = wr;
            if constexpr (E::cplx) {
                *(unsigned*)(out + a.part_stride) = wi;
                *(unsigned*)(out + 2 * a.part_stride) = wd;
            }
            return;
        }
        unsigned wr = 0, wi = 0, wd = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // ceil(|x| * 2^s) in FP64 (v_ldexp_f64, v_ceil_f64, v_cvt_i32_f64: exact, the value is below 2^7 by construction of s) instead of
            // the ~25 integer operations of upper_bound_i8 (kept in oz2_device.hpp; -DOZ2_BOUND_FLOAT=0)
            auto ub = [&](double x) { return OZ2_BOUND_FLOAT ? (int)ceil(ldexp(fabs(x), s)) : upper_bound_i8(x, s); };
            const int br = ub(E::re(v[e]));
            wr |= ((unsigned)br & 0xFFu) << (8 * e);
            if constexpr (E::cplx) {
                const int bi = ub(E::im(v[e]));
                wi |= ((unsigned)bi & 0xFFu) << (8 * e);
                wd |= ((unsigned)(br - bi) & 0xFFu) << (8 * e);
            }
        }
        *(unsigned*)out = wr;
        if constexpr (E::cplx) {
            *(unsigned*)(out + a.part_stride) = wi;
            *(unsigned*)(out + 2 * a.part_stride) = wd;
        }
    } else if constexpr (OZ2_STAGE_FLOATRES) {
        emit4_mod_float<T>(a, out, v, s);
    } else {
        uint64_t Mr[4], Mi[4];
        int Er[4], Ei[4];
        bool nr[4], ni[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const ScaledInt x = trunc_scale(E::re(v[e]), s);
            Mr[e] = x.M;
            Er[e] = x.E;
            nr[e] = x.neg;
            if constexpr (E::cplx) {
                const ScaledInt y = trunc_scale(E::im(v[e]), s);
                Mi[e] = y.M;
                Ei[e] = y.E;
                ni[e] = (a.conj && y.M != 0) ? !y.neg : y.neg;  // a zero stays +0 (two's-complement residue path)
            }
        }
        // E > 0 (|x|*2^s >= 2^53) cannot happen for num_moduli <= 15; one wave-uniform test keeps it out of the common path
        bool anyE = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            anyE |= Er[e] > 0;
            if constexpr (E::cplx) anyE |= Ei[e] > 0;
        }
        const bool fastE = !__any(anyE);
        unsigned rlo[4], rhi[4], ilo[4], ihi[4];  // fastE: M or its 56-bit two's complement
        Bytes128 Xr[4], Xi[4];                    // otherwise: M*2^E or its 120-bit two's complement
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (fastE) {
                const uint64_t mt = nr[e] ? (1ull << 56) - Mr[e] : Mr[e];
                rlo[e] = (unsigned)mt, rhi[e] = (unsigned)(mt >> 32);
                if constexpr (E::cplx) {
                    const uint64_t it = ni[e] ? (1ull << 56) - Mi[e] : Mi[e];
                    ilo[e] = (unsigned)it, ihi[e] = (unsigned)(it >> 32);
                }
            } else {
                Xr[e] = shifted_bytes(Mr[e], Er[e], nr[e]);
                if constexpr (E::cplx) Xi[e] = shifted_bytes(Mi[e], Ei[e], ni[e]);
            }
        }
        auto put_fp8 = [&](int8_t* o, int t, const int (&rr)[4]) {
            if (a.f6) put_f6_planes(a, o, t, rr);
            else put_fp8_planes(a, o, t, rr);
        };
        // one pass over the moduli; FAST / WIDE are compile-time so the residue code is branch-free.  Complex: the residues of
        // Re, Im and wrapping(Re + Im) go to the three parts (INT8: the sum of the int8-cast values, mod.hpp:321-325).
        auto planes = [&]<bool FAST, bool WIDE>() {
            auto residues = [&](int t, int (&rr)[4], int (&ri)[4], int (&rs)[4]) {
                const ModConst mc = a.mt.mc[t];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rr[e] = FAST ? residue_sym_bytes_e0<WIDE>(rlo[e], rhi[e], nr[e], mc) : residue_sym_bytes128<WIDE>(Xr[e], nr[e], mc);
                    if constexpr (E::cplx) {
                        ri[e] = FAST ? residue_sym_bytes_e0<WIDE>(ilo[e], ihi[e], ni[e], mc) : residue_sym_bytes128<WIDE>(Xi[e], ni[e], mc);
                        rs[e] = WIDE ? wrapping(rr[e] + ri[e], mc.p) : wrapping((int)(int8_t)rr[e] + (int)(int8_t)ri[e], mc.p);
                    }
                }
            };
            auto pack = [](const int (&r)[4]) {
                return ((unsigned)r[0] & 0xFFu) | (((unsigned)r[1] & 0xFFu) << 8) | (((unsigned)r[2] & 0xFFu) << 16) | ((unsigned)r[3] << 24);
            };
            if constexpr (!WIDE && OZ2_STAGE_V16) {
                // four planes per trip; the dwords of a lane quad (16 consecutive k of this row) are transposed so that lane q stores the
                // 16 bytes of plane t + q: 256 contiguous bytes per plane and store instruction, a quarter of the store instructions
                typedef unsigned V4 __attribute__((ext_vector_type(4)));
                const unsigned q = threadIdx.x & 3u;
                int8_t* oq = out - 4 * q;  // first byte of the quad's 16-byte run
                for (int t = a.t_begin; t < a.t_end; t += 4) {
                    unsigned wr[4] = {0u, 0u, 0u, 0u}, wi[4] = {0u, 0u, 0u, 0u}, ws[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (t + i < a.t_end) {
                            int rr[4], ri[4], rs[4];
                            residues(t + i, rr, ri, rs);
                            wr[i] = pack(rr);
                            if constexpr (E::cplx) wi[i] = pack(ri), ws[i] = pack(rs);
                        }
                    }
                    quad_transpose4(wr, q);
                    if constexpr (E::cplx) quad_transpose4(wi, q), quad_transpose4(ws, q);
                    if (t + (int)q < a.t_end) {
                        int8_t* o = oq + (size_t)(t + (int)q) * a.plane_stride;
                        *(V4*)o = V4{wr[0], wr[1], wr[2], wr[3]};
                        if constexpr (E::cplx) {
                            *(V4*)(o + a.part_stride) = V4{wi[0], wi[1], wi[2], wi[3]};
                            *(V4*)(o + 2 * a.part_stride) = V4{ws[0], ws[1], ws[2], ws[3]};
                        }
                    }
                }
                return;
            }
            for (int t = a.t_begin; t < a.t_end; ++t) {
                int rr[4], ri[4], rs[4];
                residues(t, rr, ri, rs);
                if constexpr (WIDE) {
                    int8_t* o = out + (size_t)(t < 6 ? 2 * t : 12 + 3 * (t - 6)) * a.plane_stride;
                    put_fp8(o, t, rr);
                    if constexpr (E::cplx) {
                        put_fp8(o + a.part_stride, t, ri);
                        put_fp8(o + 2 * a.part_stride, t, rs);
                    }
                } else {
                    int8_t* o = out + (size_t)t * a.plane_stride;
                    *(unsigned*)o = pack(rr);
                    if constexpr (E::cplx) {
                        *(unsigned*)(o + a.part_stride) = pack(ri);
                        *(unsigned*)(o + 2 * a.part_stride) = pack(rs);
                    }
                }
            }
        };
        if (a.backend == kFP8) {
            if (fastE) planes.template operator()<true, true>();
            else planes.template operator()<false, true>();
        } else {
            if (fastE) planes.template operator()<true, false>();
            else planes.template operator()<false, false>();
        }
    }
}

// K-major operand: one 256-thread block per row; thread = 4 consecutive k per 1024-wide sweep
template <typename T, int MODE>
__device__ __forceinline__ void stage_kmajor_body(const StageArgs& a, const unsigned bid) {
    using E = ET<T>;
    using U = typename E::U;
    if constexpr (MODE == MODE_MOD) {
        if (a.f6) {
            // FP6 panel images: one workgroup per (8 rows, one 128-element K-step), the K-step index fastest: the eight rows' 16-byte X slots and
            // 8-byte Y slot pairs of a K group are adjacent in the image, so every plane leaves as complete 128-byte lines (row by row, a K-step's
            // 96 bytes of one row are scattered over six lines that seven other rows complete later)
            const unsigned nks = (unsigned)(a.kp / 128);
            const unsigned rg = bid / nks;
            const size_t row = (size_t)rg * 8 + (threadIdx.x >> 5);
            const size_t k0 = (size_t)(bid - rg * nks) * 128 + (size_t)(threadIdx.x & 31) * 4;
            if (row >= a.rows) return;  // (uniform over the 32 lanes of a row)
            const T* x = (const T*)((const char*)a.X + OZ2_ZX) + row * a.ld;
            const int s = OZ2_ROW_SHIFT(a, row, bid == rg * nks && (threadIdx.x & 31) == 0);
            T v[4];
            load4<T>(x, k0, a.k, v);
            emit4<T, MODE>(a, row, k0, v, s);
            return;
        }
    }
    if constexpr (MODE == MODE_MOD && OZ2_STAGE_KCHUNK && sizeof(T) <= 8) {  // (16-byte elements measured 4 % better with the row loop)
        // quantise: one workgroup per 1024-wide k chunk of a row, the chunk index fastest: the workgroups in flight walk through
        // memory together (one row after the other) instead of streaming ~2000 rows at once
        const unsigned nch = (unsigned)(a.kp / 1024 + (a.kp % 1024 != 0));
        const size_t row = bid / nch;
        const size_t k0 = (size_t)(bid - row * nch) * 1024 + (size_t)threadIdx.x * 4;
        if (k0 >= a.kp) return;
        const T* x = (const T*)((const char*)a.X + OZ2_ZX) + row * a.ld;
        const int s = OZ2_ROW_SHIFT(a, row, k0 == 0);
        T v[4];
        load4<T>(x, k0, a.k, v);
        emit4<T, MODE>(a, row, k0, v, s);
        return;
    }
    const size_t row = bid;
    const T* x = (const T*)((const char*)a.X + OZ2_ZX) + row * a.ld;
    int s;
    if constexpr (MODE == MODE_BOUND) {
        __shared__ U sm[4];
        // rows that fit NC x 1024 elements stay in registers between the amax pass and the extract pass: one read of the row
        constexpr int NC = sizeof(T) == 16 ? 4 : 8;
        if (a.kp <= (size_t)1024 * NC) {
            T vb[NC][4];
            U am = 0;
#pragma unroll
            for (int it = 0; it < NC; ++it) {
                const size_t k0 = (size_t)threadIdx.x * 4 + (size_t)it * 1024;
                load4<T>(x, k0, a.k, vb[it]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const U ar = (U)fabs(E::re(vb[it][e])), ai = (U)fabs(E::im(vb[it][e]));
                    am = ar > am ? ar : am;
                    am = ai > am ? ai : am;
                }
            }
            am = wave_max(am);
            if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = am;
            __syncthreads();
            am = sm[0];
            am = sm[1] > am ? sm[1] : am;
            am = sm[2] > am ? sm[2] : am;
            am = sm[3] > am ? sm[3] : am;
            s = (a.backend == kINT8 ? 5 : 7) - ilogb0(am);
            if (threadIdx.x == 0) {
                ((int16_t*)((char*)a.sft0 + OZ2_ZW))[row] = (int16_t)s;
                if (a.sft0_keep) ((int16_t*)((char*)a.sft0_keep + OZ2_ZW))[row] = (int16_t)s;
            }
#pragma unroll
            for (int it = 0; it < NC; ++it) {
                const size_t k0 = (size_t)threadIdx.x * 4 + (size_t)it * 1024;
                if (k0 < a.kp) emit4<T, MODE>(a, row, k0, vb[it], s);
            }
            return;
        }
        U am = 0;
        for (size_t kk = threadIdx.x; !OZ2_AMAX_PREFETCH && kk < a.k; kk += 256) {
            const T v = x[kk];
            const U ar = (U)fabs(E::re(v)), ai = (U)fabs(E::im(v));
            am = ar > am ? ar : am;
            am = ai > am ? ai : am;
        }
        for (size_t k0 = (size_t)threadIdx.x * 4; OZ2_AMAX_PREFETCH && k0 < a.k; k0 += 2048) {  // two 4-element groups in flight per thread
            T v0[4], v1[4];
            load4<T, false>(x, k0, a.k, v0);
            load4<T, false>(x, k0 + 1024 < a.k ? k0 + 1024 : k0, a.k, v1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const U ar = (U)fabs(E::re(v0[e])), ai = (U)fabs(E::im(v0[e])), br = (U)fabs(E::re(v1[e])), bi = (U)fabs(E::im(v1[e]));
                am = ar > am ? ar : am;
                am = ai > am ? ai : am;
                am = br > am ? br : am;
                am = bi > am ? bi : am;
            }
        }
        am = wave_max(am);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = am;
        __syncthreads();
        am = sm[0];
        am = sm[1] > am ? sm[1] : am;
        am = sm[2] > am ? sm[2] : am;
        am = sm[3] > am ? sm[3] : am;
        s = (a.backend == kINT8 ? 5 : 7) - ilogb0(am);
        if (threadIdx.x == 0) {
            ((int16_t*)((char*)a.sft0 + OZ2_ZW))[row] = (int16_t)s;
            if (a.sft0_keep) ((int16_t*)((char*)a.sft0_keep + OZ2_ZW))[row] = (int16_t)s;
        }
    } else {
        s = OZ2_ROW_SHIFT(a, row, threadIdx.x == 0);
    }
    for (size_t k0 = (size_t)threadIdx.x * 4; k0 < a.kp; k0 += 1024) {
        T v[4];
        load4<T>(x, k0, a.k, v);
        emit4<T, MODE>(a, row, k0, v, s);
    }
}

// Row-strided operand: 1-D grid of ceil(kp/TK) * ceil(rows/TR) workgroups, the k-tile index fastest (a 2-D grid would cap the
// row-tile count at 65535, i.e. operands of ~1M rows); 256 threads; tile TR rows x TK k staged RAW in LDS
template <typename T> struct StageTile {
    // 16.6 KiB of LDS per workgroup for every type (8 workgroups per CU): 32 rows of float, 16 rows of the 8- and 16-byte
    // types (a 16-row read segment of doubles is still one full 128-B cache line)
#ifndef OZ2_STAGE_TR8
#define OZ2_STAGE_TR8 16  // rows per tile of the 8- and 16-byte element types
#endif
#ifndef OZ2_STAGE_Z_WIDE
#define OZ2_STAGE_Z_WIDE 1  // 16-byte elements: 8 rows x 128 k (128-byte runs per row and plane on the way out) instead of 16 rows x 64 k (64-byte runs)
#endif
    static constexpr int TR = sizeof(T) == 4 ? 32 : (sizeof(T) == 16 && OZ2_STAGE_Z_WIDE) ? 8 : OZ2_STAGE_TR8;
    static constexpr int TK = (sizeof(T) == 16 && !OZ2_STAGE_Z_WIDE) ? 64 : 128;
};
template <typename T, int MODE>
__device__ __forceinline__ void stage_strided_body(const StageArgs& a, const unsigned bid) {
    using E = ET<T>;
    using U = typename E::U;
    constexpr int TR = StageTile<T>::TR, TK = StageTile<T>::TK;
    constexpr int PITCH = TK + (sizeof(T) >= 16 ? 1 : 16 / sizeof(T));  // rows stay 16-B aligned
    constexpr int CH = TK / 4;                                          // 4-wide k chunks per row
    constexpr int RPP = 256 / CH;                                       // rows per pass
    __shared__ __attribute__((aligned(16))) T tile[TR][PITCH];
    using UBm = typename std::conditional<sizeof(U) == 8, unsigned long long, unsigned>::type;
    [[maybe_unused]] __shared__ UBm rowam[TR];
#if OZ2_STAGE_RTFAST
    const unsigned nrt = (unsigned)((a.rows + TR - 1) / TR);
    const unsigned kt = bid / nrt, rt = bid - kt * nrt;
#else
    const unsigned nkt = (unsigned)(a.kp / TK);
    const unsigned rt = bid / nkt, kt = bid - rt * nkt;
#endif
    const size_t r0 = (size_t)rt * TR;
    const size_t kb = (size_t)kt * TK;
    if constexpr (MODE == MODE_BOUND) {
        // row maxima of the tile's rows = the maximum over the k splits' partial arrays (amax_pair_kernel): 256 threads = TR rows x PL lanes, one
        // load per lane and partial (non-negative IEEE patterns order like unsigned integers), published to the passes below by the tile's barrier
        constexpr int PL = 256 / TR;
        const int rr = threadIdx.x / PL, pp = threadIdx.x % PL;
        UBm v = 0;
        if (r0 + rr < a.rows) {
            const UBm* am = (const UBm*)((const char*)a.amax + OZ2_ZW) + r0 + rr;
            for (unsigned q = pp; q < a.amax_parts; q += PL) {
                const UBm w = am[(size_t)q * a.amax_pstride];
                v = w > v ? w : v;
            }
        }
#pragma unroll
        for (int d = PL / 2; d >= 1; d >>= 1) {
            const UBm w = __shfl_xor(v, d);
            v = w > v ? w : v;
        }
        if (pp == 0) rowam[rr] = v;
    }
    if constexpr (OZ2_STAGE_PAIRLOAD && sizeof(T) <= 8) {
        // 4- and 8-byte elements: a lane fetches RPL = 4 / 2 consecutive rows with one 16-byte load (a half / a quarter of the load
        // instructions, 1 KiB per wave instruction) when the rows exist and the group is 16-byte aligned; all loads of the tile are in
        // flight at once.  (float operands took the one-element-per-lane path until round 3: quantise A 222 us = 3.3 TB/s at 8192^2,
        // against 5.4 TB/s for double.)
        constexpr int RPL = 16 / (int)sizeof(T);
        constexpr int RP = TR / RPL;     // row groups per column
        constexpr int KY = 256 / RP;     // k values fetched per pass
        const int rp = threadIdx.x % RP, ky = threadIdx.x / RP;
        const size_t row = r0 + RPL * rp;
        const T* x = (const T*)((const char*)a.X + OZ2_ZX) + row;
        const bool pair_ok = row + RPL - 1 < a.rows && ((reinterpret_cast<uintptr_t>(x) | (a.ld * sizeof(T))) & 15u) == 0;
        typedef unsigned V4 __attribute__((ext_vector_type(4)));
        V4 buf[TK / KY];
#pragma unroll
        for (int it = 0; it < TK / KY; ++it) {
            const size_t kg = kb + ky + KY * it;
            V4 v = {0u, 0u, 0u, 0u};
            if (kg < a.k) {
                if (pair_ok) {
                    v = OZ2_LOAD_NT ? __builtin_nontemporal_load((const V4*)(x + kg * a.ld)) : *(const V4*)(x + kg * a.ld);
                } else {
#pragma unroll
                    for (int e = 0; e < RPL; ++e) {
                        const T t = (row + e < a.rows) ? x[kg * a.ld + e] : E::zero();
                        __builtin_memcpy((char*)&v + e * sizeof(T), &t, sizeof(T));
                    }
                }
            }
            buf[it] = v;
        }
#pragma unroll
        for (int it = 0; it < TK / KY; ++it) {
            const int kk = ky + KY * it;
#pragma unroll
            for (int e = 0; e < RPL; ++e) __builtin_memcpy(&tile[RPL * rp + e][kk], (const char*)&buf[it] + e * sizeof(T), sizeof(T));
        }
    } else {
        constexpr int KY = 256 / TR;  // k values fetched per pass
        const int rx = threadIdx.x % TR, ky = threadIdx.x / TR;
        const size_t row = r0 + rx;
        const T* x = (const T*)((const char*)a.X + OZ2_ZX) + row;
#pragma unroll 4
        for (int it = 0; it < TK / KY; ++it) {
            const int kk = ky + KY * it;
            const size_t kg = kb + kk;
            tile[rx][kk] = (row < a.rows && kg < a.k) ? x[kg * a.ld] : E::zero();
        }
    }
    __syncthreads();
    const int c = threadIdx.x % CH;
#pragma unroll 1
    for (int pass = 0; pass < TR / RPP; ++pass) {
        const int rl = pass * RPP + threadIdx.x / CH;
        const size_t row = r0 + rl;
        if (row >= a.rows) continue;
        int s;
        if constexpr (MODE == MODE_BOUND) {
            const UBm bits = rowam[rl];
            U am;
            __builtin_memcpy(&am, &bits, sizeof(U));
            s = (a.backend == kINT8 ? 5 : 7) - ilogb0(am);
            if (kt == 0 && c == 0) {
                ((int16_t*)((char*)a.sft0 + OZ2_ZW))[row] = (int16_t)s;
                if (a.sft0_keep) ((int16_t*)((char*)a.sft0_keep + OZ2_ZW))[row] = (int16_t)s;
            }
        } else {
            s = OZ2_ROW_SHIFT(a, row, kt == 0 && c == 0);
        }
        T v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[rl][c * 4 + e];
        emit4<T, MODE>(a, row, kb + c * 4, v, s);
    }
}

// accurate-mode extract (MODE_BOUND) of BOTH operands in one launch (round 6; workgroups [0, nA) work on a, the rest on b; either share may be empty).  The
// row maxima of row-strided operands come from amax_pair_kernel's partial arrays (no atomics, hence nothing to zero for them); the zero-fill of the bound GEMM's
// maxima arrays rides on this launch (zero_p / zero_words of `a`).
template <typename T> __global__ void __launch_bounds__(256) extract_pair_kernel(const StageArgs a, const StageArgs b, const unsigned nA, const int kmA, const int kmB) {
    if (a.zero_words) {
        unsigned* zp = (unsigned*)((char*)a.zero_p + OZ2_ZW);
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < a.zero_words; i += gridDim.x * 256u) zp[i] = 0u;
    }
    if (blockIdx.x < nA) {
        if (kmA) stage_kmajor_body<T, MODE_BOUND>(a, blockIdx.x);
        else stage_strided_body<T, MODE_BOUND>(a, blockIdx.x);
    } else {
        if (kmB) stage_kmajor_body<T, MODE_BOUND>(b, blockIdx.x - nA);
        else stage_strided_body<T, MODE_BOUND>(b, blockIdx.x - nA);
    }
}
// quantise (MODE_MOD) of BOTH operands in one launch: workgroups [0, nA) work on a, the rest on b; either share may be empty (skip-scaling,
// single-operand callers).  The two halves are independent and at launch-bound sizes each of them is a ~5-10 us dispatch (1024^3: 12.7 + 9.5 us,
// together ... see DESIGN.md 3.2); the form of each operand (K-major / row-strided) is a uniform run-time branch, registers and LDS are the
// maximum of the two forms (40 / 47 VGPRs, 0 / 16.6 KiB for double).
template <typename T> __global__ void __launch_bounds__(256) quantise_pair_kernel(const StageArgs a, const StageArgs b, const unsigned nA, const int kmA, const int kmB) {
    if (blockIdx.x < nA) {
        if (kmA) stage_kmajor_body<T, MODE_MOD>(a, blockIdx.x);
        else stage_strided_body<T, MODE_MOD>(a, blockIdx.x);
    } else {
        if (kmB) stage_kmajor_body<T, MODE_MOD>(b, blockIdx.x - nA);
        else stage_strided_body<T, MODE_MOD>(b, blockIdx.x - nA);
    }
}
static_assert(2 * sizeof(StageArgs) + 16 <= 4096, "two argument blocks must fit the 4 KiB kernel-argument segment");

// ---- FP6 panel images of REAL operands: one lane = one fragment (round 5).  The generic stage kernels above give a lane four consecutive k; for the
// FP6 images that form is bound by VALU issue (config 3: 2.4 ms for 7 GB, ~19 operations per value and modulus, half of them the sign-magnitude codes,
// their 6-bit packing and the quad shuffle that assembles a dword).  Here a lane owns the 32 consecutive k of one row that make ONE fragment of the image
// (oz2_gemm_f6.hip: 16 bytes in the X region + 8 in the Y region), and gfx950 converts and packs them in hardware: v_cvt_scalef32_2xpk16_fp6_f32 takes
// two 16-float operands a, b and a scale and returns the 32 e2m3 codes of a[0]/scale, b[0]/scale, a[1]/scale, ... in six registers -- with scale 8 the code
// of an integer |v| <= 16 is sign << 5 | |v| (tools/ubench/cvt_fp6.hip; a negative zero keeps its sign bit: the same value for the MFMA).  What is left per
// value and modulus is the residue (3) and the split (3) in packed fp32 plus two FP64 operations per pair: the kernel runs at the HBM side.
//   * workgroup = 64 rows x one 128-element K-step, wave q = fragment column q: a wave's 64 X slots are one contiguous KiB, its Y slots 512 bytes in 8-byte pieces;
//   * row-strided operand (k strided by ld): lane = row, 32 coalesced loads; K-major operand: coalesced 16-byte loads of 8 rows x 128 bytes per instruction,
//     transposed through a wave-private LDS tile (row pitch 144 bytes: conflict-free 16-byte reads), 128 bytes of every row at a time.
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
#ifndef OZ2_F6_LANE_KERNEL
#define OZ2_F6_LANE_KERNEL 1  // 0: real FP6 planes through the generic stage kernels (four k per lane), as complex operands.  (A complex form was built in
                              // round 5 -- the fragment in two halves of 16 k, three plane sets per value -- and needs 282-314 registers: the 16-float operand
                              // tuples of the pack instruction with half their lanes as padding fragment the register file; one wave per SIMD: not kept.)
#endif
template <typename T> __device__ __forceinline__ void stage_f6_body(const StageArgs& a, const unsigned bid, const bool kmajor, char* tile /* this wave's 64 x 144 bytes */) {
    static_assert(!ET<T>::cplx, "real operands");
    constexpr int EPL = 16 / (int)sizeof(T);   // elements per 16-byte piece
    constexpr int CH = 128 / (int)sizeof(T);   // elements per 128-byte chunk of a row
    const unsigned lane = threadIdx.x & 63u, q = threadIdx.x >> 6;
    const unsigned nks = (unsigned)(a.kp / 128), nrb = (unsigned)((a.rows + 63) / 64);
    unsigned rb, kt;
    if (kmajor) rb = bid / nks, kt = bid - rb * nks;   // the workgroups in flight walk along the rows' memory
    else kt = bid / nrb, rb = bid - kt * nrb;           // ... down the columns'
    const size_t r0 = (size_t)rb * 64, row = r0 + lane, k0 = (size_t)kt * 128 + (size_t)q * 32;
    const T* X = (const T*)((const char*)a.X + OZ2_ZX);
    typedef unsigned V4 __attribute__((ext_vector_type(4)));
    T v[32];
    if (kmajor) {
        const unsigned pc = lane & 7u, pr = lane >> 3;
#pragma unroll
        for (int c = 0; c < 32 / CH; ++c) {
            V4 buf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t r = r0 + pr + 8 * j, kk = k0 + (size_t)c * CH + (size_t)pc * EPL;
                const T* src = X + r * a.ld + kk;
                V4 t = {0u, 0u, 0u, 0u};
                if (r < a.rows) {
                    if (kk + EPL <= a.k && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
                        t = OZ2_LOAD_NT ? __builtin_nontemporal_load((const V4*)src) : *(const V4*)src;
                    } else {
                        T e[EPL];
#pragma unroll
                        for (int u = 0; u < EPL; ++u) e[u] = kk + u < a.k ? src[u] : ET<T>::zero();
                        __builtin_memcpy(&t, e, 16);
                    }
                }
                buf[j] = t;
            }
            if (c > 0) __builtin_amdgcn_wave_barrier();  // (the tile is re-used: every lane has read its row of the previous chunk -- LDS operations of a wave execute in order)
#pragma unroll
            for (int j = 0; j < 8; ++j) *(V4*)(tile + (pr + 8 * j) * 144 + pc * 16) = buf[j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const V4 t = *(const V4*)(tile + lane * 144 + i * 16);
                __builtin_memcpy(&v[c * CH + i * EPL], &t, 16);
            }
        }
    } else {
        const T* src = X + row;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const size_t kg = k0 + kk;
            v[kk] = (row < a.rows && kg < a.k) ? (OZ2_LOAD_NT ? __builtin_nontemporal_load(src + kg * a.ld) : src[kg * a.ld]) : ET<T>::zero();
        }
    }
    if (row >= a.rows) return;
    const int s = OZ2_ROW_SHIFT(a, row, k0 == 0);   // (k0 == 0: K-step 0, fragment column 0 -- one lane per row over the grid)
    // xs = trunc(x 2^s): exact; a float operand's xs is a float again (24 significant bits), kept in 32 registers and widened pair by pair
    using XS = typename std::conditional<sizeof(T) == 4, float, double>::type;
    XS xs[32];
    bool big = false;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        if constexpr (sizeof(T) == 4) xs[e] = truncf(ldexpf(v[e], s));
        else xs[e] = trunc(ldexp((double)v[e], s));
        big |= fabs((double)xs[e]) >= 0x1.0p53;
    }
    // this lane's fragment in plane 0: X slot (q Rp + r), Y slot ((q >> 1) 2 Rp + 2 r + (q & 1))
    const unsigned tb = (unsigned)(row >> 8), r = (unsigned)row & 255u;
    const unsigned rp = tb == a.f6_last ? a.f6_rp_last : 256u;
    int8_t* const panel = a.lo + OZ2_ZW + (size_t)tb * 256u * (a.kp / 4 * 3) + (size_t)kt * rp * 96u;
    int8_t* const ox = panel + (size_t)(q * rp + r) * 16u;
    int8_t* const oy = panel + 64u * (size_t)rp + (size_t)((q >> 1) * 2u * rp + 2u * r + (q & 1u)) * 8u;
    auto put = [&](int plane, const float (&w)[32]) {
        v16f ev, od;
#pragma unroll
        for (int i = 0; i < 16; ++i) ev[i] = w[2 * i], od[i] = w[2 * i + 1];
        const v6u c = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ev, od, 8.0f);
        typedef unsigned V2 __attribute__((ext_vector_type(2)));
        *(V4*)(ox + (size_t)plane * a.plane_stride) = V4{c[0], c[1], c[2], c[3]};
        *(V2*)(oy + (size_t)plane * a.plane_stride) = V2{c[4], c[5]};
    };
    // (the loops over the 32 values are cut into groups of eight by scheduling barriers: left alone the scheduler interleaves all 32 chains and the
    // kernel needs 263 registers -- one wave per SIMD)
    auto run = [&]<bool BIG>() {
        for (int t = a.t_begin; t < a.t_end; t += 2) {
            const double P = a.pairP[(t - a.t_begin) >> 1], invP = a.pairInvP[(t - a.t_begin) >> 1];
            float Rf[32];
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {
#pragma unroll
                for (int e = 8 * g8; e < 8 * g8 + 8; ++e) {
                    const double x = (double)xs[e];
                    double R = fma(-rint(x * invP), P, x);
                    if constexpr (BIG) R = fma(-rint(R * invP), P, R);
                    Rf[e] = (float)R;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
                const int tt = t + u;
                if (tt >= a.t_end) break;
                const ModConst mc = a.mt.mc[tt];
                const float pf = (float)mc.p;
                const bool even = !(mc.p & 1);  // (uniform; p = 1024 only) a tie takes +p/2
                const bool sqm = tt < 6;
                float lo[32], hi[32];
                // the residue of every value first (into lo): q = the quotient of residue_from_small, r = R - q p; for the one even modulus (p = 1024) a tie
                // (-p/2) takes +p/2 -- in a real branch (written as a select the compiler ran it for every modulus: a third of the block)
                if (even) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        const float qq = fmaf(Rf[e], mc.invp, 12582912.0f) - 12582912.0f;
                        const float rr = fmaf(-qq, pf, Rf[e]);
                        lo[e] = rr == -0.5f * pf ? 0.5f * pf : rr;
                    }
                } else {
#pragma unroll
                    for (int g8 = 0; g8 < 4; ++g8) {
#pragma unroll
                        for (int e = 8 * g8; e < 8 * g8 + 8; ++e) {
                            const float qq = fmaf(Rf[e], mc.invp, 12582912.0f) - 12582912.0f;
                            lo[e] = fmaf(-qq, pf, Rf[e]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (sqm) {  // fp8_split_sq: hi = rint(r / s), lo = r - s hi
                    const float sq = (float)a.sqrtp[tt], inv = 1.0f / sq;
#pragma unroll
                    for (int g8 = 0; g8 < 4; ++g8) {
#pragma unroll
                        for (int e = 8 * g8; e < 8 * g8 + 8; ++e) {
                            hi[e] = rintf(lo[e] * inv);
                            lo[e] = fmaf(-sq, hi[e], lo[e]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {    // fp8_split_kara: hi = sign(r) ceil(|r| / 16), lo = r - 16 hi
#pragma unroll
                    for (int g8 = 0; g8 < 4; ++g8) {
#pragma unroll
                        for (int e = 8 * g8; e < 8 * g8 + 8; ++e) {
                            hi[e] = copysignf(ceilf(fabsf(lo[e]) * 0.0625f), lo[e]);
                            lo[e] = fmaf(-16.0f, hi[e], lo[e]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                const int plane = sqm ? 2 * tt : 12 + 3 * (tt - 6);
                put(plane, hi);
                put(plane + 1, lo);
                if (!sqm) {
#pragma unroll
                    for (int e = 0; e < 32; ++e) hi[e] += lo[e];
                    put(plane + 2, hi);
                }
            }
        }
    };
    if (__any(big)) run.template operator()<true>();
    else run.template operator()<false>();
}
#ifdef OZ2_F6_WAVES  // experiment: force the register budget of that many waves per SIMD
#define OZ2_F6_KATTR __attribute__((amdgpu_waves_per_eu(OZ2_F6_WAVES, OZ2_F6_WAVES)))
#else
#define OZ2_F6_KATTR
#endif
template <typename T> __global__ void __launch_bounds__(256) OZ2_F6_KATTR quantise_f6_pair_kernel(const StageArgs a, const StageArgs b, const unsigned nA, const int kmA, const int kmB) {
    __shared__ __attribute__((aligned(16))) char tile[4][64 * 144];
    char* const mine = tile[threadIdx.x >> 6];
    if (blockIdx.x < nA) stage_f6_body<T>(a, blockIdx.x, kmA != 0, mine);
    else stage_f6_body<T>(b, blockIdx.x - nA, kmB != 0, mine);
}

// per-row amax of row-strided operands, BOTH operands in one launch (round 6): workgroup = 64 rows x one of `parts` k splits, 256 threads = 64 rows x 4 k-lanes;
// the maxima of split y go to the partial array y (amax + y * pstride): no atomics, nothing to zero; the extract takes the maximum over the splits.
struct AmaxOperand {
    const void* X;
    size_t ld, rows, bx;
    void* amax;
    size_t pstride;
    unsigned parts, row_groups;
};
template <typename T> __device__ __forceinline__ void amax_strided_body(const AmaxOperand& o, const unsigned bid, size_t k, size_t bw) {
    const T* X = (const T*)((const char*)o.X + blockIdx.z * o.bx);  // batched launch: item blockIdx.z
    using E = ET<T>;
    using U = typename E::U;
    using UB = typename std::conditional<sizeof(U) == 8, unsigned long long, unsigned>::type;
    UB* amax = (UB*)((char*)o.amax + blockIdx.z * bw);
    const unsigned by = bid / o.row_groups, bxr = bid - by * o.row_groups;  // the row group is the fast index: the workgroups in flight read whole columns
    const size_t rows = o.rows, ld = o.ld;
    // a lane owns RPL consecutive rows = 16 bytes of a column (one 16-byte load when they exist and are aligned); 64 rows per workgroup,
    // the remaining lanes spread over KL k-lanes
    constexpr int RPL = OZ2_STAGE_PAIRLOAD ? 16 / (int)sizeof(T) : 1;
    constexpr int LPC = 64 / RPL, KL = 256 / LPC;
    __shared__ U sm[KL][64];
    const int rl = threadIdx.x % LPC, ky = threadIdx.x / LPC;
    const size_t row0 = (size_t)bxr * 64 + (size_t)rl * RPL;
    const size_t kper = (k + o.parts - 1) / o.parts;
    const size_t kbeg = (size_t)by * kper, kend = (kbeg + kper < k) ? kbeg + kper : k;
    U am[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) am[j] = 0;
    auto take = [&](const T& v, int j) {
        const U ar = (U)fabs(E::re(v)), ai = (U)fabs(E::im(v));
        am[j] = ar > am[j] ? ar : am[j];
        am[j] = ai > am[j] ? ai : am[j];
    };
    if (row0 < rows) {
        const T* x = X + row0;
        const bool vec = RPL > 1 && row0 + RPL <= rows && ((reinterpret_cast<uintptr_t>(x) | (ld * sizeof(T))) & 15u) == 0;
        size_t kk = kbeg + ky;
        if (vec) {
            typedef unsigned V4 __attribute__((ext_vector_type(4)));
            for (; OZ2_AMAX_PREFETCH && kk + 3 * KL < kend; kk += 4 * KL) {  // four strided 16-byte loads in flight per thread
                V4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) raw[u] = *(const V4*)(x + (kk + (size_t)KL * u) * ld);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    T v[RPL];
                    __builtin_memcpy(v, &raw[u], 16);
#pragma unroll
                    for (int j = 0; j < RPL; ++j) take(v[j], j);
                }
            }
            for (; kk < kend; kk += KL) {
                T v[RPL];
                const V4 raw = *(const V4*)(x + kk * ld);
                __builtin_memcpy(v, &raw, 16);
#pragma unroll
                for (int j = 0; j < RPL; ++j) take(v[j], j);
            }
        } else {
            for (; kk < kend; kk += KL) {
#pragma unroll
                for (int j = 0; j < RPL; ++j)
                    if (row0 + j < rows) take(x[kk * ld + j], j);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RPL; ++j) sm[ky][rl * RPL + j] = am[j];
    __syncthreads();
    const size_t row = (size_t)bxr * 64 + threadIdx.x;
    if (threadIdx.x < 64 && row < rows) {
        U m = sm[0][threadIdx.x];
#pragma unroll
        for (int y = 1; y < KL; ++y) m = sm[y][threadIdx.x] > m ? sm[y][threadIdx.x] : m;
        UB bits;
        __builtin_memcpy(&bits, &m, sizeof(U));
        amax[(size_t)by * o.pstride + row] = bits;  // (an empty k split writes 0: the identity of the maximum)
    }
}
template <typename T> __global__ void __launch_bounds__(256) amax_pair_kernel(const AmaxOperand a, const AmaxOperand b, const unsigned nA, size_t k, size_t bw) {
    if (blockIdx.x < nA) amax_strided_body<T>(a, blockIdx.x, k, bw);
    else amax_strided_body<T>(b, blockIdx.x - nA, k, bw);
}

template <typename T, int MODE> static size_t stage_blocks(bool kmajor, const StageArgs& a) {
    if (a.rows == 0) return 0;
    if (kmajor && MODE == MODE_MOD && a.f6) return ((a.rows + 7) / 8) * (a.kp / 128);
    if (kmajor) return a.rows * ((MODE == MODE_MOD && OZ2_STAGE_KCHUNK && sizeof(T) <= 8) ? (a.kp + 1023) / 1024 : 1);
    return (a.kp / StageTile<T>::TK) * ((a.rows + StageTile<T>::TR - 1) / StageTile<T>::TR);
}
template <typename T> static hipError_t launch_quantise_stage(hipStream_t stream, bool kmA, const StageArgs& a, bool kmB, const StageArgs& b) {
    if constexpr (OZ2_F6_LANE_KERNEL && !ET<T>::cplx) {
        // real operands, FP6 panel images on both sides (an operand that is skipped has no rows): one lane per fragment
        if ((a.rows == 0 || a.f6) && (b.rows == 0 || b.f6) && (a.rows != 0 || b.rows != 0) && (a.rows ? a.backend : b.backend) == kFP8) {
            const size_t nA = a.rows ? ((a.rows + 63) / 64) * (a.kp / 128) : 0, nB = b.rows ? ((b.rows + 63) / 64) * (b.kp / 128) : 0;
            if (nA + nB > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
            hipLaunchKernelGGL(quantise_f6_pair_kernel<T>, dim3((unsigned)(nA + nB), 1, g_batch.batch), dim3(256), 0, stream, a, b, (unsigned)nA, (int)kmA, (int)kmB);
            return hipGetLastError();
        }
    }
    const size_t nA = stage_blocks<T, MODE_MOD>(kmA, a), nB = stage_blocks<T, MODE_MOD>(kmB, b);
    if (nA + nB == 0) return hipSuccess;
    if (nA + nB > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(quantise_pair_kernel<T>, dim3((unsigned)(nA + nB), 1, g_batch.batch), dim3(256), 0, stream, a, b, (unsigned)nA, (int)kmA, (int)kmB);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) zero_words_kernel(unsigned* p, size_t nwords, size_t bw) {
    p = (unsigned*)((char*)p + blockIdx.z * bw);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nwords) p[i] = 0u;
}

hipError_t launch_zero(hipStream_t stream, void* p, size_t bytes) {
    const size_t nwords = bytes / 4;
    if (nwords == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((nwords + 255) / 256), 1, g_batch.batch), dim3(256), 0, stream, (unsigned*)p, nwords, g_batch.ws);
    return hipGetLastError();
}

// k splits of the row-maxima pass of a row-strided operand: enough workgroups to fill the chip (~2048) -- the per-thread chain of dependent strided loads,
// not bandwidth, bounds that kernel when the grid is small -- at least 16 k values per workgroup, at most 32 splits (the extract reduces them per tile)
// and at most what the caller's scratch holds (max_parts).
unsigned amax_parts_for(size_t rows, size_t k, size_t max_parts) {
    const size_t row_groups = (rows + 63) / 64;
    size_t ks = (2048 + row_groups - 1) / row_groups;
    ks = std::min(ks, (k + 15) / 16);
    ks = std::min<size_t>(ks, 32);
    ks = std::min(ks, max_parts);
    return (unsigned)std::max<size_t>(ks, 1);
}

static StageArgs extract_args(int backend, size_t k, size_t kp, const ExtractOperand& o) {
    StageArgs a{};
    a.bx = o.xstride;
    a.bw = g_batch.ws;
    a.X = o.X;
    a.ld = o.ld;
    a.rows = o.rows;
    a.k = k;
    a.kp = kp;
    a.lo = o.lo;
    a.part_stride = o.part_stride;
    a.sft0 = o.sft0;
    a.sft0_keep = o.sft0_keep;
    a.amax = o.amax;
    a.amax_parts = o.parts;
    a.amax_pstride = o.pstride;
    a.backend = backend;
    a.conj = o.conj;
    return a;
}
template <typename T> static hipError_t launch_amax_pair_t(hipStream_t stream, size_t k, const ExtractOperand& A, const ExtractOperand& B) {
    auto mk = [](const ExtractOperand& o) {
        AmaxOperand q{};
        if (o.rows && !o.kmajor) q = AmaxOperand{o.X, o.ld, o.rows, o.xstride, o.amax, o.pstride, o.parts, (unsigned)((o.rows + 63) / 64)};
        return q;
    };
    const AmaxOperand a = mk(A), b = mk(B);
    const size_t nA = (size_t)a.row_groups * a.parts, nB = (size_t)b.row_groups * b.parts;
    if (nA + nB == 0) return hipSuccess;
    if (nA + nB > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(amax_pair_kernel<T>, dim3((unsigned)(nA + nB), 1, g_batch.batch), dim3(256), 0, stream, a, b, (unsigned)nA, k, g_batch.ws);
    return hipGetLastError();
}
hipError_t launch_amax_pair(hipStream_t stream, int dtype, size_t k, const ExtractOperand& A, const ExtractOperand& B) {
    switch (dtype) {
    case kF32: return launch_amax_pair_t<float>(stream, k, A, B);
    case kF64: return launch_amax_pair_t<double>(stream, k, A, B);
    case kC32: return launch_amax_pair_t<float2>(stream, k, A, B);
    case kC64: return launch_amax_pair_t<double2>(stream, k, A, B);
    }
    return hipErrorInvalidValue;
}
template <typename T> static hipError_t launch_extract_pair_t(hipStream_t stream, bool kmA, StageArgs a, bool kmB, const StageArgs& b, void* zero_p, size_t zero_bytes) {
    const size_t nA = stage_blocks<T, MODE_BOUND>(kmA, a), nB = stage_blocks<T, MODE_BOUND>(kmB, b);
    if (nA + nB == 0) return zero_bytes ? launch_zero(stream, zero_p, zero_bytes) : hipSuccess;
    if (nA + nB > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    if (zero_p && zero_bytes) a.zero_p = (unsigned*)zero_p, a.zero_words = (unsigned)(zero_bytes / 4);
    hipLaunchKernelGGL(extract_pair_kernel<T>, dim3((unsigned)(nA + nB), 1, g_batch.batch), dim3(256), 0, stream, a, b, (unsigned)nA, (int)kmA, (int)kmB);
    return hipGetLastError();
}
// extract of both operands (rows == 0: absent) in ONE launch; zero_p / zero_bytes: words the launch also zero-fills (the bound GEMM's maxima arrays)
hipError_t launch_extract_pair(hipStream_t stream, int dtype, int backend, size_t k, size_t kp, const ExtractOperand& A, const ExtractOperand& B, void* zero_p,
                               size_t zero_bytes) {
    const StageArgs a = extract_args(backend, k, kp, A), b = extract_args(backend, k, kp, B);
    switch (dtype) {
    case kF32: return launch_extract_pair_t<float>(stream, A.kmajor, a, B.kmajor, b, zero_p, zero_bytes);
    case kF64: return launch_extract_pair_t<double>(stream, A.kmajor, a, B.kmajor, b, zero_p, zero_bytes);
    case kC32: return launch_extract_pair_t<float2>(stream, A.kmajor, a, B.kmajor, b, zero_p, zero_bytes);
    case kC64: return launch_extract_pair_t<double2>(stream, A.kmajor, a, B.kmajor, b, zero_p, zero_bytes);
    }
    return hipErrorInvalidValue;
}

static StageArgs quantise_args(int backend, int t_begin, int t_end, size_t k, size_t kp, const QuantOperand& o) {
    StageArgs a{};
    a.bx = o.xstride;
    a.bw = g_batch.ws;
    a.X = o.X;
    a.ld = o.ld;
    a.rows = o.rows;
    a.k = k;
    a.kp = kp;
    a.lo = o.lo;
    a.plane_stride = o.plane_stride;
    a.part_stride = o.part_stride;
    a.sft = o.sft;
    if (o.fin_max) {  // accurate mode: shift finalize folded into this launch
        a.fin_sft0 = o.fin_sft0;
        a.fin_max = o.fin_max;
        a.fin_out = o.sft;
        a.fin_log2P = o.fin_log2P;
        a.fin_float_max = backend == kFP8 ? 1 : 0;
    }
    a.backend = backend;
    a.conj = o.conj;
    a.t_begin = t_begin;
    a.t_end = t_end;
    a.mt = make_mod_table(backend);
    for (int t = t_begin, j = 0; t < t_end && j < 10; t += 2, ++j) {
        const double P = (double)a.mt.mc[t].p * (t + 1 < t_end ? (double)a.mt.mc[t + 1].p : 1.0);
        a.pairP[j] = P;
        a.pairInvP[j] = 1.0 / P;
    }
    for (int t = 0; t < 6; ++t) a.sqrtp[t] = GEMMUL8_SQRT_MODULI_FP8[t];
    if (backend == kFP8 && o.f6_rows > 0) {
        a.f6 = 1;
        a.f6_last = (unsigned)((o.f6_rows - 1) / 256);
        a.f6_rp_last = (unsigned)((o.f6_rows - 256 * (size_t)a.f6_last + 15) / 16 * 16);
    }
    return a;
}

hipError_t launch_quantise_pair(hipStream_t stream, int dtype, int backend, int t_begin, int t_end, size_t k, size_t kp, const QuantOperand& A,
                                const QuantOperand& B) {
    if ((A.rows == 0 && B.rows == 0) || t_end <= t_begin) return hipSuccess;
    const StageArgs a = quantise_args(backend, t_begin, t_end, k, kp, A), b = quantise_args(backend, t_begin, t_end, k, kp, B);
    switch (dtype) {
    case kF32: return launch_quantise_stage<float>(stream, A.kmajor, a, B.kmajor, b);
    case kF64: return launch_quantise_stage<double>(stream, A.kmajor, a, B.kmajor, b);
    case kC32: return launch_quantise_stage<float2>(stream, A.kmajor, a, B.kmajor, b);
    case kC64: return launch_quantise_stage<double2>(stream, A.kmajor, a, B.kmajor, b);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------ accurate-mode shift from the bound maxima
// rows of A (blocks 0 .. blocksA-1) and columns of B (the remaining blocks) in ONE launch; either count may be 0
__global__ void shift_finalize_kernel(size_t rowsA, const int* maxA, int16_t* sftA, unsigned blocksA, size_t rowsB, const int* maxB,
                                      int16_t* sftB, float log2P, int float_max, size_t bw) {
    {  // batched launch: item blockIdx.z (all four arrays live in the item's workspace)
        const size_t o = blockIdx.z * bw;
        maxA = (const int*)((const char*)maxA + o), maxB = (const int*)((const char*)maxB + o);
        sftA = (int16_t*)((char*)sftA + o), sftB = (int16_t*)((char*)sftB + o);
    }
    const bool isB = blockIdx.x >= blocksA;
    const size_t r = (size_t)(isB ? blockIdx.x - blocksA : blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= (isB ? rowsB : rowsA)) return;
    const int amax = (isB ? maxB : maxA)[r];  // INT8: int32 maximum; FP8: bit pattern of a non-negative float
    int f = 0;  // all-zero row/column: the reference is undefined here (log2(0)); any shift is valid
    if (amax > 0) {
        const float l = __log2f(float_max ? __int_as_float(amax) : __int2float_rn(amax));
        f = __float2int_rd(__fmaf_rd(-0x1.000006p-1f, l, log2P));
    }
    int16_t* sft = isB ? sftB : sftA;
    sft[r] = (int16_t)(-((int)sft[r] + f));
}
hipError_t launch_shift_finalize(hipStream_t stream, int backend, unsigned N, size_t rowsA, const int* maxA, int16_t* sftA, size_t rowsB,
                                 const int* maxB, int16_t* sftB) {
    const unsigned bA = (unsigned)((rowsA + 255) / 256), bB = (unsigned)((rowsB + 255) / 256);
    if (bA + bB == 0) return hipSuccess;
    const float log2P = backend == kINT8 ? GEMMUL8_LOG2P_INT8[N - 2] : GEMMUL8_LOG2P_FP8[N - 2];
    hipLaunchKernelGGL(shift_finalize_kernel, dim3(bA + bB, 1, g_batch.batch), dim3(256), 0, stream, rowsA, maxA, sftA, bA, rowsB, maxB, sftB, log2P,
                       backend == kFP8 ? 1 : 0, g_batch.ws);
    return hipGetLastError();
}

// ------------------------------------------------------------------ fast-mode shifts
// The round-up sums are order dependent; both kernels keep the reference's order so that the
// shifts are the ones its HIP build would produce: per-lane strided chains, width-32 shuffle
// trees, then a tree over the group sums (find_max.hpp:258-341, template_math.hpp:179-212).
template <typename U> __device__ __forceinline__ U add_ru(U a, U b);
template <> __device__ __forceinline__ float add_ru<float>(float a, float b) { return __fadd_ru(a, b); }
template <> __device__ __forceinline__ double add_ru<double>(double a, double b) { return __dadd_ru(a, b); }
template <typename U> __device__ __forceinline__ U sqr_add_ru(U x, U s);
template <> __device__ __forceinline__ float sqr_add_ru<float>(float x, float s) { return __fmaf_ru(x, x, s); }
template <> __device__ __forceinline__ double sqr_add_ru<double>(double x, double s) { return __fma_ru(x, x, s); }

template <typename U> __device__ __forceinline__ U tree32_sum_ru(U v) {  // result valid in lane (l & 31) == 0
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v = add_ru<U>(v, __shfl_down(v, off, 32));
    return v;
}
template <typename U> __device__ __forceinline__ U tree32_max(U v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        const U o = __shfl_down(v, off, 32);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ int fast_sft(double amax, double vecnrm, float log2P) {
    const int exponent = ilogb0(vecnrm);
    const float vecnrmf = __double2float_ru(scalbn(vecnrm, -exponent));
    const float log2vsum = __fadd_ru(__log2f(vecnrmf), (float)exponent);
    const float log2vnrm = __fmul_ru(0x1.000006p-1f, log2vsum);
    const float exp1 = __fsub_rd(__fsub_rd(log2P, 1.5f), fmaxf(1.0f, log2vnrm));
    return __float2int_rd(exp1) - ilogb0((float)amax);
}
__device__ __forceinline__ int fast_sft(float amax, float vecnrm, float log2P) {
    const float log2vsum = __log2f(vecnrm);
    const float log2vnrm = __fmul_ru(0x1.000006p-1f, log2vsum);
    const float exp1 = __fsub_rd(__fsub_rd(log2P, 1.5f), fmaxf(1.0f, log2vnrm));
    return __float2int_rd(exp1) - ilogb0(amax);
}

// K-major: one 256-thread block per row (scaling_fast_real.hpp:142-164)
template <typename T> __device__ __forceinline__ void fast_shift_kmajor_body(const T* X, size_t ld, size_t k, int16_t* sft, float log2P, const unsigned bid) {
    using E = ET<T>;
    using U = typename E::U;
    __shared__ U samax[32], ssum[32];
    const T* x = X + (size_t)bid * ld;
    U amax = 0, sum = 0;
    size_t i = threadIdx.x;
    for (; i + 3 * 256 < k; i += 4 * 256) {  // four loads ahead of their (sequential) round-up FMAs
        T v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = x[i + 256 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const U ar = (U)fabs(E::re(v[u]));
            amax = ar > amax ? ar : amax;
            sum = sqr_add_ru<U>(ar, sum);
            if constexpr (E::cplx) {
                const U ai = (U)fabs(E::im(v[u]));
                amax = ai > amax ? ai : amax;
                sum = sqr_add_ru<U>(ai, sum);
            }
        }
    }
    for (; i < k; i += 256) {
        const T v = x[i];
        const U ar = (U)fabs(E::re(v));
        amax = ar > amax ? ar : amax;
        sum = sqr_add_ru<U>(ar, sum);
        if constexpr (E::cplx) {
            const U ai = (U)fabs(E::im(v));
            amax = ai > amax ? ai : amax;
            sum = sqr_add_ru<U>(ai, sum);
        }
    }
    amax = tree32_max(amax);
    sum = tree32_sum_ru(sum);
    if ((threadIdx.x & 31) == 0) {
        samax[threadIdx.x >> 5] = amax;
        ssum[threadIdx.x >> 5] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        amax = threadIdx.x < 8 ? samax[threadIdx.x] : (U)0;
        sum = threadIdx.x < 8 ? ssum[threadIdx.x] : (U)0;
        amax = tree32_max(amax);
        sum = tree32_sum_ru(sum);
        if (threadIdx.x == 0) sft[bid] = (int16_t)(-fast_sft(amax, sum, log2P));
    }
}

// Row-strided (scaling_fast_real.hpp:27-49: 32 rows x 32 k-lanes per block).  The reduction ORDER is the reference's: lane ty of a row adds
// the squares of columns ty, ty + 32, ... in sequence, then a width-32 tree over the 32 lanes.  Geometry here: 256 threads = 8 lanes per
// column x 32 k-lanes; a lane owns RPL = 16 / sizeof(T) consecutive rows (one 16-byte load per column when aligned), a workgroup 8 RPL rows
// = one 128-byte line per column -- 512 workgroups for 8192 rows of doubles where the 32 x 32 form had 256 and 8-byte loads.  The loads of
// eight chain steps are issued ahead of their round-up FMAs: the chain itself is sequential, and with one load per step it ran at one
// memory latency per element (289 us for 1024 x 16384 doubles; 128 MiB).
template <typename T> __device__ __forceinline__ void fast_shift_strided_body(const T* X, size_t ld, size_t rows, size_t k, int16_t* sft, float log2P, const unsigned bid) {
    using E = ET<T>;
    using U = typename E::U;
    constexpr int RPL = 16 / (int)sizeof(T), RPB = 8 * RPL;
    __shared__ U samax[32][RPB + 1], ssum[32][RPB + 1];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const size_t row0 = (size_t)bid * RPB + (size_t)tx * RPL;
    U amax[RPL], sum[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) amax[j] = 0, sum[j] = 0;
    auto take = [&](const T& v, int j) {
        const U ar = (U)fabs(E::re(v));
        amax[j] = ar > amax[j] ? ar : amax[j];
        sum[j] = sqr_add_ru<U>(ar, sum[j]);
        if constexpr (E::cplx) {
            const U ai = (U)fabs(E::im(v));
            amax[j] = ai > amax[j] ? ai : amax[j];
            sum[j] = sqr_add_ru<U>(ai, sum[j]);
        }
    };
    if (row0 < rows) {
        const T* x = X + row0;
        const bool vec = row0 + RPL <= rows && ((reinterpret_cast<uintptr_t>(x) | (ld * sizeof(T))) & 15u) == 0;
        size_t col = ty;
        if (vec) {
            typedef unsigned V4 __attribute__((ext_vector_type(4)));
            // loads in flight per thread: 8 x 16 bytes; 4-byte elements have half as many workgroups (8 x 4 rows each: 256 at 8192 rows,
            // one per CU), so they keep 16 in flight (8192^2 float: 88 us = 3.0 TB/s with 8).  The order of the accumulation -- columns
            // ascending per thread -- does not depend on the depth.
            constexpr int UNR = sizeof(T) == 4 ? 16 : 8;
            if constexpr (UNR > 8) {
                for (; col + (UNR - 1) * 32 < k; col += UNR * 32) {
                    V4 raw[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) raw[u] = *(const V4*)(x + (col + 32 * u) * ld);
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        T v[RPL];
                        __builtin_memcpy(v, &raw[u], 16);
#pragma unroll
                        for (int j = 0; j < RPL; ++j) take(v[j], j);
                    }
                }
            }
            for (; col + 7 * 32 < k; col += 8 * 32) {
                V4 raw[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) raw[u] = *(const V4*)(x + (col + 32 * u) * ld);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    T v[RPL];
                    __builtin_memcpy(v, &raw[u], 16);
#pragma unroll
                    for (int j = 0; j < RPL; ++j) take(v[j], j);
                }
            }
            for (; col < k; col += 32) {
                T v[RPL];
                const V4 raw = *(const V4*)(x + col * ld);
                __builtin_memcpy(v, &raw, 16);
#pragma unroll
                for (int j = 0; j < RPL; ++j) take(v[j], j);
            }
        } else {
            for (; col < k; col += 32) {
#pragma unroll
                for (int j = 0; j < RPL; ++j)
                    if (row0 + j < rows) take(x[col * ld + j], j);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
        samax[ty][tx * RPL + j] = amax[j];
        ssum[ty][tx * RPL + j] = sum[j];
    }
    __syncthreads();
    // thread r * 32 + l holds lane l of row r: width-32 trees, 8 rows per sweep
    const int ll = threadIdx.x & 31;
    for (int rr = threadIdx.x >> 5; rr < RPB; rr += 8) {
        const U s = tree32_sum_ru(ssum[ll][rr]);
        const U m = tree32_max(samax[ll][rr]);
        const size_t row = (size_t)bid * RPB + rr;
        if (row < rows && ll == 0) sft[row] = (int16_t)(-fast_sft(m, s, log2P));
    }
}

// both operands in one launch: workgroups [0, a.blocks) take the rows of A, the rest the columns of B; either share may be empty
struct ShiftOperand {
    const void* X;
    size_t ld, rows, bx;
    int16_t* sft;
    unsigned blocks;
    int kmajor;
};
template <typename T> __global__ void __launch_bounds__(256) fast_shift_pair_kernel(const ShiftOperand a, const ShiftOperand b, size_t k, float log2P, size_t bw) {
    const bool isB = blockIdx.x >= a.blocks;
    const ShiftOperand& o = isB ? b : a;  // (kernel arguments: a scalar select per field, no copy)
    const unsigned bid = isB ? blockIdx.x - a.blocks : blockIdx.x;
    const T* X = (const T*)((const char*)o.X + blockIdx.z * o.bx);  // batched launch: item blockIdx.z
    int16_t* sft = (int16_t*)((char*)o.sft + blockIdx.z * bw);
    if (o.kmajor) fast_shift_kmajor_body<T>(X, o.ld, k, sft, log2P, bid);
    else fast_shift_strided_body<T>(X, o.ld, o.rows, k, sft, log2P, bid);
}

hipError_t launch_fast_shift_pair(hipStream_t stream, int dtype, int backend, unsigned N, size_t k, const QuantOperand& A, const QuantOperand& B) {
    if (A.rows == 0 && B.rows == 0) return hipSuccess;
    const float log2P = backend == kINT8 ? GEMMUL8_LOG2P_INT8[N - 2] : GEMMUL8_LOG2P_FP8[N - 2];
    const size_t rpb = 8 * (16 / (is_f32(dtype) ? 4 : 8) / (is_complex(dtype) ? 2 : 1));  // rows per workgroup of the row-strided form
    auto operand = [&](const QuantOperand& o) {
        const size_t blocks = o.kmajor ? o.rows : (o.rows + rpb - 1) / rpb;
        return ShiftOperand{o.X, o.ld, o.rows, o.xstride, o.sft, (unsigned)blocks, o.kmajor ? 1 : 0};
    };
    if (A.rows + B.rows > 0x7FFFFFFFull) return hipErrorInvalidConfiguration;
    const ShiftOperand a = operand(A), b = operand(B);
    dim3 grid(a.blocks + b.blocks, 1, g_batch.batch);
    switch (dtype) {
    case kF32: hipLaunchKernelGGL(fast_shift_pair_kernel<float>, grid, dim3(256), 0, stream, a, b, k, log2P, g_batch.ws); break;
    case kF64: hipLaunchKernelGGL(fast_shift_pair_kernel<double>, grid, dim3(256), 0, stream, a, b, k, log2P, g_batch.ws); break;
    case kC32: hipLaunchKernelGGL(fast_shift_pair_kernel<float2>, grid, dim3(256), 0, stream, a, b, k, log2P, g_batch.ws); break;
    case kC64: hipLaunchKernelGGL(fast_shift_pair_kernel<double2>, grid, dim3(256), 0, stream, a, b, k, log2P, g_batch.ws); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace oz2
