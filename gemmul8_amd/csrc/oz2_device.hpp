// Device-side helpers shared by the HIP kernels of the Ozaki-II emulation (gfx950 only).
//
// What the reference does with compile-time unrolled templates over num_moduli
// (GEMMul8/src/mod.hpp:638-877, scaling.hpp:237-280) is done here with ONE exact integer
// representation of trunc(x*2^s) = +-M*2^E (M < 2^53, E >= 0) and a byte-wise residue that runs at
// full VALU rate (v_dot4_u32_u8 byte sums, one fma quotient, one 24-bit multiply-add); num_moduli stays a run-time loop bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tables.inc"

namespace oz2 {

constexpr int kINT8 = 0;
constexpr int kFP8 = 1;

// per-modulus constants of the byte-wise residue (built on the host, passed by value in kernargs)
struct ModConst {
    int p;        // modulus
    float invp;   // RN(1/p)
    unsigned cb[4];  // cb[i] = bytes (c_{4i}, .., c_{4i+3}), c_j = 256^j mod p in [0, p) for the INT8 moduli (p <= 256), the LOW 5
                     // bits of it for the FP8 moduli (p <= 1089); byte 15 is 0
    unsigned cbh[4]; // FP8 moduli: the high part c_j >> 5 (< 35); 0 for INT8
    unsigned k56;    //              (-2^56) mod p in [0, p): correction for the 56-bit two's complement of a negative value
    unsigned k120;   //              (-2^120) mod p: the same for the 120-bit form used when E > 0
};
struct ModTable {
    ModConst mc[20];
};

__device__ __forceinline__ int ilogb0(double x) { return x == 0.0 ? 0 : ilogb(x); }
__device__ __forceinline__ int ilogb0(float x) { return x == 0.0f ? 0 : ilogbf(x); }

// exact trunc(|x| * 2^sft) = M * 2^E  (restates scaling.hpp:99-235 as one representation)
struct ScaledInt {
    uint64_t M;
    int E;
    bool neg;
};
__device__ __forceinline__ ScaledInt trunc_scale(double x, int sft) {
    const uint64_t bits = (uint64_t)__double_as_longlong(x);
    ScaledInt r;
    r.neg = (bits >> 63) != 0;
    int e = (int)((bits >> 52) & 0x7FF);
    uint64_t mant = bits & 0xFFFFFFFFFFFFFull;
    if (e) mant |= (1ull << 52);
    else e = 1;
    const int x2 = e - 1075 + sft;  // value = mant * 2^x2
    if (x2 >= 0) {
        r.M = mant;
        r.E = x2;
    } else {
        r.M = (-x2 >= 64) ? 0ull : (mant >> (-x2));
        r.E = 0;
    }
    if (r.M == 0) r.neg = false, r.E = 0;
    return r;
}

// ceil(|x| * 2^sft) as int8: exact ceiling, tiny non-zero -> 1 (scaling.hpp:3-46)
__device__ __forceinline__ int upper_bound_i8(double x, int sft) {
    const uint64_t bits = ((uint64_t)__double_as_longlong(x)) & ~(1ull << 63);
    if (bits == 0) return 0;
    int e = (int)(bits >> 52);
    uint64_t mant = bits & 0xFFFFFFFFFFFFFull;
    if (e) mant |= (1ull << 52);
    else e = 1;
    const int x2 = e - 1075 + sft;
    if (x2 >= 0) return (int)(int8_t)(mant << (x2 > 63 ? 63 : x2));
    if (-x2 >= 64) return 1;
    const uint64_t fl = mant >> (-x2);
    const uint64_t has = (mant & ((1ull << (-x2)) - 1)) != 0;
    return (int)(int8_t)(fl + has);
}

// INT8 moduli (p <= 256): symmetric residue of +-M*2^E (M < 2^53, E < 64) from the BYTES of the 120-bit integer M*2^E
// -- or of its two's complement 2^120 - M*2^E for a negative value, corrected by k120 = (-2^120 mod p) in the accumulator
// input: sum_i b_i * (256^i mod p) < 2^20 by four v_dot4_u32_u8, then one quotient step from a single fma
// (float(s)/p + 2^23 rounds to 2^23 + q; s * |RN(1/p) - 1/p| <= 1/(16p) keeps it clear of the 1/(2p) tie distance of an
// odd p, so the result is the canonical representative; p = 256: a tie gives +-128, the same int8 byte).  No 2^E mod p
// table, no second reduction.  Only needed for num_moduli > 15 (|A'| >= 2^53).
struct Bytes128 {
    unsigned w[4];
};
__device__ __forceinline__ Bytes128 shifted_bytes(uint64_t M, int E, bool neg) {
    E = E < 63 ? E : 63;  // the algorithm keeps E <= ~26 (|A'| < sqrt(P)); the clamp only guards the shift itself
    uint64_t lo = M << E;
    uint64_t hi = E ? (M >> (64 - E)) : 0ull;
    if (neg) {  // 2^120 - X (X != 0 whenever neg is set)
        lo = ~lo + 1ull;
        hi = ~hi + (lo == 0 ? 1ull : 0ull);
        hi &= 0x00FFFFFFFFFFFFFFull;
    }
    return Bytes128{{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)}};
}
// WIDE = false: INT8 moduli.  WIDE = true: FP8 moduli (up to 1089): the constants 256^j mod p do not fit a byte, so they are
// split c = 32 c_hi + c_lo and the sum is s = sum b c_lo + 32 sum b c_hi < 2^23 (120-bit form; < 2^21 for 56 bits) -- still
// exact in fp32, and s * |RN(1/p) - 1/p| <= 2^23 * 2^-24 / p < 1/(2p) keeps the single fma quotient exact for odd p.
// p = 1024 is even: the tie -512 is moved to the reference's representative +512 ((-p/2, p/2], mod.hpp:8-12).
template <bool WIDE> __device__ __forceinline__ int finish_residue(unsigned s, const ModConst& mc) {
    const float qf = fmaf((float)s, mc.invp, 8388608.0f);
    // pinned to the full-rate 24-bit multiply-add: left to itself the compiler sees that only the low bits of the result
    // are stored and picks the quarter-rate v_mul_lo_u32 / v_mad_u64_u32
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(__float_as_int(qf)), "s"(-mc.p), "v"(s));
    if constexpr (WIDE) {
        if (!(mc.p & 1)) r = (r == -(mc.p >> 1)) ? (mc.p >> 1) : r;
    }
    return r;
}
template <bool WIDE> __device__ __forceinline__ int residue_sym_bytes128(const Bytes128& x, bool neg, const ModConst& mc) {
    unsigned s = __builtin_amdgcn_udot4(x.w[3], mc.cb[3], neg ? mc.k120 : 0u, false);
    s = __builtin_amdgcn_udot4(x.w[2], mc.cb[2], s, false);
    s = __builtin_amdgcn_udot4(x.w[1], mc.cb[1], s, false);
    s = __builtin_amdgcn_udot4(x.w[0], mc.cb[0], s, false);
    if constexpr (WIDE) {
        unsigned h = __builtin_amdgcn_udot4(x.w[3], mc.cbh[3], 0u, false);
        h = __builtin_amdgcn_udot4(x.w[2], mc.cbh[2], h, false);
        h = __builtin_amdgcn_udot4(x.w[1], mc.cbh[1], h, false);
        h = __builtin_amdgcn_udot4(x.w[0], mc.cbh[0], h, false);
        s += h << 5;
    }
    return finish_residue<WIDE>(s, mc);
}

// The common case, E = 0 (always true for num_moduli <= 15): Mt is M, or the 56-bit two's complement 2^56 - M of a negative
// value, whose byte sum plus k56 = (-2^56 mod p) is congruent to -M.  The quotient comes from ONE fma: float(s)/p + 2^23
// rounds to the integer 2^23 + q (RN-even at unit spacing), whose low 24 bits are q, exactly what v_mad_i32_i24 reads.
// INT8 moduli: 6 VALU operations (v_cndmask, 2 x v_dot4_u32_u8, v_cvt_f32_u32, v_fma_f32, v_mad_i32_i24).
template <bool WIDE> __device__ __forceinline__ int residue_sym_bytes_e0(unsigned Mt_lo, unsigned Mt_hi, bool neg, const ModConst& mc) {
    unsigned s = __builtin_amdgcn_udot4(Mt_lo, mc.cb[0], __builtin_amdgcn_udot4(Mt_hi, mc.cb[1], neg ? mc.k56 : 0u, false), false);
    if constexpr (WIDE) s += __builtin_amdgcn_udot4(Mt_lo, mc.cbh[0], __builtin_amdgcn_udot4(Mt_hi, mc.cbh[1], 0u, false), false) << 5;
    return finish_residue<WIDE>(s, mc);
}

// wrapping (mod.hpp:8-12)
__device__ __forceinline__ int wrapping(int a, int p) {
    const int h = p >> 1;
    return (a > h) ? a - p : ((a < -h) ? a + p : a);
}

// symmetric residue of an int32 accumulator: the reference's mod_small (mod.hpp:15-21,58-60);
// result in [-p/2, p/2] with the same representative (checked exhaustively in tests)
__device__ __forceinline__ int mod_i32_sym(int a, int p, int pinv32) {
    const int rem = a - p * __mulhi(a, pinv32);
    return wrapping(rem, p);
}

// The same residue for ODD p and ANY int32 a with full-rate instructions only (v_mul_hi/v_mul_lo_u32 are quarter rate; FP64
// VALU runs at the FP32 rate on gfx950): ONE quotient step in FP64.  a*invp is within 2^-21 of a/p, which is at least 1/(2p)
// away from a rounding tie for odd p, so q = rint(a/p) exactly and a - q*p (integers below 2^53: the fma is exact) is the
// canonical representative in [-(p-1)/2, (p-1)/2] -- identical to mod_i32_sym (CPU model over the whole int32 range in
// tests/test_residue_math.py, bit-for-bit by the GPU parity tests).  Five instructions against ten for two fp32 steps.
__device__ __forceinline__ int mod_i32_sym_odd_f64(int a, double p, double invp) {
    const double x = (double)a;
    const double q = rint(x * invp);
    return (int)fma(-q, p, x);
}
// 0 <= s < 2^22, odd p: s * |RN(1/p) - 1/p| <= 2^22 2^-24 / p < 1/(2p), so fma(s, RN(1/p), 2^23) rounds (RN-even at unit spacing) to
// 2^23 + q with q = rint(s / p) exactly; the low 24 bits of its bit pattern are q, which v_mad_i32_i24(bits, -p, s) = s - q p turns
// into the canonical residue in [-(p-1)/2, (p-1)/2] (same step as finish_residue).
__device__ __forceinline__ int mod_small_sym_u(unsigned s, int p, float invp) {
    const float qf = fmaf((float)s, invp, 8388608.0f);
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(__float_as_int(qf)), "s"(-p), "v"(s));
    return r;
}
// |a| < 2^16: one exact step
__device__ __forceinline__ int mod_small_sym_odd(int a, int p, float invp) {
    return a - __mul24((int)rintf((float)a * invp), p);
}

// ---------------------------------------------------------------- OCP FP8 e4m3 helpers (FP8 backend)
// two small integers (|v| <= 16, exactly representable) -> two e4m3 bytes in the low half of the result
__device__ __forceinline__ unsigned fp8x2_from_ints(int a, int b) {
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32((float)a, (float)b, 0, false) & 0xFFFFu;
}
// smallest e4m3 value >= a for 0 <= a < 448 (restates fp8_e4m3_ru, scaling.hpp:48-54: RN conversion, then
// one encoding step up if the result fell below a)
template <typename U> __device__ __forceinline__ unsigned fp8_round_up(U a) {
    unsigned r = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32((float)a, 0.0f, 0, false) & 0xFFu;
    const float y = __builtin_amdgcn_cvt_f32_fp8((int)r, 0);
    return r + ((U)y < a ? 1u : 0u);
}
__device__ __forceinline__ float fp8_to_float(unsigned byte) { return __builtin_amdgcn_cvt_f32_fp8((int)byte, 0); }

// FP8 residue splitting (mod.hpp:159-189).  Square moduli p = s^2 (t < 6): a = s*hi + lo, hi = rint(a/s);
// otherwise a = 16*hi + lo with hi = sign(a)*ceil(|a|/16) and a third value hi + lo.
// (hi keeps the sign of zero: rintf(-0.09) = -0.0f is stored as the e4m3 byte 0x80, exactly like the reference)
__device__ __forceinline__ void fp8_split_sq(int a, int s, float inv_s, float& hi, float& lo) {
    const float af = (float)a;
    hi = rintf(af * inv_s);
    lo = fmaf(-(float)s, hi, af);
}
__device__ __forceinline__ unsigned fp8x2_from_floats(float a, float b) {
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xFFFFu;
}
__device__ __forceinline__ void fp8_split_kara(int a, int& hi, int& lo) {
    const unsigned absu = (unsigned)(a < 0 ? -a : a);
    const int q = (int)((absu + 15u) >> 4);
    hi = a < 0 ? -q : q;
    lo = a - 16 * hi;
}

}  // namespace oz2
