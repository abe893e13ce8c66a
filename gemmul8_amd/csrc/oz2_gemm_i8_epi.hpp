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
#if defined(OZ2_PRODUCT_BUILD) && (defined(OZ2_LAB_HOOKS) || defined(OZ2_LAB_FUSED_CRT) || defined(OZ2_LAB_SHORTK) || defined(OZ2_LAB_SELFPIPE) || defined(OZ2_LAB_W4))
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
#if defined(OZ2_PRODUCT_BUILD) && OZ2_HOOK_SKIP_EPILOGUE
#error "timing probes (OZ2_HOOK_SKIP_*) compute something else: not allowed in the product build of libgemmul8.so"
#endif

enum { EPI_MOD = 0, EPI_MAX = 1, EPI_CPLX = 2, EPI_MOD256 = 3 };  // EPI_MOD256 (round 6): EPI_MOD of launches with K <= 256, accumulators carried as float patterns (RED_MAGIC)

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
    TileMapArgs map;       // the same with the divisors' magic numbers (make_tile_map; filled by launch<EPI>)
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
    unsigned m_ppi;        // floor(2^32 / ppi) (map_magic): plane_ref divides on the scalar unit
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
    unsigned b, tt;  // (a plain p / ppi leaves a hoisted float reciprocal in a VGPR across every K loop: oz2_gemm_common.hpp, udivmod_magic)
    udivmod_magic((unsigned)p, (unsigned)args.ppi, args.m_ppi, b, tt);
    return {(size_t)b * args.bstride, (int)tt};
}
// the per-modulus constants of the residue epilogues (scalar loads from the argument block, indexed by the plane): the persistent kernels fetch them
// at the START of a tile, so that their latency passes behind the K loop instead of in front of the epilogue
struct PlaneConsts {
    int p;
    unsigned dotw, dotc;
};
__device__ __forceinline__ PlaneConsts plane_consts(const GemmArgs& args, PlaneRef pl) {
    const int t = args.t_begin + pl.tt;
    return {args.moduli[t], args.dotw[t], args.dotc[t]};
}
enum { RED_ODD = 1, RED_ODD_SMALL = 3, RED_MAGIC = 4 };
// RED selects how an accumulator is reduced to its residue's low byte -- one form per kernel instantiation (i8_epilogue): RED_ODD for any int32
// accumulator (byte dot product on the biased accumulator + one fp32 quotient, below), RED_ODD_SMALL for launches with K <= 512.  p = 256 runs through
// the same forms: every quotient leaves the low byte of the accumulator in place.  History: an FP64 quotient step (5 instructions; the two-step fp32
// form before it cost 10 and 12 % of the kernel time at k = 1024); reading the FP64 quotient from the low dword of fma(a, 1/p, 1.5 * 2^52) --
// three instructions -- measured 10 % SLOWER at k = 1024 (the dependent FP64 chains no longer overlap); separate run-time forms for p = 256 (low
// byte) and for even p (32-bit multiply-high, no INT8 modulus needs it) existed until round 4 and cost far more than they saved (i8_epilogue).
// Hook: called as hook(s, 0) before and hook(s, 1) after the stores of sub-block s = 2 tj + tg (64 rows x 16 columns; 8 per wave tile); NoHook
// (nothing) in every product kernel -- the laboratory short-K kernel places its workgroup barriers there (tools/experiments/shortk).
struct NoHook {
    __device__ __forceinline__ void operator()(int, int) const {}
};
// 16-byte store of the lanes whose column exists (col < n), WITHOUT a branch: the execution mask is narrowed and restored inside one asm block.
// A divergent `if (col < n)` around every store made the whole epilogue a non-uniform region, and the structurizer then LINEARISED the (uniform) chain
// of reduction variants around it -- variant after variant in one straight line, each guarded by its predicate --, so that the accumulators stayed
// live through the complete epilogue of every variant but the last: no accumulator register could be reused, the residues, the X / Y values and the
// store addresses of the complex combine went to scratch (64-144 bytes), and every reload in between is a vector-memory wait that serialises the
// epilogue's loads (see the EPI_CPLX branch below).  The waitcnt pass does not see the store; hidden stores only make its vmcnt waits conservative.
// The s_nop 1 behind the store is REQUIRED (round 6): a VALU write to the data registers of a store of more than 8 bytes needs two wait states on gfx940 / gfx950
// (LLVM GCNHazardRecognizer, VMEM store-data hazard); the compiler cannot see a store inside an asm block, so it neither counts nor inserts them.  Found when a new
// kernel instantiation re-used the data registers right behind the block: non-temporal stores (read a little later) wrote half-overwritten residues
// (tests/test_kernel_resources.py::test_inline_asm_stores_carry_their_wait_states pins the two wait states).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#ifndef OZ2_HOOK_SKIP_STORES
#define OZ2_HOOK_SKIP_STORES 0  // timing probe (lab_hooks.hpp, OZ2_PROBE & 64): the epilogue's arithmetic without its stores
#endif
#if defined(OZ2_PRODUCT_BUILD) && OZ2_HOOK_SKIP_STORES
#error "timing probes (OZ2_HOOK_SKIP_*) compute something else: not allowed in the product build of libgemmul8.so"
#endif
template <bool NT> __device__ __forceinline__ void store16_cols(void* ptr, v4u d, int col, int n) {
#if OZ2_HOOK_SKIP_STORES
    asm volatile("" ::"v"(ptr), "v"(d), "v"(col), "s"(n));
    return;
#endif
    unsigned long long saved;
    if constexpr (NT)
        asm volatile("s_mov_b64 %0, exec\n\tv_cmp_gt_i32_e32 vcc, %1, %2\n\ts_and_b64 exec, exec, vcc\n\tglobal_store_dwordx4 %3, %4, off nt\n\ts_nop 1\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "s"(n), "v"(col), "v"(ptr), "v"(d) : "vcc", "scc", "memory");
    else
        asm volatile("s_mov_b64 %0, exec\n\tv_cmp_gt_i32_e32 vcc, %1, %2\n\ts_and_b64 exec, exec, vcc\n\tglobal_store_dwordx4 %3, %4, off\n\ts_nop 1\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "s"(n), "v"(col), "v"(ptr), "v"(d) : "vcc", "scc", "memory");
}
template <int EPI, int RED, typename Hook = NoHook>
__device__ __forceinline__ void i8_epilogue_mod(const v4i (&acc)[8][4], const GemmArgs& args, PlaneRef pl, PlaneConsts pc, int i0, int j0, int lane, Hook hook = {}) {
    const int c16 = lane & 15;
    const int q = lane >> 4;
    const int p = pc.p;
    const float invp = 1.0f / (float)p;
    [[maybe_unused]] const unsigned dotw = pc.dotw, dotc = pc.dotc;
    static_assert(RED == RED_ODD || RED == RED_ODD_SMALL || RED == RED_MAGIC, "one of the three reduction forms");
    constexpr bool MODLIKE = EPI == EPI_MOD || EPI == EPI_MOD256;
    // RED_ODD_SMALL, short K (kp * nseg <= 512: |x| <= 512 * 127^2 < 2^23; the accumulators start at 0, GemmArgs.acc0): the quotient comes
    // straight from the accumulator -- v_cvt_f32_i32, one fma against 1.5 * 2^23 (its low 24 bits are 2^22 + q for either sign
    // of q), v_mad_i32_i24: the canonical residue minus p 2^22, i.e. the canonical LOW BYTE, which is all the epilogue stores.
    // Three instructions.  The bound is 2^23, not the 2^24 of fp32 exactness: |x| |RN(1/p) - 1/p| must stay
    // below the 1/(2p) that separates x / p from a rounding tie (exhaustive CPU model: first wrong byte at |x| = 8 454 907 for
    // p = 255, tests/test_residue_math.py; tests/test_gpu_parity.py::test_epilogue_reduction_on_extreme_accumulators).
    [[maybe_unused]] auto red = [&](int x) {
        const float qf = fmaf((float)x, invp, 12582912.0f);
        int r;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(__float_as_int(qf)), "s"(-p), "v"(x));
        return r;
    };
    // RED_ODD.  The accumulators start at -2^31 (GemmArgs.acc0): read as unsigned the register holds u = x + 2^31 for ANY int32 sum x, and
    // s = sum_j byte_j(u) (256^j mod p) + ((-2^31) mod p) == x (mod p), 0 <= s < 2^18: v_dot4_u32_u8; one fp32 quotient RN(s RN(1/p)) and the 24-bit
    // multiply-add s - q p give the canonical residue (mod_small_sym_u, oz2_device.hpp: CPU model over the whole int32 range in
    // tests/test_residue_math.py).  Since round 4 on two accumulators at a time, three instructions per accumulator instead of four: the dot product's constant carries
    // the bit pattern of 2^23 (dotc | 0x4B000000; s < 2^18), so its result READ AS A FLOAT is 2^23 + s: no v_cvt_f32_u32.  s as a float and the
    // quotient q = RN(s RN(1/p)) + 2^23 (the very fma of mod_small_sym_u: same operands, same rounding) are packed-FP32 instructions on the pair
    // (v_pk_add_f32, v_pk_fma_f32: two lanes' worth per issue); v_mad_i32_i24 finishes on the low 24 bits of q's pattern (= q) with the dot product's
    // pattern as addend -- the 0x4B000000 above bit 23 does not reach the low byte, which is all the epilogue keeps.
    [[maybe_unused]] const unsigned dotc_f = dotc + 0x4B000000u;
    [[maybe_unused]] auto red_odd_pair = [&](int x0, int x1, int& r0, int& r1) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const unsigned u0 = __builtin_amdgcn_udot4((unsigned)x0, dotw, dotc_f, false), u1 = __builtin_amdgcn_udot4((unsigned)x1, dotw, dotc_f, false);
        const v2f sm = {__uint_as_float(u0), __uint_as_float(u1)};
        const v2f sf = sm - v2f{8388608.0f, 8388608.0f};
        const v2f qm = __builtin_elementwise_fma(sf, v2f{invp, invp}, v2f{8388608.0f, 8388608.0f});
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r0) : "v"(__float_as_int(qm[0])), "s"(-p), "v"(u0));
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r1) : "v"(__float_as_int(qm[1])), "s"(-p), "v"(u1));
    };
    // RED_MAGIC, K <= 256 (round 6): the accumulators start at 0x4B400000 (GemmArgs.acc0), the bit pattern of the float 1.5 * 2^23; an int32 sum |x| <= 256 * 128^2 = 2^22
    // added to it as an INTEGER leaves the pattern of the float 1.5 * 2^23 + x (ulp 1 on [2^23, 2^24], both ends representable).  So the conversion of RED_ODD_SMALL
    // becomes a packed subtraction on two accumulators, its fma a packed fma with the very same operands (same quotient, bit for bit), and v_mad_i32_i24 finishes on
    // the biased register: the bias has no bit below 2^22, the low byte -- all the epilogue keeps -- is that of x - q p.  Two instructions per accumulator instead of
    // three (~20 % of the epilogue's vector-ALU work; CPU model: tests/test_residue_math.py::test_magic_bias_reduction).
    [[maybe_unused]] auto red_magic_pair = [&](int x0, int x1, int& r0, int& r1) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f xm = {__int_as_float(x0), __int_as_float(x1)};
        const v2f xf = xm - v2f{12582912.0f, 12582912.0f};
        const v2f qm = __builtin_elementwise_fma(xf, v2f{invp, invp}, v2f{12582912.0f, 12582912.0f});
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r0) : "v"(__float_as_int(qm[0])), "s"(-p), "v"(x0));
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r1) : "v"(__float_as_int(qm[1])), "s"(-p), "v"(x1));
    };
    [[maybe_unused]] auto red_small = [&](int x) { return mod_small_sym_odd(x, p, invp); };  // |x| < 2^16 (complex combine); p = 256: the low byte survives
    // After the 4 x 4 dword transpose below lane (q, c16) owns the 16 consecutive rows i0 + 64 tg + 16 q .. + 15 of column
    // j0 + 16 tj + c16: one 64-bit element offset per lane for the whole block, the (tg, tj) sub-blocks add 64 tg and 16 tj * ldo --
    // no per-store multiplies (v_mul_lo_u32 / v_mad_u64_u32 are quarter rate)
    const size_t e00 = (size_t)(j0 + c16) * args.ldo + i0 + q * 16;
    const size_t po = pl.boff + (size_t)pl.tt * args.strideO, pr = pl.boff + (size_t)pl.tt * args.strideR;  // wave-uniform: scalar multiplies
    const size_t ejs = (size_t)16 * args.ldo;
    // residues of the 64 x 16 sub-block (tj, tg) of the wave tile: z[0..3] = this lane's 16 consecutive rows (first row i0 + 64 tg + 16 q) of column j0 + 16 tj + c16
    auto reduce_block = [&](int tj, int tg, unsigned (&z)[4]) {
        unsigned d[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            int r[4];
            if constexpr (RED == RED_ODD) {
                red_odd_pair(acc[tg * 4 + ti][tj][0], acc[tg * 4 + ti][tj][1], r[0], r[1]);
                red_odd_pair(acc[tg * 4 + ti][tj][2], acc[tg * 4 + ti][tj][3], r[2], r[3]);
            } else if constexpr (RED == RED_MAGIC) {
                red_magic_pair(acc[tg * 4 + ti][tj][0], acc[tg * 4 + ti][tj][1], r[0], r[1]);
                red_magic_pair(acc[tg * 4 + ti][tj][2], acc[tg * 4 + ti][tj][3], r[2], r[3]);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) r[b] = red(acc[tg * 4 + ti][tj][b]);
            }
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
        z[0] = w01[0], z[1] = w01[1], z[2] = w23[0], z[3] = w23[1];
    };
    if constexpr (MODLIKE) {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const int col = j0 + tj * 16 + c16;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                unsigned z[4];
                reduce_block(tj, tg, z);
                hook(2 * tj + tg, 0);
                const size_t e = e00 + tj * ejs + tg * 64;  // first of 16 consecutive rows
                if (pl.tt < args.nt_planes) store16_cols<true>(args.out + po + e, v4u{z[0], z[1], z[2], z[3]}, col, args.n);  // wave-uniform
                else store16_cols<false>(args.out + po + e, v4u{z[0], z[1], z[2], z[3]}, col, args.n);
                hook(2 * tj + tg, 1);
            }
        }
    } else {
        // Complex combine in TWO passes (round 4).  Vector-memory operations of a wave complete in issue order (one vmcnt for loads and stores on
        // gfx9), and a reload of a spilled register is a vector-memory load too: the sub-block-by-sub-block form of rounds 1-3 (load X / Y, combine,
        // store; its store addresses reloaded from scratch) waited for a full memory round trip per 8 rows -- sixteen serialised latencies per tile,
        // the 40 % by which the combine launch ran longer than a plain residue launch (profiles/archive/r04_pmc_cplx_combine.txt: waves parked at s_waitcnt).
        // Pass 1 reduces the accumulators sub-block by sub-block to packed residues and issues the two 16-byte X / Y loads of a sub-block as soon as
        // its accumulators are dead (the loads land in those registers): sixteen loads in flight behind the residue arithmetic, none behind a store.
        // One wait, then pass 2 combines and stores (the stores are invisible to the compiler's vmcnt bookkeeping -- store16_cols -- so it must not be
        // left to wait for the loads one by one: each of those waits would also drain the stores issued before it).
        unsigned z[8][4];
        v4u X[8], Y[8];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            // columns beyond n read the last existing column instead (their results are never stored): no branch
            const int colc = min(j0 + tj * 16 + c16, args.n - 1);
            const size_t ec = (size_t)colc * args.ldo + i0 + q * 16;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                const int sb = 2 * tj + tg;
                reduce_block(tj, tg, z[sb]);
                X[sb] = *(const v4u*)(args.rx + pr + ec + tg * 64);
                Y[sb] = *(const v4u*)(args.ry + pr + ec + tg * 64);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        // (the store offsets are recomputed from an opaque copy of e00: kept live from pass 1 they are sixteen more registers at its peak)
        size_t e00s = e00;
        asm volatile("" : "+v"(e00s));
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            const int col = j0 + tj * 16 + c16;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                const int sb = 2 * tj + tg;
                // (Cr, Ci) = (X - Y, Z - X - Y) mod p on PAIRS of rows in packed FP32 (v_pk_add_f32 / v_pk_fma_f32: two values per issue): the differences are
                // integers of magnitude <= 381, the quotient is RN(d RN(1/p)) read from an fma against 1.5 * 2^23 (no tie is reachable: p is odd, or 256
                // where every representative has the same byte), d - q p is exact, and adding 1.5 * 2^23 once more leaves the residue's two's-complement
                // byte in the low byte of the pattern.  Ten instructions per element where the scalar form (two fp32 quotient steps with conversions both
                // ways and a quarter-rate v_mul_lo_u32 each) took eighteen: the combine pass is VALU-bound.
                typedef float v2f __attribute__((ext_vector_type(2)));
                const v2f invp2 = {invp, invp}, mg = {12582912.0f, 12582912.0f}, np2 = {-(float)p, -(float)p};
                unsigned o[8];
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {  // four rows per dword of X, Y, Z -> two dwords of (Cr, Ci) byte pairs
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        auto sx = [](unsigned v, int b) { return (float)(int)(int8_t)(v >> (8 * b)); };
                        const v2f xf = {sx(X[sb][w4], 2 * pr), sx(X[sb][w4], 2 * pr + 1)}, yf = {sx(Y[sb][w4], 2 * pr), sx(Y[sb][w4], 2 * pr + 1)},
                                  zf = {sx(z[sb][w4], 2 * pr), sx(z[sb][w4], 2 * pr + 1)};
                        const v2f d1 = xf - yf, d2 = zf - (xf + yf);
                        const v2f q1 = __builtin_elementwise_fma(d1, invp2, mg) - mg, q2 = __builtin_elementwise_fma(d2, invp2, mg) - mg;
                        const v2f r1 = __builtin_elementwise_fma(q1, np2, d1) + mg, r2 = __builtin_elementwise_fma(q2, np2, d2) + mg;
                        o[2 * w4 + pr] = __builtin_amdgcn_perm(__float_as_uint(r2[0]), __float_as_uint(r1[0]), 0x0c0c0400u) |
                                         __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r1[1]), 0x04000c0cu);
                    }
                }
                hook(sb, 0);
                const size_t e = e00s + tj * ejs + tg * 64;
                v4u* const dst = (v4u*)(args.out + po + 2 * e);
                if (pl.tt < args.nt_planes) {  // wave-uniform
                    store16_cols<true>(dst, v4u{o[0], o[1], o[2], o[3]}, col, args.n);
                    store16_cols<true>(dst + 1, v4u{o[4], o[5], o[6], o[7]}, col, args.n);
                } else {
                    store16_cols<false>(dst, v4u{o[0], o[1], o[2], o[3]}, col, args.n);
                    store16_cols<false>(dst + 1, v4u{o[4], o[5], o[6], o[7]}, col, args.n);
                }
                hook(sb, 1);
            }
        }
    }
}

template <int EPI, typename Hook = NoHook, int SMALLK = -1>
__device__ __forceinline__ void i8_epilogue(const v4i (&acc)[8][4], const GemmArgs& args, PlaneRef pl, PlaneConsts pc, int i0, int j0, int lane, Hook hook = {}) {
    const int c16 = lane & 15;
    const int q = lane >> 4;

    if constexpr (EPI == EPI_MOD256) {
        i8_epilogue_mod<EPI, RED_MAGIC, Hook>(acc, args, pl, pc, i0, j0, lane, hook);
    } else if constexpr (EPI == EPI_MOD || EPI == EPI_CPLX) {
        // ONE reduction form per kernel instantiation (SMALLK: the launch's accumulators start at 0, K <= 512).  p = 256 takes the odd-modulus forms too:
        // the only thing the epilogue keeps of a residue is its low byte, and for p = 256 every quotient leaves the low byte of the accumulator
        // in place (dotw = 1, dotc = 0).  Until round 4 the epilogue was a run-time chain of four forms (256 / small / odd / generic); the
        // structurizer lays such a chain out as a straight line of predicated blocks, which keeps the accumulators live through the whole
        // epilogue of every form but the last -- no accumulator register could be reused inside an epilogue.
        // SMALLK < 0: the form is chosen at run time from GemmArgs.acc0 (laboratory kernels).
        if constexpr (SMALLK > 0) i8_epilogue_mod<EPI, RED_ODD_SMALL, Hook>(acc, args, pl, pc, i0, j0, lane, hook);
        else if constexpr (SMALLK == 0) i8_epilogue_mod<EPI, RED_ODD, Hook>(acc, args, pl, pc, i0, j0, lane, hook);
        else if (args.acc0 == 0) i8_epilogue_mod<EPI, RED_ODD_SMALL, Hook>(acc, args, pl, pc, i0, j0, lane, hook);
        else i8_epilogue_mod<EPI, RED_ODD, Hook>(acc, args, pl, pc, i0, j0, lane, hook);
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


template <int EPI, typename Hook = NoHook, int SMALLK = -1>
__device__ __forceinline__ void i8_epilogue(const v4i (&acc)[8][4], const GemmArgs& args, PlaneRef pl, int i0, int j0, int lane, Hook hook = {}) {
    i8_epilogue<EPI, Hook, SMALLK>(acc, args, pl, EPI == EPI_MAX ? PlaneConsts{} : plane_consts(args, pl), i0, j0, lane, hook);
}

#ifdef OZ2_LAB_SHORTK  // laboratory kernel (tools/experiments/shortk/oz2_gemm_i8_shortk.hip); `a` complete as launch<EPI> of oz2_gemm_i8.hip leaves it
hipError_t launch_gemm_i8_shortk(hipStream_t stream, const GemmArgs& a, int epi);
#endif

}  // namespace oz2
