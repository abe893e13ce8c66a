// CRT accumulation + inverse scaling + axpby (HBM-bound: N int8 planes in, one FP matrix out).
//
// Replaces GEMMul8/src/inverse_scaling_real.hpp:8-278 and inverse_scaling_complex.hpp:8-326:
//   S  = sum_{t=0}^{N-1} fma(qPi_t, double(C_mid[t]), S)            fixed order t = 0..N-1
//   TP = double  (float outputs, or N <= 6 [INT8] / 5 [FP8]):  R = fma(Pneg, rint(invP*S), S)
//   TP = double2 (otherwise): hi sum error-free, lo sum rounded;
//        R = fma(Pneg_lo, q, fma(Pneg_hi, q, Sh) + Sl),  q = rint(invP*Sh)
//   AB = scalbn((T)R, sftA[i] + sftB[j])   (shift arrays hold the NEGATED exponents)
//   C  = AB | C+AB | -AB | C-AB  for host scalars alpha=+-1, beta in {0,1};
//        otherwise fma(beta, C, alpha*AB) (complex: nested fma order of template_math.hpp:61-75);
//   device-pointer scalars always take the general form (inverse_scaling_real.hpp:211-216).
//   beta == 0 in the general form: C is NOT read and enters the fma as +0 (BLAS semantics -- the hook serves callers such as
//   torch.baddbmm(torch.empty(..), beta=0) whose C holds NaN garbage; the reference evaluates fma(0, C, alpha*AB), the same value for
//   every finite C except the sign of an exact zero when C < 0).
// Each thread handles ROWS consecutive rows of one column (LB = 8 bytes of every residue plane: 8 / 4 / 4 / 2 rows): one vector load per residue plane, all
// N loads issued before the first use (the plane loop is fully unrolled over the 20-moduli maximum with a uniform guard); the results
// leave through a per-wave LDS transposition so that every store instruction writes 1 KiB of contiguous memory (crt_wave_store).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "oz2_crt_common.hpp"
#include "oz2_kernels.h"
#include "oz2_knobs.hpp"

namespace oz2 {

// U = float|double ; CPLX ; MID = int8_t|int16_t ; LB = bytes of every residue plane a thread loads (8 or 16)
#ifndef OZ2_CRT_BLOCK
#define OZ2_CRT_BLOCK 256  // threads per block of crt_kernel: 128 / 256 / 512 / 1024 -> 455 / 377 / 414 / 380 us at config 2 (tools/crt_ab.py)
#endif
#ifndef OZ2_CRT_NT
#define OZ2_CRT_NT 1      // non-temporal residue loads (plain loads: 393 us)
#endif
#ifndef OZ2_CRT_LB
#define OZ2_CRT_LB 8      // bytes per residue-plane load and thread (tools/crt_ab.py measures 8 against 16)
#endif
#ifndef OZ2_CRT_LDS_STORE
#define OZ2_CRT_LDS_STORE 1  // 1: the wave's results go through LDS so that every store instruction writes 1 KiB of contiguous memory
#endif

// A thread's NV results are NV * sizeof(U) = J * 16 contiguous bytes of C, a wave's 64 threads own 64 * J * 16 contiguous bytes (when
// they sit in one column).  Stored directly, store instruction j of the J would scatter 16-byte pieces at a J * 16-byte lane stride
// (every instruction touches all of the wave's cache lines, a quarter or an eighth of each).  crt_wave_store sends the pieces through
// the wave's private LDS slice and reads them back transposed: instruction j then writes bytes [1024 j, 1024 (j + 1)) -- lane-linear,
// whole lines.  Chunk (lane l, piece j) lives at slot l * J + (j ^ f(l)), f(l) = (l / (16 / J)) % J: the 8 lanes of a
// ds_write_b128 group hit 8 different 16-byte bank slots, and the read side is a permutation inside 256-byte blocks (conflict-free).
template <int J> __device__ __forceinline__ unsigned crt_slot(unsigned l, unsigned j) {
    if constexpr (J >= 2 && J <= 16) return l * J + (j ^ ((l / (16 / J)) % J));
    else return l * J + j;
}
template <int J> __device__ __forceinline__ void crt_wave_store(char* lds_wave, const void* vals, char* gdst, unsigned lane) {
    typedef unsigned V4 __attribute__((ext_vector_type(4)));
    const V4* v = (const V4*)vals;
    V4* ls = (V4*)lds_wave;
#pragma unroll
    for (unsigned j = 0; j < (unsigned)J; ++j) ls[crt_slot<J>(lane, j)] = v[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (unsigned j = 0; j < (unsigned)J; ++j) {
        const unsigned c = j * 64u + lane;  // logical 16-byte chunk of the wave's output
        const V4 x = ls[crt_slot<J>(c / J, c % J)];
        __builtin_nontemporal_store(x, (V4*)(gdst + (size_t)c * 16));  // streaming: C is written once and not re-read by this kernel
    }
}

#ifndef OZ2_CRT_UNITS
#define OZ2_CRT_UNITS 1  // row groups per thread; > 1: the residue vectors of group u + 1 are requested before group u is accumulated
#endif

template <typename MID, int LB, int COMPS> struct CrtVec {
    static constexpr int NV = LB / (int)sizeof(MID);
    struct alignas(LB) Vec {
        MID v[NV];
    };
    // 64-bit lanes: with 32-bit vector elements the byte extraction keeps every plane's dwords live (112 -> 240 VGPRs)
    typedef unsigned long long RawV2 __attribute__((ext_vector_type(2)));
    using RawV = typename std::conditional<LB == 8, unsigned long long, RawV2>::type;
};

// position of thread-unit `gid`: column, first row (idle lanes of the last wave shadow the last valid unit; they never store)
struct CrtPos {
    size_t col, i0;
    bool active;
};
template <int ROWS> __device__ __forceinline__ CrtPos crt_pos(size_t gid, size_t total, unsigned row_groups) {
    CrtPos p;
    p.active = gid < total;
    const size_t g = p.active ? gid : total - 1;
    p.col = g / row_groups;
    p.i0 = (g - p.col * row_groups) * ROWS;
    return p;
}

// every plane's ROWS values of one unit (planes are padded to 256 rows: the vector load never leaves the plane)
template <typename MID, int LB, int COMPS, int NMAX>
__device__ __forceinline__ void crt_load(const CrtArgs& a, const CrtPos& p, size_t gid, typename CrtVec<MID, LB, COMPS>::Vec (&c)[NMAX]) {
    using RawV = typename CrtVec<MID, LB, COMPS>::RawV;
    const size_t zw = blockIdx.z * a.bw;  // batched launch: item blockIdx.z
    const MID* base = (const MID*)((const char*)a.Cmid + zw) + (p.col * a.ld_mid + p.i0) * COMPS;
#pragma unroll
    for (unsigned t = 0; t < (unsigned)NMAX; ++t)
        if (t < a.N) {
            const RawV* src_ = (const RawV*)(base + (size_t)t * a.plane_stride * COMPS);
#if defined(OZ2_CRT_ABL) && (OZ2_CRT_ABL & 2)  // timing probe: no residue loads
            RawV raw;
            {
                const unsigned long long fake = (unsigned long long)gid * 0x9E3779B97F4A7C15ull + t;
                __builtin_memcpy(&raw, &fake, 8);
                if (LB == 16) __builtin_memcpy((char*)&raw + 8, &fake, 8);
            }
            (void)src_;
#else
            const RawV raw = OZ2_CRT_NT ? __builtin_nontemporal_load(src_) : *src_;
#endif
            __builtin_memcpy(&c[t], &raw, LB);
        }
}

template <typename U, bool CPLX, typename MID, int LB, bool LDS_STORE, int NMAX>
__device__ __forceinline__ void crt_unit(const CrtArgs& a, const CrtPos& p, size_t gid, size_t total, unsigned row_groups,
                                         const typename CrtVec<MID, LB, (CPLX ? 2 : 1)>::Vec (&c)[NMAX], char* stage) {
    constexpr int COMPS = CPLX ? 2 : 1;
    constexpr int ROWS = LB / (COMPS * (int)sizeof(MID));
    constexpr int NV = ROWS * COMPS;
    constexpr int OUTB = NV * (int)sizeof(U);
    constexpr int J = OUTB / 16;
    const size_t col = p.col, i0 = p.i0;
    const bool active = p.active;
    const size_t zw = blockIdx.z * a.bw;

    U al[2] = {(U)a.alpha[0], (U)a.alpha[1]}, be[2] = {(U)a.beta[0], (U)a.beta[1]};
    int mode = a.mode;
    if (mode == 5) {
        al[0] = ((const U*)a.alpha_dev)[0];
        be[0] = ((const U*)a.beta_dev)[0];
        if (CPLX) {
            al[1] = ((const U*)a.alpha_dev)[1];
            be[1] = ((const U*)a.beta_dev)[1];
        }
        mode = 0;
    }
    const int16_t* sftA_z = (const int16_t*)((const char*)a.sftA + zw);
    const int sB = (int)((const int16_t*)((const char*)a.sftB + zw))[col];
    U* Cc = (U*)((char*)a.C + blockIdx.z * a.bc) + (col * a.ldc) * COMPS;
    const bool full = i0 + ROWS <= a.m;  // all ROWS rows exist: the old and new C values move as one vector per thread
    const bool beta0 = be[0] == (U)0 && (!CPLX || be[1] == (U)0);
    const bool reads_c = (mode == 0 && !beta0) || mode == 2 || mode == 4;
    U outv[NV];

    // accumulate at most 8 values at a time (LB = 16: two passes over the loaded vectors): 2 x 8 FP64 accumulators
#ifndef OZ2_CRT_PASS
#define OZ2_CRT_PASS 8
#endif
    constexpr int PASS = NV > OZ2_CRT_PASS ? OZ2_CRT_PASS : NV;
#pragma unroll
    for (int p0 = 0; p0 < NV; p0 += PASS) {
        double Sh[PASS], Sl[PASS];
#pragma unroll
        for (int e = 0; e < PASS; ++e) Sh[e] = 0.0, Sl[e] = 0.0;
#pragma unroll
        for (unsigned t = 0; t < (unsigned)NMAX; ++t) {
            if (t < a.N) {
                if (a.use_dd) {
                    const double qh = a.qh[t], ql = a.ql[t];
#pragma unroll
                    for (int e = 0; e < PASS; ++e) {
                        const double cd = (double)c[t].v[p0 + e];
                        Sh[e] = fma(qh, cd, Sh[e]);
                        Sl[e] = fma(ql, cd, Sl[e]);
                    }
                } else {
                    const double q1 = a.q1[t];
#pragma unroll
                    for (int e = 0; e < PASS; ++e) Sh[e] = fma(q1, (double)c[t].v[p0 + e], Sh[e]);
                }
            }
        }
        // the old C values of this pass (after the accumulation: they are not live beside the residue vectors)
        U oldc[NV];
#pragma unroll
        for (int e = 0; e < PASS; ++e) oldc[p0 + e] = (U)0;
        if (reads_c && active) {
            if (full) {
                __builtin_memcpy(oldc + p0, Cc + i0 * COMPS + p0, PASS * sizeof(U));
            } else {
#pragma unroll
                for (int e = p0; e < p0 + PASS; ++e)
                    if (i0 + e / COMPS < a.m) oldc[e] = Cc[i0 * COMPS + e];
            }
        }
#pragma unroll
        for (int r = 0; r < PASS / COMPS; ++r) {
            const int e = p0 / COMPS + r;  // row inside the thread's group
            const size_t row = i0 + e;
            const int sft = (row < a.m ? (int)sftA_z[row] : 0) + sB;
            if constexpr (!CPLX) {
                const U AB = scalb<U>((U)crt_reduce(a, Sh[r], Sl[r]), sft);
                switch (mode) {
                case 1: outv[e] = AB; break;
                case 2: outv[e] = oldc[e] + AB; break;
                case 3: outv[e] = -AB; break;
                case 4: outv[e] = oldc[e] - AB; break;
                default: outv[e] = fmaU<U>(be[0], oldc[e], al[0] * AB); break;
                }
            } else {
                const U x = scalb<U>((U)crt_reduce(a, Sh[2 * r], Sl[2 * r]), sft);
                const U y = scalb<U>((U)crt_reduce(a, Sh[2 * r + 1], Sl[2 * r + 1]), sft);
                const U cx = oldc[2 * e], cy = oldc[2 * e + 1];
                switch (mode) {
                case 1: outv[2 * e] = x, outv[2 * e + 1] = y; break;
                case 2: outv[2 * e] = cx + x, outv[2 * e + 1] = cy + y; break;
                case 3: outv[2 * e] = -x, outv[2 * e + 1] = -y; break;
                case 4: outv[2 * e] = cx - x, outv[2 * e + 1] = cy - y; break;
                default:
                    outv[2 * e] = fmaU<U>(-be[1], cy, fmaU<U>(be[0], cx, fmaU<U>(-al[1], y, al[0] * x)));
                    outv[2 * e + 1] = fmaU<U>(be[1], cx, fmaU<U>(be[0], cy, fmaU<U>(al[1], x, al[0] * y)));
                    break;
                }
            }
        }
    }

    if constexpr (LDS_STORE) {
        // wave-uniform test: all 64 threads exist, sit in one column with all their rows inside m, and the wave's first byte of C is
        // 16-byte aligned -> the wave's output is one contiguous, aligned block of 64 * OUTB bytes
        const unsigned lane = threadIdx.x & 63u;
        const size_t first = gid - lane, last = first + 63;
        const size_t colf = first / row_groups;
        const size_t i0f = (first - colf * row_groups) * ROWS;
        char* wdst = (char*)((U*)((char*)a.C + blockIdx.z * a.bc) + (colf * a.ldc + i0f) * COMPS);
        const bool wave_fast = last < total && last / row_groups == colf && i0f + (size_t)64 * ROWS <= a.m && ((uintptr_t)wdst & 15u) == 0;
#if defined(OZ2_CRT_ABL) && (OZ2_CRT_ABL & 1)  // timing probe: no stores (the condition is never true)
        if (wave_fast && outv[0] != (U)123.456) return;
#endif
        if (wave_fast) {
            crt_wave_store<J>(stage + (size_t)(threadIdx.x >> 6) * 64 * OUTB, outv, wdst, lane);
            return;
        }
    }
    if (!active) return;
    if (full) {
        typedef U VecU __attribute__((ext_vector_type(NV)));
        VecU ov;
#pragma unroll
        for (int e = 0; e < NV; ++e) ov[e] = outv[e];
        if ((reinterpret_cast<uintptr_t>(Cc + i0 * COMPS) & (sizeof(VecU) - 1)) == 0) __builtin_nontemporal_store(ov, (VecU*)(Cc + i0 * COMPS));
        else __builtin_memcpy(Cc + i0 * COMPS, outv, sizeof(outv));
    } else {
#pragma unroll
        for (int e = 0; e < NV; ++e)
            if (i0 + e / COMPS < a.m) Cc[i0 * COMPS + e] = outv[e];
    }
}

// UNITS row groups per thread (consecutive 256-thread slabs of one workgroup): the loads of unit u + 1 are issued before unit u is
// accumulated, so every wave keeps residue vectors in flight while its FP64 pipe is busy (the kernel's FP64 work -- byte extraction,
// conversion and two FMAs per residue -- is ~2/3 of its HBM time: without the overlap they add up instead of hiding each other)
#ifndef OZ2_CRT_WAVES
#define OZ2_CRT_WAVES 4  // waves per SIMD the register allocation aims at (experiment switch)
#endif
template <typename U, bool CPLX, typename MID, int LB, int UNITS, int NMAX>
__global__ void __launch_bounds__(OZ2_CRT_BLOCK) __attribute__((amdgpu_waves_per_eu(OZ2_CRT_WAVES, 8))) crt_kernel(const CrtArgs a) {
    constexpr int COMPS = CPLX ? 2 : 1;
    constexpr int ROWS = LB / (COMPS * (int)sizeof(MID));  // LB = 8: 8 / 4 / 4 / 2 rows; wider vectors cost registers
    constexpr int OUTB = ROWS * COMPS * (int)sizeof(U);     // bytes of C per thread
    static_assert(OUTB % 16 == 0, "a thread's results are whole 16-byte pieces");
    constexpr bool LDS_STORE = OZ2_CRT_LDS_STORE && OUTB > 16;
    __shared__ __attribute__((aligned(16))) char stage[LDS_STORE ? OZ2_CRT_BLOCK * OUTB : 16];
    using Vec = typename CrtVec<MID, LB, COMPS>::Vec;

    const unsigned row_groups = (unsigned)((a.m + ROWS - 1) / ROWS);
    const size_t total = (size_t)row_groups * a.n;
    const size_t gid0 = (size_t)blockIdx.x * (OZ2_CRT_BLOCK * UNITS) + threadIdx.x;
    if (!LDS_STORE && UNITS == 1 && gid0 >= total) return;
    Vec cb[UNITS > 1 ? 2 : 1][NMAX];  // NMAX = 14 or 20: registers are reserved for every plane the instantiation may load
    CrtPos pos[2];
    pos[0] = crt_pos<ROWS>(gid0, total, row_groups);
    crt_load<MID, LB, COMPS, NMAX>(a, pos[0], gid0, cb[0]);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        const size_t gid = gid0 + (size_t)u * OZ2_CRT_BLOCK;
        if (u + 1 < UNITS) {
            pos[(u + 1) & 1] = crt_pos<ROWS>(gid + OZ2_CRT_BLOCK, total, row_groups);
            crt_load<MID, LB, COMPS, NMAX>(a, pos[(u + 1) & 1], gid + OZ2_CRT_BLOCK, cb[(u + 1) & 1]);
        }
        crt_unit<U, CPLX, MID, LB, LDS_STORE, NMAX>(a, pos[u & 1], gid, total, row_groups, cb[u & 1], stage);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA form of the CRT kernel for INT8 residues (round 3).  One wave per workgroup; a unit = 1024 consecutive bytes of one column
// of every residue plane (1024 rows, or 512 complex elements).  The N slices go global -> LDS with `global_load_lds_dwordx4` (1 KiB per
// instruction, no VGPRs), the lanes then read single bytes with `ds_read_i8` -- the SIGN-EXTENDING byte load replaces the v_bfe_i32 of
// the register form (one of four VALU operations per residue), the residue vectors no longer occupy 2 x N registers, and up to eleven
// workgroups per CU (N KiB of LDS each) keep HBM requests in flight while others accumulate.  Lane l reads bytes 128 j + 2 l + h
// (j = 0..7, h = 0, 1): two consecutive rows (or Re / Im of one complex element), so its results for one j are 16 contiguous bytes of C
// (8 for float) and a store instruction writes 1 KiB of contiguous memory without any transposition.  Accumulation order, reduction and
// axpby forms are those of crt_unit: the results are bit-identical (tests/test_gpu_parity.py runs both kernels).
// Eligibility (launch_crt): int8 residues, (m * COMPS) % 1024 == 0, 16-byte aligned plane slices and C columns.
#ifndef OZ2_CRT_DMA
#define OZ2_CRT_DMA 1
#endif

#ifndef OZ2_CRT_DMA_WAVES
#define OZ2_CRT_DMA_WAVES 2  // waves per workgroup sharing one unit (1 / 2 / 4: 289 / 266 / 262 us real x 14, 728 / 668 / 697 us complex x 20): each wave fetches half the planes and accumulates half the j (twice
                             // the waves per CU for the same LDS)
#endif
template <typename U, bool CPLX>
__global__ void __launch_bounds__(64 * OZ2_CRT_DMA_WAVES) crt_dma_kernel(const CrtArgs a, unsigned units_per_col) {
    extern __shared__ __attribute__((aligned(16))) char dma_lds[];
    constexpr int COMPS = CPLX ? 2 : 1;
    constexpr int WPB = OZ2_CRT_DMA_WAVES;
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const size_t unit = blockIdx.x;
    const size_t col = unit / units_per_col;
    const size_t ub = (unit - col * units_per_col) * 1024;  // first byte of the unit inside the column of a plane
    const size_t zw = blockIdx.z * a.bw;                  // batched launch: item blockIdx.z
    const char* src = (const char*)a.Cmid + zw + col * a.ld_mid * COMPS + ub + lane * 16;
    const unsigned N = a.N;
    for (unsigned t = wv; t < N; t += WPB)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)t * a.plane_stride * COMPS),
                                         (__attribute__((address_space(3))) void*)(dma_lds + t * 1024), 16, 0, 0);

    U al[2] = {(U)a.alpha[0], (U)a.alpha[1]}, be[2] = {(U)a.beta[0], (U)a.beta[1]};
    int mode = a.mode;
    if (mode == 5) {
        al[0] = ((const U*)a.alpha_dev)[0];
        be[0] = ((const U*)a.beta_dev)[0];
        if (CPLX) {
            al[1] = ((const U*)a.alpha_dev)[1];
            be[1] = ((const U*)a.beta_dev)[1];
        }
        mode = 0;
    }
    const bool beta0 = be[0] == (U)0 && (!CPLX || be[1] == (U)0);
    const bool reads_c = (mode == 0 && !beta0) || mode == 2 || mode == 4;
    const int16_t* sftA_z = (const int16_t*)((const char*)a.sftA + zw);
    const int sB = (int)((const int16_t*)((const char*)a.sftB + zw))[col];
    const size_t e0 = ub / COMPS;  // first element (row) of the unit
    U* Cc = (U*)((char*)a.C + blockIdx.z * a.bc) + (col * a.ldc + e0) * COMPS;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slices this wave fetched have landed
    if constexpr (WPB > 1) __syncthreads();           // ... and the other waves' too
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // volatile: single sign-extending byte reads (ds_read_i8); merged into wider reads the compiler would need a v_bfe_i32 per byte again
    typedef const volatile __attribute__((address_space(3))) signed char* LdsBytes;
    LdsBytes lb = (LdsBytes)dma_lds + 2 * lane;
#ifndef OZ2_CRT_DMA_J
#define OZ2_CRT_DMA_J 8  // j per pass (2 values each): 8 -> 32 FP64 accumulators, one pass over the weights (2 / 4 / 8: 330 / 304 / 289 us; LDS, not registers, bounds the occupancy)
#endif
    constexpr int PJ = (OZ2_CRT_DMA_J * WPB > 8) ? 8 / WPB : OZ2_CRT_DMA_J, PV = 2 * PJ;
    const int j_begin = (int)wv * (8 / WPB), j_end = j_begin + 8 / WPB;
#pragma unroll 1
    for (int jp = j_begin; jp < j_end; jp += PJ) {
        double Sh[PV], Sl[PV];
#pragma unroll
        for (int e = 0; e < PV; ++e) Sh[e] = 0.0, Sl[e] = 0.0;
        // Rolled loop over the planes, weights from scalar loads.  Measured and not adopted (tools/hbm_ab.py, 14 real / 20 complex planes): fully
        // unrolled with the next plane's byte reads issued ahead (SGPR weights spill to VGPR lanes; weights staged in LDS: 319 / 902 us),
        // rolled with LDS weights (305 / 805 us), with or without the read-ahead -- this plain form: 295 / 765 us.
        if (a.use_dd) {
            for (unsigned t = 0; t < N; ++t) {
                const double qh = a.qh[t], ql = a.ql[t];
                LdsBytes p = lb + t * 1024 + jp * 128;
                int cb[PV];
#pragma unroll
                for (int e = 0; e < PV; ++e) cb[e] = (int)p[(e >> 1) * 128 + (e & 1)];
#pragma unroll
                for (int e = 0; e < PV; ++e) {
                    const double cd = (double)cb[e];
                    Sh[e] = fma(qh, cd, Sh[e]);
                    Sl[e] = fma(ql, cd, Sl[e]);
                }
            }
        } else {
            for (unsigned t = 0; t < N; ++t) {
                const double q1 = a.q1[t];
                LdsBytes p = lb + t * 1024 + jp * 128;
                int cb[PV];
#pragma unroll
                for (int e = 0; e < PV; ++e) cb[e] = (int)p[(e >> 1) * 128 + (e & 1)];
#pragma unroll
                for (int e = 0; e < PV; ++e) Sh[e] = fma(q1, (double)cb[e], Sh[e]);
            }
        }
#pragma unroll
        for (int jj = 0; jj < PJ; ++jj) {
            const int j = jp + jj;
            U* dst = Cc + (size_t)(CPLX ? 64 * j + lane : 128 * j + 2 * lane) * COMPS;  // two values: 2 rows, or (Re, Im)
            U oldc[2] = {(U)0, (U)0};
            if (reads_c) __builtin_memcpy(oldc, dst, 2 * sizeof(U));
            int s0, s1;  // this lane's shifts: real -> rows 128 j + 2 l, + 1; complex -> row 64 j + l
            if constexpr (CPLX) {
                s0 = s1 = (int)sftA_z[e0 + 64 * j + lane];
            } else {
                const unsigned w = *(const unsigned*)(sftA_z + e0 + 128 * j + 2 * lane);
                s0 = (int)(int16_t)(w & 0xFFFFu), s1 = (int)(int16_t)(w >> 16);
            }
            const U x = scalb<U>((U)crt_reduce(a, Sh[2 * jj], Sl[2 * jj]), s0 + sB);
            const U y = scalb<U>((U)crt_reduce(a, Sh[2 * jj + 1], Sl[2 * jj + 1]), s1 + sB);
            U o[2];
            if constexpr (!CPLX) {
                switch (mode) {
                case 1: o[0] = x, o[1] = y; break;
                case 2: o[0] = oldc[0] + x, o[1] = oldc[1] + y; break;
                case 3: o[0] = -x, o[1] = -y; break;
                case 4: o[0] = oldc[0] - x, o[1] = oldc[1] - y; break;
                default: o[0] = fmaU<U>(be[0], oldc[0], al[0] * x), o[1] = fmaU<U>(be[0], oldc[1], al[0] * y); break;
                }
            } else {
                const U cx = oldc[0], cy = oldc[1];
                switch (mode) {
                case 1: o[0] = x, o[1] = y; break;
                case 2: o[0] = cx + x, o[1] = cy + y; break;
                case 3: o[0] = -x, o[1] = -y; break;
                case 4: o[0] = cx - x, o[1] = cy - y; break;
                default:
                    o[0] = fmaU<U>(-be[1], cy, fmaU<U>(be[0], cx, fmaU<U>(-al[1], y, al[0] * x)));
                    o[1] = fmaU<U>(be[1], cx, fmaU<U>(be[0], cy, fmaU<U>(al[1], x, al[0] * y)));
                    break;
                }
            }
            typedef U V2 __attribute__((ext_vector_type(2)));
            __builtin_nontemporal_store(V2{o[0], o[1]}, (V2*)dst);  // streaming: C is written once and not re-read by this kernel
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-GPU exchange variant (A) of BASELINE.json's north_star / SURVEY.md 8(e): each rank accumulates the CRT sum over ITS
// moduli only and the ranks add the FP64 partials (RCCL reduce-scatter, sum) before the mod-P reduction.
//   crt_partial: (Sh, Sl)[j][i] = sum over t in [t_begin, t_end) of fma(q_t, double(C_mid[t][j][i]), .) in ascending t, the same
//                two chains as crt_kernel restricted to the rank's moduli; the hi chain is error-free by construction of the
//                tables (every partial sum of hi parts is exact), the lo chain (and the single chain of the TP = double case)
//                is rounded, so its value depends on how the moduli are grouped -- the only place variant (A) can differ from
//                the single-GPU result.  Output planes: out_hi / out_lo, element (i, j) at j * ld_out + i (complex: 2 values).
//   crt_finish : R = crt_reduce(Sh, Sl) on the summed partials, then scalbn + axpby exactly as crt_kernel.
template <bool CPLX, typename MID>
__global__ void __launch_bounds__(256) crt_partial_kernel(const CrtArgs a, unsigned t_begin, unsigned t_end, double* out_hi, double* out_lo,
                                                          size_t ld_out, size_t col_block, size_t block_stride) {
    constexpr int COMPS = CPLX ? 2 : 1;
    constexpr int ROWS = 8 / (COMPS * (int)sizeof(MID));
    constexpr int NV = ROWS * COMPS;
    const unsigned row_groups = (unsigned)((a.m + ROWS - 1) / ROWS);
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)row_groups * a.n) return;
    const size_t col = gid / row_groups;
    const size_t i0 = (gid - col * row_groups) * ROWS;
    struct alignas(sizeof(MID) * NV) Vec {
        MID v[NV];
    };
    const MID* base = (const MID*)a.Cmid + (col * a.ld_mid + i0) * COMPS;
    double Sh[NV], Sl[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) Sh[e] = 0.0, Sl[e] = 0.0;
    for (unsigned t = t_begin; t < t_end; ++t) {
        Vec c;
        const unsigned long long raw = *(const unsigned long long*)(base + (size_t)(t - t_begin) * a.plane_stride * COMPS);
        __builtin_memcpy(&c, &raw, 8);
        if (a.use_dd) {
            const double qh = a.qh[t], ql = a.ql[t];
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const double cd = (double)c.v[e];
                Sh[e] = fma(qh, cd, Sh[e]);
                Sl[e] = fma(ql, cd, Sl[e]);
            }
        } else {
            const double q1 = a.q1[t];
#pragma unroll
            for (int e = 0; e < NV; ++e) Sh[e] = fma(q1, (double)c.v[e], Sh[e]);
        }
    }
    // destination: column blocks of col_block columns, block b at b * block_stride doubles (the reduce-scatter unit of a rank)
    const size_t blk = col / col_block, cin = col - blk * col_block;
    double* oh = out_hi + blk * block_stride + (cin * ld_out + i0) * COMPS;
    double* ol = out_lo + blk * block_stride + (cin * ld_out + i0) * COMPS;
#pragma unroll
    for (int e = 0; e < NV; ++e)
        if (i0 + e / COMPS < a.m) oh[e] = Sh[e], ol[e] = Sl[e];
}

template <typename U, bool CPLX>
__global__ void __launch_bounds__(256) crt_finish_kernel(const CrtArgs a, const double* in_hi, const double* in_lo, size_t ld_in) {
    constexpr int COMPS = CPLX ? 2 : 1;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.m * a.n) return;
    const size_t col = gid / a.m, row = gid - col * a.m;
    U al[2] = {(U)a.alpha[0], (U)a.alpha[1]}, be[2] = {(U)a.beta[0], (U)a.beta[1]};
    int mode = a.mode;
    if (mode == 5) {
        al[0] = ((const U*)a.alpha_dev)[0];
        be[0] = ((const U*)a.beta_dev)[0];
        if (CPLX) al[1] = ((const U*)a.alpha_dev)[1], be[1] = ((const U*)a.beta_dev)[1];
        mode = 0;
    }
    const int sft = (int)a.sftA[row] + (int)a.sftB[col];
    U* Cc = (U*)a.C + (col * a.ldc + row) * COMPS;
    const double* ph = in_hi + (col * ld_in + row) * COMPS;
    const double* pl = in_lo + (col * ld_in + row) * COMPS;
    const bool beta0 = be[0] == (U)0 && (!CPLX || be[1] == (U)0);
    const bool reads_c = (mode == 0 && !beta0) || mode == 2 || mode == 4;
    if constexpr (!CPLX) {
        const U AB = scalb<U>((U)crt_reduce(a, ph[0], pl[0]), sft);
        const U old = reads_c ? Cc[0] : (U)0;
        U o;
        switch (mode) {
        case 1: o = AB; break;
        case 2: o = old + AB; break;
        case 3: o = -AB; break;
        case 4: o = old - AB; break;
        default: o = fmaU<U>(be[0], old, al[0] * AB); break;
        }
        Cc[0] = o;
    } else {
        const U x = scalb<U>((U)crt_reduce(a, ph[0], pl[0]), sft);
        const U y = scalb<U>((U)crt_reduce(a, ph[1], pl[1]), sft);
        const U cx = reads_c ? Cc[0] : (U)0, cy = reads_c ? Cc[1] : (U)0;
        U ox, oy;
        switch (mode) {
        case 1: ox = x, oy = y; break;
        case 2: ox = cx + x, oy = cy + y; break;
        case 3: ox = -x, oy = -y; break;
        case 4: ox = cx - x, oy = cy - y; break;
        default:
            ox = fmaU<U>(-be[1], cy, fmaU<U>(be[0], cx, fmaU<U>(-al[1], y, al[0] * x)));
            oy = fmaU<U>(be[1], cx, fmaU<U>(be[0], cy, fmaU<U>(al[1], x, al[0] * y)));
            break;
        }
        Cc[0] = ox, Cc[1] = oy;
    }
}

void fill_crt_tables(CrtArgs& a, int dtype, int backend, unsigned N) {
    const bool f32 = is_f32(dtype);
    const int pdbl = backend == kINT8 ? 6 : 5;
    a.N = N;
    a.use_dd = !(f32 || (int)N <= pdbl);
    const bool i8 = backend == kINT8;
    a.Phi = (i8 ? GEMMUL8_PNEG_HI_INT8 : GEMMUL8_PNEG_HI_FP8)[N - 2];
    a.Plo = (i8 ? GEMMUL8_PNEG_LO_INT8 : GEMMUL8_PNEG_LO_FP8)[N - 2];
    a.invP = (i8 ? GEMMUL8_INVP_INT8 : GEMMUL8_INVP_FP8)[N - 2];
    for (unsigned t = 0; t < N; ++t) {
        a.q1[t] = (i8 ? GEMMUL8_QPI1_INT8 : GEMMUL8_QPI1_FP8)[N - 2][t];
        a.qh[t] = (i8 ? GEMMUL8_QPI2_HI_INT8 : GEMMUL8_QPI2_HI_FP8)[N - 2][t];
        a.ql[t] = (i8 ? GEMMUL8_QPI2_LO_INT8 : GEMMUL8_QPI2_LO_FP8)[N - 2][t];
    }
}
void fill_crt_scalars(CrtArgs& a, int dtype, const void* alpha, const void* beta, bool scalars_on_device) {
    const bool f32 = is_f32(dtype), cplx = is_complex(dtype);
    if (scalars_on_device) {
        a.mode = 5;
        a.alpha_dev = alpha;
        a.beta_dev = beta;
        return;
    }
    double ar, ai = 0, br, bi = 0;
    if (f32) {
        ar = ((const float*)alpha)[0];
        br = ((const float*)beta)[0];
        if (cplx) ai = ((const float*)alpha)[1], bi = ((const float*)beta)[1];
    } else {
        ar = ((const double*)alpha)[0];
        br = ((const double*)beta)[0];
        if (cplx) ai = ((const double*)alpha)[1], bi = ((const double*)beta)[1];
    }
    a.alpha[0] = ar, a.alpha[1] = ai, a.beta[0] = br, a.beta[1] = bi;
    a.mode = 0;
    if (ai == 0 && bi == 0) {
        if (ar == 1 && br == 0) a.mode = 1;
        else if (ar == 1 && br == 1) a.mode = 2;
        else if (ar == -1 && br == 0) a.mode = 3;
        else if (ar == -1 && br == 1) a.mode = 4;
    }
}

hipError_t launch_crt_partial(hipStream_t stream, int dtype, int backend, unsigned N, unsigned t_begin, unsigned t_end, size_t m, size_t n,
                              const void* Cmid, size_t ld_mid, size_t plane_stride, double* out_hi, double* out_lo, size_t ld_out,
                              size_t col_block, size_t block_stride) {
    if (m == 0 || n == 0) return hipSuccess;
    CrtArgs a{};
    a.Cmid = Cmid;
    a.ld_mid = ld_mid;
    a.plane_stride = plane_stride;
    a.m = m;
    a.n = n;
    fill_crt_tables(a, dtype, backend, N);
    const bool cplx = is_complex(dtype), i8 = backend == kINT8;
    const size_t rows_per_thread = 8 / ((cplx ? 2 : 1) * (i8 ? 1 : 2));
    const size_t threads = ((m + rows_per_thread - 1) / rows_per_thread) * n;
    dim3 grid((unsigned)((threads + 255) / 256));
#define OZ2_CRTP(CP, MID) hipLaunchKernelGGL((crt_partial_kernel<CP, MID>), grid, dim3(256), 0, stream, a, t_begin, t_end, out_hi, out_lo, ld_out, col_block, block_stride)
    if (i8) {
        if (cplx) OZ2_CRTP(true, int8_t);
        else OZ2_CRTP(false, int8_t);
    } else {
        if (cplx) OZ2_CRTP(true, int16_t);
        else OZ2_CRTP(false, int16_t);
    }
#undef OZ2_CRTP
    return hipGetLastError();
}

hipError_t launch_crt_finish(hipStream_t stream, int dtype, int backend, unsigned N, size_t m, size_t n, const double* in_hi,
                             const double* in_lo, size_t ld_in, const int16_t* sftA, const int16_t* sftB, const void* alpha,
                             const void* beta, bool scalars_on_device, void* C, size_t ldc) {
    if (m == 0 || n == 0) return hipSuccess;
    CrtArgs a{};
    a.m = m;
    a.n = n;
    a.sftA = sftA;
    a.sftB = sftB;
    a.C = C;
    a.ldc = ldc;
    fill_crt_tables(a, dtype, backend, N);
    fill_crt_scalars(a, dtype, alpha, beta, scalars_on_device);
    dim3 grid((unsigned)((m * n + 255) / 256));
    switch (dtype) {
    case kF32: hipLaunchKernelGGL((crt_finish_kernel<float, false>), grid, dim3(256), 0, stream, a, in_hi, in_lo, ld_in); break;
    case kF64: hipLaunchKernelGGL((crt_finish_kernel<double, false>), grid, dim3(256), 0, stream, a, in_hi, in_lo, ld_in); break;
    case kC32: hipLaunchKernelGGL((crt_finish_kernel<float, true>), grid, dim3(256), 0, stream, a, in_hi, in_lo, ld_in); break;
    case kC64: hipLaunchKernelGGL((crt_finish_kernel<double, true>), grid, dim3(256), 0, stream, a, in_hi, in_lo, ld_in); break;
    }
    return hipGetLastError();
}

// D(i, j) += bias[i]: the broadcast bias vector of a hipblasLtMatmul BIAS epilogue, applied after the emulated GEMM (the hook's only use)
template <typename U> __global__ void __launch_bounds__(256) row_bias_kernel(U* D, size_t ldd, size_t m, const U* bias) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) D[(size_t)blockIdx.y * ldd + i] += bias[i];
}
// dst[i] += src[i] on doubles: the running sum of the reduced FP64 partial planes of the pipelined fp64sum plan (oz2_dist.cpp)
__global__ void __launch_bounds__(256) add_f64_kernel(double* dst, const double* src, size_t count) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < count) {
        double2 d = *(double2*)(dst + i);
        const double2 s = *(const double2*)(src + i);
        d.x += s.x, d.y += s.y;
        *(double2*)(dst + i) = d;
    } else if (i < count) {
        dst[i] += src[i];
    }
}
hipError_t launch_add_f64(hipStream_t stream, double* dst, const double* src, size_t count) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(add_f64_kernel, dim3((unsigned)((count + 511) / 512)), dim3(256), 0, stream, dst, src, count);
    return hipGetLastError();
}

hipError_t launch_row_bias(hipStream_t stream, int dtype, size_t m, size_t n, void* D, size_t ldd, const void* bias) {
    if (m == 0 || n == 0) return hipSuccess;
    if (n > 65535) return hipErrorInvalidValue;
    dim3 grid((unsigned)((m + 255) / 256), (unsigned)n);
    if (dtype == kF32) hipLaunchKernelGGL(row_bias_kernel<float>, grid, dim3(256), 0, stream, (float*)D, ldd, m, (const float*)bias);
    else if (dtype == kF64) hipLaunchKernelGGL(row_bias_kernel<double>, grid, dim3(256), 0, stream, (double*)D, ldd, m, (const double*)bias);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_crt(hipStream_t stream, int dtype, int backend, unsigned N, size_t m, size_t n, const void* Cmid, size_t ld_mid,
                      size_t plane_stride, const int16_t* sftA, const int16_t* sftB, const void* alpha, const void* beta,
                      bool scalars_on_device, void* C, size_t ldc) {
    if (m == 0 || n == 0) return hipSuccess;
    CrtArgs a{};
    a.Cmid = Cmid;
    a.ld_mid = ld_mid;
    a.plane_stride = plane_stride;
    a.m = m;
    a.n = n;
    a.sftA = sftA;
    a.sftB = sftB;
    a.C = C;
    a.ldc = ldc;
    fill_crt_tables(a, dtype, backend, N);
    fill_crt_scalars(a, dtype, alpha, beta, scalars_on_device);
    a.bw = g_batch.ws;
    a.bc = g_batch.sc;
    const bool cplx = is_complex(dtype), i8 = backend == kINT8;
    {
        // LDS-DMA form: int8 residues, whole 1024-byte units per column, 16-byte aligned slices and C columns
        const size_t comps = cplx ? 2 : 1, usz = is_f32(dtype) ? 4 : 8;
        const int force = knobs().crt_kernel;  // 1 = dma, 2 = reg: testing switch (default: dma when eligible and large enough)
        const bool eligible = OZ2_CRT_DMA && i8 && (m * comps) % 1024 == 0 && (ld_mid * comps) % 16 == 0 && (plane_stride * comps) % 16 == 0 &&
                              ((uintptr_t)Cmid & 15u) == 0 && ((uintptr_t)C & 15u) == 0 && ((uintptr_t)sftA & 3u) == 0 /* dword loads of two int16 shifts */ && (ldc * comps * usz) % 16 == 0 && (g_batch.ws & 15u) == 0 &&
                              (g_batch.sc & 15u) == 0;
        const size_t units = m * comps / 1024 * n;
        const bool want = force == 1 ? true : force == 2 ? false : units >= 4096;
        if (eligible && want && units <= 0x7FFFFFFFull) {
            const unsigned upc = (unsigned)(m * comps / 1024);
            const size_t lds = (size_t)N * 1024;
            dim3 grid((unsigned)units, 1, g_batch.batch);
#define OZ2_CRT_DMA_LAUNCH(U, CP) hipLaunchKernelGGL((crt_dma_kernel<U, CP>), grid, dim3(64 * OZ2_CRT_DMA_WAVES), lds, stream, a, upc)
            switch (dtype) {
            case kF32: OZ2_CRT_DMA_LAUNCH(float, false); break;
            case kF64: OZ2_CRT_DMA_LAUNCH(double, false); break;
            case kC32: OZ2_CRT_DMA_LAUNCH(float, true); break;
            case kC64: OZ2_CRT_DMA_LAUNCH(double, true); break;
            }
#undef OZ2_CRT_DMA_LAUNCH
            return hipGetLastError();
        }
    }
    const size_t rows_per_thread = OZ2_CRT_LB / ((cplx ? 2 : 1) * (i8 ? 1 : 2));
    const size_t threads = ((m + rows_per_thread - 1) / rows_per_thread) * n;
    // small problems keep one unit per thread (more workgroups); large ones take the prefetching form
    const bool multi = OZ2_CRT_UNITS > 1 && threads >= (size_t)OZ2_CRT_BLOCK * OZ2_CRT_UNITS * 2048;
    const size_t per_block = (size_t)OZ2_CRT_BLOCK * (multi ? OZ2_CRT_UNITS : 1);
    dim3 grid((unsigned)((threads + per_block - 1) / per_block), 1, g_batch.batch);
#define OZ2_CRT(U, CP, MID)                                                                                                      \
    do {                                                                                                                         \
        if (multi) hipLaunchKernelGGL((crt_kernel<U, CP, MID, OZ2_CRT_LB, OZ2_CRT_UNITS, 20>), grid, dim3(OZ2_CRT_BLOCK), 0, stream, a); \
        else if (N <= 14) hipLaunchKernelGGL((crt_kernel<U, CP, MID, OZ2_CRT_LB, 1, 14>), grid, dim3(OZ2_CRT_BLOCK), 0, stream, a);       \
        else hipLaunchKernelGGL((crt_kernel<U, CP, MID, OZ2_CRT_LB, 1, 20>), grid, dim3(OZ2_CRT_BLOCK), 0, stream, a);                    \
    } while (0)
    if (i8) {
        switch (dtype) {
        case kF32: OZ2_CRT(float, false, int8_t); break;
        case kF64: OZ2_CRT(double, false, int8_t); break;
        case kC32: OZ2_CRT(float, true, int8_t); break;
        case kC64: OZ2_CRT(double, true, int8_t); break;
        }
    } else {
        switch (dtype) {
        case kF32: OZ2_CRT(float, false, int16_t); break;
        case kF64: OZ2_CRT(double, false, int16_t); break;
        case kC32: OZ2_CRT(float, true, int16_t); break;
        case kC64: OZ2_CRT(double, true, int16_t); break;
        }
    }
#undef OZ2_CRT
    return hipGetLastError();
}

}  // namespace oz2
