// Host orchestration + C ABI (include/gemmul8_c.h).
//
// Restates the reference's pipeline drivers (GEMMul8/src/gemmul8_real.hpp:8-211,
// src/gemmul8_complex.hpp:8-226): workspace carving, phase order, skip-scaling semantics -- with
// three differences that are the point of the MI355X build:
//   * the low-precision GEMMs are our own MFMA kernels with the requantise / bound-max epilogues
//     fused (no C_hi round trip, no vendor BLAS, no 32 MiB BLAS workspace use);
//   * all moduli of a call go out in ONE batched launch;
//   * no host synchronisation unless the caller asks for the 4 phase timers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <atomic>
#include <mutex>

#include "../../include/gemmul8_c.h"
#include "oz2_kernels.h"
#include "oz2_knobs.hpp"

using namespace oz2;

namespace {

inline char* align256(void* p) {
    uintptr_t x = reinterpret_cast<uintptr_t>(p);
    x = (x + 255) & ~uintptr_t(255);
    return reinterpret_cast<char*>(x);
}
inline int norm_op(int op) {
    if (op >= 111 && op <= 113) return op - 111;  // hipblasOperation_t
    return op;
}
inline size_t low_size(int) { return 1; }
inline size_t mid_size(int backend, bool cplx) { return (backend == kINT8 ? 1 : 2) * (cplx ? 2 : 1); }

// num_moduli range of a type: 2..20 for double / complex-double, 2..13 for float / complex-float (reference contract,
// GEMMul8/include/gemmul8.hpp:30; its float pipeline accepts more but overflows: P reaches 2^128 at 16 moduli -- the oracle's
// restatement returns inf there, tests/test_gpu_parity.py::test_float_types_reject_more_than_13_moduli)
static inline bool moduli_ok(int dtype, unsigned N) { return N >= 2 && N <= (is_f32(dtype) ? 13u : 20u); }

// FP8 backend: C0 + C1 of a square modulus may share one FP32 accumulator (K-concatenation) while every partial sum stays an exact
// integer: 2 k products of magnitude <= 16 * 16 (src/mod.hpp:159-189) <= 2^24
// -- and while the four operand planes of such a GEMM stay inside the Infinity Cache: interleaved (profiles/archive/r04_f8_concat_ab.txt) SGEMM 8192^2 x 4096 /
// 8192, 6 moduli: +4.5 / +2.3 % of the whole call, 16384^3 (1 GiB of planes per concatenated GEMM): -0.4 %
static inline bool f8_concat_ok(size_t k, size_t m, size_t n, size_t kp) { return k <= 32768 && 2 * (m + n) * kp <= ((size_t)384 << 20); }

#define OZ2_HIP(expr)                       \
    do {                                    \
        hipError_t e__ = (expr);            \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// Phase-timer events: one set per (host thread, device) -- an event may only be recorded on a stream of the device it was
// created on, and one host thread may drive several GPUs through this API.
struct Timer {
    hipEvent_t ev[4];
    bool made = false, ok = false;
};
Timer* thread_timer() {
    constexpr int kMaxDev = 64;
    static thread_local Timer t[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    Timer& T = t[dev];
    if (!T.made) {
        T.made = T.ok = true;
        for (auto& e : T.ev)
            if (hipEventCreate(&e) != hipSuccess) T.ok = false;
    }
    return T.ok ? &T : nullptr;
}

}  // namespace

static bool scalars_on_device(const void* alpha);

// ---- testing / A-B knobs (oz2_knobs.hpp): parsed once; gemmul8_reload_knobs publishes a fresh snapshot (old snapshots are kept: a
// launch in another thread may still be reading one -- a handful of tiny structs per process at most)
namespace oz2 {
static Knobs parse_knobs() {
    Knobs k;
    auto num = [](const char* name, int dflt) {
        const char* s = getenv(name);
        return s && *s ? atoi(s) : dflt;
    };
    if (const char* e = getenv("GEMMUL8_EPI_NT"); e && (e[0] == '0' || e[0] == '1') && !e[1]) k.epi_nt = e[0] - '0';
    if (const int t = num("GEMMUL8_BOUND_TILE", 0); t == 128 || t == 256) k.bound_tile = t;
    if (num("GEMMUL8_CPLX_BOUND_LAUNCHES", 1) == 2) k.cplx_bound_launches = 2;
    if (const int c = num("GEMMUL8_CPLX_CHUNK", 0); c >= 1) k.cplx_chunk = c;
    if (const char* e = getenv("GEMMUL8_CRT_KERNEL")) k.crt_kernel = e[0] == 'd' ? 1 : e[0] == 'r' ? 2 : 0;
    if (const int c = num("GEMMUL8_GEMM_CUS", 0); c >= 8) k.gemm_cus = c & ~7;
    if (const char* e = getenv("GEMMUL8_FP8_FUSED"); e && e[0] == '0') k.fp8_fused = 0;
    if (const char* e = getenv("GEMMUL8_FP8_PLANES")) k.fp8_planes = e[0] == 'e' ? 1 : 0;
    if (const char* e = getenv("GEMMUL8_SCALE_FOLD"); e && e[0] == '0') k.scale_fold = 0;
    if (const char* e = getenv("GEMMUL8_CRT_PANELS"); e && atoi(e) > 1) {
        k.crt_panels = std::min(atoi(e), 256);
        k.crt_panels_ring = e[strlen(e) - 1] == 'r';
    }
    if (const char* e = getenv("GEMMUL8_MAP_COLBLOCK"); e && *e) k.map_colblock = atoi(e) > 0 ? atoi(e) : 0;
    return k;
}
static std::atomic<const Knobs*> g_knobs{nullptr};
static std::mutex g_knobs_mtx;
void reload_knobs() {
    std::lock_guard<std::mutex> lk(g_knobs_mtx);
    g_knobs.store(new Knobs(parse_knobs()), std::memory_order_release);
}
const Knobs& knobs() {
    const Knobs* k = g_knobs.load(std::memory_order_acquire);
    if (!k) {
        std::lock_guard<std::mutex> lk(g_knobs_mtx);
        k = g_knobs.load(std::memory_order_acquire);
        if (!k) g_knobs.store(k = new Knobs(parse_knobs()), std::memory_order_release);
    }
    return *k;
}
}  // namespace oz2

extern "C" {

void gemmul8_reload_knobs(void) { oz2::reload_knobs(); }

int gemmul8_set_fp8_bound_mode(int mode) {
    if (mode < 0 || mode > 2) return GEMMUL8_E_ARG;
    const int old = get_f8_bound_mode();
    set_f8_bound_mode(mode);
    return old;
}

int gemmul8_abi_version(void) { return GEMMUL8_ABI_VERSION; }
size_t gemmul8_layout_bytes(void) { return sizeof(gemmul8_layout); }

const char* gemmul8_version(void) { return "gemmul8-mi355x 0.5 (gfx950; v_mfma_i32_16x16x64_i8 / v_mfma_scale_f32_16x16x128_f8f6f4, fused epilogues)"; }

size_t gemmul8_work_size(int is_complex, int backend, size_t m, size_t n, size_t k, unsigned N, int enA, int enB, size_t* wA,
                         size_t* wB) {
    const size_t kp = padding256(k), mp = padding256(m);
    const size_t sizeA = kp * mp, sizeB = kp * n, sizeC = mp * n;
    const size_t nm = num_mat(backend, N);
    const size_t parts = is_complex ? 3 : 1;
    const size_t midsz = mid_size(backend, is_complex != 0);
    const size_t nhi = (backend == kINT8 ? 1 : 3) * parts;
    const size_t lwork = size_t(1) << 25;
    size_t tA = 255, tB = 255, tC = 255;
    tA += sizeA * (nm + (enA ? 1 : 0)) * parts + 2 * mp;
    tB += sizeB * (nm + (enB ? 1 : 0)) * parts + 2 * padding256(n);
    tC += midsz * sizeC * (N - 1) + std::max(lwork, midsz * sizeC);
    tC += 4 * sizeC * nhi;
    if (wA) *wA = tA;
    if (wB) *wB = tB;
    return tA + tB + tC;
}

int gemmul8_get_layout(int dtype, int backend, size_t m, size_t n, size_t k, unsigned N, void* work, void* workA, void* workB,
                       int enA, int enB, gemmul8_layout* L) {
    if (!L || !work) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N)) return GEMMUL8_E_NUM_MODULI;
    const bool cplx = is_complex(dtype);
    memset(L, 0, sizeof(*L));
    L->kp = padding256(k);
    L->mp = padding256(m);
    L->num_mat = num_mat(backend, N);
    L->parts = cplx ? 3 : 1;
    L->sizeA = L->kp * L->mp;
    L->sizeB = L->kp * n;
    L->sizeC = L->mp * n;
    const size_t offsetA = L->sizeA * L->num_mat, offsetB = L->sizeB * L->num_mat;
    const size_t size_vecA = L->mp, size_vecB = padding256(n);
    char* w = align256(work);
    char* wa = workA ? align256(workA) : nullptr;
    char* wb = workB ? align256(workB) : nullptr;
    char* A_lo = wa ? wa : w;
    char* sftA = A_lo + L->parts * offsetA + (enA ? L->parts * L->sizeA : 0);
    char* B_lo = wb ? wb : (wa ? w : sftA + 2 * size_vecA);
    char* sftB = B_lo + L->parts * offsetB + (enB ? L->parts * L->sizeB : 0);
    char* C_mid = wb ? (wa ? w : sftA + 2 * size_vecA) : sftB + 2 * size_vecB;
    const size_t midsz = mid_size(backend, cplx);
    char* work_native = C_mid + (N - 1) * L->sizeC * midsz;
    const size_t native_sz = std::max(size_t(1) << 25, midsz * L->sizeC);
    char* C_hi = work_native + native_sz;
    const size_t hi_bytes = 4 * L->sizeC * (backend == kINT8 ? 1 : 3) * L->parts;
    L->A_lo = A_lo;
    L->B_lo = B_lo;
    L->part_strideA = offsetA;
    L->part_strideB = offsetB;
    L->A_bound = enA ? A_lo + L->parts * offsetA : A_lo;
    L->B_bound = enB ? B_lo + L->parts * offsetB : B_lo;
    L->sftA = reinterpret_cast<int16_t*>(sftA);
    L->sftB = reinterpret_cast<int16_t*>(sftB);
    L->C_mid = C_mid;
    // scratch: the unused tail of the 32 MiB BLAS-workspace block and the C_hi region behind it are one contiguous range
    // (so that tall-skinny problems, whose C_hi region alone is smaller than the row-maxima arrays, still fit)
    const size_t tail = (native_sz - midsz * L->sizeC) & ~size_t(255);
    L->scratch = C_hi - tail;
    L->scratch_bytes = tail + hi_bytes;
    // FP6 panel images need both operands in that encoding, and whether they fit depends on n: with skip-scaling enabled an operand's planes outlive
    // the call and may meet a partner of another shape (the reference's use: one A against changing B), so cached planes keep the e4m3 bytes
    L->lo_format = (backend == kFP8 && !enA && !enB && f6_planes_ok(n)) ? 1 : 0;
    return GEMMUL8_OK;
}

// scratch carving shared by the two halves of the scaling phase: rowmax int32[mp] | colmax int32[pad(n)] | sft0 copies int16[mp] | int16[pad(n)] | (256-aligned)
// the partial row-maxima arrays of the row-strided operands (dead once the extract launch has run: the complex FP8 bound's float plane reuses the space).
// The copies of the preliminary shifts are what the shift finalize folded into the quantise launch reads: L->sftA / sftB receive the final values there.
static inline size_t scale_fixed_bytes(size_t mp, size_t np) { return (4 * (mp + np) + 2 * (mp + np) + 255) / 256 * 256; }
static inline size_t scale_scratch_bytes(size_t mp, size_t np) { return scale_fixed_bytes(mp, np) + 8 * (mp + np); }  // at least one partial array per operand
static int scale_scratch(const gemmul8_layout* L, size_t n, int** rowmax, int** colmax, void** amax, int16_t** s0A = nullptr, int16_t** s0B = nullptr) {
    const size_t np = padding256(n);
    if (L->scratch_bytes < scale_scratch_bytes(L->mp, np)) return GEMMUL8_E_ARG;
    *rowmax = (int*)L->scratch;
    *colmax = *rowmax + L->mp;
    int16_t* s0 = (int16_t*)(*colmax + np);
    if (s0A) *s0A = s0;
    if (s0B) *s0B = s0 + L->mp;
    *amax = (char*)L->scratch + scale_fixed_bytes(L->mp, np);
    return GEMMUL8_OK;
}

int gemmul8_scale_bounds(void* stream_, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void* A,
                         size_t lda, const void* B, size_t ldb, unsigned N, size_t col_begin, size_t col_end, const gemmul8_layout* L,
                         int skipA, int skipB) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!L || !A || !B) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N)) return GEMMUL8_E_NUM_MODULI;
    if (backend == kFP8 && k > 65536) return GEMMUL8_E_ARG;  // exact FP32 accumulation needs k*16*16 <= 2^24
    op_A = norm_op(op_A);
    op_B = norm_op(op_B);
    if (op_A < 0 || op_A > 2 || op_B < 0 || op_B > 2 || col_begin > col_end || col_end > n) return GEMMUL8_E_ARG;
    if (skipA && skipB) return GEMMUL8_OK;
    const bool cplx = is_complex(dtype);
    const bool kmajA = op_A != 0, kmajB = op_B == 0;
    const bool conjA = cplx && op_A == 2, conjB = cplx && op_B == 2;
    int *rowmax, *colmax;
    void* amax;
    int16_t *s0A, *s0B;
    int rc = scale_scratch(L, n, &rowmax, &colmax, &amax, &s0A, &s0B);
    if (rc) return rc;
    const size_t np = padding256(n);
    const size_t bstrideA = cplx ? L->sizeA : 0, bstrideB = cplx ? L->sizeB : 0;
    // Round 6: (1) row maxima of the row-strided operands, both in one launch, as per-k-split partial arrays (no atomics: nothing to zero for them);
    // (2) the extract of BOTH operands in one launch, which also zero-fills the bound GEMM's maxima arrays.  GEMMUL8_SCALE_FOLD=0 (testing): the zero-fill
    // and the two extracts as launches of their own (the round-5 launch count).
    const bool runA = !skipA, runB = !skipB;
    const size_t ub = is_f32(dtype) ? 4 : 8;  // bytes of a row maximum
    const size_t srA = runA && !kmajA ? L->mp : 0, srB = runB && !kmajB ? np : 0;  // padded row counts of the operands that need the pass
    const size_t room = (L->scratch_bytes - scale_fixed_bytes(L->mp, np)) / ub;
    ExtractOperand ea, eb;
    if (runA) {
        ea = ExtractOperand{kmajA, conjA, m, A, lda, (int8_t*)L->A_bound, bstrideA, L->sftA, s0A, g_batch.sa, nullptr, 1, L->mp};
        if (srA) ea.amax = amax, ea.parts = amax_parts_for(m, k, room * srA / (srA + srB) / srA);
    }
    if (runB) {
        eb = ExtractOperand{kmajB, conjB, n, B, ldb, (int8_t*)L->B_bound, bstrideB, L->sftB, s0B, g_batch.sb, nullptr, 1, np};
        if (srB) eb.amax = (char*)amax + (srA ? (size_t)ea.parts * srA * ub : 0), eb.parts = amax_parts_for(n, k, room * srB / (srA + srB) / srB);
    }
    if (srA + srB) OZ2_HIP(launch_amax_pair(stream, dtype, k, ea, eb));
    const size_t zero_bytes = 4 * (L->mp + np);
    if (knobs().scale_fold) {
        OZ2_HIP(launch_extract_pair(stream, dtype, backend, k, L->kp, ea, eb, rowmax, zero_bytes));
    } else {
        OZ2_HIP(launch_zero(stream, rowmax, zero_bytes));
        OZ2_HIP(launch_extract_pair(stream, dtype, backend, k, L->kp, ea, ExtractOperand{}, nullptr, 0));
        OZ2_HIP(launch_extract_pair(stream, dtype, backend, k, L->kp, ExtractOperand{}, eb, nullptr, 0));
    }
    if (col_end > col_begin) {
        const int8_t* Ab = (const int8_t*)L->A_bound;
        const int8_t* Bb = (const int8_t*)L->B_bound + col_begin * L->kp;
        if (backend == kFP8 && !cplx) {
            OZ2_HIP(launch_gemm_f8_max(stream, Ab, Bb, L->kp, k, m, col_end - col_begin, rowmax, colmax + col_begin));
        } else if (backend == kFP8) {
            // complex FP8 (find_max.hpp, complex FP8 overloads; scaling_accu_complex.hpp:150-175): three separately inflated
            // products ArBi, AiBr and (Ar-Ai)(Br-Bi) combined with round-up additions; the first two pass through a float
            // scratch plane [ncols][mp] behind the maxima arrays.
            const size_t ncols = col_end - col_begin;
            const size_t foff = scale_fixed_bytes(L->mp, np);  // (the partial row maxima behind it are dead by now)
            if (L->scratch_bytes < foff + 4 * L->mp * ncols) return GEMMUL8_E_ARG;
            float* fbuf = (float*)((char*)L->scratch + foff);
            OZ2_HIP(launch_gemm_f8_bound_cplx(stream, 1, Ab, Bb + L->sizeB, L->kp, k, m, ncols, fbuf, L->mp, rowmax, colmax + col_begin));
            OZ2_HIP(launch_gemm_f8_bound_cplx(stream, 2, Ab + L->sizeA, Bb, L->kp, k, m, ncols, fbuf, L->mp, rowmax, colmax + col_begin));
            OZ2_HIP(launch_gemm_f8_bound_cplx(stream, 3, Ab + 2 * L->sizeA, Bb + 2 * L->sizeB, L->kp, k, m, ncols, fbuf, L->mp, rowmax,
                                              colmax + col_begin));
        } else if (!cplx) {
            const int8_t* As[1] = {Ab};
            const int8_t* Bs[1] = {Bb};
            OZ2_HIP(launch_gemm_i8_max(stream, 1, As, Bs, L->kp, m, col_end - col_begin, rowmax, colmax + col_begin));
        } else {
            // bound planes: 0 = |Re|, 1 = |Im|, 2 = |Re|-|Im| (scaling_accu_complex.hpp:441-460, find_max.hpp:99-114):
            //   C1 = ArBi + AiBr (K-concatenation of two products), C1 + C0 = ArBr + AiBi with C0 = (Ar-Ai)(Br-Bi);
            //   the maxima of both matrices accumulate into the same rowmax/colmax.
            const int8_t* As[3] = {Ab, Ab + L->sizeA, Ab + 2 * L->sizeA};
            const int8_t* Bs[3] = {Bb + L->sizeB, Bb, Bb + 2 * L->sizeB};
            //   One launch over the three segments; the maxima are taken after the second (C1) and after the third (C1 + C0).
            const bool two_launches = knobs().cplx_bound_launches == 2;  // A/B and testing switch (oz2_knobs.hpp)
            if (two_launches) {
                OZ2_HIP(launch_gemm_i8_max(stream, 2, As, Bs, L->kp, m, col_end - col_begin, rowmax, colmax + col_begin));
                OZ2_HIP(launch_gemm_i8_max(stream, 3, As, Bs, L->kp, m, col_end - col_begin, rowmax, colmax + col_begin));
            } else {
                OZ2_HIP(launch_gemm_i8_max(stream, 3, As, Bs, L->kp, m, col_end - col_begin, rowmax, colmax + col_begin, 2));
            }
        }
    }
    return GEMMUL8_OK;
}

int gemmul8_scale_finish(void* stream_, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void* A,
                         size_t lda, const void* B, size_t ldb, unsigned N, int fastmode, unsigned t_begin, unsigned t_end,
                         const gemmul8_layout* L, int skipA, int skipB) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!L || !A || !B) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N) || t_end > N || t_begin > t_end) return GEMMUL8_E_NUM_MODULI;
    if (backend == kFP8 && k > 65536) return GEMMUL8_E_ARG;
    op_A = norm_op(op_A);
    op_B = norm_op(op_B);
    if (op_A < 0 || op_A > 2 || op_B < 0 || op_B > 2) return GEMMUL8_E_ARG;
    if (skipA && skipB) return GEMMUL8_OK;
    const bool cplx = is_complex(dtype);
    const bool kmajA = op_A != 0, kmajB = op_B == 0;
    const bool conjA = cplx && op_A == 2, conjB = cplx && op_B == 2;
    QuantOperand oa, ob;  // rows == 0: the operand is skipped
    const bool f6 = backend == kFP8 && L->lo_format == 1;  // FP6 panel images (oz2_gemm_f6.hip): A's blocks are whole (mp rows), B's last one may be short
    if (!skipA) oa = QuantOperand{kmajA, conjA, m, A, lda, L->sftA, (int8_t*)L->A_lo, L->sizeA, L->part_strideA, g_batch.sa, f6 ? L->mp : 0};
    if (!skipB) ob = QuantOperand{kmajB, conjB, n, B, ldb, L->sftB, (int8_t*)L->B_lo, L->sizeB, L->part_strideB, g_batch.sb, f6 ? n : 0};
    if (fastmode) {
        OZ2_HIP(launch_fast_shift_pair(stream, dtype, backend, N, k, oa, ob));
    } else {
        int *rowmax, *colmax;
        void* amax;
        int16_t *s0A, *s0B;
        int rc = scale_scratch(L, n, &rowmax, &colmax, &amax, &s0A, &s0B);
        if (rc) return rc;
        if (t_end > t_begin && knobs().scale_fold) {
            // the finalize rides on the quantise launch below (one dispatch less: 4-5 us of a launch-bound call): every workgroup derives its rows' final
            // shifts from the extract's preliminary ones (scratch copy) and the bound maxima, the first k chunk of a row publishes them to L->sftA / sftB
            const float log2P = backend == kINT8 ? GEMMUL8_LOG2P_INT8[N - 2] : GEMMUL8_LOG2P_FP8[N - 2];
            oa.fin_sft0 = s0A, oa.fin_max = rowmax, oa.fin_log2P = log2P;
            ob.fin_sft0 = s0B, ob.fin_max = colmax, ob.fin_log2P = log2P;
        } else {  // no plane to write (a rank without moduli): the shifts are still wanted by the CRT
            OZ2_HIP(launch_shift_finalize(stream, backend, N, skipA ? 0 : m, rowmax, L->sftA, skipB ? 0 : n, colmax, L->sftB));
        }
    }
    OZ2_HIP(launch_quantise_pair(stream, dtype, backend, (int)t_begin, (int)t_end, k, L->kp, oa, ob));
    return GEMMUL8_OK;
}

int gemmul8_scale(void* stream_, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void* A, size_t lda,
                  const void* B, size_t ldb, unsigned N, int fastmode, unsigned t_begin, unsigned t_end, const gemmul8_layout* L,
                  int skipA, int skipB) {
    if (!fastmode) {
        int rc = gemmul8_scale_bounds(stream_, dtype, backend, op_A, op_B, m, n, k, A, lda, B, ldb, N, 0, n, L, skipA, skipB);
        if (rc) return rc;
    }
    return gemmul8_scale_finish(stream_, dtype, backend, op_A, op_B, m, n, k, A, lda, B, ldb, N, fastmode, t_begin, t_end, L, skipA, skipB);
}

int gemmul8_lowprec_gemm(void* stream_, int dtype, int backend, size_t m, size_t n, size_t k, unsigned N, unsigned t_begin,
                         unsigned t_end, const gemmul8_layout* L) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!L) return GEMMUL8_E_ARG;
    // the layout carries the padded inner dimension the planes were built with: a k that does not pad to it is a caller error (the FP8
    // K-concatenation below is exact only while 2 k 16^2 <= 2^24, so a placeholder k must not pass)
    if (padding256(k) != L->kp) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N) || t_end > N || t_begin > t_end) return GEMMUL8_E_NUM_MODULI;
    const int8_t* A_lo = (const int8_t*)L->A_lo;
    const int8_t* B_lo = (const int8_t*)L->B_lo;
    if (backend == kFP8 && L->lo_format == 1 && n < 64) return GEMMUL8_E_ARG;  // (an image layout for a shape it cannot hold: not from gemmul8_get_layout)
    // FP8 backend: the residue GEMMs run on e4m3 byte planes or on FP6 panel images of the same integers (twice the matrix rate), as the layout says
    const bool f6 = backend == kFP8 && L->lo_format == 1;
    auto launch_gemm_f8 = [f6](hipStream_t st, int which, const int8_t* A, const int8_t* B, size_t sA, size_t sB, size_t kp, size_t mm, size_t nn, int tb, int te,
                               int16_t* out, size_t ldo, size_t sO, const int16_t* r0, const int16_t* r1, size_t sR, const int16_t* rx = nullptr,
                               const int16_t* ry = nullptr) {
        return f6 ? oz2::launch_gemm_f6(st, which, A, B, sA, sB, kp, mm, nn, tb, te, out, ldo, sO, r0, r1, sR, rx, ry)
                  : oz2::launch_gemm_f8(st, which, A, B, sA, sB, kp, mm, nn, tb, te, out, ldo, sO, r0, r1, sR, rx, ry);
    };
    // FP6 planes (round 5): the three products of a modulus run as ONE tile loop over three K segments with the accumulators reduced in between
    // (launch_gemm_f6 which = 7 / 8, f8_fill_planes): no partial-residue planes, one epilogue and one launch instead of two or three.  Every
    // accumulator stays an exact integer while kp * 256 + 2^16.1 <= 2^24; beyond that (kp > 65024) the three-launch form below.
    const bool f6_fused = f6 && L->kp <= 65024 && knobs().fp8_fused != 0;
    if (f6_fused && !is_complex(dtype)) {
        OZ2_HIP(launch_gemm_f8(stream, 7, A_lo, B_lo, L->sizeA, L->sizeB, L->kp, m, n, (int)t_begin, (int)t_end,
                               (int16_t*)L->C_mid + (size_t)t_begin * L->sizeC, L->mp, L->sizeC, nullptr, nullptr, 0));
        return GEMMUL8_OK;
    }
    if (f6_fused) {
        // complex: the parts X = ArBr, Y = AiBi to canonical int16 scratch planes, the part Z = (Ar+Ai)(Br+Bi) with the complex combine behind it
        const size_t per_mod = 2 * 2 * L->sizeC;
        const size_t chunk = L->scratch_bytes / per_mod;
        if (chunk == 0) return GEMMUL8_E_ARG;
        int16_t* rx = (int16_t*)L->scratch;
        for (unsigned t0 = t_begin; t0 < t_end; t0 += (unsigned)chunk) {
            const unsigned t1 = std::min<unsigned>(t_end, t0 + (unsigned)chunk);
            int16_t* ry = rx + (size_t)(t1 - t0) * L->sizeC;
            for (int part = 0; part < 3; ++part) {
                const int8_t* Ap = A_lo + part * L->part_strideA;
                const int8_t* Bp = B_lo + part * L->part_strideB;
                if (part < 2) {
                    OZ2_HIP(launch_gemm_f8(stream, 7, Ap, Bp, L->sizeA, L->sizeB, L->kp, m, n, (int)t0, (int)t1, part == 0 ? rx : ry, L->mp, L->sizeC, nullptr, nullptr, 0));
                } else {
                    OZ2_HIP(launch_gemm_f8(stream, 8, Ap, Bp, L->sizeA, L->sizeB, L->kp, m, n, (int)t0, (int)t1, (int16_t*)L->C_mid + (size_t)t0 * 2 * L->sizeC, L->mp,
                                           2 * L->sizeC, nullptr, nullptr, L->sizeC, rx, ry));
                }
            }
        }
        return GEMMUL8_OK;
    }
    if (backend == kFP8 && !is_complex(dtype)) {
        // three e4m3 GEMMs per modulus (gemmul8_real.hpp:159-181); the residues of the first two wait in int16 scratch planes
        // (the reference's C_hi region) for the third one's epilogue.  Moduli are chunked to the scratch size.
        const size_t per_mod = 2 * 2 * L->sizeC;
        const size_t chunk = L->scratch_bytes / per_mod;
        if (chunk == 0) return GEMMUL8_E_ARG;
        int16_t* r0 = (int16_t*)L->scratch;
        for (unsigned t0 = t_begin; t0 < t_end; t0 += (unsigned)chunk) {
            const unsigned t1 = std::min<unsigned>(t_end, t0 + (unsigned)chunk);
            int16_t* r1 = r0 + (size_t)(t1 - t0) * L->sizeC;
            // Square moduli (t < 6: value = s (C0 + C1) + C2): C0 + C1 as ONE GEMM over the K-concatenation [Ahi | Alo] x [Blo ; Bhi] (round 4) --
            // one residue plane and one epilogue less per modulus, same MACs, same bits; exact while 2 k * 16 * 16 <= 2^24
            const unsigned tc = f8_concat_ok(k, m, n, L->kp) ? std::min<unsigned>(t1, std::max<unsigned>(t0, 6u)) : t0;  // [t0, tc): concatenated form
            if (tc > t0) {
                OZ2_HIP(launch_gemm_f8(stream, 4, A_lo, B_lo, L->sizeA, L->sizeB, L->kp, m, n, (int)t0, (int)tc, r0, L->mp, L->sizeC, nullptr, nullptr, 0));
                OZ2_HIP(launch_gemm_f8(stream, 5, A_lo, B_lo, L->sizeA, L->sizeB, L->kp, m, n, (int)t0, (int)tc,
                                       (int16_t*)L->C_mid + (size_t)t0 * L->sizeC, L->mp, L->sizeC, r0, r0, L->sizeC));
            }
            if (tc < t1) {
                int16_t *q0 = r0 + (size_t)(tc - t0) * L->sizeC, *q1 = r1 + (size_t)(tc - t0) * L->sizeC;
                OZ2_HIP(launch_gemm_f8(stream, 0, A_lo, B_lo, L->sizeA, L->sizeB, L->kp, m, n, (int)tc, (int)t1, q0, L->mp, L->sizeC, nullptr, nullptr, 0));
                OZ2_HIP(launch_gemm_f8(stream, 1, A_lo, B_lo, L->sizeA, L->sizeB, L->kp, m, n, (int)tc, (int)t1, q1, L->mp, L->sizeC, nullptr, nullptr, 0));
                OZ2_HIP(launch_gemm_f8(stream, 2, A_lo, B_lo, L->sizeA, L->sizeB, L->kp, m, n, (int)tc, (int)t1,
                                       (int16_t*)L->C_mid + (size_t)tc * L->sizeC, L->mp, L->sizeC, q0, q1, L->sizeC));
            }
        }
        return GEMMUL8_OK;
    }
    if (backend == kFP8) {
        // complex FP8: nine e4m3 GEMMs per modulus (gemmul8_complex.hpp:170-195, matmult.hpp:355-404): each of the three complex
        // parts X = ArBr, Y = AiBi, Z = (Ar+Ai)(Br+Bi) is a 3-GEMM modular product as in the real case; the residues of X and Y
        // wait in int16 scratch planes for the epilogue of Z's last GEMM, which writes the interleaved (Cr, Ci) plane
        // (conv_hi2mid_complex.hpp:28-41).  Scratch per modulus: r0, r1, X, Y.
        const size_t per_mod = 4 * 2 * L->sizeC;
        const size_t chunk = L->scratch_bytes / per_mod;
        if (chunk == 0) return GEMMUL8_E_ARG;
        int16_t* r0 = (int16_t*)L->scratch;
        for (unsigned t0 = t_begin; t0 < t_end; t0 += (unsigned)chunk) {
            const unsigned t1 = std::min<unsigned>(t_end, t0 + (unsigned)chunk);
            const size_t nt = t1 - t0;
            int16_t *r1 = r0 + nt * L->sizeC, *rx = r1 + nt * L->sizeC, *ry = rx + nt * L->sizeC;
            const unsigned tc = f8_concat_ok(k, m, n, L->kp) ? std::min<unsigned>(t1, std::max<unsigned>(t0, 6u)) : t0;  // [t0, tc): square moduli, concatenated form (real path above)
            for (int part = 0; part < 3; ++part) {
                const int8_t* Ap = A_lo + part * L->part_strideA;
                const int8_t* Bp = B_lo + part * L->part_strideB;
                // moduli [a0, a1) with partial residues in planes po.. of the chunk's scratch; cc = concatenated C0 + C1
                auto product = [&](unsigned a0, unsigned a1, bool cc) -> int {
                    if (a0 >= a1) return 0;
                    const size_t po = (size_t)(a0 - t0) * L->sizeC;
                    if (cc) {
                        OZ2_HIP(launch_gemm_f8(stream, 4, Ap, Bp, L->sizeA, L->sizeB, L->kp, m, n, (int)a0, (int)a1, r0 + po, L->mp, L->sizeC, nullptr, nullptr, 0));
                    } else {
                        OZ2_HIP(launch_gemm_f8(stream, 0, Ap, Bp, L->sizeA, L->sizeB, L->kp, m, n, (int)a0, (int)a1, r0 + po, L->mp, L->sizeC, nullptr, nullptr, 0));
                        OZ2_HIP(launch_gemm_f8(stream, 1, Ap, Bp, L->sizeA, L->sizeB, L->kp, m, n, (int)a0, (int)a1, r1 + po, L->mp, L->sizeC, nullptr, nullptr, 0));
                    }
                    if (part < 2) {
                        OZ2_HIP(launch_gemm_f8(stream, cc ? 5 : 2, Ap, Bp, L->sizeA, L->sizeB, L->kp, m, n, (int)a0, (int)a1, (part == 0 ? rx : ry) + po, L->mp,
                                               L->sizeC, r0 + po, r1 + po, L->sizeC));
                    } else {
                        OZ2_HIP(launch_gemm_f8(stream, cc ? 6 : 3, Ap, Bp, L->sizeA, L->sizeB, L->kp, m, n, (int)a0, (int)a1,
                                               (int16_t*)L->C_mid + (size_t)a0 * 2 * L->sizeC, L->mp, 2 * L->sizeC, r0 + po, r1 + po, L->sizeC, rx + po, ry + po));
                    }
                    return 0;
                };
                if (int rc = product(t0, tc, true)) return rc;
                if (int rc = product(tc, t1, false)) return rc;
            }
        }
        return GEMMUL8_OK;
    }
    if (!is_complex(dtype)) {
        OZ2_HIP(launch_gemm_i8_mod(stream, A_lo + (size_t)t_begin * L->sizeA, B_lo + (size_t)t_begin * L->sizeB, L->sizeA, L->sizeB, L->kp, m, n,
                                   (int)t_begin, (int)t_end, (int8_t*)L->C_mid + (size_t)t_begin * L->sizeC, L->mp, L->sizeC, true));
        return GEMMUL8_OK;
    }
    // complex (gemmul8_complex.hpp:154-206, conv_hi2mid_complex.hpp:9-127): per modulus X = ArBr, Y = AiBi,
    // Z = (Ar+Ai)(Br+Bi).  The residues of X and Y go to scratch planes (the reference's C_hi region), the Z GEMM
    // combines them in its epilogue into the interleaved (Cr, Ci) plane.  Moduli are chunked to the scratch size.
    const size_t per_mod = 2 * L->sizeC;
    size_t chunk = L->scratch_bytes / per_mod;
    if (chunk == 0) return GEMMUL8_E_ARG;
    if (const size_t want = (size_t)knobs().cplx_chunk; want >= 1 && want < chunk) chunk = want;  // A/B switch (oz2_knobs.hpp)
    int8_t* rx = (int8_t*)L->scratch;
    for (unsigned t0 = t_begin; t0 < t_end; t0 += (unsigned)chunk) {
        const unsigned t1 = std::min<unsigned>(t_end, t0 + (unsigned)chunk);
        int8_t* ry = rx + (size_t)(t1 - t0) * L->sizeC;
        OZ2_HIP(launch_gemm_i8_mod(stream, A_lo + (size_t)t0 * L->sizeA, B_lo + (size_t)t0 * L->sizeB, L->sizeA, L->sizeB, L->kp, m, n,
                                   (int)t0, (int)t1, rx, L->mp, L->sizeC, false));  // X, Y: re-read by the Z launch right away
        OZ2_HIP(launch_gemm_i8_mod(stream, A_lo + L->part_strideA + (size_t)t0 * L->sizeA, B_lo + L->part_strideB + (size_t)t0 * L->sizeB,
                                   L->sizeA, L->sizeB, L->kp, m, n, (int)t0, (int)t1, ry, L->mp, L->sizeC, false));
        OZ2_HIP(launch_gemm_i8_cplx(stream, A_lo + 2 * L->part_strideA + (size_t)t0 * L->sizeA,
                                    B_lo + 2 * L->part_strideB + (size_t)t0 * L->sizeB, L->sizeA, L->sizeB, L->kp, m, n, (int)t0, (int)t1, rx,
                                    ry, L->sizeC, (int8_t*)L->C_mid + (size_t)t0 * 2 * L->sizeC, L->mp, 2 * L->sizeC));
    }
    return GEMMUL8_OK;
}

int gemmul8_crt(void* stream_, int dtype, int backend, unsigned N, size_t m, size_t n, const void* C_mid, size_t ld_mid,
                size_t plane_stride, const int16_t* sftA, const int16_t* sftB, const void* alpha, const void* beta, void* C, size_t ldc) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!C_mid || !sftA || !sftB || !alpha || !beta || !C) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N)) return GEMMUL8_E_NUM_MODULI;
    OZ2_HIP(launch_crt(stream, dtype, backend, N, m, n, C_mid, ld_mid, plane_stride, sftA, sftB, alpha, beta, scalars_on_device(alpha), C, ldc));
    return GEMMUL8_OK;
}

static bool scalars_on_device(const void* alpha) {
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, alpha) == hipSuccess)
        return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeArray;
    (void)hipGetLastError();  // unregistered host pointer: clear the sticky error
    return false;
}

int gemmul8_crt_partial(void* stream_, int dtype, int backend, unsigned N, unsigned t_begin, unsigned t_end, size_t m, size_t n,
                        const void* C_mid, size_t ld_mid, size_t plane_stride, double* out_hi, double* out_lo, size_t ld_out,
                        size_t col_block, size_t block_stride) {
    if (!C_mid || !out_hi || !out_lo || col_block == 0) return GEMMUL8_E_ARG;
    if (dtype < 0 || dtype > 3 || backend < 0 || backend > 1) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N) || t_end > N || t_begin > t_end) return GEMMUL8_E_NUM_MODULI;
    OZ2_HIP(launch_crt_partial((hipStream_t)stream_, dtype, backend, N, t_begin, t_end, m, n, C_mid, ld_mid, plane_stride, out_hi, out_lo,
                               ld_out, col_block, block_stride));
    return GEMMUL8_OK;
}

int gemmul8_crt_finish(void* stream_, int dtype, int backend, unsigned N, size_t m, size_t n, const double* in_hi, const double* in_lo,
                       size_t ld_in, const int16_t* sftA, const int16_t* sftB, const void* alpha, const void* beta, void* C, size_t ldc) {
    if (!in_hi || !in_lo || !sftA || !sftB || !alpha || !beta || !C) return GEMMUL8_E_ARG;
    if (dtype < 0 || dtype > 3 || backend < 0 || backend > 1) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N)) return GEMMUL8_E_NUM_MODULI;
    OZ2_HIP(launch_crt_finish((hipStream_t)stream_, dtype, backend, N, m, n, in_hi, in_lo, ld_in, sftA, sftB, alpha, beta,
                              scalars_on_device(alpha), C, ldc));
    return GEMMUL8_OK;
}

int gemmul8_gemm(void* stream_, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void* alpha,
                 const void* A, size_t lda, const void* B, size_t ldb, const void* beta, void* C, size_t ldc, unsigned N, int fastmode,
                 void* work, void* workA, void* workB, int enA, int enB, int skip_scalA, int skip_scalB, double* timers_ns) {
    hipStream_t stream = (hipStream_t)stream_;
    if (timers_ns) timers_ns[0] = timers_ns[1] = timers_ns[2] = timers_ns[3] = 0.0;
    if (dtype < 0 || dtype > 3 || backend < 0 || backend > 1) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N)) return GEMMUL8_E_NUM_MODULI;
    if (!alpha || !beta || !A || !B || !C || !work) return GEMMUL8_E_ARG;
    if (k > (size_t(1) << 17)) return GEMMUL8_E_ARG;
    if (m == 0 || n == 0 || k == 0) return GEMMUL8_OK;  // success, C untouched: what the reference's hook does (hook.cu:616-617); its gemm() itself has no check
    gemmul8_layout L;
    int rc = gemmul8_get_layout(dtype, backend, m, n, k, N, work, workA, workB, enA, enB, &L);
    if (rc) return rc;
    const bool skipA = skip_scalA && enA, skipB = skip_scalB && enB;
    // timers are a convenience: if the events cannot be used (e.g. the stream belongs to another device than the current
    // one) the GEMM still runs, untimed, and the four slots stay 0
    Timer* T = timers_ns ? thread_timer() : nullptr;
    if (T && hipEventRecord(T->ev[0], stream) != hipSuccess) {
        (void)hipGetLastError();
        T = nullptr;
    }
    rc = gemmul8_scale(stream, dtype, backend, op_A, op_B, m, n, k, A, lda, B, ldb, N, fastmode, 0, N, &L, skipA, skipB);
    if (rc) return rc;
    if (T) OZ2_HIP(hipEventRecord(T->ev[1], stream));
    // Column panels (testing knob GEMMUL8_CRT_PANELS; SURVEY 8 f3 "by cache residency", measured in profiles/r06_panel_crt*.txt): residue GEMMs of panel p,
    // then its CRT, so that a panel's C_mid could stay in the Infinity Cache in between.  Real INT8 only; panel edges on tile columns.
    const int P = (backend == kINT8 && !is_complex(dtype) && g_batch.batch <= 1) ? std::min<int>(knobs().crt_panels, (int)((n + 255) / 256)) : 0;
    if (P > 1) {
        const size_t tiles = (n + 255) / 256, esz = is_f32(dtype) ? 4 : 8;
        for (int p = 0; p < P; ++p) {
            const size_t c0 = std::min(n, tiles * p / P * 256), c1 = std::min(n, tiles * (p + 1) / P * 256);
            if (c1 <= c0) continue;
            gemmul8_layout Lp = L;
            Lp.B_lo = (char*)L.B_lo + c0 * L.kp;
            if (!knobs().crt_panels_ring) Lp.C_mid = (char*)L.C_mid + c0 * L.mp;
            rc = gemmul8_lowprec_gemm(stream, dtype, backend, m, c1 - c0, k, N, 0, N, &Lp);
            if (rc) return rc;
            rc = gemmul8_crt(stream, dtype, backend, N, m, c1 - c0, Lp.C_mid, L.mp, L.sizeC, L.sftA, L.sftB + c0, alpha, beta, (char*)C + c0 * ldc * esz, ldc);
            if (rc) return rc;
        }
        if (T) OZ2_HIP(hipEventRecord(T->ev[2], stream));  // the phases interleave: everything is booked under the low-precision GEMMs
    } else {
        rc = gemmul8_lowprec_gemm(stream, dtype, backend, m, n, k, N, 0, N, &L);
        if (rc) return rc;
        if (T) OZ2_HIP(hipEventRecord(T->ev[2], stream));
        rc = gemmul8_crt(stream, dtype, backend, N, m, n, L.C_mid, L.mp, L.sizeC, L.sftA, L.sftB, alpha, beta, C, ldc);
        if (rc) return rc;
    }
    if (T) {
        OZ2_HIP(hipEventRecord(T->ev[3], stream));
        OZ2_HIP(hipEventSynchronize(T->ev[3]));
        float ms;
        OZ2_HIP(hipEventElapsedTime(&ms, T->ev[0], T->ev[1]));
        timers_ns[0] = ms * 1e6;
        OZ2_HIP(hipEventElapsedTime(&ms, T->ev[1], T->ev[2]));
        timers_ns[1] = ms * 1e6;
        timers_ns[2] = 0.0;
        OZ2_HIP(hipEventElapsedTime(&ms, T->ev[2], T->ev[3]));
        timers_ns[3] = ms * 1e6;
    }
    return GEMMUL8_OK;
}

int gemmul8_add_f64(void* stream_, double* dst, const double* src, size_t count) {
    if (!dst || !src || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return GEMMUL8_E_ARG;
    OZ2_HIP(launch_add_f64((hipStream_t)stream_, dst, src, count));
    return GEMMUL8_OK;
}

int gemmul8_add_row_bias(void* stream_, int dtype, size_t m, size_t n, void* D, size_t ldd, const void* bias) {
    if (!D || !bias) return GEMMUL8_E_ARG;
    if (dtype != kF32 && dtype != kF64) return GEMMUL8_E_UNSUPPORTED;
    OZ2_HIP(launch_row_bias((hipStream_t)stream_, dtype, m, n, D, ldd, bias));
    return GEMMUL8_OK;
}

// ---- strided batch as ONE set of launches (no counterpart in the reference; hipblas{S,D,C,Z}gemmStridedBatched in the hook).
// The items' workspaces are consecutive blocks of gemmul8_batched_item_bytes; every kernel of the pipeline takes the item from
// gridDim.z (the persistent GEMM kernels fold the items into their plane sequence), so that a batch of small matrices fills the
// chip and costs six launches (accurate mode) instead of six per item.  Results are bit-identical to per-item gemmul8_gemm calls.
size_t gemmul8_batched_item_bytes(int is_complex, int backend, size_t m, size_t n, size_t k, unsigned N) {
    return padding256(gemmul8_work_size(is_complex, backend, m, n, k, N, 0, 0, nullptr, nullptr));
}
size_t gemmul8_work_size_batched(int is_complex, int backend, size_t m, size_t n, size_t k, unsigned N, size_t batch) {
    return gemmul8_batched_item_bytes(is_complex, backend, m, n, k, N) * batch + 256;
}

int gemmul8_gemm_batched(void* stream_, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void* alpha,
                         const void* A, size_t lda, long long strideA, const void* B, size_t ldb, long long strideB, const void* beta,
                         void* C, size_t ldc, long long strideC, size_t batch, unsigned N, int fastmode, void* work) {
    if (dtype < 0 || dtype > 3 || backend < 0 || backend > 1) return GEMMUL8_E_ARG;
    if (!moduli_ok(dtype, N)) return GEMMUL8_E_NUM_MODULI;
    if (!alpha || !beta || !A || !B || !C || !work) return GEMMUL8_E_ARG;
    if (k > (size_t(1) << 17)) return GEMMUL8_E_ARG;
    if (backend == kFP8 && k > 65536) return GEMMUL8_E_ARG;  // exact FP32 accumulation needs k*16*16 <= 2^24
    if (m == 0 || n == 0 || k == 0 || batch == 0) return GEMMUL8_OK;
    const size_t esz = (is_f32(dtype) ? 4 : 8) * (is_complex(dtype) ? 2 : 1);
    const size_t W = gemmul8_batched_item_bytes(is_complex(dtype), backend, m, n, k, N);
    struct Guard {
        ~Guard() { g_batch = BatchCtx{}; }
    } guard;
    char* w0 = align256(work);
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {  // gridDim.z limit
        const size_t nb = std::min<size_t>(65535, batch - b0);
        const char* Ab = (const char*)A + (long long)b0 * strideA * (long long)esz;
        const char* Bb = (const char*)B + (long long)b0 * strideB * (long long)esz;
        char* Cb = (char*)C + (long long)b0 * strideC * (long long)esz;
        char* wb = w0 + b0 * W;
        gemmul8_layout L;
        int rc = gemmul8_get_layout(dtype, backend, m, n, k, N, wb, nullptr, nullptr, 0, 0, &L);
        if (rc) return rc;
        g_batch.batch = (unsigned)nb;
        g_batch.ws = W;
        g_batch.sa = (size_t)(strideA * (long long)esz);  // negative strides wrap: pointer arithmetic is modular
        g_batch.sb = (size_t)(strideB * (long long)esz);
        g_batch.sc = (size_t)(strideC * (long long)esz);
        rc = gemmul8_scale(stream_, dtype, backend, op_A, op_B, m, n, k, Ab, lda, Bb, ldb, N, fastmode, 0, N, &L, 0, 0);
        if (rc) return rc;
        rc = gemmul8_lowprec_gemm(stream_, dtype, backend, m, n, k, N, 0, N, &L);
        if (rc) return rc;
        rc = gemmul8_crt(stream_, dtype, backend, N, m, n, L.C_mid, L.mp, L.sizeC, L.sftA, L.sftB, alpha, beta, Cb, ldc);
        if (rc) return rc;
    }
    return GEMMUL8_OK;
}

}  // extern "C"
