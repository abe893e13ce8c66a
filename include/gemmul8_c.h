/*
 * gemmul8_c.h -- C ABI of the MI355X-native Ozaki-scheme-II GEMM emulator (libgemmul8.so).
 *
 * This is the drop-in boundary for the hot path of RIKEN-RCCS/GEMMul8: plain pointers and sizes,
 * no C++ / torch types.  The C++ template API of include/gemmul8.hpp (gemmul8::workSize / gemm /
 * gemmLt, reference GEMMul8/include/gemmul8.hpp:25-151) and the LD_PRELOAD hipBLAS hook
 * (reference GEMMul8/src/hook.cu:846-1055) are thin layers over these entry points.
 *
 * Conventions: column-major BLAS semantics, all matrix pointers are DEVICE pointers, `stream` is a
 * hipStream_t passed as void*.  Every function returns 0 on success, a negative GEMMUL8_E_* code
 * on bad arguments, or a positive hipError_t if the HIP runtime failed.
 */
#ifndef GEMMUL8_C_H
#define GEMMUL8_C_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define GEMMUL8_API __attribute__((visibility("default")))
#else
#define GEMMUL8_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* element type of A, B, C (reference: template parameter T of gemmul8::gemm, gemmul8.hpp:98) */
enum { GEMMUL8_S = 0, GEMMUL8_D = 1, GEMMUL8_C = 2, GEMMUL8_Z = 3 };
/* low-precision engine (reference: enum class Backend, gemmul8.hpp:19-20) */
enum { GEMMUL8_INT8 = 0, GEMMUL8_FP8 = 1 };
/* op(X): 0/1/2, or the hipblasOperation_t values 111/112/113 (both accepted) */
enum { GEMMUL8_OP_N = 0, GEMMUL8_OP_T = 1, GEMMUL8_OP_C = 2 };

enum {
    GEMMUL8_OK = 0,
    GEMMUL8_E_NUM_MODULI = -1, /* num_moduli outside 2..20 */
    GEMMUL8_E_ARG = -2,        /* null pointer / bad enum / k > 2^17 */
    GEMMUL8_E_UNSUPPORTED = -3, /* combination not built */
    GEMMUL8_E_INTERNAL = -4     /* resource failure that is NOT a property of the arguments (allocation, transport): a multi-rank caller
                                  must not treat it as "every rank declines" */
};

/* Workspace bytes; same formula as the reference so callers' allocations stay valid.
 * Replaces gemmul8::workSize<is_Complex,backend> (include/gemmul8.hpp:25-35,
 * src/gemmul8_real.hpp:8-47, src/gemmul8_complex.hpp:8-47). */
GEMMUL8_API size_t gemmul8_work_size(int is_complex, int backend, size_t m, size_t n, size_t k, unsigned num_moduli,
                         int enable_skip_scalA, int enable_skip_scalB, size_t *workSizeA, size_t *workSizeB);

/* Whole emulated GEMM: C = alpha*op(A)*op(B) + beta*C.
 * Replaces gemmul8::gemm<T,backend> / gemmLt<T,backend> (include/gemmul8.hpp:98-151,
 * src/gemmul8_real.hpp:52-211, src/gemmul8_complex.hpp:52-226).
 * alpha/beta may be host or device pointers (detected like inverse_scaling_real.hpp:211-213).
 * timers_ns: NULL = fully asynchronous; else 4 doubles [scaling, low-prec GEMM, requantise (always
 * 0: fused into the GEMM epilogue), inverse scaling] in ns, measured with events (one host sync). */
GEMMUL8_API int gemmul8_gemm(void *stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k,
                 const void *alpha, const void *A, size_t lda, const void *B, size_t ldb, const void *beta, void *C,
                 size_t ldc, unsigned num_moduli, int fastmode, void *work, void *workA, void *workB,
                 int enable_skip_scalA, int enable_skip_scalB, int skip_scalA, int skip_scalB, double *timers_ns);

/* Where the intermediates of the last/next gemmul8_gemm call live inside the workspaces (device
 * pointers; same carving as src/gemmul8_real.hpp:95-107).  Used by the parity tests and by the
 * moduli-sharded multi-GPU driver. */
typedef struct gemmul8_layout {
    size_t kp, mp;             /* padded k and m (multiples of 256) */
    size_t num_mat;            /* low-precision planes per part */
    size_t parts;              /* 1 (real) or 3 (complex: Re, Im, Re+Im) */
    size_t sizeA, sizeB, sizeC;/* elements per plane: kp*mp, kp*n, mp*n */
    void *A_lo, *B_lo;         /* plane (part,q) at X_lo + (part*num_mat_total + q)*sizeX, num_mat_total = num_mat (+1 if skip enabled) ... see part_strideX */
    size_t part_strideA, part_strideB; /* bytes between Re/Im/Re+Im plane sets */
    void *A_bound, *B_bound;   /* accurate-mode 7-bit bound planes (alias plane 0 unless skip enabled) */
    int16_t *sftA, *sftB;      /* negated shift exponents */
    void *C_mid;               /* N residue planes [n][mp] (complex: interleaved re,im) */
    void *scratch;             /* the reference's C_hi region (free for row/col maxima etc.) */
    size_t scratch_bytes;
    size_t lo_format;          /* encoding of the A_lo / B_lo residue planes: 0 = one byte per element (int8 residues / e4m3 pieces), row r at
                                  r * kp; 1 = FP6 panel images of the FP8 backend's pieces (csrc/oz2_gemm_f6.hip; n >= 64, skip-scaling
                                  not enabled -- cached planes may meet a partner of another shape --, unless GEMMUL8_FP8_PLANES=e4m3): same plane offsets and strides, 3/4 of each plane used.  The bound planes stay bytes. */
} gemmul8_layout;

GEMMUL8_API int gemmul8_get_layout(int dtype, int backend, size_t m, size_t n, size_t k, unsigned num_moduli, void *work, void *workA,
                       void *workB, int enable_skip_scalA, int enable_skip_scalB, gemmul8_layout *out);

/* ABI guard.  gemmul8_get_layout fills sizeof(gemmul8_layout) bytes of the CALLER's struct: a binding compiled against an older header
 * (the struct grew by lo_format in version 5) would be overrun.  GEMMUL8_ABI_VERSION is bumped whenever a struct of this header grows or a
 * signature changes; a binding checks gemmul8_abi_version() == GEMMUL8_ABI_VERSION (C / C++) or gemmul8_layout_bytes() against the size of
 * its own mirror of the struct (ctypes: gemmul8_amd/__init__.py does) before it calls anything else. */
#define GEMMUL8_ABI_VERSION 7   /* 7: gemmul8_add_f64; gemmul8_dist_engine grew by add_f64 */
GEMMUL8_API int gemmul8_abi_version(void);
GEMMUL8_API size_t gemmul8_layout_bytes(void);

/* ---- phase-level entry points (one per kernel family) -------------------------------------- */

/* Shifts + residue planes of both operands for moduli [t_begin, t_end).  fastmode=0 runs the
 * accurate path (extract, bound GEMM with max epilogue, shift).  Replaces fast::scaling /
 * accu::scaling (src/scaling_fast_real.hpp:222-268, src/scaling_accu_real.hpp:380-457 and the
 * complex variants). */
GEMMUL8_API int gemmul8_scale(void *stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void *A,
                  size_t lda, const void *B, size_t ldb, unsigned num_moduli, int fastmode, unsigned t_begin,
                  unsigned t_end, const gemmul8_layout *L, int skipA, int skipB);

/* The two halves of gemmul8_scale, split where a multi-GPU run must exchange data:
 *  _bounds (accurate mode only): 7-bit bound planes of A and B, then the bound GEMM restricted to
 *          columns [col_begin, col_end) of op(B); leaves int32 rowmax[mp] at L->scratch and colmax[pad256(n)]
 *          right behind it (zero outside the computed columns) -- ranks combine them with one
 *          all-reduce(MAX) over the (mp + pad256(n)) int32 words;
 *  _finish: final shifts (from the maxima, or the fast-mode norms) and the residue planes. */
GEMMUL8_API int gemmul8_scale_bounds(void *stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k,
                                     const void *A, size_t lda, const void *B, size_t ldb, unsigned num_moduli, size_t col_begin,
                                     size_t col_end, const gemmul8_layout *L, int skipA, int skipB);
GEMMUL8_API int gemmul8_scale_finish(void *stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k,
                                     const void *A, size_t lda, const void *B, size_t ldb, unsigned num_moduli, int fastmode,
                                     unsigned t_begin, unsigned t_end, const gemmul8_layout *L, int skipA, int skipB);

/* Low-precision GEMMs of moduli [t_begin, t_end) with the requantise epilogue: fills C_mid planes.
 * Replaces gemm_low_prec_* + conv_hi2mid (src/matmult.hpp:120-389, src/conv_hi2mid_real.hpp,
 * src/conv_hi2mid_complex.hpp; loop at src/gemmul8_real.hpp:144-191).  k is the inner dimension the layout was made for: a k whose
 * padding differs from L->kp is GEMMUL8_E_ARG (the FP8 backend chooses its exact-accumulation form from it). */
GEMMUL8_API int gemmul8_lowprec_gemm(void *stream, int dtype, int backend, size_t m, size_t n, size_t k, unsigned num_moduli,
                         unsigned t_begin, unsigned t_end, const gemmul8_layout *L);

/* CRT accumulation + inverse scaling + axpby on an arbitrary column block: C_mid planes given by
 * pointer/stride so that a GPU can finish the columns it owns after an exchange of residue planes.
 * Replaces inverse_scaling (src/inverse_scaling_real.hpp:242-278, inverse_scaling_complex.hpp). */
GEMMUL8_API int gemmul8_crt(void *stream, int dtype, int backend, unsigned num_moduli, size_t m, size_t n, const void *C_mid,
                size_t ld_mid, size_t plane_stride, const int16_t *sftA, const int16_t *sftB, const void *alpha,
                const void *beta, void *C, size_t ldc);

/* A strided batch of GEMMs (same shape, alpha / beta shared; strides in ELEMENTS of the matrix type, as in
 * hipblas{S,D,C,Z}gemmStridedBatched) as ONE set of launches: every kernel of the pipeline takes the item from gridDim.z, the
 * persistent GEMM kernels run over the items' residue planes in one launch.  `work` holds gemmul8_work_size_batched bytes (the items'
 * workspaces are consecutive blocks of gemmul8_batched_item_bytes).  Both backends (FP8: k <= 65536 as in gemmul8_gemm).
 * Bit-identical to per-item gemmul8_gemm calls.  No counterpart in the reference (it hooks no batched entry point). */
GEMMUL8_API size_t gemmul8_batched_item_bytes(int is_complex, int backend, size_t m, size_t n, size_t k, unsigned num_moduli);
GEMMUL8_API size_t gemmul8_work_size_batched(int is_complex, int backend, size_t m, size_t n, size_t k, unsigned num_moduli, size_t batch);
GEMMUL8_API int gemmul8_gemm_batched(void *stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k,
                         const void *alpha, const void *A, size_t lda, long long strideA, const void *B, size_t ldb,
                         long long strideB, const void *beta, void *C, size_t ldc, long long strideC, size_t batch,
                         unsigned num_moduli, int fastmode, void *work);

/* D(i, j) += bias[i] for a column-major m x n real matrix: the broadcast bias of a hipblasLtMatmul BIAS epilogue, applied by the hook after
 * the emulated GEMM (one more rounding than the vendor's fused form).  S / D only.  No counterpart in the reference. */
GEMMUL8_API int gemmul8_add_row_bias(void *stream, int dtype, size_t m, size_t n, void *D, size_t ldd, const void *bias);

/* Multi-GPU exchange variant (A) of the moduli-sharded plan (include/gemmul8_dist.h): the rank's FP64 partial CRT sums over its
 * moduli [t_begin, t_end) -- C_mid points at plane t_begin -- as two double planes (hi: error-free chain, lo: rounded chain;
 * same FMAs and order as gemmul8_crt restricted to those moduli), written in column blocks of col_block columns, block b at
 * out + b * block_stride doubles, column c of a block at c * ld_out (the reduce-scatter unit of rank b); and the finish on the
 * SUMMED partials: mod-P reduction, scalbn, axpby as in gemmul8_crt.  No counterpart in the reference (it has no multi-GPU code);
 * the arithmetic is src/inverse_scaling_real.hpp:8-89 split at the accumulator. */
GEMMUL8_API int gemmul8_crt_partial(void *stream, int dtype, int backend, unsigned num_moduli, unsigned t_begin, unsigned t_end, size_t m,
                        size_t n, const void *C_mid, size_t ld_mid, size_t plane_stride, double *out_hi, double *out_lo,
                        size_t ld_out, size_t col_block, size_t block_stride);
GEMMUL8_API int gemmul8_crt_finish(void *stream, int dtype, int backend, unsigned num_moduli, size_t m, size_t n, const double *in_hi,
                       const double *in_lo, size_t ld_in, const int16_t *sftA, const int16_t *sftB, const void *alpha,
                       const void *beta, void *C, size_t ldc);

/* dst[i] += src[i], i < count, on FP64 arrays (16-byte aligned): the running sum of reduced partial planes when the fp64sum plan of
 * include/gemmul8_dist.h exchanges its partial sums in moduli groups (GEMMUL8_DIST_FP64_GROUPS).  No counterpart in the reference. */
GEMMUL8_API int gemmul8_add_f64(void *stream, double *dst, const double *src, size_t count);

/* FP8 backend, accurate mode: inflation of the bound GEMM's sums before the row / column maxima are taken.
 *   0 (default)  ku = 7 * 2^-13 + 4 (k+1) * 2^-24: covers how gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 accumulates (products aligned
 *                to the largest of a group of 8 with 13 bits below it, truncating) -- with the reference's formula the bound can come
 *                out up to 1.2e-3 LOW, which can raise a shift by one and wrap the CRT;
 *                plus an absolute 7 kp 2^-14 (bound-plane units): a group of 8 products is aligned to the largest sum of the operands' exponent
 *                FIELDS, and an e4m3 subnormal carries the field of 2^-6 -- beside such a reference only 10 bits below the true largest product survive.
 *                Complex types: the mixed-sign product C0 = (|Ar|-|Ai|)(|Br|-|Bi|) of the bound of |Re C| is inflated by
 *                ku (|C0| + 2 (|Ar||Bi| + |Ai||Br|)) -- the engine's error on C0 scales with the magnitudes of its terms;
 *   1            ku = (k+1) * 2^-24, the reference's IEEE-FP32 summation bound (GEMMul8/src/find_max.hpp:82-96), and its ku C0 for complex;
 *   2            mode 0's ku with the reference's complex combination (the round-3 default; kept for tests/test_gpu_fp8_bound.py).
 * Process-wide; returns the previous mode (>= 0) or GEMMUL8_E_ARG.  The hook sets mode 1 when GEMMUL8_FP8_BOUND=reference. */
GEMMUL8_API int gemmul8_set_fp8_bound_mode(int mode);

/* What the hook does with a GEMM of this shape under the CURRENT value of GEMMUL8_MIN_FLOPS (oz2_hook.cpp below_floor): 1 = emulated,
 * 0 = handed to the native routine, GEMMUL8_E_ARG on bad arguments.  GEMMUL8_MIN_FLOPS unset, empty or 0: every selected call is emulated
 * (the reference's behaviour: its hook has no floor) -- always 1; "auto": the fitted cost model (tools/fit_floor.py,
 * profiles/sweeps/r04b_floor_scan_*.csv) decides per shape; a number: a plain floor on 2 m n k.  batch = items of a strided batch (1 for a
 * plain call).  No counterpart in the reference. */
GEMMUL8_API int gemmul8_hook_would_emulate(int dtype, int backend, size_t m, size_t n, size_t k, unsigned num_moduli, int fastmode,
                                           size_t batch);

/* 1 if `version` (a rocblas_get_version_string result) is a rocBLAS release the opt-in interposition of rocblas_internal_gemm_template
 * (GEMMUL8_HOOK_ROCBLAS=1; an internal, unversioned rocBLAS symbol) was tested with, else 0: with any other rocBLAS that symbol is passed
 * through untouched.  No counterpart in the reference (it hooks the hipBLAS names only). */
GEMMUL8_API int gemmul8_hook_rocblas_version_tested(const char *version);

/* Testing / A-B knobs (GEMMUL8_EPI_NT, _BOUND_TILE, _CPLX_BOUND_LAUNCHES, _CPLX_CHUNK, _CRT_KERNEL, _MAP_COLBLOCK; INTEGRATION.md
 * "Testing switches"): every one selects between bit-identical code paths.  They are parsed from the environment ONCE, at the first
 * launch; a test harness that changes the environment inside one process calls this afterwards.  No counterpart in the reference. */
GEMMUL8_API void gemmul8_reload_knobs(void);

/* Library identification (build arch, version) */
GEMMUL8_API const char *gemmul8_version(void);

#ifdef __cplusplus
}
#endif
#endif
