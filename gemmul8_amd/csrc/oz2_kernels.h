// Internal launcher prototypes (host side) of the HIP kernels.  The public boundary is
// include/gemmul8_c.h (C ABI) and include/gemmul8.hpp (C++ API); nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "oz2_device.hpp"
#include "oz2_knobs.hpp"

namespace oz2 {

enum DType { kF32 = 0, kF64 = 1, kC32 = 2, kC64 = 3 };
inline bool is_complex(int dt) { return dt >= 2; }
inline bool is_f32(int dt) { return dt == kF32 || dt == kC32; }
inline size_t padding256(size_t x) { return (x + 255) / 256 * 256; }
inline unsigned planes_of(int backend, int t) { return backend == kINT8 ? 1u : (t < 6 ? 2u : 3u); }
inline unsigned num_mat(int backend, unsigned N) {
    unsigned s = 0;
    for (unsigned t = 0; t < N; ++t) s += planes_of(backend, (int)t);
    return s;
}
ModTable make_mod_table(int backend);

// Batched calls (gemmul8_gemm_batched: the items of a strided batch as ONE set of launches).  Every launcher puts the item index in
// gridDim.z (the persistent GEMM kernels fold it into their tile index); item b works on workspace + b * ws and on operand + b *
// xstride (a launcher argument, 0 by default).  Set by the batched driver around the phase calls of one host thread.
struct BatchCtx {
    unsigned batch = 1;
    size_t ws = 0;                // bytes between the items' workspaces
    size_t sa = 0, sb = 0, sc = 0;  // bytes between the items' A, B, C
};
inline thread_local BatchCtx g_batch;

// ---- INT8 MFMA GEMM (oz2_gemm_i8.hip)
hipError_t launch_gemm_i8_mod(hipStream_t stream, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                              size_t n, int t_begin, int t_end, int8_t* out, size_t ldo, size_t strideO, bool stream_out);
                              // stream_out: the planes are read next by the CRT pass only (non-temporal stores allowed, see the definition)
hipError_t launch_gemm_i8_cplx(hipStream_t stream, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                               size_t n, int t_begin, int t_end, const int8_t* rx, const int8_t* ry, size_t strideR, int8_t* out,
                               size_t ldo, size_t strideO);
hipError_t launch_gemm_i8_max(hipStream_t stream, int nseg, const int8_t* const* A, const int8_t* const* B, size_t kp, size_t m, size_t n,
                              int* rowmax, int* colmax, int mid_seg = 0);

// 128 x 128-tile form of the bound GEMM for products whose 256 x 256 tiles would not fill the chip (oz2_gemm_i8_small.hip)
hipError_t launch_gemm_i8_max_small(hipStream_t stream, int nseg, const int8_t* const* A, const int8_t* const* B, size_t kp, size_t m, size_t n,
                                    int* rowmax, int* colmax, int mid_seg = 0);

// ---- FP8 MFMA GEMM (oz2_gemm_f8.hip)
hipError_t launch_gemm_f8(hipStream_t stream, int which, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                          size_t n, int t_begin, int t_end, int16_t* out, size_t ldo, size_t strideO, const int16_t* r0, const int16_t* r1,
                          size_t strideR, const int16_t* rx = nullptr, const int16_t* ry = nullptr);
// the same residue GEMMs on FP6 (e2m3) panel images of the operand planes (oz2_gemm_f6.hip): same `which`, same outputs, twice the matrix rate
hipError_t launch_gemm_f6(hipStream_t stream, int which, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                          size_t n, int t_begin, int t_end, int16_t* out, size_t ldo, size_t strideO, const int16_t* r0, const int16_t* r1,
                          size_t strideR, const int16_t* rx = nullptr, const int16_t* ry = nullptr);
void set_f8_bound_mode(int mode);  // 0 = engine-safe inflation (default), 1 = the reference's (k+1) * 2^-24
int get_f8_bound_mode();
hipError_t launch_gemm_f8_max(hipStream_t stream, const int8_t* A, const int8_t* B, size_t kp, size_t k, size_t m, size_t n, int* rowmax,
                              int* colmax);
hipError_t launch_gemm_f8_bound_cplx(hipStream_t stream, int stage, const int8_t* A, const int8_t* B, size_t kp, size_t k, size_t m, size_t n,
                                     float* fbuf, size_t ldf, int* rowmax, int* colmax);

// ---- scale / quantise (oz2_scale.hip).  An operand has `rows` logical rows (m for A, n for B) of
// length k; K-major: element (r,kk) at X[r*ld+kk]; row-strided: X[kk*ld+r].  lo planes are
// [rows_pad][kp] int8, zero-filled for kk in [k,kp).
// zero `bytes` (a multiple of 4, 4-byte aligned) with a kernel: hipMemsetAsync nodes misbehave under HIP-graph replay on ROCm 7.2
// (tests/test_gpu_graph.py), and a kernel launch is cheaper on the host than the runtime's memset path
hipError_t launch_zero(hipStream_t stream, void* p, size_t bytes);
// one operand of the accurate mode's extract launches; rows == 0 = absent (skip-scaling)
struct ExtractOperand {
    bool kmajor = false, conj = false;
    size_t rows = 0;
    const void* X = nullptr;
    size_t ld = 0;
    int8_t* lo = nullptr;      // bound plane(s)
    size_t part_stride = 0;    // complex: bytes between the |Re|, |Im|, |Re| - |Im| planes
    int16_t* sft0 = nullptr;   // preliminary shifts (workspace array) ...
    int16_t* sft0_keep = nullptr;  // ... and their scratch copy (what the finalize folded into the quantise launch reads)
    size_t xstride = 0;        // batched launch: bytes between the items' operands
    void* amax = nullptr;      // row-strided operand: `parts` partial row-maxima arrays, pstride elements apart
    unsigned parts = 1;
    size_t pstride = 0;
};
unsigned amax_parts_for(size_t rows, size_t k, size_t max_parts);
hipError_t launch_amax_pair(hipStream_t stream, int dtype, size_t k, const ExtractOperand& A, const ExtractOperand& B);  // row-strided operands only
hipError_t launch_extract_pair(hipStream_t stream, int dtype, int backend, size_t k, size_t kp, const ExtractOperand& A, const ExtractOperand& B, void* zero_p,
                               size_t zero_bytes);
hipError_t launch_shift_finalize(hipStream_t stream, int backend, unsigned N, size_t rowsA, const int* maxA, int16_t* sftA, size_t rowsB,
                                 const int* maxB, int16_t* sftB);
// one operand of the quantise / fast-shift launches; rows == 0 = absent (skip-scaling: the cached planes and shifts are kept)
struct QuantOperand {
    bool kmajor = false, conj = false;
    size_t rows = 0;
    const void* X = nullptr;
    size_t ld = 0;
    int16_t* sft = nullptr;   // fast shift: written; quantise: read (negated final shifts)
    int8_t* lo = nullptr;
    size_t plane_stride = 0, part_stride = 0;
    size_t xstride = 0;       // batched launch: bytes between the items' operands
    size_t f6_rows = 0;       // FP8 backend: > 0 = write FP6 panel images (oz2_gemm_f6.hip) of a plane with this many image rows (A: mp, B: n)
    // accurate mode, quantise only: fin_max != nullptr folds the shift finalize into the launch -- the final shifts are derived from the preliminary
    // ones (fin_sft0: the scratch copy the extract kept) and the bound maxima, and published to `sft` by the launch itself
    const int16_t* fin_sft0 = nullptr;
    const int* fin_max = nullptr;
    float fin_log2P = 0.0f;
};
// FP8 backend: the residue planes are FP6 panel images whenever B's last row block fits its share of the reference's plane size
// (16-row granules at 3/4 byte per element: n >= 45; 64 keeps whole wave tiles); GEMMUL8_FP8_PLANES=e4m3 keeps the e4m3 byte planes
inline bool f6_planes_ok(size_t n) { return n >= 64 && knobs().fp8_planes != 1; }
// both operands in ONE launch each (the A and B halves are independent; at launch-bound sizes every dispatch costs 5-10 us)
hipError_t launch_fast_shift_pair(hipStream_t stream, int dtype, int backend, unsigned N, size_t k, const QuantOperand& A, const QuantOperand& B);
hipError_t launch_quantise_pair(hipStream_t stream, int dtype, int backend, int t_begin, int t_end, size_t k, size_t kp, const QuantOperand& A,
                                const QuantOperand& B);

// ---- CRT accumulation + inverse scaling (oz2_crt.hip)
hipError_t launch_crt(hipStream_t stream, int dtype, int backend, unsigned N, size_t m, size_t n, const void* Cmid, size_t ld_mid,
                      size_t plane_stride, const int16_t* sftA, const int16_t* sftB, const void* alpha, const void* beta,
                      bool scalars_on_device, void* C, size_t ldc);

hipError_t launch_row_bias(hipStream_t stream, int dtype, size_t m, size_t n, void* D, size_t ldd, const void* bias);
hipError_t launch_add_f64(hipStream_t stream, double* dst, const double* src, size_t count);  // dst += src (16-byte aligned arrays)

// multi-GPU exchange variant (A): per-rank FP64 partial CRT sums + the finish on the summed partials (oz2_crt.hip)
hipError_t launch_crt_partial(hipStream_t stream, int dtype, int backend, unsigned N, unsigned t_begin, unsigned t_end, size_t m, size_t n,
                              const void* Cmid, size_t ld_mid, size_t plane_stride, double* out_hi, double* out_lo, size_t ld_out,
                              size_t col_block, size_t block_stride);
hipError_t launch_crt_finish(hipStream_t stream, int dtype, int backend, unsigned N, size_t m, size_t n, const double* in_hi,
                             const double* in_lo, size_t ld_in, const int16_t* sftA, const int16_t* sftB, const void* alpha,
                             const void* beta, bool scalars_on_device, void* C, size_t ldc);

}  // namespace oz2
