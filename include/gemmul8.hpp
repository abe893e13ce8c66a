// gemmul8.hpp -- C++ API of the MI355X-native Ozaki-scheme-II GEMM emulator.
//
// Drop-in for the HIP half of the reference header (GEMMul8/include/gemmul8.hpp:19-35,98-151):
// same namespace, enum, template parameters, argument lists and return type, so code written
// against RIKEN-RCCS/GEMMul8 recompiles unchanged.  Differences a caller may notice:
//   * all eight gemm<T,Backend> / gemmLt<T,Backend> combinations are defined (the reference's HIP
//     build leaves gemm<T,FP8> unsupported and gemmLt<T,INT8> undefined, src/gemmul8.cu:136-149);
//     the BLAS / BLASLt handle is only used to fetch the stream -- no vendor GEMM is ever called;
//   * the 4 returned phase timers [scaling, low-prec GEMM, requantise, inverse scaling] (ns) come
//     from HIP events with ONE host synchronisation per call instead of 2N+3; the requantise slot is
//     0 because that phase is fused into the GEMM epilogue.  Set GEMMUL8_ASYNC=1 to skip the
//     synchronisation entirely (timers are then all 0);
//   * invalid num_moduli (outside 2..20) prints a diagnostic on stderr and returns without touching C.
// Everything is a thin layer over the C ABI in gemmul8_c.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hipblas/hipblas.h>
#include <hipblaslt/hipblaslt.h>
#include <hip/hip_complex.h>
#include <cstddef>
#include <vector>

namespace gemmul8 {

enum class Backend { INT8,
                     FP8 };

/***
 * workSize returns the required workspace size in bytes.
 */
template <bool is_Complex = false, Backend backend = Backend::INT8>
size_t workSize(
    size_t m,                         // Number of rows of C
    size_t n,                         // Number of columns of C
    size_t k,                         // Inner dimension <= 2^17
    unsigned num_moduli,              // #moduli, 2 <= num_moduli <= 20 for FP64, 2 <= num_moduli <= 13 for FP32
    bool enable_skip_scalA = false,   // [optional] Reserve extra space for A to allow skip_scalA
    bool enable_skip_scalB = false,   // [optional] Reserve extra space for B to allow skip_scalB
    size_t *workSizeA      = nullptr, // [optional] Output: workspace size used for A8i and sftA
    size_t *workSizeB      = nullptr  // [optional] Output: workspace size used for B8i and sftB
);

/***
 * GEMM emulation on INT8 / FP8 matrix cores (hand-written gfx950 MFMA kernels)
 */
template <typename T, Backend backend = Backend::INT8>
std::vector<double> gemm(
    hipblasHandle_t handle,           // hipBLAS handle: only its stream is used
    hipblasOperation_t op_A,          // HIPBLAS_OP_N, HIPBLAS_OP_T, or HIPBLAS_OP_C
    hipblasOperation_t op_B,          // HIPBLAS_OP_N, HIPBLAS_OP_T, or HIPBLAS_OP_C
    size_t m,                         // Number of rows of C
    size_t n,                         // Number of columns of C
    size_t k,                         // Inner dimension <= 2^17
    const T *alpha,                   // Scaling factor for op(A)*op(B) (host or device pointer)
    const T *const A,                 // 1-D device array of dimensions lda*k (HIPBLAS_OP_N) or lda*m (HIPBLAS_OP_T/C)
    size_t lda,                       // Leading dimension of A
    const T *const B,                 // 1-D device array of dimensions ldb*n (HIPBLAS_OP_N) or ldb*k (HIPBLAS_OP_T/C)
    size_t ldb,                       // Leading dimension of B
    const T *beta,                    // Scaling factor for C (host or device pointer)
    T *const C,                       // 1-D device array of dimensions ldc*n
    size_t ldc,                       // Leading dimension of C
    unsigned num_moduli,              // #moduli, 2 <= num_moduli <= 20 for FP64, 2 <= num_moduli <= 13 for FP32
    bool fastmode,                    // false (accurate mode) or true (fast mode)
    void *const work,                 // Preallocated workspace
    void *const workA      = nullptr, // [optional] Separate workspace for A (if nullptr, uses work)
    void *const workB      = nullptr, // [optional] Separate workspace for B (if nullptr, uses work)
    bool enable_skip_scalA = false,   // [optional] Enables scaling-skip mechanism for A
    bool enable_skip_scalB = false,   // [optional] Enables scaling-skip mechanism for B
    bool skip_scalA        = false,   // [optional] If true, skip preprocessing for A
    bool skip_scalB        = false    // [optional] If true, skip preprocessing for B
);

template <typename T, Backend backend = Backend::INT8>
std::vector<double> gemmLt(
    hipblasLtHandle_t handle,         // hipBLASLt handle (unused: kept for signature compatibility)
    hipblasOperation_t op_A,          // HIPBLAS_OP_N, HIPBLAS_OP_T, or HIPBLAS_OP_C
    hipblasOperation_t op_B,          // HIPBLAS_OP_N, HIPBLAS_OP_T, or HIPBLAS_OP_C
    size_t m,                         // Number of rows of C
    size_t n,                         // Number of columns of C
    size_t k,                         // Inner dimension <= 2^17
    const T *alpha,                   // Scaling factor for op(A)*op(B)
    const T *const A,                 // 1-D device array of dimensions lda*k (HIPBLAS_OP_N) or lda*m (HIPBLAS_OP_T/C)
    size_t lda,                       // Leading dimension of A
    const T *const B,                 // 1-D device array of dimensions ldb*n (HIPBLAS_OP_N) or ldb*k (HIPBLAS_OP_T/C)
    size_t ldb,                       // Leading dimension of B
    const T *beta,                    // Scaling factor for C
    T *const C,                       // 1-D device array of dimensions ldc*n
    size_t ldc,                       // Leading dimension of C
    unsigned num_moduli,              // #moduli, 2 <= num_moduli <= 20 for FP64, 2 <= num_moduli <= 13 for FP32
    bool fastmode,                    // false (accurate mode) or true (fast mode)
    void *const work,                 // Preallocated workspace
    void *const workA      = nullptr, // [optional] Separate workspace for A (if nullptr, uses work)
    void *const workB      = nullptr, // [optional] Separate workspace for B (if nullptr, uses work)
    bool enable_skip_scalA = false,   // [optional] Enables scaling-skip mechanism for A
    bool enable_skip_scalB = false,   // [optional] Enables scaling-skip mechanism for B
    bool skip_scalA        = false,   // [optional] If true, skip preprocessing for A
    bool skip_scalB        = false,   // [optional] If true, skip preprocessing for B
    hipStream_t stream     = 0        // [optional] stream identifier
);

} // namespace gemmul8
