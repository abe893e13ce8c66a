// Explicit specialisations of the C++ API (include/gemmul8.hpp) over the C ABI.
// Replaces GEMMul8/src/gemmul8.cu:95-157.  Exported (Itanium-mangled) names match the reference's
// HIP object and add the four gemm<T,FP8> / gemmLt<T,INT8> it leaves undefined.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#pragma GCC visibility push(default)
#include "../../include/gemmul8.hpp"
#pragma GCC visibility pop
#include "../../include/gemmul8_c.h"

namespace {

template <typename T> struct TypeCode;
template <> struct TypeCode<float> { static constexpr int v = GEMMUL8_S; };
template <> struct TypeCode<double> { static constexpr int v = GEMMUL8_D; };
template <> struct TypeCode<hipFloatComplex> { static constexpr int v = GEMMUL8_C; };
template <> struct TypeCode<hipDoubleComplex> { static constexpr int v = GEMMUL8_Z; };

// hipblasGetStream is looked up at run time so that libgemmul8.so carries no link-time dependency
// on hipBLAS (it never calls a BLAS routine; the handle is only a stream carrier).
hipStream_t stream_of(hipblasHandle_t handle) {
    if (!handle) return nullptr;
    using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipStream_t*);
    static Fn fn = [] {
        void* p = dlsym(RTLD_DEFAULT, "hipblasGetStream");
        if (!p) {
            void* h = dlopen("libhipblas.so.3", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libhipblas.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) p = dlsym(h, "hipblasGetStream");
        }
        return reinterpret_cast<Fn>(p);
    }();
    hipStream_t s = nullptr;
    if (!fn || fn(handle, &s) != HIPBLAS_STATUS_SUCCESS) {
        std::fprintf(stderr, "[GEMMUL8] hipblasGetStream unavailable: using the default stream\n");
        return nullptr;
    }
    return s;
}

bool async_mode() {
    static const bool v = [] {
        const char* s = std::getenv("GEMMUL8_ASYNC");
        return s && std::strcmp(s, "1") == 0;
    }();
    return v;
}

template <typename T>
std::vector<double> run(hipStream_t stream, int backend, hipblasOperation_t op_A, hipblasOperation_t op_B, size_t m, size_t n, size_t k,
                        const T* alpha, const T* A, size_t lda, const T* B, size_t ldb, const T* beta, T* C, size_t ldc,
                        unsigned num_moduli, bool fastmode, void* work, void* workA, void* workB, bool enA, bool enB, bool skA, bool skB) {
    std::vector<double> timer(4, 0.0);
    const int rc = gemmul8_gemm(stream, TypeCode<T>::v, backend, (int)op_A, (int)op_B, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc,
                                num_moduli, fastmode ? 1 : 0, work, workA, workB, enA, enB, skA, skB, async_mode() ? nullptr : timer.data());
    if (rc != 0) {
        // The reference's API has no error channel (its gemm validates nothing, include/gemmul8.hpp:98-151).  Here a failed call
        // leaves C untouched and says so; GEMMUL8_ABORT_ON_ERROR=1 turns it into an abort for callers that cannot check stderr.
        std::fprintf(stderr, "[GEMMUL8] gemm failed with status %d (m=%zu n=%zu k=%zu num_moduli=%u): C was not computed\n", rc, m, n, k, num_moduli);
        const char* s = std::getenv("GEMMUL8_ABORT_ON_ERROR");
        if (s && std::strcmp(s, "1") == 0) std::abort();
    }
    return timer;
}

}  // namespace

#pragma GCC visibility push(default)
namespace gemmul8 {

#define OZ2_WS(CPLX, BE)                                                                                                  \
    template <> size_t workSize<CPLX, Backend::BE>(size_t m, size_t n, size_t k, unsigned num_moduli, bool enA, bool enB, \
                                                   size_t * wA, size_t * wB) {                                            \
        return gemmul8_work_size(CPLX, (int)Backend::BE, m, n, k, num_moduli, enA, enB, wA, wB);                          \
    }
OZ2_WS(false, INT8)
OZ2_WS(false, FP8)
OZ2_WS(true, INT8)
OZ2_WS(true, FP8)
#undef OZ2_WS

#define OZ2_GEMM(T, BE)                                                                                                                 \
    template <>                                                                                                                         \
    std::vector<double> gemm<T, Backend::BE>(hipblasHandle_t handle, hipblasOperation_t op_A, hipblasOperation_t op_B, size_t m,       \
                                             size_t n, size_t k, const T* alpha, const T* const A, size_t lda, const T* const B,       \
                                             size_t ldb, const T* beta, T* const C, size_t ldc, unsigned num_moduli, bool fastmode,    \
                                             void* const work, void* const workA, void* const workB, bool enA, bool enB, bool skA,     \
                                             bool skB) {                                                                               \
        return run<T>(stream_of(handle), (int)Backend::BE, op_A, op_B, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, num_moduli,       \
                      fastmode, work, workA, workB, enA, enB, skA, skB);                                                               \
    }                                                                                                                                   \
    template <>                                                                                                                         \
    std::vector<double> gemmLt<T, Backend::BE>(hipblasLtHandle_t, hipblasOperation_t op_A, hipblasOperation_t op_B, size_t m, size_t n, \
                                               size_t k, const T* alpha, const T* const A, size_t lda, const T* const B, size_t ldb,   \
                                               const T* beta, T* const C, size_t ldc, unsigned num_moduli, bool fastmode,              \
                                               void* const work, void* const workA, void* const workB, bool enA, bool enB, bool skA,   \
                                               bool skB, hipStream_t stream) {                                                         \
        return run<T>(stream, (int)Backend::BE, op_A, op_B, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, num_moduli, fastmode, work,  \
                      workA, workB, enA, enB, skA, skB);                                                                               \
    }
OZ2_GEMM(float, INT8)
OZ2_GEMM(double, INT8)
OZ2_GEMM(hipFloatComplex, INT8)
OZ2_GEMM(hipDoubleComplex, INT8)
OZ2_GEMM(float, FP8)
OZ2_GEMM(double, FP8)
OZ2_GEMM(hipFloatComplex, FP8)
OZ2_GEMM(hipDoubleComplex, FP8)
#undef OZ2_GEMM

}  // namespace gemmul8
#pragma GCC visibility pop
