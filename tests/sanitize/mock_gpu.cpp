// TEST INFRASTRUCTURE for tests/test_sanitizers.py: a host-memory stand-in for what oz2_hook.cpp calls -- the HIP runtime subset it
// uses, the "real" hipBLAS / hipBLASLt routines it forwards to (found through dlsym(RTLD_NEXT)), and the libgemmul8 C ABI -- so that
// the hook's host logic (per-handle state, grow-only stream-ordered buffers, skip-scaling cache, stream switches, plan cache,
// descriptor decoding) can run under AddressSanitizer / UBSan on a machine without a GPU.  gemmul8_gemm here WRITES every byte of
// the three workspaces at the size gemmul8_work_size reports, so an under-sized or freed buffer is a sanitizer error.
#include <hip/hip_runtime.h>
#include <hipblas/hipblas.h>
#include <hipblaslt/hipblaslt.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "../../include/gemmul8_dist.h"

#define EXPORT extern "C" __attribute__((visibility("default")))

std::atomic<long> g_native_calls{0}, g_emulated_calls{0}, g_live_allocs{0};
EXPORT long mock_native_calls() { return g_native_calls.load(); }
EXPORT long mock_emulated_calls() { return g_emulated_calls.load(); }
EXPORT long mock_live_allocs() { return g_live_allocs.load(); }

// ---- HIP runtime subset
EXPORT hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { *p = std::malloc(n ? n : 1); ++g_live_allocs; return *p ? hipSuccess : hipErrorOutOfMemory; }
EXPORT hipError_t hipFreeAsync(void* p, hipStream_t) { std::free(p); --g_live_allocs; return hipSuccess; }
EXPORT hipError_t hipFree(void* p) { std::free(p); --g_live_allocs; return hipSuccess; }
EXPORT const char* hipGetErrorString(hipError_t) { return "mock"; }
EXPORT hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
EXPORT hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { std::memset((void*)e, 1, 8); return hipSuccess; }
EXPORT hipError_t hipEventSynchronize(hipEvent_t e) { return e ? hipSuccess : hipErrorInvalidValue; }
EXPORT hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return *(volatile char*)e == 1 ? hipSuccess : hipErrorInvalidValue; }
EXPORT hipError_t hipEventDestroy(hipEvent_t e) { std::free((void*)e); return hipSuccess; }
EXPORT hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
EXPORT hipError_t hipStreamDestroy(hipStream_t s) { std::free((void*)s); return hipSuccess; }
EXPORT hipError_t hipDeviceSynchronize() { return hipSuccess; }
EXPORT hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
EXPORT hipError_t hipGetLastError() { return hipSuccess; }
EXPORT hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) std::memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}

// ---- "real" hipBLAS: handles are small heap objects carrying a stream
struct MockHandle { hipStream_t stream; };
EXPORT hipblasStatus_t mock_create(hipblasHandle_t* h) { *h = (hipblasHandle_t) new MockHandle{nullptr}; return HIPBLAS_STATUS_SUCCESS; }
EXPORT hipblasStatus_t mock_set_stream(hipblasHandle_t h, hipStream_t s) { ((MockHandle*)h)->stream = s; return HIPBLAS_STATUS_SUCCESS; }
EXPORT hipblasStatus_t hipblasGetStream(hipblasHandle_t h, hipStream_t* s) { *s = ((MockHandle*)h)->stream; return HIPBLAS_STATUS_SUCCESS; }
EXPORT hipblasStatus_t hipblasDestroy(hipblasHandle_t h) { delete (MockHandle*)h; return HIPBLAS_STATUS_SUCCESS; }
#define NATIVE_GEMM(NAME, T, I)                                                                                              \
    EXPORT hipblasStatus_t NAME(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, I m, I n, I, const T*, const T*, I, const T*, I, \
                                const T*, T* C, I ldc) {                                                                     \
        ++g_native_calls;                                                                                                    \
        for (I j = 0; j < n; ++j) std::memset(C + (size_t)j * ldc, 0, sizeof(T) * m);                                        \
        return HIPBLAS_STATUS_SUCCESS;                                                                                       \
    }
NATIVE_GEMM(hipblasSgemm, float, int)
NATIVE_GEMM(hipblasDgemm, double, int)
NATIVE_GEMM(hipblasCgemm, hipComplex, int)
NATIVE_GEMM(hipblasZgemm, hipDoubleComplex, int)
NATIVE_GEMM(hipblasDgemm_64, double, int64_t)
EXPORT hipblasStatus_t hipblasGemmEx(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int, const void*, const void*, hipDataType, int,
                                     const void*, hipDataType, int, const void*, void*, hipDataType, int, hipblasComputeType_t, hipblasGemmAlgo_t) {
    ++g_native_calls;
    return HIPBLAS_STATUS_SUCCESS;
}
EXPORT hipblasStatus_t hipblasDgemmStridedBatched(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int, const double*, const double*, int,
                                                  long long, const double*, int, long long, const double*, double*, int, long long, int) {
    ++g_native_calls;
    return HIPBLAS_STATUS_SUCCESS;
}

// ---- "real" hipBLASLt: descriptors are plain structs
struct MockDesc { int32_t ta, tb; uint32_t epi; const void* bias = nullptr; int32_t bias_type = -1; };
struct MockLayout { uint32_t type; int32_t order, batch; uint64_t rows, cols; int64_t ld; };
EXPORT void* mock_lt_desc(int ta, int tb, unsigned epi) { return new MockDesc{ta, tb, epi}; }
EXPORT void* mock_lt_desc_bias(int ta, int tb, const void* bias, int bias_type) { return new MockDesc{ta, tb, (unsigned)HIPBLASLT_EPILOGUE_BIAS, bias, bias_type}; }
EXPORT void mock_lt_free_desc(void* p) { delete (MockDesc*)p; }
EXPORT hipblasStatus_t hipblasLtMatmulDescGetAttribute(hipblasLtMatmulDesc_t d, hipblasLtMatmulDescAttributes_t a, void* buf, size_t n, size_t* w) {
    const MockDesc* D = (const MockDesc*)d;
    if (w) *w = n;
    switch (a) {
    case HIPBLASLT_MATMUL_DESC_TRANSA: std::memcpy(buf, &D->ta, 4); return HIPBLAS_STATUS_SUCCESS;
    case HIPBLASLT_MATMUL_DESC_TRANSB: std::memcpy(buf, &D->tb, 4); return HIPBLAS_STATUS_SUCCESS;
    case HIPBLASLT_MATMUL_DESC_EPILOGUE: std::memcpy(buf, &D->epi, 4); return HIPBLAS_STATUS_SUCCESS;
    case HIPBLASLT_MATMUL_DESC_BIAS_POINTER: std::memcpy(buf, &D->bias, sizeof(void*)); return HIPBLAS_STATUS_SUCCESS;
    case HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE: std::memcpy(buf, &D->bias_type, 4); return HIPBLAS_STATUS_SUCCESS;
    default: std::memset(buf, 0, n); return HIPBLAS_STATUS_SUCCESS;  // pointers NULL, pointer mode host
    }
}
EXPORT hipblasStatus_t hipblasLtMatrixLayoutCreate(hipblasLtMatrixLayout_t* l, hipDataType type, uint64_t rows, uint64_t cols, int64_t ld) {
    *l = (hipblasLtMatrixLayout_t) new MockLayout{(uint32_t)type, 0, 1, rows, cols, ld};
    return HIPBLAS_STATUS_SUCCESS;
}
EXPORT hipblasStatus_t hipblasLtMatrixLayoutSetAttribute(hipblasLtMatrixLayout_t l, hipblasLtMatrixLayoutAttribute_t a, const void* buf, size_t) {
    if (a == HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT) std::memcpy(&((MockLayout*)l)->batch, buf, 4);
    return HIPBLAS_STATUS_SUCCESS;
}
EXPORT hipblasStatus_t hipblasLtMatrixLayoutDestroy(const hipblasLtMatrixLayout_t l) { delete (MockLayout*)l; return HIPBLAS_STATUS_SUCCESS; }
EXPORT hipblasStatus_t hipblasLtDestroy(const hipblasLtHandle_t) { return HIPBLAS_STATUS_SUCCESS; }
EXPORT hipblasStatus_t hipblasLtMatmul(hipblasLtHandle_t, hipblasLtMatmulDesc_t, const void*, const void*, hipblasLtMatrixLayout_t, const void*,
                                       hipblasLtMatrixLayout_t, const void*, const void*, hipblasLtMatrixLayout_t, void*, hipblasLtMatrixLayout_t,
                                       const hipblasLtMatmulAlgo_t*, void*, size_t, hipStream_t) {
    ++g_native_calls;
    return HIPBLAS_STATUS_SUCCESS;
}

// ---- libgemmul8 C ABI: sizes from a simple formula, the "GEMM" touches every workspace byte and every element of C
static size_t pad(size_t x) { return (x + 255) / 256 * 256; }
EXPORT size_t gemmul8_work_size(int cplx, int backend, size_t m, size_t n, size_t k, unsigned N, int enA, int enB, size_t* wA, size_t* wB) {
    const size_t parts = cplx ? 3 : 1, nm = backend ? 3 * N : N;
    const size_t a = 255 + pad(k) * pad(m) * (nm + (enA ? 1 : 0)) * parts + 2 * pad(m), b = 255 + pad(k) * n * (nm + (enB ? 1 : 0)) * parts + 2 * pad(n);
    const size_t c = 255 + pad(m) * n * N + (1u << 20);
    if (wA) *wA = a;
    if (wB) *wB = b;
    return a + b + c;
}
EXPORT int gemmul8_gemm(void*, int dtype, int backend, int, int, size_t m, size_t n, size_t k, const void* alpha, const void* A, size_t, const void* B, size_t,
                        const void* beta, void* C, size_t ldc, unsigned N, int, void* work, void* workA, void* workB, int enA, int enB, int skA, int skB, double*) {
    if (!alpha || !beta || !A || !B || !C || !work) return GEMMUL8_E_ARG;
    if (k > (size_t(1) << 17)) return GEMMUL8_E_ARG;
    size_t wa = 0, wb = 0;
    const size_t tot = gemmul8_work_size(dtype >= 2, backend, m, n, k, N, enA, enB, &wa, &wb);
    if (workA && !skA) std::memset(workA, 0x11, wa);
    if (workB && !skB) std::memset(workB, 0x22, wb);
    if (workA && skA && *(volatile unsigned char*)workA != 0x11) return 7;  // a skipped operand's planes must still be there
    if (workB && skB && *(volatile unsigned char*)workB != 0x22) return 7;
    std::memset(work, 0x33, workA && workB ? tot - wa - wb : tot);
    const size_t es = dtype == 0 ? 4 : dtype == 3 ? 16 : 8;
    for (size_t j = 0; j < n; ++j) std::memset((char*)C + j * ldc * es, 0x44, m * es);
    ++g_emulated_calls;
    return GEMMUL8_OK;
}
EXPORT size_t gemmul8_work_size_batched(int cplx, int backend, size_t m, size_t n, size_t k, unsigned N, size_t batch) {
    return pad(gemmul8_work_size(cplx, backend, m, n, k, N, 0, 0, nullptr, nullptr)) * batch + 256;
}
EXPORT int gemmul8_gemm_batched(void*, int dtype, int backend, int, int, size_t m, size_t n, size_t k, const void* alpha, const void* A, size_t, long long,
                                const void* B, size_t, long long, const void* beta, void* C, size_t ldc, long long sc, size_t batch, unsigned N, int, void* work) {
    if (!alpha || !beta || !A || !B || !C || !work) return GEMMUL8_E_ARG;
    if (backend != 0) return GEMMUL8_E_UNSUPPORTED;
    std::memset(work, 0x55, gemmul8_work_size_batched(dtype >= 2, backend, m, n, k, N, batch));  // every byte of the batched workspace
    const size_t es = dtype == 0 ? 4 : dtype == 3 ? 16 : 8;
    for (size_t b = 0; b < batch; ++b)
        for (size_t j = 0; j < n; ++j) std::memset((char*)C + ((long long)b * sc + (long long)(j * ldc)) * (long long)es, 0x44, m * es);
    g_emulated_calls += (long)batch;
    return GEMMUL8_OK;
}
EXPORT int gemmul8_add_row_bias(void*, int dtype, size_t m, size_t n, void* D, size_t ldd, const void* bias) {
    if (!D || !bias) return GEMMUL8_E_ARG;
    const size_t es = dtype == 0 ? 4 : 8;
    volatile unsigned char sink = 0;
    for (size_t i = 0; i < m * es; ++i) sink = sink + ((const unsigned char*)bias)[i];  // the whole bias vector must be readable
    for (size_t j = 0; j < n; ++j) std::memset((char*)D + j * ldd * es, 0x66, m * es);
    return GEMMUL8_OK;
}
EXPORT int gemmul8_comm_rccl_from_env(gemmul8_comm**) { return GEMMUL8_E_UNSUPPORTED; }
EXPORT int gemmul8_dist_create(const gemmul8_comm*, const gemmul8_dist_engine*, int, int, int, int, int, int, size_t, size_t, size_t, unsigned, int,
                               gemmul8_dist_plan**) { return GEMMUL8_E_UNSUPPORTED; }
EXPORT int gemmul8_dist_gemm(gemmul8_dist_plan*, void*, const void*, const void*, size_t, const void*, size_t, const void*, void*, size_t) { return GEMMUL8_E_UNSUPPORTED; }
EXPORT int gemmul8_dist_allgather_c(gemmul8_dist_plan*, void*, void*, size_t) { return GEMMUL8_E_UNSUPPORTED; }
EXPORT void gemmul8_dist_destroy(gemmul8_dist_plan*) {}
EXPORT int gemmul8_set_fp8_bound_mode(int mode) { return mode == 0 || mode == 1 ? 0 : -2; }
