// Hook round trip (the reference's GEMMul8/debug/test_hijack.cu:33-99,163-177): an unmodified
// hipBLAS program run under LD_PRELOAD=libgemmul8.so.  This program links ONLY hipBLAS + the HIP
// runtime; the direct emulation it compares with is reached through dlsym on the preloaded library.
//   GEMMUL8_NUM_MOD_D=15 (set by the caller): hipblasDgemm == gemmul8_gemm(N=15, accurate) bit for bit
//   S<->D switches and repeated A/B pointers with GEMMUL8_SKIP_SCALE_A/B=1: same bits again
//   unset at run time (setenv NUM_MOD_D=0): native passthrough, differs from the emulation only by rounding
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipblas/hipblas.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x)                                                       \
    do {                                                               \
        if (!(x)) {                                                    \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); \
            return 1;                                                  \
        }                                                              \
    } while (0)

using gemm_fn = int (*)(void*, int, int, int, int, size_t, size_t, size_t, const void*, const void*, size_t, const void*, size_t, const void*,
                        void*, size_t, unsigned, int, void*, void*, void*, int, int, int, int, double*);
using ws_fn = size_t (*)(int, int, size_t, size_t, size_t, unsigned, int, int, size_t*, size_t*);

template <typename T> T* dev(const std::vector<T>& h) {
    T* d;
    hipMalloc(&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

int main() {
    auto direct = (gemm_fn)dlsym(RTLD_DEFAULT, "gemmul8_gemm");
    auto wsize = (ws_fn)dlsym(RTLD_DEFAULT, "gemmul8_work_size");
    if (!direct || !wsize) {
        std::printf("FAILED: run me with LD_PRELOAD=libgemmul8.so\n");
        return 1;
    }
    hipSetDevice(0);
    hipblasHandle_t handle;
    hipblasCreate(&handle);
    hipStream_t s1;
    hipStreamCreate(&s1);
    const int m = 160, n = 130, k = 500;
    std::mt19937 gen(0);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<double> hA((size_t)m * k), hB((size_t)k * n), hC((size_t)m * n, 0.0);
    for (auto& x : hA) x = U(gen);
    for (auto& x : hB) x = U(gen);
    std::vector<float> fA(hA.begin(), hA.end()), fB(hB.begin(), hB.end()), fC(hC.begin(), hC.end());
    double *A = dev(hA), *B = dev(hB), *C1 = dev(hC), *C2 = dev(hC);
    float *sA = dev(fA), *sB = dev(fB), *sC = dev(fC);
    const double one = 1, zero = 0;
    const float onef = 1, zerof = 0;
    void* work;
    hipMalloc(&work, wsize(0, 0, m, n, k, 15, 0, 0, nullptr, nullptr));
    CHECK(direct(nullptr, 1, 0, 0, 0, m, n, k, &one, A, m, B, k, &zero, C2, m, 15, 0, work, nullptr, nullptr, 0, 0, 0, 0, nullptr) == 0);
    std::vector<double> ref(hC.size()), got(hC.size());
    hipMemcpy(ref.data(), C2, ref.size() * 8, hipMemcpyDeviceToHost);

    // the sequence of test_hijack.cu: D, D (same pointers -> skip-scaling cache), S, D on another stream, GemmEx
    // reps 5-8: the ILP64 twins and the WithFlags variants that ROCm 7's hipBLAS exports as well
    for (int rep = 0; rep < 9; ++rep) {
        if (rep == 2) CHECK(hipblasSgemm(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &onef, sA, m, sB, k, &zerof, sC, m) == HIPBLAS_STATUS_SUCCESS);
        if (rep == 3) hipblasSetStream(handle, s1);
        hipMemset(C1, 0, hC.size() * 8);
        hipDeviceSynchronize();
        if (rep == 4)
            CHECK(hipblasGemmEx(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, A, HIP_R_64F, m, B, HIP_R_64F, k, &zero, C1, HIP_R_64F, m,
                                HIPBLAS_COMPUTE_64F, HIPBLAS_GEMM_DEFAULT) == HIPBLAS_STATUS_SUCCESS);
        else if (rep == 5)
            CHECK(hipblasDgemm_64(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, (int64_t)m, (int64_t)n, (int64_t)k, &one, A, (int64_t)m, B, (int64_t)k, &zero,
                                  C1, (int64_t)m) == HIPBLAS_STATUS_SUCCESS);
        else if (rep == 6)
            CHECK(hipblasGemmEx_64(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, (int64_t)m, (int64_t)n, (int64_t)k, &one, A, HIP_R_64F, (int64_t)m, B,
                                   HIP_R_64F, (int64_t)k, &zero, C1, HIP_R_64F, (int64_t)m, HIPBLAS_COMPUTE_64F, HIPBLAS_GEMM_DEFAULT) ==
                  HIPBLAS_STATUS_SUCCESS);
        else if (rep == 7)
            CHECK(hipblasGemmExWithFlags(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, A, HIP_R_64F, m, B, HIP_R_64F, k, &zero, C1,
                                         HIP_R_64F, m, HIPBLAS_COMPUTE_64F, HIPBLAS_GEMM_DEFAULT, HIPBLAS_GEMM_FLAGS_NONE) == HIPBLAS_STATUS_SUCCESS);
        else if (rep == 8)
            CHECK(hipblasGemmExWithFlags_64(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, (int64_t)m, (int64_t)n, (int64_t)k, &one, A, HIP_R_64F, (int64_t)m,
                                            B, HIP_R_64F, (int64_t)k, &zero, C1, HIP_R_64F, (int64_t)m, HIPBLAS_COMPUTE_64F, HIPBLAS_GEMM_DEFAULT,
                                            HIPBLAS_GEMM_FLAGS_NONE) == HIPBLAS_STATUS_SUCCESS);
        else
            CHECK(hipblasDgemm(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, A, m, B, k, &zero, C1, m) == HIPBLAS_STATUS_SUCCESS);
        hipDeviceSynchronize();
        hipMemcpy(got.data(), C1, got.size() * 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < got.size(); ++i)
            if (got[i] != ref[i]) {
                std::printf("FAILED rep %d: hooked result differs from the direct call at %zu: %a vs %a\n", rep, i, got[i], ref[i]);
                return 1;
            }
    }
    std::printf("hooked hipblasDgemm/GemmEx (+ _64, WithFlags) == direct gemmul8_gemm (bitwise), incl. skip-scaling and stream switch\n");

    // strided batched (torch.bmm's entry point; the reference has none): 7 items spread over the hook's stream lanes
    // (GEMMUL8_BATCH_STREAMS, default 4), then the serial loop -- every item bitwise equal to the direct call on that item
    {
        const int nb = 7;
        std::vector<double> bA((size_t)nb * m * k), bB((size_t)nb * k * n), bC((size_t)nb * m * n, 0.0), bgot(bC.size());
        for (auto& x : bA) x = U(gen) - 0.5;
        for (auto& x : bB) x = U(gen) - 0.5;
        double *dA = dev(bA), *dB = dev(bB), *dC = dev(bC);
        std::vector<double> bref(bC.size());
        for (int b = 0; b < nb; ++b) {
            CHECK(direct(nullptr, 1, 0, 0, 0, m, n, k, &one, dA + (size_t)b * m * k, m, dB + (size_t)b * k * n, k, &zero, C2, m, 15, 0, work, nullptr,
                         nullptr, 0, 0, 0, 0, nullptr) == 0);
            hipMemcpy(bref.data() + (size_t)b * m * n, C2, (size_t)m * n * 8, hipMemcpyDeviceToHost);
        }
        for (int pass = 0; pass < 5; ++pass) {  // 0: one set of launches (gemmul8_gemm_batched), 1-2: stream lanes, 3: serial loop,
                                                // 4: one launch set per chunk of 2 items (bounded workspace)
            if (pass == 1) setenv("GEMMUL8_BATCH_FUSED", "0", 1);
            if (pass == 3) setenv("GEMMUL8_BATCH_STREAMS", "1", 1);
            if (pass == 4) unsetenv("GEMMUL8_BATCH_FUSED"), setenv("GEMMUL8_BATCH_WORKSPACE_MB", "80", 1);  // an item needs ~33 MiB
            hipMemset(dC, 0, bC.size() * 8);
            hipDeviceSynchronize();
            CHECK(hipblasDgemmStridedBatched(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, dA, m, (long long)m * k, dB, k, (long long)k * n, &zero,
                                             dC, m, (long long)m * n, nb) == HIPBLAS_STATUS_SUCCESS);
            hipStreamSynchronize(s1);  // the handle's stream: the side lanes must have been joined into it
            hipMemcpy(bgot.data(), dC, bgot.size() * 8, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < bgot.size(); ++i)
                if (bgot[i] != bref[i]) {
                    std::printf("FAILED batched pass %d: item %zu differs from the direct call: %a vs %a\n", pass, i / ((size_t)m * n), bgot[i], bref[i]);
                    return 1;
                }
        }
        unsetenv("GEMMUL8_BATCH_STREAMS");
        unsetenv("GEMMUL8_BATCH_FUSED");
        unsetenv("GEMMUL8_BATCH_WORKSPACE_MB");
        std::printf("hooked hipblasDgemmStridedBatched (7 items; one launch set, stream lanes, serial) == direct gemmul8_gemm per item (bitwise)\n");
        hipFree(dA), hipFree(dB), hipFree(dC);
    }

    // float result sanity (S path emulated with GEMMUL8_NUM_MOD_S from the environment)
    std::vector<float> gs(fC.size());
    hipMemcpy(gs.data(), sC, gs.size() * 4, hipMemcpyDeviceToHost);
    double l2 = 0;
    for (size_t i = 0; i < gs.size(); ++i) l2 += (gs[i] - ref[i]) * (gs[i] - ref[i]);
    std::printf("SGEMM vs DGEMM-emulation L2^2 = %e\n", l2);
    CHECK(l2 < 1e-3);  // threshold of test_hijack.cu:95

    // early outs and passthrough
    CHECK(hipblasDgemm(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, 0, n, k, &one, A, m, B, k, &zero, C1, m) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasDgemm(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, nullptr, m, B, k, &zero, C1, m) == HIPBLAS_STATUS_INVALID_VALUE);
    setenv("GEMMUL8_NUM_MOD_D", "0", 1);  // read on every call -> native routine
    CHECK(hipblasDgemm(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, A, m, B, k, &zero, C1, m) == HIPBLAS_STATUS_SUCCESS);
    hipDeviceSynchronize();
    hipMemcpy(got.data(), C1, got.size() * 8, hipMemcpyDeviceToHost);
    double e = 0;
    size_t ndiff = 0;
    for (size_t i = 0; i < got.size(); ++i) e = std::fmax(e, std::fabs(got[i] - ref[i]) / std::fabs(ref[i])), ndiff += got[i] != ref[i];
    std::printf("native passthrough vs emulation: max rel diff %e, %zu differing elements\n", e, ndiff);
    CHECK(e < 1e-12);
    CHECK(hipblasDestroy(handle) == HIPBLAS_STATUS_SUCCESS);
    std::printf("ALL OK\n");
    return 0;
}
