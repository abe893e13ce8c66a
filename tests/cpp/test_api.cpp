// C++-API drop-in test: code written against the reference header compiles and runs unchanged.
// Mirrors the reference's own harnesses: the 4x5x3 known-answer sample
// (GEMMul8/sample/dgemm_cuBLAS_int8.cu:17-81), the op x (alpha,beta) x odd-size matrix of
// GEMMul8/debug/test.cu:106-141,247-299 (checked against native hipblasDgemm) and the direct
// skip-scaling calls of README.md:163-195.
#include <hipblas/hipblas.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/gemmul8.hpp"

#define CHECK(x)                                                     \
    do {                                                             \
        if (!(x)) {                                                  \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); \
            return 1;                                                \
        }                                                            \
    } while (0)

template <typename T> T* dev(const std::vector<T>& h) {
    T* d;
    hipMalloc(&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

int main() {
    hipSetDevice(0);
    hipblasHandle_t handle;
    hipblasCreate(&handle);

    {  // ---- known-answer sample
        const int m = 4, n = 3, k = 5;
        std::vector<double> hA = {0x1.13491b78f7ff1p-1, 0x1.d5797d024f750p+0, -0x1.2121e4d9576a2p+1, 0x1.b96ec80cedfb6p-1,
                                  0x1.466a65212f053p-2, -0x1.4ec4a901fe3c4p+0, -0x1.bbff8c0e700a1p-2, 0x1.5ed8f2ba5f2dbp-2,
                                  0x1.ca08e9321d439p+1, 0x1.627ce99fd7ed1p+1, -0x1.599230c5450f8p+0, 0x1.84785f44e10f1p+1,
                                  0x1.73682ebd0c291p-1, -0x1.0245d3a33f7d8p-4, 0x1.6df2c829f659fp-1, -0x1.a3c53ea980203p-3,
                                  -0x1.fc7ec8b9281f7p-4, 0x1.7d5cd28a5e35bp+0, 0x1.68b67bfca10cfp+0, 0x1.6acd1f3bd1cafp+0};
        std::vector<double> hB = {0x1.57ce78e868ad7p-1, -0x1.351ddceb47a8bp+0, 0x1.6f39e78dc4de4p-1, 0x1.a1571993bf63bp+0,
                                  0x1.f4a0918ad43eep-2, 0x1.08e1a41eff3c4p+0, 0x1.742a49c7a8c1fp-1, -0x1.36b937c0e54f0p-2,
                                  0x1.2ceca451a1789p-2, -0x1.9316bb4db16cfp-1, 0x1.c6dbcad09ddd8p-1, -0x1.25a662f3a6d75p+0,
                                  -0x1.11a17e8d7e02fp+0, -0x1.9e769ce56b489p-1, -0x1.78de4dacf30d6p+1};
        std::vector<double> hX = {0x1.d51136ef01e9dp+1, 0x1.5b07528da2db2p+2, -0x1.b7d034d197c42p-4, 0x1.59b0e0e988db5p+1,
                                  0x1.ad784e3b16dc5p-7, -0x1.15b1323003b06p+0, -0x1.922e5c1c4b38bp+1, -0x1.e95843f74c224p-1,
                                  -0x1.f79e85fefa19bp+1, -0x1.0a9fa599dc6d9p+2, -0x1.32cc3fa2fc921p+2, -0x1.b82c3fad3ab16p+2};
        std::vector<double> hC(m * n, 0.0);
        double *A = dev(hA), *B = dev(hB), *C = dev(hC);
        const unsigned num_moduli = 15;
        const size_t lwork = gemmul8::workSize<false, gemmul8::Backend::INT8>(m, n, k, num_moduli);
        void* work;
        hipMalloc(&work, lwork);
        const double alpha = 1.0, beta = 0.0;
        std::vector<double> t = gemmul8::gemm<double, gemmul8::Backend::INT8>(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &alpha, A, m, B, k,
                                                                               &beta, C, m, num_moduli, false, work);
        hipMemcpy(hC.data(), C, hC.size() * 8, hipMemcpyDeviceToHost);
        double nrm = 0;
        for (int i = 0; i < m * n; ++i) nrm = std::fma(hX[i] - hC[i], hX[i] - hC[i], nrm);
        nrm = std::sqrt(nrm);
        std::printf("KAT error = %e  timers(ns) = %.0f %.0f %.0f %.0f\n", nrm, t[0], t[1], t[2], t[3]);
        CHECK(nrm < 4e-15);
        CHECK(t.size() == 4 && t[0] > 0 && t[1] > 0 && t[3] > 0);
        hipFree(work), hipFree(A), hipFree(B), hipFree(C);
    }

    {  // ---- op x (alpha,beta) on odd sizes vs native hipblasDgemm; host and device scalars; gemmLt entry
        std::mt19937 gen(9999);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        const double ab[5][2] = {{1, 0}, {1, 1}, {-1, 0}, {-1, 1}, {-1.5, 1.5}};
        for (int m : {33, 47})
            for (hipblasOperation_t ta : {HIPBLAS_OP_N, HIPBLAS_OP_T})
                for (hipblasOperation_t tb : {HIPBLAS_OP_N, HIPBLAS_OP_T})
                    for (auto& s : ab) {
                        const int n = m + 3, k = m + 8;
                        const int lda = ta == HIPBLAS_OP_N ? m : k, ldb = tb == HIPBLAS_OP_N ? k : n;
                        std::vector<double> hA((size_t)lda * (ta == HIPBLAS_OP_N ? k : m)), hB((size_t)ldb * (tb == HIPBLAS_OP_N ? n : k)), hC((size_t)m * n);
                        for (auto& x : hA) x = U(gen);
                        for (auto& x : hB) x = U(gen);
                        for (auto& x : hC) x = U(gen);
                        double *A = dev(hA), *B = dev(hB), *C1 = dev(hC), *C2 = dev(hC), *C3 = dev(hC);
                        hipblasDgemm(handle, ta, tb, m, n, k, &s[0], A, lda, B, ldb, &s[1], C1, m);
                        void* work;
                        hipMalloc(&work, gemmul8::workSize<false>(m, n, k, 16));
                        gemmul8::gemm<double>(handle, ta, tb, m, n, k, &s[0], A, lda, B, ldb, &s[1], C2, m, 16, false, work);
                        // device-pointer scalars through the gemmLt entry (stream given explicitly)
                        std::vector<double> sc = {s[0], s[1]};
                        double* dsc = dev(sc);
                        gemmul8::gemmLt<double, gemmul8::Backend::INT8>(nullptr, ta, tb, m, n, k, dsc, A, lda, B, ldb, dsc + 1, C3, m, 16, true, work, nullptr,
                                                                       nullptr, false, false, false, false, 0);
                        std::vector<double> r1(hC.size()), r2(hC.size()), r3(hC.size());
                        hipMemcpy(r1.data(), C1, r1.size() * 8, hipMemcpyDeviceToHost);
                        hipMemcpy(r2.data(), C2, r2.size() * 8, hipMemcpyDeviceToHost);
                        hipMemcpy(r3.data(), C3, r3.size() * 8, hipMemcpyDeviceToHost);
                        double e2 = 0, e3 = 0;
                        for (size_t i = 0; i < r1.size(); ++i) e2 = std::fmax(e2, std::fabs(r1[i] - r2[i])), e3 = std::fmax(e3, std::fabs(r1[i] - r3[i]));
                        if (!(e2 < 1e-12 && e3 < 1e-12)) {
                            std::printf("FAILED ops ta=%d tb=%d m=%d alpha=%g beta=%g: %e %e\n", ta, tb, m, s[0], s[1], e2, e3);
                            return 1;
                        }
                        hipFree(work), hipFree(A), hipFree(B), hipFree(C1), hipFree(C2), hipFree(C3), hipFree(dsc);
                    }
        std::printf("op x alpha/beta matrix ok\n");
    }

    {  // ---- skip-scaling through the direct API: cached planes give the same bits
        const int m = 100, n = 90, k = 300;
        std::mt19937 gen(1);
        std::normal_distribution<double> G;
        std::vector<double> hA((size_t)m * k), hB((size_t)k * n), hC((size_t)m * n, 0.0);
        for (auto& x : hA) x = G(gen);
        for (auto& x : hB) x = G(gen);
        double *A = dev(hA), *B = dev(hB), *C1 = dev(hC), *C2 = dev(hC);
        size_t wA, wB;
        const size_t w = gemmul8::workSize<false>(m, n, k, 14, true, true, &wA, &wB);
        void *work, *workA, *workB;
        hipMalloc(&work, w - wA - wB), hipMalloc(&workA, wA), hipMalloc(&workB, wB);
        const double one = 1, zero = 0;
        for (bool fast : {false, true}) {
            gemmul8::gemm<double>(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, A, m, B, k, &zero, C1, m, 14, fast, work, workA, workB, true, true, false, false);
            gemmul8::gemm<double>(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, A, m, B, k, &zero, C2, m, 14, fast, work, workA, workB, true, true, true, true);
            std::vector<double> r1(hC.size()), r2(hC.size());
            hipMemcpy(r1.data(), C1, r1.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(r2.data(), C2, r2.size() * 8, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < r1.size(); ++i) CHECK(r1[i] == r2[i]);
            // skip only A, with a new B
            gemmul8::gemm<double>(handle, HIPBLAS_OP_N, HIPBLAS_OP_T, m, m, k, &one, A, m, A, m, &zero, C1, m, 14, fast, work, workA, workB, true, true, false, false);
            gemmul8::gemm<double>(handle, HIPBLAS_OP_N, HIPBLAS_OP_N, m, n, k, &one, A, m, B, k, &zero, C1, m, 14, fast, work, workA, workB, true, true, true, false);
            hipMemcpy(r1.data(), C1, r1.size() * 8, hipMemcpyDeviceToHost);
            double e = 0;
            for (size_t i = 0; i < r1.size(); ++i) e = std::fmax(e, std::fabs(r1[i] - r2[i]));
            CHECK(e < 1e-11);
        }
        std::printf("skip-scaling ok\n");
    }

    hipblasDestroy(handle);
    std::printf("ALL OK\n");
    return 0;
}
