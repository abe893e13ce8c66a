// rocBLAS entry points of the LD_PRELOAD hook (GEMMUL8_HOOK_ROCBLAS=1): an application that calls rocblas_dgemm / rocblas_gemm_ex /
// rocblas_dgemm_strided_batched directly.  Run by tests/test_gpu_cpp.py under LD_PRELOAD=libgemmul8.so with GEMMUL8_NUM_MOD_D=15:
//   argv[1] = "on"  : GEMMUL8_HOOK_ROCBLAS=1 -- every call below must be EMULATED (error vs the long-double product ~1e-16, far below
//                     what the native FP64 routine leaves at k = 1500; and the stats line at exit counts them)
//   argv[1] = "off" : variable unset -- every call goes to rocBLAS untouched
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                        \
    do {                                                                \
        if (!(x)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x);  \
            return 1;                                                   \
        }                                                               \
    } while (0)

static double max_rel_err(const std::vector<double>& A, const std::vector<double>& B, const std::vector<double>& C, int m, int n, int k) {
    double err = 0, scale = 0;
    for (int j = 0; j < n; j += 7)
        for (int i = 0; i < m; i += 5) {
            long double s = 0, sa = 0;
            for (int l = 0; l < k; ++l) {
                s += (long double)A[(size_t)l * m + i] * B[(size_t)j * k + l];
                sa += fabsl((long double)A[(size_t)l * m + i] * B[(size_t)j * k + l]);
            }
            err = std::fmax(err, (double)(fabsl(s - C[(size_t)j * m + i]) / sa));
            scale = 1;
        }
    return scale ? err : 1;
}

int main(int argc, char** argv) {
    const bool on = argc > 1 && !std::strcmp(argv[1], "on");
    const int m = 300, n = 260, k = 1500, batch = 3;
    std::vector<double> A((size_t)m * k * batch), B((size_t)k * n * batch), C((size_t)m * n * batch);
    srand(1);
    for (auto& x : A) x = rand() / (double)RAND_MAX - 0.5;
    for (auto& x : B) x = rand() / (double)RAND_MAX - 0.5;
    double *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, A.size() * 8) == hipSuccess && hipMalloc(&dB, B.size() * 8) == hipSuccess && hipMalloc(&dC, C.size() * 8) == hipSuccess);
    CHECK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice) == hipSuccess);
    rocblas_handle h;
    CHECK(rocblas_create_handle(&h) == rocblas_status_success);
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream) == hipSuccess);
    CHECK(rocblas_set_stream(h, stream) == rocblas_status_success);
    const double one = 1.0, zero = 0.0;
    // emulated (15 moduli): ~2e-17 relative to sum |a||b|; the native FP64 routine at k = 1500: 2e-16 .. 5e-16
    const double tol_emulated = 1e-16;
    double errs[3];
    // 1. rocblas_dgemm
    CHECK(hipMemset(dC, 0, C.size() * 8) == hipSuccess);
    CHECK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, m, n, k, &one, dA, m, dB, k, &zero, dC, m) == rocblas_status_success);
    CHECK(hipStreamSynchronize(stream) == hipSuccess);
    CHECK(hipMemcpy(C.data(), dC, (size_t)m * n * 8, hipMemcpyDeviceToHost) == hipSuccess);
    errs[0] = max_rel_err(A, B, C, m, n, k);
    // 2. rocblas_gemm_ex, in place (c == d)
    CHECK(hipMemset(dC, 0, C.size() * 8) == hipSuccess);
    CHECK(rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_none, m, n, k, &one, dA, rocblas_datatype_f64_r, m, dB, rocblas_datatype_f64_r, k,
                          &zero, dC, rocblas_datatype_f64_r, m, dC, rocblas_datatype_f64_r, m, rocblas_datatype_f64_r, rocblas_gemm_algo_standard, 0,
                          0) == rocblas_status_success);
    CHECK(hipStreamSynchronize(stream) == hipSuccess);
    CHECK(hipMemcpy(C.data(), dC, (size_t)m * n * 8, hipMemcpyDeviceToHost) == hipSuccess);
    errs[1] = max_rel_err(A, B, C, m, n, k);
    // 3. rocblas_dgemm_strided_batched
    CHECK(rocblas_dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, m, n, k, &one, dA, m, (rocblas_stride)m * k, dB, k,
                                        (rocblas_stride)k * n, &zero, dC, m, (rocblas_stride)m * n, batch) == rocblas_status_success);
    CHECK(hipStreamSynchronize(stream) == hipSuccess);
    CHECK(hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost) == hipSuccess);
    errs[2] = 0;
    for (int b = 0; b < batch; ++b) {
        std::vector<double> Ab(A.begin() + (size_t)b * m * k, A.begin() + (size_t)(b + 1) * m * k), Bb(B.begin() + (size_t)b * k * n, B.begin() + (size_t)(b + 1) * k * n),
            Cb(C.begin() + (size_t)b * m * n, C.begin() + (size_t)(b + 1) * m * n);
        errs[2] = std::fmax(errs[2], max_rel_err(Ab, Bb, Cb, m, n, k));
    }
    std::printf("rocblas hook %s: max error relative to sum|a||b|: dgemm %.2e  gemm_ex %.2e  dgemm_strided_batched %.2e\n", on ? "ON" : "off", errs[0],
                errs[1], errs[2]);
    for (double e : errs) {
        if (on) CHECK(e < tol_emulated);
        else CHECK(e > tol_emulated && e < 1e-13);  // the native routine's rounding at k = 1500
    }
    // 4. other types + device-pointer scalars: rocblas_zgemm with alpha / beta in device memory (rocblas_pointer_mode_device), op C / T
    {
        const int mz = 96, nz = 80, kz = 200;
        std::vector<rocblas_double_complex> Az((size_t)kz * mz), Bz((size_t)nz * kz), Cz((size_t)mz * nz);
        for (auto& x : Az) x = rocblas_double_complex(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
        for (auto& x : Bz) x = rocblas_double_complex(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
        rocblas_double_complex *dAz, *dBz, *dCz, *dsc;
        CHECK(hipMalloc(&dAz, Az.size() * 16) == hipSuccess && hipMalloc(&dBz, Bz.size() * 16) == hipSuccess && hipMalloc(&dCz, Cz.size() * 16) == hipSuccess &&
              hipMalloc(&dsc, 32) == hipSuccess);
        CHECK(hipMemcpy(dAz, Az.data(), Az.size() * 16, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(dBz, Bz.data(), Bz.size() * 16, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemset(dCz, 0, Cz.size() * 16) == hipSuccess);
        const rocblas_double_complex sc[2] = {rocblas_double_complex(1.0, 0.0), rocblas_double_complex(0.0, 0.0)};
        CHECK(hipMemcpy(dsc, sc, 32, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(rocblas_set_pointer_mode(h, rocblas_pointer_mode_device) == rocblas_status_success);
        // C = A^H * B^T : A stored k x m (lda = kz), B stored n x k (ldb = nz)
        CHECK(rocblas_zgemm(h, rocblas_operation_conjugate_transpose, rocblas_operation_transpose, mz, nz, kz, dsc, dAz, kz, dBz, nz, dsc + 1, dCz, mz) ==
              rocblas_status_success);
        CHECK(rocblas_set_pointer_mode(h, rocblas_pointer_mode_host) == rocblas_status_success);
        CHECK(hipStreamSynchronize(stream) == hipSuccess);
        CHECK(hipMemcpy(Cz.data(), dCz, Cz.size() * 16, hipMemcpyDeviceToHost) == hipSuccess);
        double ez = 0;
        for (int j = 0; j < nz; j += 3)
            for (int i = 0; i < mz; i += 5) {
                long double sr = 0, si = 0, sa = 0;
                for (int l = 0; l < kz; ++l) {
                    const long double ar = std::real(Az[(size_t)i * kz + l]), ai = -std::imag(Az[(size_t)i * kz + l]);  // conj(A(l, i))
                    const long double br = std::real(Bz[(size_t)l * nz + j]), bi = std::imag(Bz[(size_t)l * nz + j]);   // B(j, l)
                    sr += ar * br - ai * bi;
                    si += ar * bi + ai * br;
                    sa += fabsl(ar * br) + fabsl(ai * bi) + fabsl(ar * bi) + fabsl(ai * br);
                }
                ez = std::fmax(ez, (double)(std::fmax(fabsl(sr - std::real(Cz[(size_t)j * mz + i])), fabsl(si - std::imag(Cz[(size_t)j * mz + i]))) / sa));
            }
        std::printf("rocblas hook %s: zgemm (op C, op T, device scalars) max error relative to sum|a||b|: %.2e\n", on ? "ON" : "off", ez);
        if (on) CHECK(ez < tol_emulated);      // GEMMUL8_NUM_MOD_Z set by the test: emulated
        else CHECK(ez < 1e-13);
        (void)hipFree(dAz), (void)hipFree(dBz), (void)hipFree(dCz), (void)hipFree(dsc);
    }
    CHECK(rocblas_destroy_handle(h) == rocblas_status_success);
    std::printf("rocblas hook test passed (%s)\n", on ? "emulated" : "native");
    return 0;
}
