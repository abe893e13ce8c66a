// CPU test of the rocBLAS ABI guard (oz2_hook.cpp rocblas_internal_abi_ok): run with
//   LD_PRELOAD="libgemmul8.so libfake_rocblas.so" GEMMUL8_HOOK_ROCBLAS=1 GEMMUL8_NUM_MOD_D=14 FAKE_ROCBLAS_VERSION=<v>
// Calls rocblas_internal_gemm_template<double> by its mangled name, as rocSOLVER does.  With an UNTESTED rocBLAS version the hook must hand
// the call to the next definition (the fake) without touching it -- not even asking for the handle's stream.  Prints one line for the
// Python side: "native=<calls> stream_queries=<n>".
#include <cstdio>
extern "C" {
extern int fake_rocblas_native_calls, fake_rocblas_stream_queries;
int internal_gemm_d(void*, int, int, int, int, int, const double*, const double*, long, int, long, const double*, long, int, long, const double*, double*, long,
                    int, long, int)
    __asm__("_Z30rocblas_internal_gemm_templateIdE15rocblas_status_P15_rocblas_handle18rocblas_operation_S3_iiiPKT_S6_lilS6_lilS6_PS4_lili");
}
int main() {
    double one = 1.0, zero = 0.0, a[4] = {1, 2, 3, 4}, b[4] = {1, 0, 0, 1}, c[4] = {0, 0, 0, 0};
    int handle_storage = 0;
    const int rc = internal_gemm_d(&handle_storage, 111, 111, 2, 2, 2, &one, a, 0, 2, 0, b, 0, 2, 0, &zero, c, 0, 2, 0, 1);
    std::printf("rc=%d native=%d stream_queries=%d\n", rc, fake_rocblas_native_calls, fake_rocblas_stream_queries);
    return 0;
}
