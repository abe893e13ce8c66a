// A stand-in for librocblas.so used by tests/test_hook_abi_guard.py ONLY (no GPU, no rocBLAS): it exports the three entry points the hook's
// rocBLAS interposition consults -- rocblas_get_version_string{_size}, rocblas_get_stream -- and the mangled
// rocblas_internal_gemm_template<double> of ROCm 7.2's rocBLAS, which here only counts its calls.  The version string comes from
// FAKE_ROCBLAS_VERSION so that one binary serves the "tested" and the "untested release" case.
#include <cstdio>
#include <cstdlib>
#include <cstring>
extern "C" {
__attribute__((visibility("default"))) int fake_rocblas_native_calls = 0;
static const char* version() {
    const char* v = getenv("FAKE_ROCBLAS_VERSION");
    return v ? v : "9.9.9.deadbeef";
}
__attribute__((visibility("default"))) int rocblas_get_version_string_size(size_t* len) {
    *len = strlen(version()) + 1;
    return 0;
}
__attribute__((visibility("default"))) int rocblas_get_version_string(char* buf, size_t len) {
    if (len < strlen(version()) + 1) return 1;
    strcpy(buf, version());
    return 0;
}
// an untested release must never get as far as asking for the handle's stream
__attribute__((visibility("default"))) int fake_rocblas_stream_queries = 0;
__attribute__((visibility("default"))) int rocblas_get_stream(void*, void** s) {
    ++fake_rocblas_stream_queries;
    *s = nullptr;
    return 0;
}
__attribute__((visibility("default"))) int fake_internal_gemm_d(void*, int, int, int, int, int, const double*, const double*, long, int, long, const double*, long, int,
                                                                long, const double*, double*, long, int, long, int)
    __asm__("_Z30rocblas_internal_gemm_templateIdE15rocblas_status_P15_rocblas_handle18rocblas_operation_S3_iiiPKT_S6_lilS6_lilS6_PS4_lili");
int fake_internal_gemm_d(void*, int, int, int, int, int, const double*, const double*, long, int, long, const double*, long, int, long, const double*, double*,
                         long, int, long, int) {
    ++fake_rocblas_native_calls;
    return 0;
}
}
