// LABORATORY host code of libgemmul8_lab.so: the C entry points of the in-kernel CRT forms (see gemmul8_lab.h).  Not product code.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../../gemmul8_amd/csrc/oz2_kernels.h"
#include "gemmul8_lab.h"

using namespace oz2;

static bool lab_scalars_on_device(const void* alpha) {
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, alpha) == hipSuccess)
        return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeArray;
    (void)hipGetLastError();
    return false;
}

extern "C" {

int gemmul8_fused_crt_selected(int dtype, int backend, size_t m, size_t n, unsigned N) {
    if (backend != kINT8 || dtype < 0 || dtype > 3 || is_complex(dtype) || N < 2 || N > 20) return 0;
    const char* s = getenv("GEMMUL8_FUSED_CRT");
    if (s && (s[0] == '1' || s[0] == '2')) return s[0] - '0';          // whenever it is legal (2: the consumer-tail variant)
    if (s && s[0] == 'a') return gemm_i8_crt_fusable(m, n, N) ? 1 : 0;  // "auto": when the tiles of one plane fill the chip
    return 0;
}

int gemmul8_lowprec_gemm_crt(void* stream_, int dtype, int backend, size_t m, size_t n, size_t k, unsigned N, const gemmul8_layout* L,
                             const void* alpha, const void* beta, void* C, size_t ldc) {
    (void)k;
    if (!L || !alpha || !beta || !C) return GEMMUL8_E_ARG;
    if (dtype < 0 || dtype > 3 || backend < 0 || backend > 1) return GEMMUL8_E_ARG;
    if (N < 2 || N > 20) return GEMMUL8_E_NUM_MODULI;
    if (backend != kINT8 || is_complex(dtype)) return GEMMUL8_E_UNSUPPORTED;
    const char* env_fused = getenv("GEMMUL8_FUSED_CRT");
    const hipError_t e = launch_gemm_i8_mod_crt((hipStream_t)stream_, dtype, (const int8_t*)L->A_lo, (const int8_t*)L->B_lo, L->sizeA, L->sizeB, L->kp, m,
                                                n, N, (int8_t*)L->C_mid, L->mp, L->sizeC, L->sftA, L->sftB, alpha, beta, lab_scalars_on_device(alpha), C,
                                                ldc, (env_fused && env_fused[0] == '2') ? 2 : 1);
    return e == hipSuccess ? GEMMUL8_OK : (int)e;  // positive = hipError_t, as in the product driver
}

int gemmul8_lab_gemm(void* stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void* alpha, const void* A,
                     size_t lda, const void* B, size_t ldb, const void* beta, void* C, size_t ldc, unsigned N, int fastmode, void* work,
                     void* workA, void* workB, int enA, int enB, int skip_scalA, int skip_scalB, double* timers_ns) {
    if (m == 0 || n == 0 || k == 0 || !gemmul8_fused_crt_selected(dtype, backend, m, n, N))
        return gemmul8_gemm(stream, dtype, backend, op_A, op_B, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, N, fastmode, work, workA, workB, enA,
                            enB, skip_scalA, skip_scalB, timers_ns);
    if (timers_ns) timers_ns[0] = timers_ns[1] = timers_ns[2] = timers_ns[3] = 0.0;
    if (!alpha || !beta || !A || !B || !C || !work) return GEMMUL8_E_ARG;
    if (k > (size_t(1) << 17)) return GEMMUL8_E_ARG;
    gemmul8_layout L;
    int rc = gemmul8_get_layout(dtype, backend, m, n, k, N, work, workA, workB, enA, enB, &L);
    if (rc) return rc;
    rc = gemmul8_scale(stream, dtype, backend, op_A, op_B, m, n, k, A, lda, B, ldb, N, fastmode, 0, N, &L, skip_scalA && enA, skip_scalB && enB);
    if (rc) return rc;
    return gemmul8_lowprec_gemm_crt(stream, dtype, backend, m, n, k, N, &L, alpha, beta, C, ldc);  // residue GEMMs + CRT in one launch
}

}  // extern "C"
