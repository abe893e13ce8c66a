/* LABORATORY interface -- exported by tools/experiments/fused_crt/lib/libgemmul8_lab.so only, NOT by libgemmul8.so.
 * The in-kernel CRT forms of the INT8 GEMM: bit-exact, measured 10-28 % slower than the two-launch path
 * (DESIGN.md 3.4), kept buildable and tested (tests/test_gpu_fused_crt.py) as the record of that measurement. */
#ifndef GEMMUL8_LAB_H
#define GEMMUL8_LAB_H
#include "../../../include/gemmul8_c.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Low-precision GEMMs of ALL moduli + CRT accumulation + inverse scaling + axpby in ONE launch (real types on
 * the INT8 backend): the kernel keeps an output tile, runs its num_moduli residue GEMMs back to back and accumulates the CRT for the
 * tile itself -- the stand-alone pass over C_mid and its launch disappear; results are bit-identical to gemmul8_lowprec_gemm followed
 * by gemmul8_crt (the residue planes still land in L->C_mid).  Returns GEMMUL8_E_UNSUPPORTED for complex types and the FP8 backend.
 * gemmul8_fused_crt_selected: 1 when gemmul8_gemm takes this path for the shape: in gemmul8_lab_gemm: GEMMUL8_FUSED_CRT=1 | 2 (whenever legal) or
 * =auto (when the tiles of one plane fill the chip); unset = the two-launch path.  Replaces the loop at src/gemmul8_real.hpp:144-204 as a whole. */
GEMMUL8_API int gemmul8_lowprec_gemm_crt(void *stream, int dtype, int backend, size_t m, size_t n, size_t k, unsigned num_moduli,
                             const gemmul8_layout *L, const void *alpha, const void *beta, void *C, size_t ldc);
GEMMUL8_API int gemmul8_fused_crt_selected(int dtype, int backend, size_t m, size_t n, unsigned num_moduli);

/* gemmul8_gemm of the laboratory library: gemmul8_scale, then gemmul8_lowprec_gemm_crt when gemmul8_fused_crt_selected says so
 * (GEMMUL8_FUSED_CRT=1: CRT on the producer waves, =2: CRT tail on the consumer waves, =auto), else the product's gemmul8_gemm. */
GEMMUL8_API int gemmul8_lab_gemm(void *stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k,
                                 const void *alpha, const void *A, size_t lda, const void *B, size_t ldb, const void *beta, void *C,
                                 size_t ldc, unsigned num_moduli, int fastmode, void *work, void *workA, void *workB, int enA,
                                 int enB, int skip_scalA, int skip_scalB, double *timers_ns);

#ifdef __cplusplus
}
#include <hip/hip_runtime.h>
namespace oz2 {
// tile-stationary variant with the CRT accumulation inside the kernel (real types)
bool gemm_i8_crt_fusable(size_t m, size_t n, unsigned N);
hipError_t launch_gemm_i8_mod_crt(hipStream_t stream, int dtype, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp,
                                  size_t m, size_t n, unsigned N, int8_t* out, size_t ldo, size_t strideO, const int16_t* sftA,
                                  const int16_t* sftB, const void* alpha, const void* beta, bool scalars_on_device, void* C, size_t ldc,
                                  int variant);
}  // namespace oz2
#endif
#endif
