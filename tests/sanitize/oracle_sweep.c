/* ASan + UBSan run of the CPU oracle (the reference's debug/Makefile:49-53 has compute-sanitizer targets for its
 * device code, the host-side counterpart here is the oracle and the hook): every dtype x backend x mode x op combination on small
 * ragged shapes, results only checked for finiteness -- the point is that no read or write leaves its buffer and no undefined
 * integer / shift / conversion occurs.  Built and run by tests/test_sanitizers.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oz2_gemm(int dtype, int backend, int opA, int opB, size_t m, size_t n, size_t k, const void *alpha, const void *A, size_t lda, const void *B,
             size_t ldb, const void *beta, void *C, size_t ldc, unsigned N, int fastmode, int scalar_mode, const int16_t *sftA_in,
             const int16_t *sftB_in, int16_t *sftA_out, int16_t *sftB_out, uint8_t *A_lo_out, uint8_t *B_lo_out, void *C_mid_out);
unsigned oz2_num_mat(int backend, unsigned N);

static double rnd(unsigned *s) {
    *s = *s * 1664525u + 1013904223u;
    return ((*s >> 8) / 16777216.0 - 0.5) * exp(((*s >> 3) % 13) - 6.0);
}

int main(void) {
    unsigned seed = 7;
    int cases = 0;
    const size_t shapes[][3] = {{1, 1, 1}, {5, 3, 17}, {9, 7, 40}};
    for (int dtype = 0; dtype < 4; ++dtype)
        for (int backend = 0; backend < 2; ++backend)
            for (int fast = 0; fast < 2; ++fast)
                for (int op = 0; op < 9; ++op)
                    for (int sh = 0; sh < 3; ++sh) {
                        const int opA = op / 3, opB = op % 3;
                        const size_t m = shapes[sh][0], n = shapes[sh][1], k = shapes[sh][2];
                        const int cplx = dtype >= 2, f32 = dtype == 0 || dtype == 2;
                        const size_t es = (f32 ? 4 : 8) * (cplx ? 2 : 1), comps = cplx ? 2 : 1;
                        const unsigned N = f32 ? (sh == 2 ? 13 : 6) : (sh == 0 ? 2 : sh == 1 ? 14 : 20);
                        const size_t ra = opA ? k : m, ca = opA ? m : k, rb = opB ? n : k, cb = opB ? k : n;
                        /* exact-size buffers so that any overrun is a heap-buffer-overflow */
                        void *A = malloc(ra * ca * es), *B = malloc(rb * cb * es), *C = malloc(m * n * es);
                        for (size_t i = 0; i < ra * ca * comps; ++i) {
                            if (f32) ((float *)A)[i] = (float)rnd(&seed);
                            else ((double *)A)[i] = rnd(&seed);
                        }
                        for (size_t i = 0; i < rb * cb * comps; ++i) {
                            if (f32) ((float *)B)[i] = (float)rnd(&seed);
                            else ((double *)B)[i] = rnd(&seed);
                        }
                        for (size_t i = 0; i < m * n * comps; ++i) {
                            if (f32) ((float *)C)[i] = (float)rnd(&seed);
                            else ((double *)C)[i] = rnd(&seed);
                        }
                        double al[2] = {-1.5, 0.5}, be[2] = {0.75, -0.25};
                        float alf[2] = {-1.5f, 0.5f}, bef[2] = {0.75f, -0.25f};
                        const unsigned nm = oz2_num_mat(backend, N), parts = cplx ? 3 : 1;
                        int16_t *sA = malloc(2 * m), *sB = malloc(2 * n);
                        uint8_t *Alo = malloc((size_t)parts * nm * m * k), *Blo = malloc((size_t)parts * nm * n * k);
                        void *Cmid = malloc((size_t)N * m * n * comps * (backend ? 2 : 1));
                        if (oz2_gemm(dtype, backend, opA, opB, m, n, k, f32 ? (void *)alf : (void *)al, A, ra, B, rb, f32 ? (void *)bef : (void *)be, C, m, N, fast,
                                     sh & 1, NULL, NULL, sA, sB, Alo, Blo, Cmid) != 0) {
                            printf("FAILED: oz2_gemm returned an error (dtype %d backend %d)\n", dtype, backend);
                            return 1;
                        }
                        for (size_t i = 0; i < m * n * comps; ++i) {
                            const double v = f32 ? ((float *)C)[i] : ((double *)C)[i];
                            if (!isfinite(v)) {
                                printf("FAILED: non-finite result (dtype %d backend %d fast %d op %d shape %d)\n", dtype, backend, fast, op, sh);
                                return 1;
                            }
                        }
                        free(A), free(B), free(C), free(sA), free(sB), free(Alo), free(Blo), free(Cmid);
                        ++cases;
                    }
    printf("oracle sweep under sanitizers: %d cases ALL OK\n", cases);
    return 0;
}
