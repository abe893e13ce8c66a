// Testing / A-B knobs of the kernel launchers, parsed ONCE from the environment into one struct (first use), not per launch: the
// launch path of a small GEMM is latency-bound, and getenv is not safe against a concurrent setenv from a host thread (a Python
// os.environ write).  Every knob selects between bit-identical code paths; none is needed in production.  A test harness that
// changes the environment inside one process calls gemmul8_reload_knobs() (include/gemmul8_c.h) afterwards.  INTEGRATION.md lists them.
#pragma once

namespace oz2 {

struct Knobs {
    int epi_nt = -1;               // GEMMUL8_EPI_NT=0|1: force default / non-temporal residue stores of the INT8 GEMM (-1: per-launch rule)
    int bound_tile = 0;            // GEMMUL8_BOUND_TILE=128|256: force the small-tile / persistent kernel for the INT8 bound GEMM (0: by tile count)
    int cplx_bound_launches = 1;   // GEMMUL8_CPLX_BOUND_LAUNCHES=2: complex INT8 bound as a 2-segment + a 3-segment launch (1: one launch, mid-tile maxima)
    int cplx_chunk = 0;            // GEMMUL8_CPLX_CHUNK=<n>: moduli per X / Y / Z launch group of the complex INT8 path (0: as many as the scratch holds)
    int crt_kernel = 0;            // GEMMUL8_CRT_KERNEL=dma|reg: force the LDS-DMA (1) / register (2) form of the CRT kernel (0: by eligibility and size)
    int fp8_planes = 0;            // GEMMUL8_FP8_PLANES=e4m3|fp6: FP8 backend's residue planes as e4m3 bytes (1) / FP6 panel images where they fit (0, default)
    int fp8_fused = 1;             // GEMMUL8_FP8_FUSED=0: FP6 planes, but the three products of a modulus as two or three launches with int16 partial-residue planes (the round-4 structure) instead of one three-segment tile loop
    int gemm_cus = 0;              // GEMMUL8_GEMM_CUS=<n>: workgroups (= CUs) of the persistent INT8 residue-GEMM launches, a multiple of 8 (0: every CU); the phase-overlap measurements leave CUs to a second stream with it
    int crt_panels = 0;            // GEMMUL8_CRT_PANELS=<P>[r]: real INT8 whole call as P column panels, gemm(p) crt(p) back to back (SURVEY 8 f3 by cache residency); suffix r: every panel's residues go to panel 0's columns of C_mid (0: one GEMM launch, one CRT launch)
    int crt_panels_ring = 0;
    int scale_fold = 1;            // GEMMUL8_SCALE_FOLD=0: accurate mode's zero-fill, two extracts and shift finalize as launches of their own (9 launches) instead of the extract-pair launch that also zero-fills and the quantise launch that also finalizes (6)
    int map_colblock = -1;         // GEMMUL8_MAP_COLBLOCK=<w>: tile-columns per column block of the GEMM tile walk, 0 = full width (-1: map_colblock's rule)
};
const Knobs& knobs();  // oz2_driver.hip
void reload_knobs();

}  // namespace oz2
