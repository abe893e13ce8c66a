// LABORATORY translation unit -- builds the INT8 GEMM object of tools/experiments/fused_crt/lib/libgemmul8_lab.so: the product kernel file
// with the in-kernel CRT forms switched on (FUSE = 1 | 2 instantiations, oz2_gemm_i8_fused.inc) plus their host launchers.  Nothing here
// is linked into gemmul8_amd/lib/libgemmul8.so.  See tools/experiments/fused_crt/README.md.
#define OZ2_LAB_FUSED_CRT "../../tools/experiments/fused_crt/oz2_gemm_i8_fused.inc"
#include "../../../gemmul8_amd/csrc/oz2_gemm_i8.hip"

#include "gemmul8_lab.h"

namespace oz2 {

// Tile-stationary GEMM + requantise + CRT in one launch (real types, all N moduli).  Worth it when the tiles of ONE plane fill the
// chip about as well as the tiles of all planes do: the unit of work per workgroup is N times larger.
bool gemm_i8_crt_fusable(size_t m, size_t n, unsigned N) {
    const long tiles = (long)((m + BM - 1) / BM) * (long)((n + BN - 1) / BN);
    long grid = num_cus() & ~7;
    if (grid <= 0) grid = 8;
    const long rounds_fused = (tiles + grid - 1) / grid * (long)N;    // tile-times on the busiest workgroup
    const long rounds_plain = (tiles * (long)N + grid - 1) / grid;
    return tiles >= grid && rounds_fused * 100 <= rounds_plain * 104;
}

hipError_t launch_gemm_i8_mod_crt(hipStream_t stream, int dtype, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp,
                                  size_t m, size_t n, unsigned N, int8_t* out, size_t ldo, size_t strideO, const int16_t* sftA,
                                  const int16_t* sftB, const void* alpha, const void* beta, bool scalars_on_device, void* C, size_t ldc,
                                  int variant) {
    if (dtype != kF64 && dtype != kF32) return hipErrorInvalidValue;
    GemmArgs a{};
    a.A[0] = A;
    a.B[0] = B;
    a.nseg = 1;
    a.strideA = strideA;
    a.strideB = strideB;
    a.t_begin = 0;
    a.out = out;
    a.ldo = ldo;
    a.strideO = strideO;
    fill_common(a, kp, m, n);
    a.planes = (int)N;
    a.ppi = (int)N;
    a.m_ppi = map_magic((unsigned)N);
    a.total_tiles = a.tiles_m * a.tiles_n;
    if (a.total_tiles <= 0) return hipSuccess;
    a.colblock = map_colblock((size_t)a.tiles_n, (size_t)a.kp);
    a.map = make_tile_map(a.tiles_m, a.tiles_n, a.colblock);
    a.acc0 = ((size_t)a.kp <= 512) ? 0 : (int)0x80000000u;
    CrtArgs c{};
    c.m = m;
    c.n = n;
    c.sftA = sftA;
    c.sftB = sftB;
    c.C = C;
    c.ldc = ldc;
    fill_crt_tables(c, dtype, kINT8, N);
    fill_crt_scalars(c, dtype, alpha, beta, scalars_on_device);
    // variant 1: CRT on the producer waves beside the next tile's MFMAs (K-step-barrier schedule at every k); variant 2: CRT tail on
    // the consumer waves after each tile (ping-pong schedule; kept as the measured baseline of DESIGN.md 3.4)
    const bool kbar = variant != 2;
    if (dtype == kF64) return kbar ? launch_sched<EPI_MOD, true, 1>(stream, a, c) : launch_sched<EPI_MOD, false, 1>(stream, a, c);
    return kbar ? launch_sched<EPI_MOD, true, 2>(stream, a, c) : launch_sched<EPI_MOD, false, 2>(stream, a, c);
}

}  // namespace oz2
