// LABORATORY hook definitions for timing probes of the INT8 GEMM kernel on REAL data (the results of a probe build are WRONG on purpose;
// nothing here is part of libgemmul8.so: the product build refuses OZ2_LAB_HOOKS).  Build with tools/build_probes.sh, e.g.
//   tools/build_probes.sh nodma="-DOZ2_PROBE=4" l2res="-DOZ2_PROBE=16" noepi="-DOZ2_PROBE=8"
// and compare against the shipped build with tools/gemm_ab.py (interleaved).  Bits of OZ2_PROBE:
//   4   the LDS-DMA is issued during a workgroup's first tile only (what the L2 -> LDS operand path costs: DESIGN.md 3.1)
//   8   no epilogue, the accumulators stay live
//   16  operands come from the first 8 K-steps of a panel only (every fetch an L2 hit: what the L2 misses cost)
//   64  the epilogue without its stores (what the store path costs at short k)
//   32  dose-response of the operand path: after a workgroup's first tile the LDS-DMA of every OZ2_DMA_SKIP-th K-step is left out
//       (OZ2_DMA_SKIP = 6: -16.7 % of the L2 -> LDS bytes per MAC, what a 384 x 256 CU tile would save; 2: -50 %)
// OZ2_KSTAG=<1..4> staggers the K-step a workgroup starts from (1: XCD x starts at x KT1 / 8; 2: plus (CU & 3) K-steps; 3: (CU & 7)
// K-steps only; 4: odd XCDs start at KT1 / 2) -- bit-identical results (INT32 sums are order-independent), measured neutral
// (profiles/archive/r04_gemm_ab_kstag_order_l2.txt).
#pragma once
#ifndef OZ2_PROBE
#define OZ2_PROBE 0
#endif
#ifndef OZ2_KSTAG
#define OZ2_KSTAG 0
#endif
#ifndef OZ2_DMA_SKIP
#define OZ2_DMA_SKIP 6
#endif
#if OZ2_PROBE & 4
#define OZ2_HOOK_DMA_ON(first_tile) (first_tile)
#elif OZ2_PROBE & 32
#define OZ2_HOOK_DMA_ON(first_tile) ((first_tile) || (kt % OZ2_DMA_SKIP) != 0)
#endif
#if OZ2_PROBE & 8
#define OZ2_HOOK_SKIP_EPILOGUE 1
#endif
#if OZ2_PROBE & 64
#define OZ2_HOOK_SKIP_STORES 1  // round 6: the residue epilogue's arithmetic, transposes and address work, but no global_store
#endif
#if OZ2_PROBE & 16
#define OZ2_HOOK_KSTEP(kin) ((kin) & 7)
#elif OZ2_KSTAG
#define OZ2_HOOK_KSTEP(kin)                                                                                                        \
    (((kin) + (OZ2_KSTAG == 1   ? (int)(((blockIdx.x & 7u) * (unsigned)KT1) >> 3)                                                  \
               : OZ2_KSTAG == 2 ? (int)(((blockIdx.x & 7u) * (unsigned)KT1) >> 3) + (int)((blockIdx.x >> 3) & 3u)                   \
               : OZ2_KSTAG == 3 ? (int)((blockIdx.x >> 3) & 7u)                                                                    \
                                : (int)((blockIdx.x & 1u) * (unsigned)KT1 >> 1))) %                                                 \
     KT1)
#endif
