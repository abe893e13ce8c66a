// Shared pieces of the low-precision MFMA GEMM kernels (INT8: oz2_gemm_i8.hip, FP8: oz2_gemm_f8.hip):
// tile constants, XCD-aware workgroup->tile mapping and the LDS-DMA producer-wave loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oz2 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 128;
constexpr int TILE_BYTES = BM * BK;          // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // two stages = 128 KiB
constexpr int WS_THREADS = 768;              // 8 consumer + 4 producer waves
#ifndef OZ2_PSLOTS
#define OZ2_PSLOTS 4
#endif
constexpr int PSLOTS = OZ2_PSLOTS;  // slots (of 8 per K-step) over which a producer spreads its 16 DMA instructions


struct TileMap {
    int plane, tm, tn;
};
// block b runs on XCD b%8: give each XCD a contiguous tile range; inside a plane, groups of 8 tile-rows with the
// tile-row index fastest, so the 32 CUs of an XCD work on 8 x 4 tiles sharing 8 + 4 operand panels.
__device__ __forceinline__ TileMap map_tile(int tiles_m, int tiles_n) {
    const int tiles_per_plane = tiles_m * tiles_n;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    TileMap t;
    t.plane = bid / tiles_per_plane;
    int rem = bid - t.plane * tiles_per_plane;
    constexpr int GM = 8;
    const int group_sz = GM * tiles_n;
    const int g = rem / group_sz;
    const int first_m = g * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    rem -= g * group_sz;
    t.tm = first_m + rem % gm;
    t.tn = rem / gm;
    return t;
}

// One LDS-DMA instruction of a K-tile: instruction Q = 0..63 moves the 64 16-byte slots p = Q*64 + lane of a stage
// (4096 slots: slot p <-> operand (p>=2048: B), row = (p&2047)>>3, physical chunk = p&7,
// logical chunk = physical ^ ((row>>1)&7)).  tA/tB: this workgroup's first row at the K offset of the tile.
__device__ __forceinline__ void dma_issue(const int8_t* tA, const int8_t* tB, int Q, char* stage, int kp, int nB_valid, int lane) {
    const int p = Q * 64 + lane;
    const bool isB = p >= 2048;
    const int pp = p & 2047;
    int row = pp >> 3;
    const int c = (pp & 7) ^ ((row >> 1) & 7);
    if (isB) row = row < nB_valid ? row : nB_valid - 1;  // B planes have exactly n rows: clamp instead of padding
    const int8_t* src = (isB ? tB : tA) + (size_t)row * kp + c * 16;
    char* dst = stage + (Q * 64) * 16;  // wave-uniform; the hardware adds lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// Producer wave pw = 0..3: DMA instructions Q = pw*16 .. pw*16+15 of every K-tile.
// gA[s]/gB[s]: K-segment s, already offset to this workgroup's first row.  Executes 8*KT + 2 barriers.
__device__ __forceinline__ void producer_loop(const int8_t* const (&gA)[3], const int8_t* const (&gB)[3], int kp, int KT1, int KT,
                                              int nB_valid, char* smem, int pw, int lane) {
    auto issue = [&](const int8_t* tA, const int8_t* tB, int q, char* stage) { dma_issue(tA, tB, pw * 16 + q, stage, kp, nB_valid, lane); };
#pragma unroll
    for (int q = 0; q < 16; ++q) issue(gA[0], gB[0], q, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int seg = 0, kin = 0;  // segment / K-step inside the segment of tile kt+1 (no divisions in the loop)
    for (int kt = 0; kt < KT; ++kt) {
        char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        const bool more = kt + 1 < KT;
        if (++kin == KT1) kin = 0, ++seg;
        const int sg = seg < 3 ? seg : 2;
        const int8_t* tA = gA[sg] + (size_t)kin * BK;
        const int8_t* tB = gB[sg] + (size_t)kin * BK;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            if (sl < PSLOTS && more) {
#pragma unroll
                for (int q = 0; q < 16 / PSLOTS; ++q) issue(tA, tB, sl * (16 / PSLOTS) + q, nxt);
            }
            if (sl == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    __builtin_amdgcn_s_barrier();
}

}  // namespace oz2
