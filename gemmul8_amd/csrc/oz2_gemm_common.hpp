// Shared pieces of the low-precision MFMA GEMM kernels (INT8: oz2_gemm_i8.hip, FP8: oz2_gemm_f8.hip):
// tile constants, XCD-aware workgroup->tile mapping and the row-maxima reduction of the bound-GEMM epilogues.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "oz2_knobs.hpp"

namespace oz2 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 128;
constexpr int TILE_BYTES = BM * BK;  // 32 KiB per operand panel
constexpr int WS_THREADS = 768;      // 8 consumer + 4 producer waves

struct TileMap {
    int plane, tm, tn;
};
// block b runs on XCD b%8 (round-robin dispatch).  The tile sequence (plane-major; inside a plane groups of 8 tile-rows
// with the tile-row index fastest) is cut into chunks of 256 = the tiles in flight at once; inside a chunk XCD x takes
// 32 consecutive tiles = 8 x 4 tiles sharing 8 + 4 operand panels in its L2, and all 8 XCDs work on the SAME plane and
// the same 8 A panels, so the L2 misses of one chunk (<= 8 + 32 panels of one plane) are served by the 256 MiB
// Infinity Cache instead of HBM.  The tail (< 264 blocks) is split contiguously over the XCDs.
#ifndef OZ2_MAP_CHUNKED
#define OZ2_MAP_CHUNKED 1
#endif
#ifndef OZ2_MAP_COLBLOCK
#define OZ2_MAP_COLBLOCK 1  // 0: always walk the full width of a plane
#endif
// Tile-columns per column block for operand panels of kbytes bytes per row (0 = full width).  Walking the full width, every group of
// 8 tile-rows touches ALL B panels of the plane; they stay in the 256 MiB Infinity Cache between row groups as long as they are not
// much more than half of it (n = 16384, k = 8192: 128 MiB -- blocking costs 3 % there, A is re-streamed once per block).  Beyond
// that (16384^2 x 16384: 256 MiB of B per plane) every row group re-read B from HBM: blocks of ~128 MiB of B panels recover it
// (6 planes 16384^2 x 16384: 19.19 -> 18.32 ms, 12288^2 x 16384: 10.47 -> 10.24; profiles/archive/r03_map_colblock_ab.txt).
inline int map_colblock(size_t tiles_n, size_t kbytes) {
    if (!OZ2_MAP_COLBLOCK) return 0;
    if (const int w = knobs().map_colblock; w >= 0)  // testing switch (oz2_knobs.hpp): tile-columns per block, 0 = full width
        return w > 0 && (size_t)w < tiles_n ? w : 0;
    const size_t panel = (size_t)BN * kbytes;
    if (tiles_n * panel <= ((size_t)160 << 20)) return 0;
    size_t w = (((size_t)128 << 20) / panel) & ~(size_t)3;
    if (w < 8) w = 8;
    return w >= tiles_n ? 0 : (int)w;
}
// bid: (virtual) workgroup id -- blockIdx.x + round * gridDim.x in the persistent kernels, gridDim.x a multiple of 8
// whenever there is more than one round -- nwg: total number of tiles.
__device__ __forceinline__ TileMap map_tile(int bid, int nwg, int tiles_m, int tiles_n, int colblock) {
    const int tiles_per_plane = tiles_m * tiles_n;
    {
        const int xcd = bid & 7, idx = bid >> 3;
        const int fc = OZ2_MAP_CHUNKED ? ((nwg >> 3) >> 5) : 0;  // full chunks of 256
        if (idx < fc * 32) {
            bid = (idx >> 5) * 256 + xcd * 32 + (idx & 31);
        } else {
            const int rem = nwg - fc * 256, q = rem >> 3, r = rem & 7, i2 = idx - fc * 32;
            bid = fc * 256 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i2;
        }
    }
    TileMap t;
    t.plane = bid / tiles_per_plane;
    int rem = bid - t.plane * tiles_per_plane;
    // Column blocks (colblock > 0, chosen by map_colblock on the host): the plane is walked one block of colblock tile-columns at a
    // time, ALL row groups of a block before the next block, so that the block's B panels stay in the Infinity Cache while the row
    // groups stream past.
    int tn0 = 0, w = tiles_n;
    if (colblock > 0 && tiles_n > colblock) {
        const int per_block = tiles_m * colblock;
        const int b = rem / per_block;
        rem -= b * per_block;
        tn0 = b * colblock;
        w = (tiles_n - tn0) < colblock ? (tiles_n - tn0) : colblock;
    }
    constexpr int GM = 8;
    const int group_sz = GM * w;
    const int g = rem / group_sz;
    const int first_m = g * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    rem -= g * group_sz;
    t.tm = first_m + rem % gm;
    t.tn = tn0 + rem / gm;
    return t;
}

// The same map with its divisions on the SCALAR unit (round 4).  A division by a run-time value expands to a float-reciprocal sequence on the
// vector ALU (v_cvt / v_rcp_iflag / v_mul / v_cvt, v_readfirstlane, a correction): five of them per tile, and -- the divisors being loop
// invariant -- reciprocals hoisted out of the persistent tile loop that sat in VGPRs across every K loop, were spilled there and reloaded
// (a vector-memory wait) at every epilogue.  The host supplies M = floor(2^32 / d) per divisor; q0 = mulhi(x, M) is q or q - 1 for any
// x < 2^32 (x (2^32 / d - M) / 2^32 < 1), one compare-and-correct finishes: s_mul_hi_u32, s_mul_i32, s_sub, s_cmp, s_cselect.
struct TileMapArgs {
    int tiles_m, tiles_n, colblock;
    unsigned tpp, m_tpp;          // tiles per plane
    unsigned pb, m_pb;            // tiles per column block (colblock > 0)
    unsigned m_gs_full, m_gs_tail;  // 8 * (block width): full blocks (or the whole width), the last (narrower) block
};
inline unsigned map_magic(unsigned d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)(0x100000000ull / d); }
inline TileMapArgs make_tile_map(int tiles_m, int tiles_n, int colblock) {
    TileMapArgs a{};
    a.tiles_m = tiles_m, a.tiles_n = tiles_n;
    a.colblock = (colblock > 0 && tiles_n > colblock) ? colblock : 0;
    a.tpp = (unsigned)tiles_m * (unsigned)tiles_n;
    a.m_tpp = map_magic(a.tpp);
    const int wfull = a.colblock ? a.colblock : tiles_n, wtail = a.colblock ? tiles_n % a.colblock : 0;
    a.pb = (unsigned)tiles_m * (unsigned)wfull;
    a.m_pb = map_magic(a.pb);
    a.m_gs_full = map_magic(8u * (unsigned)wfull);
    a.m_gs_tail = map_magic(8u * (unsigned)(wtail ? wtail : wfull));
    return a;
}
__device__ __forceinline__ void udivmod_magic(unsigned x, unsigned d, unsigned M, unsigned& q, unsigned& r) {
    q = __umulhi(x, M);
    r = x - q * d;
    if (r >= d) ++q, r -= d;
}
__device__ __forceinline__ TileMap map_tile(int bid, int nwg, const TileMapArgs a) {  // by value: the laboratory kernels read it from the kernel-argument address space
    {
        const int xcd = bid & 7, idx = bid >> 3;
        const int fc = OZ2_MAP_CHUNKED ? ((nwg >> 3) >> 5) : 0;  // full chunks of 256
        if (idx < fc * 32) {
            bid = (idx >> 5) * 256 + xcd * 32 + (idx & 31);
        } else {
            const int rem = nwg - fc * 256, q = rem >> 3, r = rem & 7, i2 = idx - fc * 32;
            bid = fc * 256 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i2;
        }
    }
    TileMap t;
    unsigned plane, rem;
    udivmod_magic((unsigned)bid, a.tpp, a.m_tpp, plane, rem);
    t.plane = (int)plane;
    int tn0 = 0, w = a.tiles_n;
    unsigned m_gs = a.m_gs_full;
    if (a.colblock > 0) {
        unsigned b;
        udivmod_magic(rem, a.pb, a.m_pb, b, rem);
        tn0 = (int)b * a.colblock;
        if (a.tiles_n - tn0 < a.colblock) w = a.tiles_n - tn0, m_gs = a.m_gs_tail;
        else w = a.colblock;
    }
    constexpr int GM = 8;
    unsigned g;
    udivmod_magic(rem, (unsigned)(GM * w), m_gs, g, rem);
    const int first_m = (int)g * GM;
    if (a.tiles_m - first_m >= GM) {  // full row group
        t.tm = first_m + (int)(rem & (GM - 1));
        t.tn = tn0 + (int)(rem >> 3);
    } else {                          // the last, shorter row group of a plane whose tile-row count is not a multiple of 8
        const unsigned gm = (unsigned)(a.tiles_m - first_m);
        t.tm = first_m + (int)(rem % gm);
        t.tn = tn0 + (int)(rem / gm);
    }
    return t;
}

// Row maxima of ONE 16-row accumulator tile row of the 16x16 MFMA shapes for the bound-GEMM epilogues: v[r], r = 0..3, is this lane's
// (column-masked, non-negative) maximum over the wave's column tiles for row i0 + 4 (lane >> 4) + r; the 16 lanes of a quad hold the
// 16 columns.  Reduce-scatter over lane bits 3, 2 (4 -> 2 -> 1 values), butterfly over bits 1, 0: lane l then holds the finished
// maximum of row r = 2 b3 + b2 (b3, b2 = bits 3, 2 of l); the four lanes with (l & 3) == 0 of every quad issue the atomicMax.
// Working tile by tile keeps only four values live beside the accumulators (a 32-entry array for the whole wave tile pushed the
// FP8 kernel's LDS-DMA offsets into scratch, with a vmcnt(0) in front of every DMA instruction).
__device__ __forceinline__ void tile_rowmax_atomic16(const int (&v)[4], int* rowmax, int i0, int m, int lane) {
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
    // bit 3: lanes with b3 keep r = 2, 3
    const int s0 = b3 ? v[0] : v[2], s1 = b3 ? v[1] : v[3];
    const int k0 = b3 ? v[2] : v[0], k1 = b3 ? v[3] : v[1];
    const int g0 = __shfl_xor(s0, 8), g1 = __shfl_xor(s1, 8);
    const int a0 = g0 > k0 ? g0 : k0, a1 = g1 > k1 ? g1 : k1;  // r = 2 b3, 2 b3 + 1
    // bit 2: lanes with b2 keep the odd one
    const int s = b2 ? a0 : a1, k = b2 ? a1 : a0;
    const int g = __shfl_xor(s, 4);
    int x = g > k ? g : k;  // r = 2 b3 + b2
    int o = __shfl_xor(x, 2);
    x = o > x ? o : x;
    o = __shfl_xor(x, 1);
    x = o > x ? o : x;
    const int row = i0 + 4 * (lane >> 4) + 2 * (b3 ? 1 : 0) + (b2 ? 1 : 0);
    if ((lane & 3) == 0 && row < m && x > 0) atomicMax(rowmax + row, x);
}

}  // namespace oz2
