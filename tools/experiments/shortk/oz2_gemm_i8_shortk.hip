// LABORATORY kernel (round 4; bit-exact, measured SLOWER than the shipped kernels: see README.md next to this file) -- not part of libgemmul8.so.
// Short-K form of the INT8 residue GEMM (padded k <= 1024; EPI_MOD and EPI_CPLX): the shape a hooked LU / QR issues as
// its trailing update.  Same arithmetic, same epilogues, same tile walk and the same bits as oz2_gemm_i8.hip
// (GEMMul8/src/matmult.hpp:120-175 + src/conv_hi2mid_real.hpp:9-25 of the reference); what changes is WHO is in the epilogue WHEN.
//
// The 256 x 256-tile kernel runs both waves of a SIMD through the same tile in lock-step, so both reach the epilogue together and the
// matrix pipe idles while they reduce, pack and store: ~5 us of a ~15 us tile at k = 1024, of a ~10 us tile at k = 512.
// Here a workgroup is two GROUPS of four waves (one wave per SIMD each) that work on DIFFERENT half-tiles (128 x 256: rows 0-127 / 128-255 of
// the 256 x 256 tile; wave tile 128 x 64 as before) half a period apart: while one group runs the K loop of its half-tile -- alone on the
// matrix pipes, 64 MFMAs per K-step and wave back to back -- the other group runs the epilogue of the half-tile it has just finished on the
// vector ALUs.  No second accumulator set is needed: the two half-tiles in flight belong to different waves.
//   tick t = one K-step (128 bytes of K) of one half-tile; ONE workgroup barrier per tick; half-tile h of a CU owns ticks [h KT, (h+1) KT)
//   group g runs the K loop of half-tiles h = g, g + 2, ...; during the other group's K loop it runs its epilogue, cut into the 8 sub-blocks
//     of i8_epilogue_mod with the KT barriers of those ticks spread between them (Hook)
//   LDS: ring of three 48 KiB slots (A half-panel 128 x 128 B + B panel 256 x 128 B, same XOR swizzle); tick t lives in slot t % 3
//   LDS-DMA: the K-loop group of tick t fetches tick t + 2 (12 instructions per wave, 3 per MFMA segment) -- also across a half-tile
//     boundary, where tick t + 2 belongs to the OTHER group.  RAW: panels of tick t + 1 were issued during tick t - 1; their issuer waits
//     for them before the barrier that ends tick t (K-loop group: s_waitcnt vmcnt(12) -- everything but the 12 just issued; a group that
//     has moved on to its epilogue: vmcnt(0) in front of its first store).  WAR: slot (t + 2) % 3 was last read in tick t - 1, and every
//     ds_read of a tick is complete (lgkmcnt(0)) before the barrier that ends it.
//   In-wave software pipeline (the partner wave of the SIMD is in an epilogue and hides nothing): fragments are double-buffered -- the
//     ds_reads of segment s + 1 are issued before the 16 MFMAs of segment s; 128 accumulators + 64 fragment registers + 12 DMA offsets:
//     8 waves x 256 VGPRs, no producer waves.
// Price: the B panel is fetched once per HALF-tile: 1.5 x the L2 -> LDS bytes per MAC of the 256 x 256 kernel -- which is why this form is
// for short K only, where the epilogue, not the operand path, is what the tile time is made of.
// Numbers: README.md here, DESIGN.md 3.1 "short K", profiles/archive/r04_shortk_ab.txt.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "oz2_gemm_common.hpp"
#include "oz2_gemm_i8_epi.hpp"
#include "oz2_kernels.h"

namespace oz2 {

constexpr int SK_THREADS = 512;            // 8 waves = 2 groups x 4 (one wave of each group per SIMD)
constexpr int SK_A_BYTES = 128 * BK;       // A half-panel: 128 rows x 128 B
constexpr int SK_SLOT = SK_A_BYTES + BN * BK;  // + B panel: 256 rows x 128 B = 48 KiB
constexpr int SK_SLOTS = 3;
constexpr int SK_DMA = SK_SLOT / 1024 / 4;  // 12 LDS-DMA instructions (1 KiB each) per wave of the fetching group and tick

// Epilogue hook of the group that is NOT in a K loop: wait for the LDS-DMA this wave still has in flight before its first store (the
// panels it fetched for the other group's second tick), and execute the `kt` workgroup barriers of the other group's K loop, spread
// evenly behind the 8 sub-blocks.
struct SkHook {
    int kt;      // barriers to execute during this epilogue (0: the last half-tile of the workgroup)
    int* slot;   // the wave's tick counter mod 3
    __device__ __forceinline__ void operator()(int s, int phase) const {
        if (phase == 0) {
            if (s == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        const int nb = ((s + 1) * kt) / 8 - (s * kt) / 8;  // wave-uniform
        for (int b = 0; b < nb; ++b) {
            __builtin_amdgcn_s_barrier();
            *slot = *slot == SK_SLOTS - 1 ? 0 : *slot + 1;
        }
    }
};

template <int EPI>
__global__ void __launch_bounds__(SK_THREADS) gemm_i8_shortk_kernel(const GemmArgs args) {
    static_assert(EPI == EPI_MOD || EPI == EPI_CPLX, "the bound GEMM keeps the 256 x 256 kernels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int KT = args.kp / BK;  // >= 2 (kp is a multiple of 256), one K segment
    const int total = args.total_tiles;
    const int G = gridDim.x;
    const int ntiles = (total - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup: blockIdx.x, + G, ...
    const int H = 2 * ntiles;                                   // its half-tiles: h = 2 i + (row half)

    // ---- LDS-DMA state: the half-tile / K-step to fetch next, its base pointers and this lane's 12 byte offsets
    auto uniform = [](const int8_t* ptr) {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const int8_t*)(((unsigned long long)hi << 32) | lo);
    };
    // Piece Q = 12 wn + q of a slot (instruction q of wave wn) is 1 KiB = rows 8 Q .. 8 Q + 7 of the 384-row slot (0-127: A rows, 128-383: B
    // rows), lane l = row 8 Q + (l >> 3), 16-byte chunk (l & 7) ^ swizzle(row); the swizzle (row >> 1) & 7 = 4 (Q & 1) | (l >> 4) depends on
    // the parity of Q only, so the source offset of a lane is  (l >> 3) kp + 16 chunk  (two VGPRs: even / odd Q) + a wave-uniform
    // 8 Q' kp + 128 kt added to the base pointer in SGPRs -- three address VGPRs instead of twelve (the kernel is register-bound: 128
    // accumulators + 64 fragment registers).  B_lo has exactly n rows: in the last tile column the piece that holds row nvalid - 1 and
    // the pieces behind it read the rows 8 Qb .. with the lane's row clamped to nvalid - 1 (third VGPR); those columns are never stored.
    const int8_t *gA = nullptr, *gB = nullptr;
    const int l3 = lane >> 3;
    unsigned lpE = 0, lpO = 0, lpC = 0;
    int qb = 48, rbrow = 0;  // first slot piece (16 ..) that is not made of 8 valid B rows (48 = none), and the row base the pieces from qb on read
    int fht = 0, fkt = 0;
    auto set_ht = [&](int ht) {
        const TileMap tm = map_tile((int)blockIdx.x + (ht >> 1) * G, total, args.tiles_m, args.tiles_n, args.colblock);
        const PlaneRef pr = plane_ref(args, tm.plane);
        gA = uniform(args.A[0] + pr.boff + (size_t)pr.tt * args.strideA + ((size_t)tm.tm * BM + (size_t)(ht & 1) * 128) * args.kp);
        gB = uniform(args.B[0] + pr.boff + (size_t)pr.tt * args.strideB + (size_t)tm.tn * BN * args.kp);
        const int nvalid = (args.n - tm.tn * BN) < BN ? (args.n - tm.tn * BN) : BN;
        const unsigned kp = (unsigned)args.kp;
        lpE = (unsigned)l3 * kp + (unsigned)(((lane & 7) ^ (l3 >> 1)) << 4);
        lpO = (unsigned)l3 * kp + (unsigned)(((lane & 7) ^ (4 | (l3 >> 1))) << 4);
        qb = 16 + (nvalid >> 3);  // pieces 16 .. qb - 1 hold 8 valid rows each
        rbrow = (nvalid - 1) & ~7;  // the piece that holds the last valid row
        const int lc = (nvalid - 1 - rbrow) < l3 ? (nvalid - 1 - rbrow) : l3;  // lane row clamped to the last valid row
        lpC = (unsigned)lc * kp + (unsigned)(((lane & 7) ^ ((4 * ((rbrow >> 3) & 1)) | (l3 >> 1))) << 4);
    };
    // issue pieces [q0, q1) of tick (fht, fkt) into slot `ds`
#define SK_ISSUE(q0_, q1_, ds_)                                                                                              \
    do {                                                                                                                     \
        _Pragma("unroll") for (int q = (q0_); q < (q1_); ++q) {                                                              \
            const int Q_ = wn * SK_DMA + q;                                                                                  \
            const bool edge_ = Q_ >= qb;                                                                                     \
            const int rows_ = Q_ < 16 ? Q_ * 8 : edge_ ? rbrow : (Q_ - 16) * 8;                                              \
            const int8_t* src_ = (Q_ < 16 ? gA : gB) + (size_t)rows_ * args.kp + (size_t)OZ2_HOOK_KSTEP(fkt) * BK;           \
            const unsigned off_ = edge_ ? lpC : (Q_ & 1) ? lpO : lpE;                                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_ + off_),                   \
                                             (__attribute__((address_space(3))) void*)(smem + (ds_) * SK_SLOT + Q_ * 1024), 16, 0, 0); \
        }                                                                                                                    \
    } while (0)
    auto advance = [&]() {
        if (++fkt == KT) {
            fkt = 0;
            ++fht;
            if (fht < H) set_ht(fht);
        }
    };

    const int r16 = lane & 15;
    const int q4 = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int a_off = r16 * BK;                              // + (ah * 4 + i) * 16 rows
    const int b_off = SK_A_BYTES + (wn * 64 + r16) * BK;     // + j * 16 rows
    const int c0 = ((q4) ^ sw) << 4, c1 = ((4 | q4) ^ sw) << 4;  // this lane's 16-byte chunk of the first / second K half

    int slot = 0;  // tick counter mod 3: the slot of the tick in progress

    // ---- prologue: group 0 fetches ticks 0 and 1
    if (grp == 0) {
        set_ht(0);
        SK_ISSUE(0, SK_DMA, 0);
        advance();
        if (fht < H) SK_ISSUE(0, SK_DMA, 1);
        if (fht < H) advance();
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // tick 0 has landed (tick 1 may still be in flight)
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) {  // idle through group 0's first K loop
        for (int kt = 0; kt < KT; ++kt) {
            __builtin_amdgcn_s_barrier();
            slot = slot == SK_SLOTS - 1 ? 0 : slot + 1;
        }
    }

    for (int i = 0; i < ntiles; ++i) {
        const int ht = 2 * i + grp;
        // the fetch cursor of this K loop starts at tick ht KT + 2
        {
            const int nht = KT == 2 ? ht + 1 : ht, nkt = KT == 2 ? 0 : 2;
            if (!(nht == fht && nkt == fkt)) {
                fht = nht, fkt = nkt;
                if (fht < H) set_ht(fht);
            }
        }
        v4i acc[8][4];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] = args.acc0;
        v4i af[2][4], bf[2][4];
#define SK_LA(buf_, sb_, ah_, co_)                                                                                           \
    _Pragma("unroll") for (int x = 0; x < 4; ++x) af[buf_][x] = *(const v4i*)((sb_) + a_off + ((ah_) * 4 + x) * 16 * BK + (co_))
#define SK_LB(buf_, sb_, co_) _Pragma("unroll") for (int x = 0; x < 4; ++x) bf[buf_][x] = *(const v4i*)((sb_) + b_off + x * 16 * BK + (co_))
// One MFMA segment = 16 MFMAs (4 A fragments x 4 B fragments) with the next segment's fragment loads and this segment's share of the
// LDS-DMA issued BEHIND its first four MFMAs: the compiler waits for every outstanding LDS read before the first MFMA that uses a loaded
// register (lgkmcnt(0): it does not count them), so the loads of segment s + 1 must not sit in front of segment s -- placed here they have
// the 12 remaining MFMAs (192 clocks) to land.
#define SK_MMA4(ab_, bb_, ah_, x0_, x1_)                                                                                     \
    _Pragma("unroll") for (int x = (x0_); x < (x1_); ++x) _Pragma("unroll") for (int y = 0; y < 4; ++y)                      \
        acc[(ah_) * 4 + x][y] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[ab_][x], bf[bb_][y], acc[(ah_) * 4 + x][y], 0, 0, 0)
#define SK_SEG(ab_, bb_, ah_, MID_)                                                                                          \
    do {                                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        SK_MMA4(ab_, bb_, ah_, 0, 1);                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        MID_;                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        SK_MMA4(ab_, bb_, ah_, 1, 4);                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    } while (0)
        __builtin_amdgcn_s_setprio(2);  // the K-loop wave owns its SIMD's issue slots; the partner is in an epilogue
        {
            const char* sb = smem + slot * SK_SLOT;  // first K-step: its panels are visible since the last barrier
            SK_LB(0, sb, c0);
            SK_LA(0, sb, 0, c0);
        }
        for (int kt = 0; kt < KT; ++kt) {
            const char* sb = smem + slot * SK_SLOT;
            const bool fetch = fht < H && OZ2_HOOK_DMA_ON(i == 0);  // wave-uniform
            const int ds = slot == 0 ? 2 : slot - 1;  // slot of tick t + 2 = (slot + 2) % 3
            SK_SEG(0, 0, 0, {
                SK_LA(1, sb, 1, c0);
                if (fetch) SK_ISSUE(0, 4, ds);
            });
            SK_SEG(1, 0, 1, {
                SK_LB(1, sb, c1);
                SK_LA(0, sb, 0, c1);
                if (fetch) SK_ISSUE(4, 8, ds);
            });
            SK_SEG(0, 1, 0, {
                SK_LA(1, sb, 1, c1);
                if (fetch) {
                    SK_ISSUE(8, SK_DMA, ds);
                    advance();
                }
            });
            // Every LDS read of tick t is complete; the panels of tick t + 1 have landed.  This wave issued them during tick t - 1 -- unless
            // this is the first K-step of the half-tile (the other group did, and waits for them in its epilogue): no VMEM wait then, which
            // would only wait for this wave's own epilogue stores of a moment ago (vmcnt counts stores as well).
            if (kt == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (fetch) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == SK_SLOTS - 1 ? 0 : slot + 1;
            const char* nb = smem + slot * SK_SLOT;
            const bool more_k = kt + 1 < KT;
            SK_SEG(1, 1, 1, {
                if (more_k) {
                    SK_LB(0, nb, c0);
                    SK_LA(0, nb, 0, c0);
                }
            });
        }
        __builtin_amdgcn_s_setprio(0);
#undef SK_LA
#undef SK_LB
#undef SK_MMA4
#undef SK_SEG
        // ---- epilogue beside the other group's K loop (its KT barriers inside; none after the workgroup's last half-tile)
        const TileMap tmap = map_tile((int)blockIdx.x + i * G, total, args.tiles_m, args.tiles_n, args.colblock);
#if OZ2_HOOK_SKIP_EPILOGUE  // laboratory probe: the accumulators stay live, the barriers of the epilogue are executed back to back
        (void)tmap;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(acc[a][b]));
        SkHook hk{ht == H - 1 ? 0 : KT, &slot};
        hk(0, 0);
        for (int sb8 = 0; sb8 < 8; ++sb8) hk(sb8, 1);
#else
        i8_epilogue<EPI, SkHook>(acc, args, plane_ref(args, tmap.plane), tmap.tm * BM + grp * 128, tmap.tn * BN + wn * 64, lane,
                                 SkHook{ht == H - 1 ? 0 : KT, &slot});
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
#undef SK_ISSUE
}

static int sk_num_cus() {
    static std::atomic<int> n{0};
    int v = n.load(std::memory_order_relaxed);
    if (!v) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 256;
        n.store(v = prop.multiProcessorCount, std::memory_order_relaxed);
    }
    return v;
}

template <int EPI> static hipError_t launch_sk(hipStream_t stream, const GemmArgs& a) {
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 0;
    if (!attr_set_dev[dev_].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_shortk_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_SLOTS * SK_SLOT);
        if (e != hipSuccess) return e;
        attr_set_dev[dev_].store(true, std::memory_order_release);
    }
    int grid = sk_num_cus() & ~7;  // persistent, one workgroup per CU; a multiple of 8 keeps block b on XCD b % 8 aligned with map_tile
    if (grid <= 0) grid = 8;
    if (a.total_tiles < grid) grid = a.total_tiles;
    hipLaunchKernelGGL(gemm_i8_shortk_kernel<EPI>, dim3(grid), dim3(SK_THREADS), SK_SLOTS * SK_SLOT, stream, a);
    return hipGetLastError();
}

// `a` is complete (launch<EPI> of oz2_gemm_i8.hip filled total_tiles, colblock, acc0, ppi, bstride); one K segment, kp <= OZ2_SHORTK_MAX_KP
hipError_t launch_gemm_i8_shortk(hipStream_t stream, const GemmArgs& a, int epi) {
    if (a.nseg != 1 || a.kp < 256) return hipErrorInvalidValue;
    return epi == EPI_CPLX ? launch_sk<EPI_CPLX>(stream, a) : launch_sk<EPI_MOD>(stream, a);
}

}  // namespace oz2
