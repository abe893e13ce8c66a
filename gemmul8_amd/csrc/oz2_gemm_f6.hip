// FP6 (e2m3) x FP6 -> FP32 "TN" GEMM on gfx950 MFMA for the residue GEMMs of the FP8 backend (round 5).
//
// The FP8 backend of the reference multiplies integers of magnitude <= 16 held as e4m3 (GEMMul8/src/mod.hpp:159-189; call sites
// src/matmult.hpp:307-389, loop src/gemmul8_real.hpp:159-181).  Every such integer v is ALSO exact as an e2m3 number: v / 8 has the
// 6-bit code sign << 5 | |v| (subnormal 0..7, normal 8..15, 16 = 2.0), and with E8M0 block scales 2^3 on both operands
// v_mfma_scale_f32_16x16x128_f8f6f4 (cbsz = blgp = 2) returns the integer product sums themselves, exact in the FP32 accumulators up
// to 2^24 like the e4m3 form (tools/ubench/f6_layout.hip checks field map, code map, scales and accumulation on the device:
// profiles/r05_f6_layout.txt).  CDNA4 runs the FP6 formats at the FP4 rate -- 16 cycles per instruction where e4m3 takes 32 -- and at
// the board's power cap a register-only loop of it sustains 6.8 POP/s on these integers against 4.5 for e4m3
// (tools/ubench/mfma_shapes.hip s16 / f16, profiles/r05_mfma_shapes_fp6.txt).  Same numbers in, same bits out: the three GEMMs per modulus,
// the epilogues (oz2_gemm_f8_epi.hpp) and every residue are those of oz2_gemm_f8.hip; only the operand planes' ENCODING differs, and it
// never leaves the workspace.  The accurate-mode bound GEMM multiplies genuine e4m3 values and stays on the e4m3 kernel.
//
// Operand planes ("FP6 panel images", written by the quantise kernels: oz2_scale.hip put_f6_planes): a plane is cut into row blocks of
// 256 rows (the CU tile; the last block of B has Rp = its rows rounded up to 16) and K-steps of 128 elements; panel (block tb, K-step kt)
// is the byte image of what the kernel wants in LDS, Rp * 96 bytes at  plane + tb * 256 * (3 kp / 4) + kt * Rp * 96:
//     X region: 16-byte slot q * Rp + r          = bytes  0..15 of the 24-byte fragment of (row r, K group q)   [ds_read_b128]
//     Y region: at 64 Rp; 8-byte slot (q >> 1) * 2 Rp + 2 r + (q & 1) = bytes 16..23 of that fragment            [ds_read_b64]
// where the fragment of (r, q) packs the codes of elements 32 q .. 32 q + 31 of the K-step, element e at bits 6 e .. 6 e + 5: exactly
// the six operand registers of lane (r & 15, q).  The LDS-DMA is then LINEAR (instruction j moves bytes 1024 j .. 1024 j + 1023 of
// the panel: eight full 128-byte lines, no address arithmetic), and both read patterns are bank-conflict free without a swizzle:
// the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots mod 256 B (slot index = r mod 16 plus a multiple of 16), the
// 32 lanes of a ds_read_b64 half-wave 32 distinct 8-byte slots (2 (r & 15) + (q & 1)).
//
// Tiling and schedule as oz2_gemm_f8.hip: persistent 256 x 256 tiles, 8 waves (2 x 4, wave tile 128 x 64, 128 accumulators), ping-pong
// LOAD / MFMA segments with the two halves one slot apart, the waves issue the LDS-DMA themselves in their LOAD segments; five 24 KiB
// panel slots (2.5 stages).  A K-step moves 48 KiB through the operand path for 2 x 32 MFMAs of 16 cycles.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "oz2_gemm_f8_epi.hpp"

namespace oz2 {

constexpr int F6_BKE = 128;          // elements per K-step (one MFMA)
constexpr int F6_ROWB = 96;          // bytes per row and K-step
constexpr int F6_SLOT = BM * F6_ROWB;  // 24 KiB per panel slot
typedef int v6i __attribute__((ext_vector_type(6)));
#ifndef OZ2_F6_SGB
#define OZ2_F6_SGB 1
#endif
#ifdef OZ2_F6_PINGPONG
constexpr int F6_NSLOT = 5;
#else
constexpr int F6_NSLOT = 6;
#endif

// The product kernel (v2): every wave is SELF-PIPELINED -- fragment sets double-buffered in its 256 registers, the reads of the next half K-step
// and its share of the LDS-DMA interleaved one by one with the MFMAs of the current half (sched_group_barrier) -- and ONE workgroup barrier per
// K-step.  With 16-cycle MFMAs a ping-pong LOAD segment (16 ds_reads + 6 DMA issues, 600+ cycles) is far longer than the partner's MFMA
// segment (256 cycles): the ping-pong form below ran the matrix pipes ~40 % busy (config 3: 3.47 POP/s).  Six 24 KiB panel slots = three full
// stages: panel A(g) in slot 2 (g % 3), B(g) behind it.  Per K-step g and wave (K-steps counted ACROSS tiles: the panel stream is continuous):
//   half 0: MFMAs of rows 0-63 (fragments aL, bcur: in registers) | reads of A rows 64-127 of panel g -> aH | DMA pieces 3-5 of panel g + 2
//   wait vmcnt(6) lgkmcnt(0); barrier B_g                            (panel g + 1 has landed for everyone; every read of panel g is complete)
//   half 1: MFMAs of rows 64-127 (aH, bcur) | reads of B(g + 1) -> bnext, A rows 0-63 of panel g + 1 -> aL | DMA pieces 0-2 of panel g + 3
// Hazards: the slot of panel g + 3 is the slot of panel g, whose last reads are complete before B_g (RAW on the fragments: lgkmcnt(0));
// pieces 0-2 / 3-5 of panel g + 2 are issued after B_{g-1} (slot of panel g - 1, free since then) and waited for before B_{g+1}
// (vmcnt(6): only the six pieces of panel g + 3 may be outstanding) -- one to 1.5 K-steps of latency budget for every piece.
// The last K-step of a tile prefetches nothing (the epilogue needs the registers): a tile's first fragments are read exposed.  The fetch stream
// runs three panels ahead of the consumers and enters the tile `lead` tiles ahead exactly once per consumer tile: that tile's addresses are
// computed at the consumer's tile start (scalar registers), so that the K loop itself contains no tile arithmetic.
template <int EPI>
__global__ void __launch_bounds__(F8_THREADS) gemm_f6_kernel(const F8Args args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT1 = args.kp / F6_BKE;  // K-steps per segment
    const int KT = KT1 * args.nseg;    // K-steps per tile (even: kp is a multiple of 256)
    const int total = args.total_tiles;
    const int G = gridDim.x;
    const size_t blockbytes = (size_t)BM * (size_t)(args.kp / 4 * 3);  // one 256-row block of a plane

    const bool isB = wave < 4;  // waves 0-3 fetch B panels, waves 4-7 A panels: six linear 1 KiB pieces each per panel
    const unsigned dbase = (unsigned)((wave & 3) * 6 * 1024 + lane * 16);  // this lane's byte in piece 0 of the wave; piece i: + 1024 i, clamped to the image
    unsigned dlast = 0;    // byte offset of the last 16-byte chunk of the panel being fetched
    const int8_t* gsrc;
    int gstep = 0;         // bytes between consecutive K-steps of the block: Rp * 96
    long long gdelta = 0;  // nseg >= 2: from the panel of K-step KT1 + j of segment 1's plane to K-step j of segment 2's plane
    long long gdelta2 = 0; // nseg == 3: the same for K-step 2 KT1 + j and segment 3's plane
    auto uniform = [](const int8_t* ptr) {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const int8_t*)(((unsigned long long)hi << 32) | lo);
    };
    auto rows_pad = [&](int tn) {  // rows of B's block tn in its panel images
        const int nr = args.n - tn * BN;
        return nr >= BN ? BN : ((nr + 15) & ~15);
    };
#define F6_SET_TILE(vb_)                                                                                                     \
    do {                                                                                                                     \
        const TileMap tmap_ = map_tile((vb_), total, args.map);                                                              \
        const F8Plane pl_ = f8_plane(args, tmap_.plane);                                                                     \
        const int rp_ = isB ? rows_pad(tmap_.tn) : BM;                                                                       \
        gstep = rp_ * F6_ROWB;                                                                                               \
        dlast = (unsigned)(rp_ * F6_ROWB - 16);                                                                              \
        gdelta2 = isB ? ((long long)args.planeB3[pl_.tt] - args.planeB[pl_.tt]) * (long long)args.strideB - 2ll * KT1 * gstep \
                      : ((long long)args.planeA3[pl_.tt] - args.planeA[pl_.tt]) * (long long)args.strideA - 2ll * KT1 * gstep; \
        gsrc = uniform(isB ? args.B + pl_.boff + (size_t)args.planeB[pl_.tt] * args.strideB + (size_t)tmap_.tn * blockbytes   \
                           : args.A + pl_.boff + (size_t)args.planeA[pl_.tt] * args.strideA + (size_t)tmap_.tm * blockbytes); \
        gdelta = isB ? ((long long)args.planeB2[pl_.tt] - args.planeB[pl_.tt]) * (long long)args.strideB - (long long)KT1 * gstep \
                     : ((long long)args.planeA2[pl_.tt] - args.planeA[pl_.tt]) * (long long)args.strideA - (long long)KT1 * gstep; \
    } while (0)
    // (the 32-bit lane offset passes through an empty asm so that its zero-extension stays next to the load: SGPR base + VGPR offset addressing.
    //  Hoisted out of the loop as a 64-bit value it costs a v_lshl_add_u64 and a register pair per piece)
#define F6_DMA(src_, q_, stage_)                                                                                             \
    do {                                                                                                                     \
        unsigned off_ = min(dbase + (q_) * 1024u, dlast);                                                                    \
        asm volatile("" : "+v"(off_));                                                                                       \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((src_) + off_),                     \
                                         (__attribute__((address_space(3))) void*)((stage_) + ((wave & 3) * 6 + (q_)) * 1024), 16, 0, 0); \
    } while (0)

    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int ax = (q * BM + wm * 128 + r16) * 16;
    const int ay = 64 * BM + ((q >> 1) * 2 * BM + 2 * (wm * 128 + r16) + (q & 1)) * 8;
    constexpr int SC3 = (int)0x82828282u;  // E8M0 scale 2^3 for every block
    typedef int v2i __attribute__((ext_vector_type(2)));
    // a fragment = the six operand registers: 16 bytes of the X region (LDS byte address xb_ + constant o_) + 8 bytes of the Y region (yb_ + o_).
    // Both reads are inline assembly.  Written as loads, the compiler fuses two fragments' 8-byte reads into one ds_read2_b64 and hoists the 16-byte
    // read of an in-place reload above the MFMAs that still read the old fragment -- either way registers have to be waited for (lgkmcnt(0)) and
    // COPIED in the middle of the MFMA stream.  The compiler's waitcnt pass does not see these reads: the K-step waits for them itself, region by
    // region (LDS operations of a wave return in order).
#define F6_FRAG(dst_, xb_, yb_, o_)                                                                                          \
    do {                                                                                                                     \
        v4i lo_;                                                                                                             \
        v2i hi_;                                                                                                             \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lo_) : "v"(xb_), "n"(o_));                                       \
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(hi_) : "v"(yb_), "n"(o_));                                        \
        dst_ = v6i{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1]};                                                          \
    } while (0)
    // the same as plain loads (visible to the compiler: waited for by it, safe to spill): only where a K-step runs into an epilogue
#define F6_FRAG_PLAIN(dst_, px_, py_, o_)                                                                                    \
    do {                                                                                                                     \
        const v4i lo_ = *(const v4i*)((px_) + (o_));                                                                         \
        const v2i hi_ = *(const v2i*)((py_) + (o_));                                                                         \
        dst_ = v6i{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1]};                                                          \
    } while (0)
    const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)smem;
    // (the builtin takes 8-register operands for every format; with cbsz / blgp = 2 the instruction reads the low six -- the upper two are left
    // undefined so that a fragment costs six registers also where it lives across loop iterations)
#define F6_MFMA(a_, b_, c_)                                                                                                  \
    c_ = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(a_, a_, 0, 1, 2, 3, 4, 5, -1, -1),         \
                                                          __builtin_shufflevector(b_, b_, 0, 1, 2, 3, 4, 5, -1, -1), c_, 2, 2, 0, SC3, 0, SC3)

    auto run = [&]<bool ISB>() {
        int vb_next = blockIdx.x, kt_next = 0;
        bool more = true;
        int fs = 0;  // (panel being fetched) % 3
        const int8_t* fsrc;
        char* fdst;
        F6_SET_TILE(vb_next);
#define F6_FETCH_ADVANCE()                                                                                                   \
    do {                                                                                                                     \
        fs = fs == 2 ? 0 : fs + 1;                                                                                           \
        if (more && ++kt_next == KT) {                                                                                       \
            kt_next = 0;                                                                                                     \
            vb_next += G;                                                                                                    \
            more = vb_next < total;                                                                                          \
            if (more) F6_SET_TILE(vb_next);                                                                                  \
            else kt_next = KT - 1;                                                                                           \
        }                                                                                                                    \
    } while (0)
        // the same step inside the K loop: the state of the tile the stream enters next was prepared at the consumer's tile start
        const int8_t* gsrcN = nullptr;
        int gstepN = 0;
        long long gdeltaN = 0, gdelta2N = 0;
        unsigned dlastN = 0;
#define F6_FETCH_ADVANCE_LOOP()                                                                                              \
    do {                                                                                                                     \
        fs = fs == 2 ? 0 : fs + 1;                                                                                           \
        if (++kt_next == KT) kt_next = 0, gsrc = gsrcN, gstep = gstepN, gdelta = gdeltaN, gdelta2 = gdelta2N, dlast = dlastN; \
    } while (0)
#define F6_FETCH_PTRS()                                                                                                      \
    do {                                                                                                                     \
        fsrc = gsrc + (long long)OZ2_HOOK_KSTEP(kt_next) * gstep + (kt_next >= 2 * KT1 ? gdelta2 : kt_next >= KT1 ? gdelta : 0); \
        fdst = smem + (2 * fs + (ISB ? 1 : 0)) * F6_SLOT;                                                                    \
    } while (0)
        // prologue: panels 0 and 1 whole, pieces 0-2 of panel 2
        F6_FETCH_PTRS();
#pragma unroll
        for (int i = 0; i < 6; ++i) F6_DMA(fsrc, i, fdst);
        F6_FETCH_ADVANCE();
        F6_FETCH_PTRS();
#pragma unroll
        for (int i = 0; i < 6; ++i) F6_DMA(fsrc, i, fdst);
        F6_FETCH_ADVANCE();
        F6_FETCH_PTRS();
#pragma unroll
        for (int i = 0; i < 3; ++i) F6_DMA(fsrc, i, fdst);
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        v6i bf[4], aL[4], aH[4];
        int cs = 0;  // (current K-step) % 3
        const int lead = (3 + KT - 1) / KT;  // tiles between the consumers' tile and the one the fetch stream enters during it (1 for KT >= 4)
        // the prologue left the stream in tile vb_next; re-base it on (consumer tile, lead): from here on every wrap takes the prepared state
        // One K-step.  PF_: prefetch the next K-step's first fragments.  The instruction stream is cut into scheduling regions of four MFMAs (one COLUMN
        // of B fragments, both halves) with the reads and the DMA piece that go beside them; the regions are what the explicit waits count (LDS
        // operations of a wave return in order; inside a region the scheduler may order the plain 16-byte reads freely, which only makes a wait stronger):
        //   half 1 issues   R0 = {aL[0], aL[1], bf[0]}  R1 = {aL[2], aL[3], bf[1]}  R2 = {bf[2]}  R3 = {bf[3]}   (two reads per fragment; bf[j] into the
        //   registers of the column whose four MFMAs were just issued), 16 reads; the next half 0 then needs lgkmcnt(10) (R0 landed) for its first two
        //   MFMAs, (4) (R1) for the next two and column 1, and -- with the reads of aH (rows 64-127) issued behind columns 1 and 2 -- (6) for column 2
        //   and (8) for column 3: the read issued behind the previous K-step's LAST MFMA is needed thirteen MFMAs into this one.
        // lgkmcnt(0) in front of the barrier (every read of panel g complete), vmcnt(6) (the wave's pieces of panel g + 1 landed).
#define F6_KSTEP(PF_)                                                                                                        \
    do {                                                                                                                     \
        const unsigned xcA_ = lds0 + (unsigned)((2 * cs) * F6_SLOT + ax), ycA_ = lds0 + (unsigned)((2 * cs) * F6_SLOT + ay);  \
        cs = cs == 2 ? 0 : cs + 1;                                                                                           \
        const unsigned xnA_ = lds0 + (unsigned)((2 * cs) * F6_SLOT + ax), ynA_ = lds0 + (unsigned)((2 * cs) * F6_SLOT + ay);  \
        const unsigned xnB_ = lds0 + (unsigned)((2 * cs + 1) * F6_SLOT + bx), ynB_ = lds0 + (unsigned)((2 * cs + 1) * F6_SLOT + by); \
        /* ---- half 0 (rows 0-63), column 0 */                                                                              \
        asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(bf[0]), "+v"(aL[0]), "+v"(aL[1]));                                       \
        F6_MFMA(aL[0], bf[0], acc[0][0]);                                                                                    \
        F6_MFMA(aL[1], bf[0], acc[1][0]);                                                                                    \
        __builtin_amdgcn_sched_barrier(0); /* (the second wait stays behind the two MFMAs that do not need it) */            \
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bf[1]), "+v"(aL[2]), "+v"(aL[3]));                                        \
        F6_MFMA(aL[2], bf[0], acc[2][0]);                                                                                    \
        F6_MFMA(aL[3], bf[0], acc[3][0]);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        /* ---- half 0, columns 1-3 */                                                                                       \
        _Pragma("unroll") for (int j = 1; j < 4; ++j) {                                                                      \
            if (j == 2 && (PF_)) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bf[2]));                                         \
            if (j == 3 && (PF_)) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bf[3]));                                         \
            if (j >= 2 && !(PF_)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[2]), "+v"(bf[3]));                           \
            if (dma_on) F6_DMA(fsrc, 2 + j, fdst);                                                                           \
            if (j < 3) {                                                                                                     \
                if (OZ2_HOOK_SKIP_AH) { /* laboratory probe: a third of the fragment reads not issued (stale registers) */     \
                } else if (PF_) {                                                                                            \
                    F6_FRAG(aH[2 * j - 2], xcA_, ycA_, (4 + 2 * j - 2) * 256);                                               \
                    F6_FRAG(aH[2 * j - 1], xcA_, ycA_, (4 + 2 * j - 1) * 256);                                               \
                } else { /* the tile's last K-step runs into the epilogue: there the compiler may spill, and it must KNOW these reads */ \
                    F6_FRAG_PLAIN(aH[2 * j - 2], smem + xcA_ - lds0, smem + ycA_ - lds0, (4 + 2 * j - 2) * 256);             \
                    F6_FRAG_PLAIN(aH[2 * j - 1], smem + xcA_ - lds0, smem + ycA_ - lds0, (4 + 2 * j - 1) * 256);             \
                }                                                                                                            \
            }                                                                                                                \
            _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                               \
                const int i = (j & 1) ? 3 - ii : ii;                                                                         \
                F6_MFMA(aL[i], bf[j], acc[i][j]);                                                                            \
            }                                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
        }                                                                                                                    \
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        F6_FETCH_ADVANCE_LOOP();                                                                                             \
        F6_FETCH_PTRS();                                                                                                     \
        /* ---- half 1 (rows 64-127) */                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
            if (dma_on && j < 3) F6_DMA(fsrc, j, fdst);                                                                      \
            if (PF_ && j < 2) {                                                                                              \
                F6_FRAG(aL[2 * j], xnA_, ynA_, (2 * j) * 256);                                                               \
                F6_FRAG(aL[2 * j + 1], xnA_, ynA_, (2 * j + 1) * 256);                                                       \
            }                                                                                                                \
            _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                               \
                const int i = (j & 1) ? 3 - ii : ii;                                                                         \
                F6_MFMA(aH[i], bf[j], acc[4 + i][j]);                                                                        \
            }                                                                                                                \
            __builtin_amdgcn_sched_barrier(0); /* the in-place reload stays behind the column's MFMAs */                     \
            if (PF_) F6_FRAG(bf[j], xnB_, ynB_, j * 256);                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
        }                                                                                                                    \
    } while (0)
        for (int vb = blockIdx.x; vb < total; vb += G) {
            const TileMap tmap = map_tile(vb, total, args.map);
            {   // state of the tile the fetch stream enters during this tile (none left: the current one again, into slots nobody reads)
                const int8_t* gs_ = gsrc;
                const int gt_ = gstep;
                const long long gd_ = gdelta, gd2_ = gdelta2;
                const unsigned dl_ = dlast;
                const int vbn_ = vb + lead * G;
                if (vbn_ < total) F6_SET_TILE(vbn_);
                gsrcN = gsrc, gstepN = gstep, gdeltaN = gdelta, gdelta2N = gdelta2, dlastN = dlast;
                gsrc = gs_, gstep = gt_, gdelta = gd_, gdelta2 = gd2_, dlast = dl_;
            }
            const int rpB = rows_pad(tmap.tn);
            const int bx = (q * rpB + wn * 64 + r16) * 16;
            const int by = 64 * rpB + ((q >> 1) * 2 * rpB + 2 * (wn * 64 + r16) + (q & 1)) * 8;
            const bool dma_on = OZ2_HOOK_DMA_ON(vb == (int)blockIdx.x);  // (laboratory hook: always true in the product)
            v4f acc[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;
            {  // the tile's first fragments, exposed (their panel landed before the last barrier)
                const unsigned xbA_ = lds0 + (unsigned)((2 * cs) * F6_SLOT + ax), ybA_ = lds0 + (unsigned)((2 * cs) * F6_SLOT + ay);
                const unsigned xbB_ = lds0 + (unsigned)((2 * cs + 1) * F6_SLOT + bx), ybB_ = lds0 + (unsigned)((2 * cs + 1) * F6_SLOT + by);
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // (in the regions the K-step's waits count: {aL0, aL1, bf0} {aL2, aL3, bf1} {bf2} {bf3})
                    if (j < 2) {
                        F6_FRAG(aL[2 * j], xbA_, ybA_, (2 * j) * 256);
                        F6_FRAG(aL[2 * j + 1], xbA_, ybA_, (2 * j + 1) * 256);
                    }
                    F6_FRAG(bf[j], xbB_, ybB_, j * 256);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (EPI == EPI_FUSED || EPI == EPI_FUSED_CPLX) {
                // three K segments (the modulus' three products); between them the accumulators -- exact integers -- are replaced by m x (their loose
                // residue mod p), three fp32 instructions + the product each (f8_fill_planes has the algebra, oz2_gemm_f8_epi.hpp red_acc the bound)
                const F8Plane plc = f8_plane(args, tmap.plane);
                const float pf = (float)args.moduli[args.t_begin + plc.tt], invp = 1.0f / pf;
                float mm = args.m1[plc.tt];
                int kseg = KT1;  // K-steps until the next transform
                for (int kt = 0; kt + 1 < KT; ++kt) {
                    F6_KSTEP(true);
                    if (--kseg == 0) {  // (uniform, twice per tile)
                        // the fragments prefetched for the next K-step are inline-asm reads the compiler does not know to be in flight: were it to
                        // spill one around the arithmetic below it would store a register that has not landed -- so they land first
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(aL[0]), "+v"(aL[1]), "+v"(aL[2]), "+v"(aL[3]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]));
#pragma unroll
                        for (int i = 0; i < 8; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float c = acc[i][j][r];
                                    acc[i][j][r] = mm * fmaf(-rintf(c * invp), pf, c);
                                }
                        mm = args.m2[plc.tt];
                        kseg = KT1;
                    }
                }
                F6_KSTEP(false);
            } else {
                for (int kt = 0; kt + 1 < KT; ++kt) F6_KSTEP(true);
                F6_KSTEP(false);
            }
#if OZ2_HOOK_SKIP_EPILOGUE
            (void)tmap;  // laboratory probe: no epilogue; the accumulators stay live
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
            const int i0 = tmap.tm * BM + wm * 128, j0 = tmap.tn * BN + wn * 64;
            const F8Plane pl = f8_plane(args, tmap.plane);
            f8_epilogue_mod<EPI>(acc, args, pl, i0, j0, lane);
#endif
        }
    };
    if (isB) run.template operator()<true>();
    else run.template operator()<false>();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
#undef F6_SET_TILE
#undef F6_DMA
#undef F6_FRAG
#undef F6_FRAG_PLAIN
#undef F6_MFMA
#undef F6_KSTEP
#undef F6_FETCH_PTRS
#undef F6_FETCH_ADVANCE
#undef F6_FETCH_ADVANCE_LOOP
}

#ifdef OZ2_F6_PINGPONG  // the first form (round 5, kept for A/B builds: tools/build_probes.sh SRC=oz2_gemm_f6 pp="-DOZ2_F6_PINGPONG"): ping-pong LOAD / MFMA segments as in oz2_gemm_f8.hip, five slots
template <int EPI>
__global__ void __launch_bounds__(F8_THREADS) gemm_f6_pingpong_kernel(const F8Args args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT1 = args.kp / F6_BKE;  // K-steps per segment
    const int KT = KT1 * args.nseg;    // K-steps per tile
    const int total = args.total_tiles;
    const int G = gridDim.x;
    const size_t blockbytes = (size_t)BM * (size_t)(args.kp / 4 * 3);  // one 256-row block of a plane

    // LDS-DMA: a panel is 24 linear instructions of 1 KiB; the four waves fetching an operand issue six each (a shorter last block of B:
    // lanes beyond the image re-read its last chunk).  Waves 0-3 fetch B (needed one K-step after issue), waves 4-7 A (two K-steps ahead).
    const bool isB = wave < 4;
    unsigned doff[6];
    const int8_t* gsrc;
    int gstep = 0;         // bytes between consecutive K-steps of the block: Rp * 96
    long long gdelta = 0;  // nseg == 2: from the panel of K-step KT1 + j of segment 1's plane to K-step j of segment 2's plane
    auto uniform = [](const int8_t* ptr) {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const int8_t*)(((unsigned long long)hi << 32) | lo);
    };
    auto rows_pad = [&](int tn) {  // rows of B's block tn in its panel images
        const int nr = args.n - tn * BN;
        return nr >= BN ? BN : ((nr + 15) & ~15);
    };
#define F6_SET_TILE(vb_)                                                                                                     \
    do {                                                                                                                     \
        const TileMap tmap_ = map_tile((vb_), total, args.map);                                                              \
        const F8Plane pl_ = f8_plane(args, tmap_.plane);                                                                     \
        const int rp_ = isB ? rows_pad(tmap_.tn) : BM;                                                                       \
        gstep = rp_ * F6_ROWB;                                                                                               \
        gsrc = uniform(isB ? args.B + pl_.boff + (size_t)args.planeB[pl_.tt] * args.strideB + (size_t)tmap_.tn * blockbytes   \
                           : args.A + pl_.boff + (size_t)args.planeA[pl_.tt] * args.strideA + (size_t)tmap_.tm * blockbytes); \
        gdelta = isB ? ((long long)args.planeB2[pl_.tt] - args.planeB[pl_.tt]) * (long long)args.strideB - (long long)KT1 * gstep \
                     : ((long long)args.planeA2[pl_.tt] - args.planeA[pl_.tt]) * (long long)args.strideA - (long long)KT1 * gstep; \
        const int last_ = rp_ * 6 - 1;                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                                                      \
            const int c_ = ((wave & 3) * 6 + q) * 64 + lane;                                                                 \
            doff[q] = (unsigned)(c_ < last_ ? c_ : last_) * 16u;                                                             \
        }                                                                                                                    \
    } while (0)
#define F6_DMA(src_, q_, stage_)                                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((src_) + doff[q_]),                     \
                                     (__attribute__((address_space(3))) void*)((stage_) + ((wave & 3) * 6 + (q_)) * 1024), 16, 0, 0)

    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    // this lane's fragment pieces inside a panel image (see the header): A blocks always have 256 rows
    const int ax = (q * BM + wm * 128 + r16) * 16;
    const int ay = 64 * BM + ((q >> 1) * 2 * BM + 2 * (wm * 128 + r16) + (q & 1)) * 8;

    auto frag = [&](const char* px, const char* py) {  // 24 bytes: the six operand registers
        const v4i lo = *(const v4i*)px;
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i hi = *(const v2i*)py;
        return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], 0, 0};
    };
    constexpr int SC3 = (int)0x82828282u;  // E8M0 scale 2^3 for every block: (v / 8 * 8) (w / 8 * 8)

    auto run = [&]<bool ISB>() {
        int vb_next = blockIdx.x, kt_next = 0;
        bool more = true;
        int hs = ISB ? 1 : 0;  // slot of the panel to fetch
        const int8_t* fsrc;
        char* fdst;
        F6_SET_TILE(vb_next);
#define F6_FETCH_ADVANCE()                                                                                                   \
    do {                                                                                                                     \
        hs = hs + 2 >= F6_NSLOT ? hs + 2 - F6_NSLOT : hs + 2;                                                                \
        if (more && ++kt_next == KT) {                                                                                       \
            kt_next = 0;                                                                                                     \
            vb_next += G;                                                                                                    \
            more = vb_next < total;                                                                                          \
            if (more) F6_SET_TILE(vb_next);                                                                                  \
            else kt_next = KT - 1;                                                                                           \
        }                                                                                                                    \
    } while (0)
#define F6_FETCH_BEGIN()                                                                                                     \
    do {                                                                                                                     \
        fsrc = gsrc + (long long)OZ2_HOOK_KSTEP(kt_next) * gstep + (kt_next >= KT1 ? gdelta : 0);                            \
        fdst = smem + hs * F6_SLOT;                                                                                          \
    } while (0)
        F6_FETCH_BEGIN();
#pragma unroll
        for (int i = 0; i < 6; ++i) F6_DMA(fsrc, i, fdst);
        if constexpr (!ISB) {
            F6_FETCH_ADVANCE();
            F6_FETCH_BEGIN();
#pragma unroll
            for (int i = 0; i < 6; ++i) F6_DMA(fsrc, i, fdst);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        // Hazards as in oz2_gemm_f8.hip: a slot was last read two (A) / one (B) K-steps before its refill is issued; every wave finishes a
        // K-step's LOAD segments (lgkmcnt(0) + barrier) before the leading half enters the next one; each wave drains the DMA the NEXT
        // K-step needs in its last LOAD segment (A waves: everything but the 6 instructions just issued).
        int sA = 0;  // slot of A(g); B(g) sits in the next slot
        for (int vb = blockIdx.x; vb < total; vb += G) {
            const TileMap tmap = map_tile(vb, total, args.map);
            const int rpB = rows_pad(tmap.tn);
            const int bx = (q * rpB + wn * 64 + r16) * 16;
            const int by = 64 * rpB + ((q >> 1) * 2 * rpB + 2 * (wn * 64 + r16) + (q & 1)) * 8;
            v4f acc[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

            for (int kt = 0; kt < KT; ++kt) {
                const char* curA = smem + sA * F6_SLOT;
                const char* curB = smem + (sA == F6_NSLOT - 1 ? 0 : sA + 1) * F6_SLOT;
                sA = sA + 2 >= F6_NSLOT ? sA + 2 - F6_NSLOT : sA + 2;
                F6_FETCH_ADVANCE();
                F6_FETCH_BEGIN();
                v8i bf[4];
#pragma unroll
                for (int ah = 0; ah < 2; ++ah) {  // LOAD segment ah of this K-step
                    v8i af[4];
                    if (OZ2_HOOK_DMA_ON(vb == (int)blockIdx.x)) {  // (laboratory hook: always true in the product)
                        if constexpr (ISB) {
                            if (ah == 0) {
#pragma unroll
                                for (int i = 0; i < 6; ++i) F6_DMA(fsrc, i, fdst);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 3; ++i) F6_DMA(fsrc, ah * 3 + i, fdst);
                        }
                    }
                    if (ah == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bf[j] = frag(curB + bx + j * 256, curB + by + j * 256);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = frag(curA + ax + (ah * 4 + i) * 256, curA + ay + (ah * 4 + i) * 256);
                    if (ah == 1) {
                        if (!ISB) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (ah == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = (i & 1) ? 3 - jj : jj;  // serpentine, as in the INT8 / e4m3 kernels
                            acc[ah * 4 + i][j] =
                                __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bf[j], acc[ah * 4 + i][j], 2, 2, 0, SC3, 0, SC3);
                        }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ah == 1 && ISB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int i0 = tmap.tm * BM + wm * 128, j0 = tmap.tn * BN + wn * 64;
            const F8Plane pl = f8_plane(args, tmap.plane);
            f8_epilogue_mod<EPI>(acc, args, pl, i0, j0, lane);
        }
    };
    if (isB) run.template operator()<true>();
    else run.template operator()<false>();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef F6_SET_TILE
#undef F6_DMA
#undef F6_FETCH_BEGIN
#undef F6_FETCH_ADVANCE
}

#endif

static int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 256;
        n = prop.multiProcessorCount;
    }
    return n;
}

template <int EPI> static hipError_t launch(hipStream_t stream, F8Args& a, int planes) {
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 0;
    if (!attr_set_dev[dev_].load(std::memory_order_acquire)) {
#ifdef OZ2_F6_PINGPONG
#define gemm_f6_kernel gemm_f6_pingpong_kernel
#endif
        hipError_t e = hipFuncSetAttribute((const void*)gemm_f6_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, F6_NSLOT * F6_SLOT);
        if (e != hipSuccess) return e;
        attr_set_dev[dev_].store(true, std::memory_order_release);
    }
    a.ppi = planes;
    a.m_ppi = map_magic((unsigned)planes);
    a.bstride = g_batch.ws;
    planes *= (int)g_batch.batch;
    a.total_tiles = planes * a.tiles_m * a.tiles_n;
    if (a.total_tiles <= 0) return hipSuccess;
    if (a.nseg < 1) a.nseg = 1;
    if (a.nres < 1) a.nres = 2;
    a.colblock = map_colblock((size_t)a.tiles_n, (size_t)a.kp / 4 * 3 * (size_t)a.nseg);
    a.map = make_tile_map(a.tiles_m, a.tiles_n, a.colblock);
    int grid = num_cus() & ~7;  // persistent: one workgroup per CU (see oz2_gemm_i8.hip)
    if (grid <= 0) grid = 8;
    if (a.total_tiles < grid) grid = a.total_tiles;
    hipLaunchKernelGGL(gemm_f6_kernel<EPI>, dim3(grid), dim3(F8_THREADS), F6_NSLOT * F6_SLOT, stream, a);
    return hipGetLastError();
}

// The residue GEMMs of launch_gemm_f8 (same `which`, same plane algebra: see there) on FP6 panel images.
hipError_t launch_gemm_f6(hipStream_t stream, int which, const int8_t* A, const int8_t* B, size_t strideA, size_t strideB, size_t kp, size_t m,
                          size_t n, int t_begin, int t_end, int16_t* out, size_t ldo, size_t strideO, const int16_t* r0, const int16_t* r1,
                          size_t strideR, const int16_t* rx, const int16_t* ry) {
    F8Args a{};
    if (f8_fill_planes(a, which, A, B, strideA, strideB, kp, m, n, t_begin, t_end, out, ldo, strideO, r0, r1, strideR, rx, ry) != 0) return hipErrorInvalidValue;
    const int planes = t_end - t_begin;
    if (which == 7) return launch<EPI_FUSED>(stream, a, planes);
    if (which == 8) return launch<EPI_FUSED_CPLX>(stream, a, planes);
    if (which == 3 || which == 6) return launch<EPI_FINAL_CPLX>(stream, a, planes);
    return (which == 2 || which == 5) ? launch<EPI_FINAL>(stream, a, planes) : launch<EPI_PART>(stream, a, planes);
}

}  // namespace oz2
