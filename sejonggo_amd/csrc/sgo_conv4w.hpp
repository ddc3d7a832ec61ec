// sgo_conv4w.hpp -- second hand-written kernel for the residual tower's 3x3 / 256 -> 256 'same' convolution with bias
// (+ skip) + ReLU fused (model.py:37-46).  Same math, layouts and per-wave MFMA tile as sgo_conv8w.hpp; different
// OCCUPANCY model:
//
//   k_conv8w: one 512-thread workgroup per CU (256 pixels x 256 channels, 150 KiB of LDS).  Its prologue (7.4k cycles) and
//             epilogue (11.5k) run with the CU's MFMA pipes idle -- 17 % of a tile's 109k cycles -- because nothing else
//             fits on the CU beside it.
//   k_conv4w: 256-thread workgroups of 256 pixels x 128 channels (4 waves = 2 pixel groups x 2 channel groups, one wave per
//             SIMD) in 78 KiB of LDS, so TWO workgroups share a CU: while one is in its prologue / epilogue / a staging
//             bubble, the other one's waves issue MFMAs on the same SIMDs.  The price: the pixel window is single-buffered
//             (restaged at the three chunk boundaries behind the last tap's MFMAs) and staged once per channel half
//             (L2 -> LDS bytes per MAC +12 %).
//
// Per K-tile (tap T of 64-channel chunk cc; weights buffer BUF = K-tile parity), all four waves in step:
//   phase A: read chan-lo, chan-hi fragments (weights[t]) and pixel-lo fragments | barrier (buffer of weights[t] is free)
//            | 32 MFMA
//   phase B: read pixel-hi fragments | stage weights[t+2] into the freed buffer | vmcnt(4): weights[t+1] have landed
//            | barrier (they are visible) | [last tap of a chunk: stage the next chunk's window] | 32 MFMA
//            | [last tap of a chunk: vmcnt(0), barrier]
// A wave issues exactly four weight DMAs per K-tile, so the counted wait is always vmcnt(4); chunk boundaries drain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgo_conv4w {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx2 __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));

#define S4_MFMA_PRIO (VAR & 1)
#define S4_AS1 __attribute__((address_space(1)))
#define S4_AS3 __attribute__((address_space(3)))

constexpr int CIN = 256, COUT = 256, CT = 128;   // CT: output channels per workgroup
constexpr int ROWB = CIN * 2, WROWB = 9 * CIN * 2, MAXW = 19;
// LDS map: two weight buffers (128 rows x 128 B; first, so that every weight fragment address is one base register + a 16-bit
// immediate: 6 VGPRs less), window (320 rows x 128 B), zero area
constexpr int LB0 = 0, LB1 = 16384, LW = 32768, LZ = 73728, LZ_BYTES = 3 * 2048 + 256, LDS_BYTES = LZ + LZ_BYTES;

#define S4_DS_READ64(dst, addr, OFF) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define S4_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define S4_DS_WRITE64(addr, val, OFF) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(OFF) : "memory")
#define S4_LGKM0()                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)
#define S4_VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define S4_GLOAD128(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define S4_DS_WRITE128(addr, val, OFF) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(OFF) : "memory")
#define S4_BARRIER()                   \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0)

// VAR: schedule variants kept selectable for A/B runs in one process (sgo_conv_tower_kernel(16 + VAR)); 7 = all on = default
//   bit 0: s_setprio(1) around the MFMA bursts; bit 1: split wait at chunk boundaries (early pieces now, late pieces one phase
//   later); bit 2: early restage of the dead window rows [0, 128) during the last tap's phase A; bit 3 (round 3, with 7 only):
//   the weights travel global -> REGISTERS -> LDS (4 x global_load_dwordx4 a K-tile ahead, 4 x ds_write_b128 into the buffer
//   barrier 1 has freed) instead of by LDS-DMA -- an LDS-DMA piece costs its wave 60-185 issue cycles inside a phase that also
//   carries fragment reads (guide, per-instruction constants), a load + a 16-byte LDS store ~20.  Same-process A/B at
//   8192 x 17 x 17 (TFLOP/s): 0 -> 1312, 4 -> 1305, 5 -> 1314, 6 -> 1324, 7 -> 1342 (k_conv8w: 1310).  Measured and
//   dropped: refilling the weight buffer after barrier 2 so that barrier 1 disappears (-4 %: the weights get less time to
//   land), prefetching bias + skip rows into dead LDS behind the last K-tile's MFMAs (-2 %), staging weights[t+2] right after
//   barrier 1 instead of behind phase A's MFMAs (-7 %: whatever sits between a barrier and the MFMA burst is exposed, what
//   follows the burst runs in its shadow), a fifth early window piece for W >= 15 (-1.3 %), s_setprio 3 (-1.9 %), the
//   priorities the other way round (read intervals at 2 or 3 above the bursts: -2.4 .. -3 %), window
//   staging unrolled with v_med3 clamps, 8 instead of ~20 instructions per piece (+-0), the next phase's fragment addresses
//   computed inside the MFMA burst, one VALU instruction behind each MFMA (sched_group_barrier; -1.5 %), refilling fragment
//   registers that die inside a burst right there (chan-hi of the next K-tile + K-half 0 of the next phase's pixels: -6 %; the
//   same with the burst's last quadrant pixel-tile-outer so that whole rows are refilled, 6 / 2 instead of 16 / 8 exposed reads
//   per phase: -5.6 %, and hipcc renames the accumulators and spills at the chunk boundaries).
//   A leaner instruction stream (v_bfe / v_bfi mask select, phase B reusing phase A's row address, scalar weight-staging
//   addresses: 37 instead of 60 VALU per K-tile) ran 3.5 % SLOWER: differences of this size are inside the band that code
//   placement alone moves a hipcc-built kernel by (guide rule 27), so nothing below ~3 % is claimed as a schedule effect.
//   Priorities by hardware wave slot (round 3; the two waves of a SIMD belong to two workgroups and sit in slots 0 / 1, HW_ID bit 0):
//   one static priority per wave for the whole kernel, no flips (-3.3 %), or flips to 2 instead of 1 in odd slots so that two
//   colliding bursts are not a tie (-4.9 %); gpurun_out/r03ax_prio.log.
//   What DOES matter is the ORDER of the fragment reads: the two K-halves of a row (addresses a, a ^ 64: complementary LDS
//   banks) back to back, as S4_READ_A issues them, is 5 % faster than all K-half-0 reads followed by all K-half-1 reads.
template <bool HAS_SKIP, int VAR>
__global__ __launch_bounds__(256, 2) void k_conv4w(const char *__restrict__ xb, const char *__restrict__ wb,
                                                    const _Float16 *__restrict__ bias, const char *__restrict__ skipb,
                                                    char *__restrict__ yb, int M, int H, int W, unsigned magicHW, unsigned magicW,
                                                    int pairs_q, int pairs_r) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    // Workgroup b runs on XCD b % 8.  Within an XCD the sequence i = b / 8 walks (tile, channel half) pairs: both halves of a
    // pixel tile are neighbours in launch order on the SAME XCD (they share the window rows in its L2), and the XCD's tiles are
    // a contiguous range (halo rows shared with the neighbouring tile).  tiles = 8 pairs_q + pairs_r.
    int tile, chalf;
    {
        const int c = blockIdx.x & 7, i = blockIdx.x >> 3;
        chalf = i & 1;
        const int ti = i >> 1;
        tile = (c < pairs_r) ? c * (pairs_q + 1) + ti : pairs_r * (pairs_q + 1) + (c - pairs_r) * pairs_q + ti;
        const int mine = (c < pairs_r) ? pairs_q + 1 : pairs_q;
        if (ti >= mine) return;                      // grid is padded to 8 x 2 x (pairs_q + 1)
    }
    const int HW = H * W, HALO = W + 1, NROWS = 256 + 2 * HALO;
    const char *wbh = wb + (size_t)chalf * CT * WROWB;     // this half's 128 filters
    if (VAR & 512) {
        // stagger probe (round 3): the two workgroups that share a CU start together and take the same time per tile, so their
        // prologues / epilogues (no MFMAs) coincide; hold the second slot's first workgroup back by about half a tile
        if (blockIdx.x < 512 && ((blockIdx.x >> 8) & 1))
            for (int i_ = 0; i_ < (VAR & 1024 ? 3 : 6); i_++) __builtin_amdgcn_s_sleep(127);
    }

    if (tid < LZ_BYTES / 16) *reinterpret_cast<intx4 *>(smem + LZ + tid * 16) = intx4{0, 0, 0, 0};
    if (tid + 256 < LZ_BYTES / 16) *reinterpret_cast<intx4 *>(smem + LZ + (tid + 256) * 16) = intx4{0, 0, 0, 0};

    // weight staging: instruction i of this wave fills rows (wid*2+i)*8 + (lane>>3) of a 64-row granule; 16-B chunk (lane&7)
    // of row r holds logical chunk (lane&7) ^ ((r>>1)&7)
    const int boff00 = (wid * 16 + (lane >> 3)) * WROWB + (((lane & 7) ^ (lane >> 4)) << 4);
    const int fragB = (((lane >> 4) ^ ((lane >> 1) & 7)) << 4);
    const int rdB0 = (wc * 32 + (lane & 15)) * 128 + fragB, rdB1 = rdB0 ^ 64;
    const int rowA = HALO + wr * 64 + (lane & 15);

    floatx4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int d = 0; d < 2; d++) acc[a][b][c][d] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 pa[4][2], wlo[2][2], whi[2][2];
    intx4 wst[4];                                        // VAR & 8: one K-tile's weight pieces on their way global -> LDS

#define S4_GLDS(src, ldsoff) \
    __builtin_amdgcn_global_load_lds((const S4_AS1 void *)(src), (S4_AS3 void *)((S4_AS3 char *)smem + (ldsoff)), 16, 0, 0)
// weights of the K-tile whose bytes start at koff_ of a filter row, granule G (64 filters) into buffer BUF
#define S4_STAGE_BK(BUF, G, koff_)                                                                    \
    do {                                                                                              \
        int bo_ = boff00;                                                                             \
        asm volatile("" : "+v"(bo_));                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; i_++) {                                            \
            const char *src_ = wbh + (unsigned)((bo_ ^ (i_ * 64)) + (i_ * 8 + (G) * 64) * WROWB + (koff_)); \
            S4_GLDS(src_, ((BUF) ? LB1 : LB0) + (G) * 8192 + (swid * 2 + i_) * 1024);                  \
        }                                                                                             \
    } while (0)
// register route (VAR & 8): the same four 1-KB pieces of a K-tile's weights, loaded into wst[] ...
#define S4_LOAD_BK(koff_)                                                                              \
    do {                                                                                              \
        int bo_ = boff00;                                                                             \
        asm volatile("" : "+v"(bo_));                                                                 \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; g_++) _Pragma("unroll") for (int i_ = 0; i_ < 2; i_++) { \
            const char *src_ = wbh + (unsigned)((bo_ ^ (i_ * 64)) + (i_ * 8 + g_ * 64) * WROWB + (koff_)); \
            S4_GLOAD128(wst[g_ * 2 + i_], src_);                                                      \
        }                                                                                             \
    } while (0)
// ... and written where the DMA would have put them: lane l of piece (g, i) owns bytes [l * 16, l * 16 + 16) of its 1-KB granule
#define S4_WRITE_BK(BUF)                                                                               \
    do {                                                                                              \
        int wl_ = lane * 16 + swid * 2048;                                                            \
        asm volatile("" : "+v"(wl_));                                                                 \
        S4_DS_WRITE128(wl_, wst[0], ((BUF) ? LB1 : LB0) + 0 * 8192 + 0 * 1024);                       \
        S4_DS_WRITE128(wl_, wst[1], ((BUF) ? LB1 : LB0) + 0 * 8192 + 1 * 1024);                       \
        S4_DS_WRITE128(wl_, wst[2], ((BUF) ? LB1 : LB0) + 1 * 8192 + 0 * 1024);                       \
        S4_DS_WRITE128(wl_, wst[3], ((BUF) ? LB1 : LB0) + 1 * 8192 + 1 * 1024);                       \
    } while (0)
// window pieces (8 rows each) pc*4 + wid for pc in [PC0, PC1) of the channel chunk at byte offset ccoff_ of a pixel row
#define S4_STAGE_W(ccoff_) S4_STAGE_WP(ccoff_, 0, 10)
#define S4_STAGE_WP(ccoff_, PC0, PC1)                                                                 \
    do {                                                                                              \
        _Pragma("nounroll") for (int pc_ = (PC0); pc_ < (PC1); pc_++) {                               \
            const int id_ = pc_ * 4 + swid;                                                           \
            if (id_ * 8 < NROWS) {                                                                    \
                int la_ = lane;                                                                       \
                asm volatile("" : "+v"(la_));                                                         \
                int q_ = tile * 256 - HALO + id_ * 8 + (la_ >> 3);                                    \
                q_ = q_ < 0 ? 0 : (q_ < M ? q_ : M - 1);                                              \
                const int wsrc_ = ((la_ & 7) ^ ((la_ >> 3) & 7)) << 4;   /* recomputed: not worth a register across the loop */ \
                const char *src_ = xb + (unsigned)(q_ * ROWB + (ccoff_) + wsrc_);                     \
                S4_GLDS(src_, LW + id_ * 1024);                                                       \
            }                                                                                         \
        }                                                                                             \
    } while (0)
#define S4_LDS16(off) (*reinterpret_cast<const half8 *>(smem + (off)))
#define S4_SHIFT(T) (((T) / 3 == 0 ? -W : (T) / 3 == 2 ? W : 0) + (T) % 3 - 1)
#define S4_READ_A(G, T)                                                                               \
    if (!(VAR & 32)) do {                                                                             \
        int ra_ = rowA;                                                                               \
        asm volatile("" : "+v"(ra_));                                                                 \
        const int rl_ = ra_ + S4_SHIFT(T);                                                            \
        const int c0_ = (((lane >> 4) ^ rl_) & 7) << 4;                                               \
        const int b0_ = LW + (G) * 16384 + (rl_ << 7) + c0_, b1_ = b0_ ^ 64;                          \
        const int z0_ = LZ + ((rl_ & 1) << 7) + c0_, z1_ = z0_ ^ 64;                                  \
        int mka_ = mk[G][0], mkb_ = mk[G][1];                                                         \
        asm volatile("" : "+v"(mka_), "+v"(mkb_));                                                    \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) {                                         \
            const bool ok_ = (((mt_ >> 1) ? mkb_ : mka_) & (1 << ((mt_ & 1) * 9 + (T)))) != 0;        \
            pa[mt_][0] = S4_LDS16((ok_ ? b0_ : z0_) + mt_ * 2048);                                    \
            pa[mt_][1] = S4_LDS16((ok_ ? b1_ : z1_) + mt_ * 2048);                                    \
        }                                                                                             \
    } while (0)
#define S4_READ_B(BUF, G, dst)                                                                        \
    if (!(VAR & (32 | 2048))) _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) {                            \
        dst[nt_][0] = S4_LDS16(((BUF) ? LB1 : LB0) + (G) * 8192 + nt_ * 2048 + rdB0);                 \
        dst[nt_][1] = S4_LDS16(((BUF) ? LB1 : LB0) + (G) * 8192 + nt_ * 2048 + rdB1);                 \
    }
#define S4_PRIO(x) __builtin_amdgcn_s_setprio(x)
// ABLATION bits (timing only, wrong results; -DSGO_CONV4W_VARIANTS builds): 16 no MFMAs, 32 no fragment reads, 64 no weight
// staging, 128 no barriers inside the K loop, 2048 no WEIGHT fragment reads (the pixel reads stay)
#define S4_MFMA(QM, QN, wfrag)                                                                         \
    if (!(VAR & 16))                                                                                   \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) acc[QM][QN][mt_][nt_] =                    \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[nt_][ks_], pa[mt_][ks_], acc[QM][QN][mt_][nt_], 0, 0, 0)

// One K-tile, tap T of chunk cc (runtime), K-tile index t = 9 cc + T; buffer parity = t & 1 = (cc + T) & 1 -> the caller
// instantiates both parities (CP = cc & 1).
#define S4_KBARRIER() do { if (!(VAR & 128)) { S4_BARRIER(); } } while (0)
#define S4_TILE(T, CP)                                                                                    \
    do {                                                                                                  \
        constexpr int BUF_ = ((T) + (CP)) & 1, T2_ = ((T) + 2) % 9, CARRY_ = ((T) + 2) / 9;               \
        int swid = wid;                                                                                   \
        asm volatile("" : "+s"(swid));                                                                    \
        const bool last2_ = cc == 3 && (T) >= 7;    /* K-tiles 34, 35: nothing left to stage */           \
        const bool boundary_ = (T) == 8 && cc < 3;  /* last tap of a chunk that has a successor */        \
        if (VAR & 256) {                                                                                  \
            /* split wait (round-3 probe): chan-hi is read LAST and waited for behind the first 16 MFMAs, which need */ \
            /* chan-lo + pixel-lo only; barrier 1 moves behind them too (the buffer is refilled in phase B at the earliest) */ \
            S4_READ_B(BUF_, 0, wlo);                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            S4_READ_A(0, T);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            S4_READ_B(BUF_, 1, whi);                                                                      \
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                            \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            S4_PRIO(S4_MFMA_PRIO);                                                                        \
            S4_MFMA(0, 0, wlo);                                                                           \
            S4_PRIO(0);                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            S4_LGKM0();                                                                                   \
            S4_KBARRIER();                                                                                \
            if ((VAR & 4) && boundary_) S4_STAGE_WP((cc + 1) * 128, 0, 4);                                \
            S4_PRIO(S4_MFMA_PRIO);                                                                        \
            S4_MFMA(0, 1, whi);                                                                           \
        } else {                                                                                          \
        S4_READ_B(BUF_, 0, wlo);                                                                          \
        S4_READ_B(BUF_, 1, whi);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        S4_READ_A(0, T);                                                                                  \
        S4_LGKM0();                                                                                       \
        /* barrier 1: every wave has read weights[t] (its buffer may be refilled) and, in the last tap, the window rows */ \
        /* [0, 128) for the last time */                                                                  \
        S4_KBARRIER();                                                                                     \
        /* last tap (shift +W+1): phase B reads window rows >= 128 + 2 (W + 1) only, so rows [0, 128) = pieces 0..15 are */ \
        /* dead from here on (whatever W) and take the next chunk's window one phase early: 4 DMAs per wave */ \
        if ((VAR & 4) && boundary_) S4_STAGE_WP((cc + 1) * 128, 0, 4);                                    \
        S4_PRIO(S4_MFMA_PRIO);                                                                            \
        S4_MFMA(0, 0, wlo);                                                                               \
        S4_MFMA(0, 1, whi);                                                                               \
        }                                                                                                 \
        S4_PRIO(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if ((VAR & 6) == 6 && (T) == 0 && cc > 0) {   /* first tap of a restaged chunk: phase A read rows [0, 128) only (the */ \
            S4_VMWAIT(0);                        /* early pieces); the late pieces, issued a whole phase ago, are needed from here */ \
            S4_KBARRIER();                                                                                 \
        }                                                                                                 \
        S4_READ_A(1, T);                                                                                  \
        if (VAR & 8) {                                                                                    \
            /* register route: weights[t+2] were loaded into wst a K-tile ago; they go into the buffer barrier 1 has freed, */ \
            /* and wst takes weights[t+3].  vmcnt(0): the four loads (long landed) and, in a boundary tap, its early pieces */ \
            constexpr int T3_ = ((T) + 3) % 9, CARRY3_ = ((T) + 3) / 9;                                   \
            if (!last2_) {                                                                                \
                S4_VMWAIT(0);                                                                             \
                S4_WRITE_BK(BUF_);                                                                        \
                if (!(cc == 3 && (T) >= 6)) S4_LOAD_BK(T3_ * (CIN * 2) + (cc + CARRY3_) * 128);           \
            } else if ((T) == 7) {                                                                        \
                S4_VMWAIT(0);                                                                             \
            }                                                                                             \
        } else if (!last2_) {                                                                             \
            const int koff_ = T2_ * (CIN * 2) + (cc + CARRY_) * 128;                                      \
            if (!(VAR & 64)) {                                                                            \
                S4_STAGE_BK(BUF_, 0, koff_);                                                              \
                S4_STAGE_BK(BUF_, 1, koff_);                                                              \
            }                                                                                             \
            /* weights[t+1] (issued one K-tile ago) have landed; younger: weights[t+2] and this tap's early window pieces */ \
            if ((VAR & 4) && boundary_) S4_VMWAIT(8);                                                     \
            else S4_VMWAIT(4);                                                                            \
        } else if ((T) == 7) {                                                                            \
            S4_VMWAIT(0);                        /* K-tile 34: K-tile 35's weights */                       \
        }                                                                                                 \
        S4_LGKM0();                                                                                       \
        S4_KBARRIER();                            /* barrier 2: weights[t+1] visible to all; this tap's window reads retired */ \
        if (boundary_) S4_STAGE_WP((cc + 1) * 128, (VAR & 4) ? 4 : 0, 10);                                \
        S4_PRIO(S4_MFMA_PRIO);                                                                            \
        S4_MFMA(1, 1, whi);                                                                               \
        S4_MFMA(1, 0, wlo);                                                                               \
        S4_PRIO(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);       /* the next K-tile's fragment reads stay below these MFMAs (registers) */ \
        if (boundary_) {                                                                                  \
            /* the next tap's phase A reads window rows [0, 128) = the EARLY pieces: the oldest of this wave's outstanding */ \
            /* DMAs ([early x 4][weights x 4][late x nlate]); the late pieces get one more phase to land */ \
            if (VAR & 8) { /* early pieces were drained by this phase's vmcnt(0); younger: wst loads + late pieces */ } \
            else if ((VAR & 6) != 6) S4_VMWAIT(0);                                                        \
            else if (nlate == 6) S4_VMWAIT(10);                                                           \
            else if (nlate == 5) S4_VMWAIT(9);                                                            \
            else if (nlate == 4) S4_VMWAIT(8);                                                            \
            else S4_VMWAIT(0);                                                                            \
            S4_KBARRIER();                        /* rows [0, 128) of the next chunk's window are in place */ \
        }                                                                                                 \
    } while (0)

    // late window pieces (pc 4..9) this wave issues at a chunk boundary: the counted wait there depends on it
    int nlate = 0;
#pragma unroll
    for (int pc = 4; pc < 10; pc++) nlate += ((pc * 4 + wid) * 8 < NROWS) ? 1 : 0;

    // ---- prologue: window of chunk 0, weights of K-tiles 0 and 1
    {
        int swid = wid;
        S4_STAGE_W(0);
        S4_STAGE_BK(0, 0, 0);
        S4_STAGE_BK(0, 1, 0);
        S4_STAGE_BK(1, 0, CIN * 2);
        S4_STAGE_BK(1, 1, CIN * 2);
        if (VAR & 8) S4_LOAD_BK(2 * (CIN * 2));          // weights of K-tile 2 (tap 2 of chunk 0) wait in registers
    }
    int mk[2][2];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            int v = 0;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int p = tile * 256 + g * 128 + wr * 64 + (h2 * 2 + e) * 16 + (lane & 15);
                // (a divisor of 1 has no 32-bit magic number -- ceil(2^32 / 1) wraps to 0 -- so x / 1 is added back by hand)
                const int q = p - (int)(__umulhi((unsigned)p, magicHW) + (HW == 1 ? (unsigned)p : 0u)) * HW;
                const int yy = (int)(__umulhi((unsigned)q, magicW) + (W == 1 ? (unsigned)q : 0u)), xx = q - yy * W;
                const int cm = (xx >= 1 ? 1 : 0) | 2 | (xx <= W - 2 ? 4 : 0);
                int m = (yy >= 1 ? cm : 0) | (cm << 3) | (yy <= H - 2 ? cm << 6 : 0);
                m = p < M ? m : 0;
                v |= m << (9 * e);
            }
            mk[g][h2] = v;
        }
    if (VAR & 8) S4_VMWAIT(8);                           // window + weights[0] landed; in flight: weights[1] (DMA) + weights[2] (registers)
    else S4_VMWAIT(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero area
    S4_BARRIER();

    for (int kk = 0; kk < 2; kk++) {
        {
            const int cc = 2 * kk;
            S4_TILE(0, 0); S4_TILE(1, 0); S4_TILE(2, 0); S4_TILE(3, 0); S4_TILE(4, 0);
            S4_TILE(5, 0); S4_TILE(6, 0); S4_TILE(7, 0); S4_TILE(8, 0);
        }
        {
            const int cc = 2 * kk + 1;
            S4_TILE(0, 1); S4_TILE(1, 1); S4_TILE(2, 1); S4_TILE(3, 1); S4_TILE(4, 1);
            S4_TILE(5, 1); S4_TILE(6, 1); S4_TILE(7, 1); S4_TILE(8, 1);
        }
    }
    S4_BARRIER();   // every wave is done with the window and the weights: the LDS becomes the output stage

    // ---- epilogue through LDS: half hf (128 pixels x 128 channels) lives at [hf*32 KiB, +32 KiB), rows of 256 B, 16-B chunk c
    //      of row r at chunk c ^ (r & 15)
    int elane = lane;
    asm volatile("" : "+v"(elane));
    intx2 bvi[2][2];
    {
        const _Float16 *bp = bias + chalf * CT + wc * 32 + (elane >> 4) * 4;
#pragma unroll
        for (int qn = 0; qn < 2; qn++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
                asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(bvi[qn][nt]) : "v"(bp), "n"((qn * 64 + nt * 16) * 2) : "memory");
    }
    if constexpr (HAS_SKIP) {
        // instruction j of this wave fills rows (wid*8+j)*4 + (lane>>4) of the half
#pragma nounroll
        for (int hf = 0; hf < 2; hf++)
#pragma nounroll
            for (int j = 0; j < 8; j++) {
                const int r_ = (wid * 8 + j) * 4 + (elane >> 4);
                int p_ = tile * 256 + hf * 128 + r_;
                p_ = p_ < M ? p_ : M - 1;
                S4_GLDS(skipb + (unsigned)(p_ * ROWB + chalf * (CT * 2) + (((elane & 15) ^ (r_ & 15)) << 4)), hf * 32768 + (wid * 8 + j) * 1024);
            }
    }
    const int epx = (wr * 64 + (elane & 15)) * 256 + ((elane >> 4) & 1) * 8;
    const int epc = ((wc * 4 + (elane >> 5)) ^ (elane & 15)) << 4;       // chunk of (qn = 0, nt = 0); qn toggles bit 3, nt bit 1
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const int a00 = hf * 32768 + epx + epc, a01 = hf * 32768 + epx + (epc ^ 32);
        const int a10 = hf * 32768 + epx + (epc ^ 128), a11 = hf * 32768 + epx + (epc ^ 128 ^ 32);
        intx2 sk[4][2][2];
        if (hf == 0) {
            if constexpr (HAS_SKIP) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // bias + the lo half's rows (the hi half's 8 DMAs may fly)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (HAS_SKIP) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the hi half's rows (younger: the 8 row stores of half 0)
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HAS_SKIP) {
            S4_BARRIER();                                          // everybody's skip rows of this half are in LDS
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                S4_DS_READ64(sk[mt][0][0], a00, mt * 4096);
                S4_DS_READ64(sk[mt][0][1], a01, mt * 4096);
                S4_DS_READ64(sk[mt][1][0], a10, mt * 4096);
                S4_DS_READ64(sk[mt][1][1], a11, mt * 4096);
            }
            S4_LGKM0();
        }
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int qn = 0; qn < 2; qn++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    floatx4 v = acc[hf][qn][mt][nt];
                    if constexpr (HAS_SKIP) {
                        const half4 s4 = __builtin_bit_cast(half4, sk[mt][qn][nt]);
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] += (float)s4[j];
                    }
                    half4 o;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float f = v[j] + (float)__builtin_bit_cast(half4, bvi[qn][nt])[j];
                        o[j] = (_Float16)(f > 0.f ? f : 0.f);
                    }
                    const intx2 oi = __builtin_bit_cast(intx2, o);
                    if (qn == 0 && nt == 0) S4_DS_WRITE64(a00, oi, mt * 4096);
                    else if (qn == 0) S4_DS_WRITE64(a01, oi, mt * 4096);
                    else if (nt == 0) S4_DS_WRITE64(a10, oi, mt * 4096);
                    else S4_DS_WRITE64(a11, oi, mt * 4096);
                }
        S4_LGKM0();
        S4_BARRIER();
        // copy-out: wave wid, instruction j, lane -> LDS bytes hf*32 KiB + wid*8192 + j*1024 + lane*16 = row wid*32 + j*4 +
        // (lane>>4), physical chunk lane&15 = logical chunk (lane&15) ^ (row & 15)
        intx4 ov[8];
        const int a2 = hf * 32768 + wid * 8192 + elane * 16;
#pragma unroll
        for (int j = 0; j < 8; j++) S4_DS_READ128(ov[j], a2, j * 1024);
        const int r0 = wid * 32 + (elane >> 4);
        const int p0 = tile * 256 + hf * 128 + r0;
        char *dst = yb + (size_t)p0 * ROWB + chalf * (CT * 2);
        S4_LGKM0();
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (p0 + j * 4 < M)
                *reinterpret_cast<intx4 *>(dst + j * 4 * ROWB + (((elane & 15) ^ ((r0 + j * 4) & 15)) << 4)) = ov[j];
    }
}

template <int VAR>
static inline int launch_var(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y,
                             hipStream_t st) {
    const long M = (long)n * h * w;
    if (M <= 0 || M * ROWB >= (1L << 31) || w > MAXW || w < 1 || h < 1) return -1;
    if ((unsigned long long)(M + 256) * (unsigned long long)(h * w) >= (1ULL << 32)) return -1;
    const int tiles = (int)((M + 255) / 256);
    const unsigned mhw = (unsigned)(((1ULL << 32) + (unsigned)(h * w) - 1) / (unsigned)(h * w)), mw = (unsigned)(((1ULL << 32) + (unsigned)w - 1) / (unsigned)w);
    const int q = tiles / 8, r = tiles % 8;
    const int per_xcd = 2 * (q + (r ? 1 : 0));        // (tile, half) pairs of the fullest XCD
    const dim3 grid(8 * per_xcd);
#define S4_ARGS (const char *)x, (const char *)wgt, (const _Float16 *)bias, (const char *)skip, (char *)y, (int)M, h, w, mhw, mw, q, r
    if (skip) hipLaunchKernelGGL((k_conv4w<true, VAR>), grid, dim3(256), 0, st, S4_ARGS);
    else hipLaunchKernelGGL((k_conv4w<false, VAR>), grid, dim3(256), 0, st, S4_ARGS);
#undef S4_ARGS
    return 0;
}

static inline int launch(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y,
                         hipStream_t st, int var = 7) {
    switch (var) {
#ifdef SGO_CONV4W_VARIANTS
    case 0: return launch_var<0>(n, h, w, x, wgt, bias, skip, y, st);
    case 4: return launch_var<4>(n, h, w, x, wgt, bias, skip, y, st);
    case 5: return launch_var<5>(n, h, w, x, wgt, bias, skip, y, st);
    case 6: return launch_var<6>(n, h, w, x, wgt, bias, skip, y, st);
    case 15: return launch_var<15>(n, h, w, x, wgt, bias, skip, y, st);
    case 23: return launch_var<23>(n, h, w, x, wgt, bias, skip, y, st);      // ablations (wrong results, timing only)
    case 39: return launch_var<39>(n, h, w, x, wgt, bias, skip, y, st);
    case 71: return launch_var<71>(n, h, w, x, wgt, bias, skip, y, st);
    case 135: return launch_var<135>(n, h, w, x, wgt, bias, skip, y, st);
    case 87: return launch_var<87>(n, h, w, x, wgt, bias, skip, y, st);
    case 231: return launch_var<231>(n, h, w, x, wgt, bias, skip, y, st);
    case 103: return launch_var<103>(n, h, w, x, wgt, bias, skip, y, st);
    case 263: return launch_var<263>(n, h, w, x, wgt, bias, skip, y, st);    // 7 + split wait in phase A
    case 519: return launch_var<519>(n, h, w, x, wgt, bias, skip, y, st);    // 7 + half-tile stagger of the CU's second slot
    case 1543: return launch_var<1543>(n, h, w, x, wgt, bias, skip, y, st);  // 7 + quarter-tile stagger
    case 2055: return launch_var<2055>(n, h, w, x, wgt, bias, skip, y, st);  // ablation: no WEIGHT fragment reads (pixel reads stay)
    case 2119: return launch_var<2119>(n, h, w, x, wgt, bias, skip, y, st);  // ... and no weight staging
    case 2247: return launch_var<2247>(n, h, w, x, wgt, bias, skip, y, st);  // ... and no K-loop barriers: bound of a register-fed weight operand
#endif
    default: return launch_var<7>(n, h, w, x, wgt, bias, skip, y, st);
    }
}

}  // namespace sgo_conv4w

#undef S4_AS1
#undef S4_AS3
#undef S4_BARRIER
#undef S4_KBARRIER
#undef S4_DS_READ128
#undef S4_DS_READ64
#undef S4_DS_WRITE64
#undef S4_GLDS
#undef S4_LDS16
#undef S4_LGKM0
#undef S4_MFMA
#undef S4_PRIO
#undef S4_MFMA_PRIO
#undef S4_READ_A
#undef S4_READ_B
#undef S4_SHIFT
#undef S4_STAGE_BK
#undef S4_STAGE_W
#undef S4_TILE
#undef S4_VMWAIT
#undef S4_GLOAD128
#undef S4_DS_WRITE128
#undef S4_LOAD_BK
#undef S4_WRITE_BK
