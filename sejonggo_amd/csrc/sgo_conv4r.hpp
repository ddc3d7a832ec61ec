// sgo_conv4r.hpp -- third hand-written kernel for the residual tower's 3x3 / 256 -> 256 'same' convolution with bias (+ skip) +
// ReLU fused (model.py:37-46).  Same math, workgroup tile (256 pixels x 128 channels, 4 waves, two workgroups per CU), pixel
// window, masks and epilogue as sgo_conv4w.hpp; different OPERAND ROUTE for the weights:
//
//   k_conv4w: weights L2 -> LDS (DMA, double-buffered per K-tile) -> registers (ds_read_b128), two barriers per K-tile.  Per
//             K-tile a wave issues 24 fragment reads (16 pixel, 8 weight) and 4 DMA pieces; the CU's LDS is busy ~89 % of the
//             MFMA time (8 waves x 24 x 8 cycles + 36 KB of DMA writes per 2 048 cycles), and the ablations of round 3
//             (profiles/r03_conv4w_ablations.json) show the MFMA bursts waiting for exactly that.
//   k_conv4r: weights L2 / L1 -> REGISTERS, one global_load_dwordx4 per fragment, from a copy of the filter bank laid out in
//             fragment order (sgo_conv3x3_tower_prepack_dev: every load of a wave is 1 KB contiguous), three rotating 16-register
//             sets: in use / next / next-next.  The LDS carries the pixel window only (-1/3 of the fragment reads, no DMA writes
//             of weights: ~53 % busy), and the K loop has NO barrier except at the three chunk boundaries where the single-
//             buffered window is restaged (as in k_conv4w), so the four waves of a workgroup drift freely in between.
//
// Per K-tile t (tap T of 64-channel chunk cc), weights lo(t) / hi(t) = filters [0, 64) / [64, 128) of this wave's channel group,
// loads numbered L(2t) = lo(t), L(2t+1) = hi(t), load j into set j % 3:
//   phase A: issue L(2t+2) (set of hi(t-1), dead) | read pixel-lo fragments | vmcnt: L(2t) landed | 16 MFMA lo | vmcnt: L(2t+1)
//            landed | 16 MFMA hi
//   phase B: read pixel-hi fragments | 16 MFMA lo | issue L(2t+3) (set of lo(t), dead from here) | 16 MFMA hi
// A wave issues exactly eight weight loads per K-tile, so the counted waits are vmcnt(8) / vmcnt(4); chunk boundaries add the
// window pieces (see R4_TILE).  The last K-tile's two look-ahead loads read the 8 KB of padding behind the packed bank.
//
// RESULT (round 3, profiles/r03_conv4r_experiment.json): identical bits, and the same speed as k_conv4w within 1 % on every box
// (2.10-2.20 ms per 8 192 x 17 x 17 launch, 0.52 of the fp16 MFMA peak), wherever the loads are issued.  The route saves cycles
// (LDS instruction cycles -29 %, kernel cycles -0.8 %) and pays them back in clock: the chip is power-limited under this kernel,
// and 64 KB per K-tile pair through the vector-memory path costs frequency even though the duplicates hit in L1
// (profiles/r03_pmc_conv4r_ablate.json: the weight loads are 5 % of the cycles and 16 % of the time; the pixel reads 13 % / 16 %).
// It is kept as a selectable route (sgo_conv3x3_tower_packed_dev), not as the default.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgo_conv4r {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx2 __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));

#define R4_AS1 __attribute__((address_space(1)))
#define R4_AS3 __attribute__((address_space(3)))

constexpr int CIN = 256, COUT = 256, CT = 128;   // CT: output channels per workgroup
constexpr int ROWB = CIN * 2, MAXW = 19;
// packed filter bank: [channel half 2][wave channel group 2][K-tile 36 = chunk-major, tap][lo / hi 2][nt 2][ks 2][lane 64][8 halves]
constexpr int WAVE_BANK = 36 * 8192, PACKED_BYTES = 4 * WAVE_BANK + 8192;
// LDS map: window (320 rows x 128 B), zero area; the epilogue reuses [0, 64 KiB)
constexpr int LW = 0, LZ = 40960, LZ_BYTES = 3 * 2048 + 256, LDS_BYTES = 65536;

#define R4_DS_READ64(dst, addr, OFF) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define R4_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define R4_DS_WRITE64(addr, val, OFF) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(OFF) : "memory")
#define R4_LGKM0()                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)
#define R4_VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define R4_VMWAITW(n) R4_VMWAIT(n)   /* a wait for WEIGHT loads */
#define R4_BARRIER()                   \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0)

// VAR bit 0: s_setprio(1) around the MFMA bursts; bit 1: the look-ahead load of hi(t+1) is issued at the END of phase B instead
// of between its two MFMA groups; bit 4 (16): the loads spread inside the bursts (R4_TILE_S).  ABLATION bits (timing only, wrong
// results; -DSGO_CONV4W_VARIANTS builds): 4 no weight loads, 32 no window DMA in the prologue, 64 no pixel fragment reads, 128 no
// restage (and no barrier) at the chunk boundaries.
// Measured and dropped (round 3): a DOUBLE-BUFFERED window for board widths <= 17 (two 292-row buffers + the zero area = 81 152 B,
// two workgroups per CU still fit; the next chunk's pieces staged two per tap at the start of taps 0..4 with the lanes beyond the
// window masked off, ONE barrier per chunk boundary, no drain): bit-identical, 252 VGPRs, and 3.2 % SLOWER (2.180 vs 2.112 ms in
// one process, gpurun_out/r03bi_ab.log) although the boundaries' restage + barriers are worth 5.9 % when ablated away together
// (variant 129): what costs at the boundaries is the window DMA itself, not the synchronisation around it.
// (There is no "loads without waits" ablation: a load that lands after the compiler has given its registers to something else --
// an address, say -- corrupts it; the one run of such a variant ended in a memory access fault.)
template <bool HAS_SKIP, int VAR>
__global__ __launch_bounds__(256, 2) void k_conv4r(const char *__restrict__ xb, const char *__restrict__ wpk,
                                                    const _Float16 *__restrict__ bias, const char *__restrict__ skipb,
                                                    char *__restrict__ yb, int M, int H, int W, unsigned magicHW, unsigned magicW,
                                                    int pairs_q, int pairs_r) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    // tile order: as k_conv4w (workgroup b runs on XCD b % 8; both channel halves of a pixel tile neighbours on one XCD; an
    // XCD's tiles a contiguous range)
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

    if (tid < LZ_BYTES / 16) *reinterpret_cast<intx4 *>(smem + LZ + tid * 16) = intx4{0, 0, 0, 0};
    if (tid + 256 < LZ_BYTES / 16) *reinterpret_cast<intx4 *>(smem + LZ + (tid + 256) * 16) = intx4{0, 0, 0, 0};

    const int rowA = HALO + wr * 64 + (lane & 15);
    // this wave's slice of the packed bank: a scalar pointer that walks 4 KB per load group, plus lane * 16
    const char *wptr = wpk + (size_t)((chalf * 2 + wc) * WAVE_BANK);
    const int wlane = lane * 16;

    floatx4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int d = 0; d < 2; d++) acc[a][b][c][d] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 pa[4][2];
    if (VAR & 64) {                                      // ablation: the pixel fragments are never read
#pragma unroll
        for (int i_ = 0; i_ < 8; i_++) asm volatile("" : "=v"(pa[i_ >> 1][i_ & 1]));
    }
    half8 ws[3][2][2];                                   // [set][nt][ks]
    if (VAR & 4) {
#pragma unroll
        for (int i_ = 0; i_ < 12; i_++) asm volatile("" : "=v"(ws[i_ / 4][(i_ >> 1) & 1][i_ & 1]));
    }

#define R4_GLDS(src, ldsoff) \
    __builtin_amdgcn_global_load_lds((const R4_AS1 void *)(src), (R4_AS3 void *)((R4_AS3 char *)smem + (ldsoff)), 16, 0, 0)
// one load group (4 KB: fragments [nt][ks] of 64 filters x 64 K) into set S; the scalar pointer moves on
#define R4_LOADW(S)                                                                                                     \
    do {                                                                                                                \
        if (!(VAR & 4)) asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"                 \
                     "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"         \
                     : "=&v"(ws[S][0][0]), "=&v"(ws[S][0][1]), "=&v"(ws[S][1][0]), "=&v"(ws[S][1][1])                   \
                     : "v"(wlane), "s"(wptr)                                                                            \
                     : "memory");                                                                                       \
        wptr += 4096;                                                                                                   \
    } while (0)
// s_waitcnt vmcnt(n), after which set S holds its fragments: the set is tied through the wait, so that nothing that uses it
// (and no copy of it) is placed above
#define R4_WAITW(n, S)                                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #n ")"                                                                              \
                 : "+v"(ws[S][0][0]), "+v"(ws[S][0][1]), "+v"(ws[S][1][0]), "+v"(ws[S][1][1])::"memory")
#define R4_TIEW(S) asm volatile("" : "+v"(ws[S][0][0]), "+v"(ws[S][0][1]), "+v"(ws[S][1][0]), "+v"(ws[S][1][1])::"memory")
// window pieces (8 rows each) pc*4 + wid for pc in [PC0, PC1) of the channel chunk at byte offset ccoff_ of a pixel row
#define R4_STAGE_WP(ccoff_, PC0, PC1)                                                                 \
    do {                                                                                              \
        _Pragma("nounroll") for (int pc_ = (PC0); pc_ < (PC1); pc_++) {                               \
            const int id_ = pc_ * 4 + swid;                                                           \
            if (id_ * 8 < NROWS) {                                                                    \
                int la_ = lane;                                                                       \
                asm volatile("" : "+v"(la_));                                                         \
                int q_ = tile * 256 - HALO + id_ * 8 + (la_ >> 3);                                    \
                q_ = q_ < 0 ? 0 : (q_ < M ? q_ : M - 1);                                              \
                const int wsrc_ = ((la_ & 7) ^ ((la_ >> 3) & 7)) << 4;                                \
                const char *src_ = xb + (unsigned)(q_ * ROWB + (ccoff_) + wsrc_);                     \
                R4_GLDS(src_, LW + id_ * 1024);                                                       \
            }                                                                                         \
        }                                                                                             \
    } while (0)
#define R4_LDS16(off) (*reinterpret_cast<const half8 *>(smem + (off)))
#define R4_SHIFT(T) (((T) / 3 == 0 ? -W : (T) / 3 == 2 ? W : 0) + (T) % 3 - 1)
#define R4_READ_A(G, T)                                                                               \
    if (!(VAR & 64)) do {                                                                             \
        int ra_ = rowA;                                                                               \
        asm volatile("" : "+v"(ra_));                                                                 \
        const int rl_ = ra_ + R4_SHIFT(T);                                                            \
        const int c0_ = (((lane >> 4) ^ rl_) & 7) << 4;                                               \
        const int b0_ = LW + (G) * 16384 + (rl_ << 7) + c0_, b1_ = b0_ ^ 64;                          \
        const int z0_ = LZ + ((rl_ & 1) << 7) + c0_, z1_ = z0_ ^ 64;                                  \
        int mka_ = mk[G][0], mkb_ = mk[G][1];                                                         \
        asm volatile("" : "+v"(mka_), "+v"(mkb_));                                                    \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) {                                         \
            const bool ok_ = (((mt_ >> 1) ? mkb_ : mka_) & (1 << ((mt_ & 1) * 9 + (T)))) != 0;        \
            pa[mt_][0] = R4_LDS16((ok_ ? b0_ : z0_) + mt_ * 2048);                                    \
            pa[mt_][1] = R4_LDS16((ok_ ? b1_ : z1_) + mt_ * 2048);                                    \
        }                                                                                             \
    } while (0)
#define R4_PRIO(x) __builtin_amdgcn_s_setprio(x)
#define R4_MFMA(QM, QN, S)                                                                             \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) acc[QM][QN][mt_][nt_] =                    \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[S][nt_][ks_], pa[mt_][ks_], acc[QM][QN][mt_][nt_], 0, 0, 0)
// vmcnt(base + nlate): the late window pieces of a chunk boundary (3..6 per wave) are among the younger loads
#define R4_VMWAIT_LATE(base)                                                      \
    do {                                                                          \
        if (nlate == 6) { R4_VMWAIT_SUM(base, 6); }                               \
        else if (nlate == 5) { R4_VMWAIT_SUM(base, 5); }                          \
        else if (nlate == 4) { R4_VMWAIT_SUM(base, 4); }                          \
        else { R4_VMWAIT_SUM(base, 0); }                                          \
    } while (0)
#define R4_VMWAIT_SUM(a, b) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((a) + (b)) : "memory")

// One K-tile, tap T of chunk cc (runtime); the set indices depend on T only (9 cc = 0 mod 3).
#define R4_TILE(T)                                                                                        \
    do {                                                                                                  \
        constexpr int SLO_ = (2 * (T)) % 3, SHI_ = (2 * (T) + 1) % 3, SNX_ = (2 * (T) + 2) % 3;           \
        int swid = wid;                                                                                   \
        asm volatile("" : "+s"(swid));                                                                    \
        const bool boundary_ = !(VAR & 128) && (T) == 8 && cc < 3;  /* last tap of a chunk that has a successor (ablation bit 128: the window is never restaged) */        \
        const bool restaged_ = !(VAR & 128) && (T) == 0 && cc > 0;  /* first tap on a restaged window */                  \
        /* ---- phase A */                                                                                \
        R4_LOADW(SNX_);                           /* L(2t+2) = lo(t+1) */                                  \
        R4_READ_A(0, T);                                                                                  \
        R4_LGKM0();                                                                                       \
        /* L(2t) has landed; younger: L(2t+1), L(2t+2) -- and, on a restaged window, the boundary's late pieces, which are */ \
        /* older than L(2t+1) and get until the next wait to land */                                      \
        if (restaged_) R4_VMWAIT_LATE(8);                                                                 \
        else R4_VMWAITW(8);                                                                               \
        R4_TIEW(SLO_);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (boundary_) {                                                                                  \
            /* every wave has read window rows [0, 128) for the last time (phase B of the last tap, shift +W+1, reads rows */ \
            /* >= 128 + 2 (W + 1) only): pieces 0..15 take the next chunk's window one phase early, 4 DMAs per wave */ \
            R4_BARRIER();                                                                                 \
            R4_STAGE_WP((cc + 1) * 128, 0, 4);                                                            \
        }                                                                                                 \
        R4_PRIO(VAR & 1);                                                                                 \
        R4_MFMA(0, 0, SLO_);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        /* L(2t+1) has landed; younger: L(2t+2) and a boundary tap's four early pieces */                 \
        if (boundary_) R4_VMWAIT(8);                                                                      \
        else R4_VMWAITW(4);                                                                               \
        R4_TIEW(SHI_);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        R4_MFMA(0, 1, SHI_);                                                                              \
        R4_PRIO(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (restaged_) {                                                                                  \
            /* phase A read the early rows [0, 128) only; every wave's late pieces were retired by its vmcnt(4) above */ \
            R4_BARRIER();                                                                                 \
        }                                                                                                 \
        /* ---- phase B */                                                                                \
        R4_READ_A(1, T);                                                                                  \
        R4_LGKM0();                                                                                       \
        if (boundary_) {                                                                                  \
            R4_BARRIER();                         /* this chunk's window reads are retired in every wave */ \
            R4_STAGE_WP((cc + 1) * 128, 4, 10);                                                           \
        }                                                                                                 \
        R4_PRIO(VAR & 1);                                                                                 \
        R4_MFMA(1, 0, SLO_);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (!(VAR & 2)) R4_LOADW(SLO_);           /* L(2t+3) = hi(t+1) into the set lo(t) has just left */ \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        R4_MFMA(1, 1, SHI_);                                                                              \
        R4_PRIO(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (VAR & 2) R4_LOADW(SLO_);                                                                      \
        if (boundary_) {                                                                                  \
            /* the next tap's phase A reads window rows [0, 128) = the EARLY pieces; younger: the late pieces and L(2t+3) */ \
            R4_VMWAIT_LATE(4);                                                                            \
            R4_BARRIER();                                                                                 \
        }                                                                                                 \
    } while (0)

// ---- VAR bit 4 (16): the weight loads are issued BETWEEN the MFMAs of a burst, one 1-KB piece behind every 8th (phase A:
// L(2t+2)) or 4th (second half of phase B: L(2t+3)) MFMA, instead of in groups of four at a phase start / in mid-burst: a VMEM
// instruction costs its wave tens of issue cycles, which beside a running MFMA (16 cycles, 8 of them holding the issue port) are
// hidden, and in the read interval between two bursts are not.
#define R4_LOADP(S, P)                                                                                                  \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (!(VAR & 4)) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"                                         \
                                     : "=&v"(ws[S][(P) >> 1][(P) & 1]) : "v"(wlane), "s"(wptr), "n"((P) * 1024) : "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define R4_M1(QM, QN, S, I)                                                                                             \
    acc[QM][QN][((I) >> 1) & 3][(I) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                      \
        ws[S][(I) & 1][(I) >> 3], pa[((I) >> 1) & 3][(I) >> 3], acc[QM][QN][((I) >> 1) & 3][(I) & 1], 0, 0, 0)
#define R4_M2(QM, QN, S, I) R4_M1(QM, QN, S, I); R4_M1(QM, QN, S, (I) + 1)
#define R4_M4(QM, QN, S, I) R4_M2(QM, QN, S, I); R4_M2(QM, QN, S, (I) + 2)
// 16 MFMAs with pieces P0, P0 + 1 of set LS behind the 4th and the 12th
#define R4_G2(QM, QN, S, LS, P0)                                                                                        \
    do {                                                                                                                \
        R4_M4(QM, QN, S, 0); R4_LOADP(LS, P0); R4_M4(QM, QN, S, 4); R4_M4(QM, QN, S, 8); R4_LOADP(LS, (P0) + 1);        \
        R4_M4(QM, QN, S, 12);                                                                                           \
    } while (0)
// 16 MFMAs with all four pieces of set LS behind the 2nd, 6th, 10th and 14th
#define R4_G4(QM, QN, S, LS)                                                                                            \
    do {                                                                                                                \
        R4_M2(QM, QN, S, 0); R4_LOADP(LS, 0); R4_M4(QM, QN, S, 2); R4_LOADP(LS, 1); R4_M4(QM, QN, S, 6); R4_LOADP(LS, 2); \
        R4_M4(QM, QN, S, 10); R4_LOADP(LS, 3); R4_M2(QM, QN, S, 14);                                                    \
    } while (0)
#define R4_TILE_S(T)                                                                                      \
    do {                                                                                                  \
        constexpr int SLO_ = (2 * (T)) % 3, SHI_ = (2 * (T) + 1) % 3, SNX_ = (2 * (T) + 2) % 3;           \
        int swid = wid;                                                                                   \
        asm volatile("" : "+s"(swid));                                                                    \
        const bool boundary_ = (T) == 8 && cc < 3;                                                        \
        const bool restaged_ = (T) == 0 && cc > 0;                                                        \
        /* ---- phase A */                                                                                \
        R4_READ_A(0, T);                                                                                  \
        R4_LGKM0();                                                                                       \
        /* L(2t) has landed; younger: L(2t+1) -- and, on a restaged window, the boundary's late pieces before it */ \
        if (restaged_) R4_VMWAIT_LATE(4);                                                                 \
        else R4_VMWAITW(4);                                                                               \
        R4_TIEW(SLO_);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (boundary_) {                                                                                  \
            R4_BARRIER();                                                                                 \
            R4_STAGE_WP((cc + 1) * 128, 0, 4);                                                            \
        }                                                                                                 \
        R4_PRIO(VAR & 1);                                                                                 \
        R4_G2(0, 0, SLO_, SNX_, 0);               /* + L(2t+2) pieces 0, 1 */                              \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        /* L(2t+1) has landed; younger: two pieces of L(2t+2), before them a boundary tap's four early window pieces */ \
        if (boundary_) R4_VMWAIT(6);                                                                      \
        else R4_VMWAITW(2);                                                                               \
        R4_TIEW(SHI_);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        R4_G2(0, 1, SHI_, SNX_, 2);               /* + L(2t+2) pieces 2, 3 */                              \
        wptr += 4096;                                                                                     \
        R4_PRIO(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (restaged_) {                                                                                  \
            R4_BARRIER();                         /* every wave's late pieces were retired by its vmcnt(2) above */ \
        }                                                                                                 \
        /* ---- phase B */                                                                                \
        R4_READ_A(1, T);                                                                                  \
        R4_LGKM0();                                                                                       \
        if (boundary_) {                                                                                  \
            R4_BARRIER();                                                                                 \
            R4_STAGE_WP((cc + 1) * 128, 4, 10);                                                           \
        }                                                                                                 \
        R4_PRIO(VAR & 1);                                                                                 \
        R4_MFMA(1, 0, SLO_);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        R4_G4(1, 1, SHI_, SLO_);                  /* + L(2t+3) = hi(t+1) into the set lo(t) has just left */ \
        wptr += 4096;                                                                                     \
        R4_PRIO(0);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (boundary_) {                                                                                  \
            /* the early pieces have landed; younger: L(2t+2), the late pieces, L(2t+3) */                \
            R4_VMWAIT_LATE(8);                                                                            \
            R4_BARRIER();                                                                                 \
        }                                                                                                 \
    } while (0)

    // late window pieces (pc 4..9) this wave issues at a chunk boundary: the counted waits there depend on it
    int nlate = 0;
#pragma unroll
    for (int pc = 4; pc < 10; pc++) nlate += ((pc * 4 + wid) * 8 < NROWS) ? 1 : 0;

    // ---- prologue: window of chunk 0, weights of K-tile 0
    {
        int swid = wid;
        if (!(VAR & 32)) R4_STAGE_WP(0, 0, 10);           // ablation bit 32: no window DMA in the prologue (a prologue hidden elsewhere)
        R4_LOADW(0);
        R4_LOADW(1);
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
    R4_VMWAIT(8);                                        // the window has landed; in flight: L(0), L(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero area
    R4_BARRIER();

#pragma nounroll
    for (int cc = 0; cc < 4; cc++) {
        if constexpr ((VAR & 16) != 0) {
            R4_TILE_S(0); R4_TILE_S(1); R4_TILE_S(2); R4_TILE_S(3); R4_TILE_S(4);
            R4_TILE_S(5); R4_TILE_S(6); R4_TILE_S(7); R4_TILE_S(8);
        } else {
            R4_TILE(0); R4_TILE(1); R4_TILE(2); R4_TILE(3); R4_TILE(4);
            R4_TILE(5); R4_TILE(6); R4_TILE(7); R4_TILE(8);
        }
    }
    // the last K-tile's look-ahead loads (padding bytes) must not land in registers the epilogue has taken over
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ws[0][0][0]), "+v"(ws[0][0][1]), "+v"(ws[0][1][0]), "+v"(ws[0][1][1]), "+v"(ws[1][0][0]), "+v"(ws[1][0][1]),
                   "+v"(ws[1][1][0]), "+v"(ws[1][1][1]), "+v"(ws[2][0][0]), "+v"(ws[2][0][1]), "+v"(ws[2][1][0]), "+v"(ws[2][1][1])::"memory");
    R4_BARRIER();   // every wave is done with the window: the LDS becomes the output stage

    // ---- epilogue through LDS (as k_conv4w): half hf (128 pixels x 128 channels) lives at [hf*32 KiB, +32 KiB), rows of 256 B,
    //      16-B chunk c of row r at chunk c ^ (r & 15)
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
                R4_GLDS(skipb + (unsigned)(p_ * ROWB + chalf * (CT * 2) + (((elane & 15) ^ (r_ & 15)) << 4)), hf * 32768 + (wid * 8 + j) * 1024);
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
            R4_BARRIER();                                          // everybody's skip rows of this half are in LDS
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                R4_DS_READ64(sk[mt][0][0], a00, mt * 4096);
                R4_DS_READ64(sk[mt][0][1], a01, mt * 4096);
                R4_DS_READ64(sk[mt][1][0], a10, mt * 4096);
                R4_DS_READ64(sk[mt][1][1], a11, mt * 4096);
            }
            R4_LGKM0();
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
                    if (qn == 0 && nt == 0) R4_DS_WRITE64(a00, oi, mt * 4096);
                    else if (qn == 0) R4_DS_WRITE64(a01, oi, mt * 4096);
                    else if (nt == 0) R4_DS_WRITE64(a10, oi, mt * 4096);
                    else R4_DS_WRITE64(a11, oi, mt * 4096);
                }
        R4_LGKM0();
        R4_BARRIER();
        // copy-out: wave wid, instruction j, lane -> LDS bytes hf*32 KiB + wid*8192 + j*1024 + lane*16 = row wid*32 + j*4 +
        // (lane>>4), physical chunk lane&15 = logical chunk (lane&15) ^ (row & 15)
        intx4 ov[8];
        const int a2 = hf * 32768 + wid * 8192 + elane * 16;
#pragma unroll
        for (int j = 0; j < 8; j++) R4_DS_READ128(ov[j], a2, j * 1024);
        const int r0 = wid * 32 + (elane >> 4);
        const int p0 = tile * 256 + hf * 128 + r0;
        char *dst = yb + (size_t)p0 * ROWB + chalf * (CT * 2);
        R4_LGKM0();
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (p0 + j * 4 < M)
                *reinterpret_cast<intx4 *>(dst + j * 4 * ROWB + (((elane & 15) ^ ((r0 + j * 4) & 15)) << 4)) = ov[j];
    }
}

// filter bank OHWI [256][3][3][256] fp16 -> fragment order (see PACKED_BYTES): one thread per 16-byte piece
__global__ void k_prepack(const char *__restrict__ w, char *__restrict__ wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // piece index in the packed bank
    if (i >= 4 * WAVE_BANK / 16) {
        if (i < PACKED_BYTES / 16) reinterpret_cast<intx4 *>(wp)[i] = intx4{0, 0, 0, 0};
        return;
    }
    const int lane = i & 63, ks = (i >> 6) & 1, nt = (i >> 7) & 1, qn = (i >> 8) & 1;
    const int t = (i >> 9) % 36, g = (i >> 9) / 36;               // g = chalf * 2 + wc
    const int cc = t / 9, T = t % 9;
    const int filter = (g >> 1) * 128 + qn * 64 + (g & 1) * 32 + nt * 16 + (lane & 15);
    const int ci = cc * 64 + ks * 32 + (lane >> 4) * 8;
    reinterpret_cast<intx4 *>(wp)[i] = *reinterpret_cast<const intx4 *>(w + ((size_t)(filter * 9 + T) * CIN + ci) * 2);
}

static inline int prepack(const void *w, void *wp, hipStream_t st) {
    const int pieces = PACKED_BYTES / 16;
    hipLaunchKernelGGL(k_prepack, dim3((pieces + 255) / 256), dim3(256), 0, st, (const char *)w, (char *)wp);
    return 0;
}

template <int VAR>
static inline int launch_var(int n, int h, int w, const void *x, const void *wpk, const void *bias, const void *skip, void *y,
                             hipStream_t st) {
    const long M = (long)n * h * w;
    if (M <= 0 || M * ROWB >= (1L << 31) || w > MAXW || w < 1 || h < 1) return -1;
    if ((unsigned long long)(M + 256) * (unsigned long long)(h * w) >= (1ULL << 32)) return -1;
    const int tiles = (int)((M + 255) / 256);
    const unsigned mhw = (unsigned)(((1ULL << 32) + (unsigned)(h * w) - 1) / (unsigned)(h * w)), mw = (unsigned)(((1ULL << 32) + (unsigned)w - 1) / (unsigned)w);
    const int q = tiles / 8, r = tiles % 8;
    const int per_xcd = 2 * (q + (r ? 1 : 0));        // (tile, half) pairs of the fullest XCD
    const dim3 grid(8 * per_xcd);
#define R4_ARGS (const char *)x, (const char *)wpk, (const _Float16 *)bias, (const char *)skip, (char *)y, (int)M, h, w, mhw, mw, q, r
    if (skip) hipLaunchKernelGGL((k_conv4r<true, VAR>), grid, dim3(256), 0, st, R4_ARGS);
    else hipLaunchKernelGGL((k_conv4r<false, VAR>), grid, dim3(256), 0, st, R4_ARGS);
#undef R4_ARGS
    return 0;
}

static inline int launch(int n, int h, int w, const void *x, const void *wpk, const void *bias, const void *skip, void *y,
                         hipStream_t st, int var = 1) {
    switch (var) {
#ifdef SGO_CONV4W_VARIANTS
    case 0: return launch_var<0>(n, h, w, x, wpk, bias, skip, y, st);
    case 3: return launch_var<3>(n, h, w, x, wpk, bias, skip, y, st);
    case 5: return launch_var<5>(n, h, w, x, wpk, bias, skip, y, st);        // ablations
    case 17: return launch_var<17>(n, h, w, x, wpk, bias, skip, y, st);      // loads spread inside the bursts
    case 65: return launch_var<65>(n, h, w, x, wpk, bias, skip, y, st);      // ablation: no pixel fragment reads (weight loads stay)
    case 69: return launch_var<69>(n, h, w, x, wpk, bias, skip, y, st);      // ablation: neither
    case 129: return launch_var<129>(n, h, w, x, wpk, bias, skip, y, st);    // ablation: no window restage / no barriers at the chunk boundaries
    case 197: return launch_var<197>(n, h, w, x, wpk, bias, skip, y, st);    // ... on top of 69 (no operand traffic at all)
    case 33: return launch_var<33>(n, h, w, x, wpk, bias, skip, y, st);      // ablation: chunk 0's window is not staged
    case 21: return launch_var<21>(n, h, w, x, wpk, bias, skip, y, st);
#endif
    default: return launch_var<1>(n, h, w, x, wpk, bias, skip, y, st);
    }
}

}  // namespace sgo_conv4r

#undef R4_AS1
#undef R4_AS3
#undef R4_BARRIER
#undef R4_DS_READ128
#undef R4_DS_READ64
#undef R4_DS_WRITE64
#undef R4_GLDS
#undef R4_LDS16
#undef R4_LGKM0
#undef R4_MFMA
#undef R4_PRIO
#undef R4_READ_A
#undef R4_SHIFT
#undef R4_STAGE_WP
#undef R4_TILE
#undef R4_VMWAIT
#undef R4_VMWAITW
#undef R4_VMWAIT_LATE
#undef R4_VMWAIT_SUM
#undef R4_LOADW
#undef R4_WAITW
#undef R4_TIEW
#undef R4_LOADP
#undef R4_M1
#undef R4_M2
#undef R4_M4
#undef R4_G2
#undef R4_G4
#undef R4_TILE_S
