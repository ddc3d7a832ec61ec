// sgo_conv8w.hpp -- hand-written MFMA implicit-GEMM kernel for the residual tower's 3x3 / 256 -> 256 'same'
// convolution with bias (+ skip) + ReLU fused (model.py:37-46 of the reference: Conv2D -> BatchNorm (folded) -> [Add]
// -> ReLU).  NHWC fp16 in/out, weights [K][3][3][C] fp16, fp32 accumulate.  gfx950 only.
//
// GEMM view: M = n*h*w output pixels, N = 256 output channels, K = 9 taps x 256 input channels = 36 K-tiles of 64,
// ordered channel-chunk-major (K-tile kt = chunk kt/9, tap kt%9).  One 512-thread workgroup (8 waves = 2 pixel groups x
// 4 channel groups) per 256 pixels x 256 channels, one workgroup per CU (144 KiB of LDS, ~235 VGPRs).
//
// What makes it a convolution kernel rather than a GEMM on an im2col view: the PIXEL WINDOW.  The nine taps of one
// 64-channel chunk read the same 256 + 2(w+1) pixel rows, shifted by (dy-1)*w + (dx-1).  The window slice (<= 320 rows
// x 128 B) is brought into LDS once per chunk (double-buffered over chunks) and every tap's fragments are ds_read from
// it at a shifted row; a tap that falls off the board reads zeros instead (per-row 9-bit masks, one v_cndmask on the
// address; the zero area mirrors the bank slot of the address it replaces, so the redirect adds no bank conflict).  Global -> LDS traffic per workgroup drops from 2 x 32 KiB per K-tile to 32 KiB of weights
// + 4.4 KiB of window, which is what bounds a 256 x 256 tile on this chip (L2 -> LDS gather rate, not MFMA rate).
//
// * Staging is LDS-DMA only (global_load_lds_dwordx4); one wave instruction fills 8 rows x 128 B.  LDS images are
//   XOR-swizzled on the SOURCE side: window chunk c of row r sits at c ^ (r & 7) (conflict-free for the b128 lane
//   groups of MI355X_MICROARCH.md at EVERY row shift), weight chunk c of row r at c ^ ((r >> 1) & 7).
// * Schedule: 2 phases per K-tile, each { fragment reads | DMA issue | counted vmcnt | lgkmcnt(0) | barrier | 32 MFMA |
//   barrier }; the two pixel groups run one barrier apart, so on every SIMD one wave issues MFMAs while its partner
//   reads LDS (intervals of 32 MFMAs = 512 cycles amortise the barrier and the partner's interference better than 16):
//        phase A: read chan-lo, chan-hi, pixel-lo (window)                                   | MFMA (lo,lo) (lo,hi)
//        phase B: read pixel-hi (window) | stage weights[t+2], window piece of the next chunk | MFMA (hi,hi) (hi,lo)
//   A wave retires its own LDS reads before the barrier that ends its read interval, so a region may be re-staged from
//   the next interval on; staged data is read one phase after the counted wait that retires it (weights[t+1] are waited
//   for in phase B of K-tile t with exactly the younger DMAs left in flight: never vmcnt(0) in the loop).
// * The K loop is unrolled over the nine taps (loop body = 18 K-tiles: taps x chunk parity): every tap shift, mask bit,
//   buffer and weight offset is an immediate, and the per-phase address arithmetic is re-derived from opaque copies so
//   that hipcc neither hoists 144 loop-invariant lane masks into SGPR pairs nor keeps 36 phases of addresses live.  The
//   chip is power-limited under this kernel (tools/convexp/datadep.py): fewer vector / scalar instructions per MFMA is what
//   this buys (+1.6 % over the runtime-decoded loop), not a shorter critical path.
// * MFMA: v_mfma_f32_16x16x32_f16 with the WEIGHTS as the row operand, so a lane ends up with 4 consecutive output
//   channels of one pixel.
// * Epilogue through LDS: the skip rows arrive by DMA (lo half behind the last K-tile, hi half behind the lo half's
//   processing), every lane adds bias/skip, applies ReLU and writes its 8-byte pieces back in place, and whole 512-B rows
//   leave as 16 B per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgo_conv8w {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx2 __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));

#define SGW_AS1 __attribute__((address_space(1)))
#define SGW_AS3 __attribute__((address_space(3)))

constexpr int CIN = 256, COUT = 256, NTILE = 36;
constexpr int ROWB = CIN * 2;        // bytes per pixel row of x / y
constexpr int WROWB = 9 * CIN * 2;   // bytes per output channel of the weights
constexpr int MAXW = 19;             // board width limit (window = 256 + 2 (w + 1) <= 296 of 320 rows)
// LDS map
constexpr int LW0 = 0, LB0 = 40960, LB1 = 73728, LW1 = 106496, LZ = 147456, LZ_BYTES = 3 * 2048 + 256, LDS_BYTES = LZ + LZ_BYTES;

// LDS accesses the compiler must not order against in-flight LDS-DMA (it would drain vmcnt to 0 before each of its own
// ds_read once a DMA is pending): issued as asm, waited for by hand (SGW_LGKM0 = lgkmcnt(0) + a scheduling fence).
#define SGW_DS_READ64(dst, addr, OFF) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define SGW_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define SGW_DS_WRITE64(addr, val, OFF) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(OFF) : "memory")
#define SGW_LGKM0()                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)
#define SGW_PRIO(x) __builtin_amdgcn_s_setprio(x)
#define SGW_VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool HAS_SKIP>
__global__ __launch_bounds__(512) void k_conv8w(const char *__restrict__ xb, const char *__restrict__ wb,
                                                 const _Float16 *__restrict__ bias, const char *__restrict__ skipb,
                                                 char *__restrict__ yb, int M, int H, int W, unsigned magicHW, unsigned magicW,
                                                 int xcd_q, int xcd_r
#ifdef SGO_CONV8_STAMPS
                                                 , long long *stamps
#endif
                                                 ) {
#ifdef SGO_CONV8_STAMPS
    const long long st0 = __builtin_amdgcn_s_memtime();
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int swid = wid;
    // Tile order.  Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8, each XCD has its own
    // L2); neighbouring tiles share 2 (w + 1) halo rows of x.  With the grid split as tiles = 8 q + r, XCD c gets the
    // CONTIGUOUS tile range [c (q+1), ...) for c < r and [r (q+1) + (c - r) q, ...) otherwise, walked in launch order: the
    // halo rows a tile needs were just fetched into the same L2 by its predecessor.  xcd_q < 0: identity order.
    int tile = blockIdx.x;
    if (xcd_q >= 0) {
        const int c = tile & 7, i = tile >> 3;
        tile = (c < xcd_r) ? c * (xcd_q + 1) + i : xcd_r * (xcd_q + 1) + (c - xcd_r) * xcd_q + i;
    }
    const int HW = H * W, HALO = W + 1, NROWS = 256 + 2 * HALO;

    if (tid < LZ_BYTES / 16) *reinterpret_cast<intx4 *>(smem + LZ + tid * 16) = intx4{0, 0, 0, 0};

    // ---- weight staging: instruction i of this wave fills rows (wid*2+i)*8 + (lane>>3) of a 128-row granule, 16-B
    //      chunk (lane&7) ^ swizzle; offsets of (granule, i) differ from (0, 0) by constants and one XOR
    const int boff00 = (wid * 16 + (lane >> 3)) * WROWB + (((lane & 7) ^ (lane >> 4)) << 4);
    // ---- window staging: piece id fills window rows id*8 + (lane>>3); pixel = tile*256 - HALO + row
    const int wsrc = ((lane & 7) ^ ((lane >> 3) & 7)) << 4;
    // ---- fragment read offsets
    const int fragB = (((lane >> 4) ^ ((lane >> 1) & 7)) << 4);
    const int rdB0 = (wc * 32 + (lane & 15)) * 128 + fragB, rdB1 = rdB0 ^ 64;
    const int rowA = HALO + wr * 64 + (lane & 15);   // window row of (G = 0, mt = 0) at tap shift 0

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

#ifdef SGW_NODMA   // ablation build: no DMA inside the K loop
#define SGW_GLDS_LOOP(src, ldsoff) asm volatile("" ::"v"(src))
#else
#define SGW_GLDS_LOOP(src, ldsoff) SGW_GLDS(src, ldsoff)
#endif
#define SGW_GLDS(src, ldsoff) \
    __builtin_amdgcn_global_load_lds((const SGW_AS1 void *)(src), (SGW_AS3 void *)((SGW_AS3 char *)smem + (ldsoff)), 16, 0, 0)

// stage the channel granule G (0 lo, 1 hi) of the K-tile whose weights start at byte koff_ of a filter row into buffer BUF
#define SGW_STAGE_BK(BUF, G, koff_)                                                                   \
    do {                                                                                              \
        int bo_ = boff00;                                                                             \
        asm volatile("" : "+v"(bo_));                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; i_++) {                                            \
            const char *src_ = wb + (unsigned)((bo_ ^ (i_ * 64)) + (i_ * 8 + (G) * 128) * WROWB + (koff_)); \
            SGW_GLDS_LOOP(src_, ((BUF) ? LB1 : LB0) + (G) * 16384 + (swid * 2 + i_) * 1024);           \
        }                                                                                             \
    } while (0)
#define SGW_STAGE_B(BUF, G, ts)                                                                       \
    do {                                                                                              \
        const int kc_ = ((ts) * 57) >> 9;                                                             \
        SGW_STAGE_BK(BUF, G, ((ts) - 9 * kc_) * (CIN * 2) + kc_ * 128);                               \
    } while (0)
// stage window piece `pc` (0..4) of the channel chunk at byte offset ccoff_ of a pixel row into window buffer WPAR: wave
// wid fills rows (pc*8+wid)*8 ..+7
#define SGW_STAGE_WK(WPAR, ccoff_, pc)                                                                \
    do {                                                                                              \
        const int id_ = (pc) * 8 + swid;                                                               \
        if (id_ * 8 < NROWS) {                                                                        \
            int la_ = lane;                                                                           \
            asm volatile("" : "+v"(la_));                                                             \
            int q_ = tile * 256 - HALO + id_ * 8 + (la_ >> 3);                                        \
            q_ = q_ < 0 ? 0 : (q_ < M ? q_ : M - 1);                                                  \
            const char *src_ = xb + (unsigned)(q_ * ROWB + (ccoff_) + wsrc);                          \
            SGW_GLDS_LOOP(src_, ((WPAR) ? LW1 : LW0) + id_ * 1024);                                   \
        }                                                                                             \
    } while (0)
#define SGW_STAGE_W(cc, pc) SGW_STAGE_WK((cc) & 1, (cc) * 128, pc)
#define SGW_LDS16(off) (*reinterpret_cast<const half8 *>(smem + (off)))
// pixel fragments of half G for tap T (compile time) from window buffer WPAR: window row = rowA + G*128 + mt*16 + shift(T),
// zeros where the tap is off the board
#define SGW_SHIFT(T) (((T) / 3 == 0 ? -W : (T) / 3 == 2 ? W : 0) + (T) % 3 - 1)
#define SGW_READ_A(G, T, WPAR)                                                                        \
    do {                                                                                              \
        int ra_ = rowA;                                                                               \
        asm volatile("" : "+v"(ra_));   /* opaque: keeps 36 phases of address arithmetic from being hoisted / kept live */ \
        const int rl_ = ra_ + SGW_SHIFT(T);                                                           \
        const int c0_ = (((lane >> 4) ^ rl_) & 7) << 4;                                               \
        const int b0_ = ((WPAR) ? LW1 : LW0) + (G) * 16384 + (rl_ << 7) + c0_, b1_ = b0_ ^ 64;        \
        /* an off-board tap reads zeros from the SAME bank slot its window address has (row parity, chunk): no new */ \
        /* bank conflict with the lanes that do read the window */                                    \
        const int z0_ = LZ + ((rl_ & 1) << 7) + c0_, z1_ = z0_ ^ 64;                                  \
        int mka_ = mk[G][0], mkb_ = mk[G][1];                                                         \
        asm volatile("" : "+v"(mka_), "+v"(mkb_));   /* opaque: 144 loop-invariant lane masks would live in SGPR pairs */ \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) {                                         \
            const bool ok_ = (((mt_ >> 1) ? mkb_ : mka_) & (1 << ((mt_ & 1) * 9 + (T)))) != 0;        \
            pa[mt_][0] = SGW_LDS16((ok_ ? b0_ : z0_) + mt_ * 2048);                                   \
            pa[mt_][1] = SGW_LDS16((ok_ ? b1_ : z1_) + mt_ * 2048);                                   \
        }                                                                                             \
    } while (0)
#define SGW_READ_B(BUF, G, dst)                                                                       \
    _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) {                                             \
        dst[nt_][0] = SGW_LDS16(((BUF) ? LB1 : LB0) + (G) * 16384 + nt_ * 2048 + rdB0);               \
        dst[nt_][1] = SGW_LDS16(((BUF) ? LB1 : LB0) + (G) * 16384 + nt_ * 2048 + rdB1);               \
    }
// end of a read/stage interval: own LDS reads retired BEFORE the barrier, so the partner group may re-stage what was
// read from the very next interval on
#define SGW_SYNC_IN()                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                 \
    __builtin_amdgcn_s_barrier();                      \
    __builtin_amdgcn_sched_barrier(0);                 \
    SGW_PRIO(1)
#ifdef SGW_NOMFMA   // ablation build: fragments kept live, no matrix work
#define SGW_MFMA(QM, QN, wfrag)                                                                        \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) {                                              \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) asm volatile("" ::"v"(pa[mt_][ks_]));      \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) asm volatile("" ::"v"(wfrag[nt_][ks_]));   \
    }
#else
#define SGW_MFMA(QM, QN, wfrag)                                                                        \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) acc[QM][QN][mt_][nt_] =                    \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[nt_][ks_], pa[mt_][ks_], acc[QM][QN][mt_][nt_], 0, 0, 0)
#endif
#define SGW_SYNC_OUT()                 \
    SGW_PRIO(0);                       \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier()
// counted wait of phase B: retires the weights of K-tile t+1 (4 DMAs issued one K-tile ago); younger than those are the
// window piece of the previous K-tile, the 4 weight DMAs and the window piece of this one
#define SGW_WAIT_B(nyoung)                     \
    do {                                       \
        if ((nyoung) == 4) SGW_VMWAIT(4);      \
        else if ((nyoung) == 5) SGW_VMWAIT(5); \
        else SGW_VMWAIT(6);                    \
    } while (0)

// One K-tile with everything but the chunk pair index kk (0 / 1) known at compile time: tap T, chunk parity CP (chunk cc =
// 2 kk + CP), weight buffer BUF = (T + CP) & 1.  Tile index t = 18 kk + 9 CP + T.  Two phases of 32 MFMAs:
//   phase A: read chan-lo, chan-hi, pixel-lo                                                | MFMA (lo,lo) (lo,hi)
//   phase B: read pixel-hi | stage weights[t+2], window piece of chunk cc+1 (taps 0..4)     | MFMA (hi,hi) (hi,lo)
// The counted wait of phase B retires weights[t+1] (4 DMAs issued one K-tile ago); younger than those are the previous
// K-tile's window piece, the 4 weight DMAs and this tile's piece -- counted per WAVE (a wave whose rows of piece 4 lie
// beyond the window issues nothing for it).
#define SGW_WP_COND(CP, TP) ((TP) >= 0 && (TP) < 5 && ((CP) == 0 || kk == 0) && ((TP) < 4 || (4 * 8 + swid) * 8 < NROWS))
#define SGW_TILE_U(T, CP)                                                                                 \
    do {                                                                                                  \
        constexpr int BUF_ = ((T) + (CP)) & 1, T2_ = ((T) + 2) % 9, CARRY_ = ((T) + 2) / 9;               \
        int swid = wid;                                                                                   \
        asm volatile("" : "+s"(swid));   /* opaque per K-tile: LDS-DMA offsets are recomputed, not kept in 100+ SGPRs */ \
        const bool last2_ = (CP) == 1 && (T) >= 7 && kk == 1;   /* K-tiles 34 and 35: nothing left to stage */ \
        SGW_READ_B(BUF_, 0, wlo);                                                                         \
        SGW_READ_B(BUF_, 1, whi);                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        SGW_READ_A(0, T, CP);                                                                             \
        if (HAS_SKIP && (CP) == 1 && (T) == 8 && kk == 1) SGW_SKIP_LO(0);                                 \
        SGW_SYNC_IN();                                                                                    \
        SGW_MFMA(0, 0, wlo);                                                                              \
        SGW_MFMA(0, 1, whi);                                                                              \
        SGW_SYNC_OUT();                                                                                   \
        SGW_READ_A(1, T, CP);                                                                             \
        if (!last2_) {                                                                                    \
            const int koff_ = T2_ * (CIN * 2) + ((CP) + CARRY_) * 128 + kk * 256;                         \
            SGW_STAGE_BK(BUF_, 0, koff_);                                                                 \
            SGW_STAGE_BK(BUF_, 1, koff_);                                                                 \
            const bool wp_ = SGW_WP_COND(CP, T), wpprev_ = SGW_WP_COND(CP, (T) - 1);                      \
            if (wp_) SGW_STAGE_WK(((CP) + 1) & 1, (2 * kk + (CP) + 1) * 128, T);                          \
            SGW_WAIT_B(4 + (wp_ ? 1 : 0) + (wpprev_ ? 1 : 0));                                            \
        } else if ((T) == 7) {                                                                            \
            SGW_VMWAIT(0);   /* K-tile 34: K-tile 35's weights */                                         \
        } else {                                                                                          \
            if (HAS_SKIP) SGW_SKIP_LO(1);                                                                 \
        }                                                                                                 \
        SGW_SYNC_IN();                                                                                    \
        SGW_MFMA(1, 1, whi);                                                                              \
        SGW_MFMA(1, 0, wlo);                                                                              \
        SGW_SYNC_OUT();                                                                                   \
    } while (0)

    // ---- prologue: window of chunk 0, K-tile 0's weights; K-tile 1's weights stay in flight
#pragma unroll
    for (int pc = 0; pc < 5; pc++) SGW_STAGE_W(0, pc);
    SGW_STAGE_B(0, 0, 0);
    SGW_STAGE_B(0, 1, 0);
    SGW_STAGE_B(1, 0, 1);
    SGW_STAGE_B(1, 1, 1);
    // ---- tap-validity masks of the lane's 8 fragment rows: row (G, mt) = G*128 + wr*64 + mt*16 + (lane&15);
    //      mk[G][mt>>1] holds 9 bits per row at bit (mt&1)*9.  Divisions by h*w and w through the host's magic numbers
    //      (ceil(2^32/d): exact for p*d < 2^32).
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
    SGW_VMWAIT(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero row
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // the hi pixel group runs one barrier behind
#ifdef SGO_CONV8_STAMPS
    const long long st1 = __builtin_amdgcn_s_memtime();
#endif

    // skip rows of the tile's half HF (128 pixels x 512 B) into LDS [HF*64 KiB, +64 KiB) by DMA: instruction j of this
    // wave fills rows (wid*8+j)*2 + (lane>>5); 16-B chunk c of row r sits at chunk c ^ (r & 15)
#define SGW_STAGE_SKIP(HF, j)                                                                          \
    do {                                                                                               \
        const int r_ = (wid * 8 + (j)) * 2 + (elane >> 5);                                             \
        int p_ = tile * 256 + (HF) * 128 + r_;                                                         \
        p_ = p_ < M ? p_ : M - 1;                                                                      \
        SGW_GLDS(skipb + (unsigned)(p_ * ROWB + (((elane & 31) ^ (r_ & 15)) << 4)), (HF) * 65536 + (wid * 8 + (j)) * 1024); \
    } while (0)
#define SGW_NOEXTRA(ph) do { } while (0)
#define SGW_SKIP_LO(ph) \
    do { SGW_STAGE_SKIP(0, 4 * (ph)); SGW_STAGE_SKIP(0, 4 * (ph) + 1); SGW_STAGE_SKIP(0, 4 * (ph) + 2); SGW_STAGE_SKIP(0, 4 * (ph) + 3); } while (0)

    int elane = lane;   // opaque copy (made so inside the loop's last iteration): keeps the epilogue's address arithmetic late
    for (int kk = 0; kk < 2; kk++) {
        if (kk == 1) asm volatile("" : "+v"(elane));
        SGW_TILE_U(0, 0); SGW_TILE_U(1, 0); SGW_TILE_U(2, 0); SGW_TILE_U(3, 0); SGW_TILE_U(4, 0);
        SGW_TILE_U(5, 0); SGW_TILE_U(6, 0); SGW_TILE_U(7, 0); SGW_TILE_U(8, 0);
        SGW_TILE_U(0, 1); SGW_TILE_U(1, 1); SGW_TILE_U(2, 1); SGW_TILE_U(3, 1); SGW_TILE_U(4, 1);
        SGW_TILE_U(5, 1); SGW_TILE_U(6, 1); SGW_TILE_U(7, 1); SGW_TILE_U(8, 1);
    }
    asm volatile("" : "+v"(elane));
    if (wr == 0) __builtin_amdgcn_s_barrier();
#ifdef SGO_CONV8_STAMPS
    const long long st2 = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue, one half (128 pixels) at a time through LDS [hf*64 KiB, +64 KiB)
    // bias: 4 channels x (qn, nt) per lane.  Loaded by asm so that the compiler does not see an ordinary load beside the
    // pending DMAs (it would wait vmcnt(0) for it, draining the skip prefetch); retired by the counted waits below.
    intx2 bvi[2][2];
    {
        const _Float16 *bp = bias + wc * 32 + (elane >> 4) * 4;
#pragma unroll
        for (int qn = 0; qn < 2; qn++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
                asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(bvi[qn][nt]) : "v"(bp), "n"((qn * 128 + nt * 16) * 2) : "memory");
    }
    if constexpr (HAS_SKIP) {
#pragma unroll
        for (int j = 0; j < 8; j++) SGW_STAGE_SKIP(1, j);
    }
    const int epx = (wr * 64 + (elane & 15)) * 512 + ((elane >> 4) & 1) * 8;
    const int epc = ((wc * 4 + (elane >> 5)) ^ (elane & 15)) << 4;
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const int a0 = hf * 65536 + epx + epc, a1 = hf * 65536 + epx + (epc ^ 32);   // nt = 0 / 1
        intx2 sk[4][2][2];
        if (!HAS_SKIP && hf == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HAS_SKIP) {
            // own DMAs of this half have landed once at most the 8 younger operations (the other half's DMAs, or the 8
            // row stores of half 0) are outstanding; the barrier publishes everyone's
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int qn = 0; qn < 2; qn++) {
                    SGW_DS_READ64(sk[mt][qn][0], a0, mt * 8192 + qn * 256);
                    SGW_DS_READ64(sk[mt][qn][1], a1, mt * 8192 + qn * 256);
                }
            SGW_LGKM0();
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
                    if (nt == 0) SGW_DS_WRITE64(a0, oi, mt * 8192 + qn * 256);
                    else SGW_DS_WRITE64(a1, oi, mt * 8192 + qn * 256);
                }
        SGW_LGKM0();
        __builtin_amdgcn_s_barrier();
        intx4 ov[8];
        const int a2 = hf * 65536 + wid * 8192 + elane * 16;
#pragma unroll
        for (int j = 0; j < 8; j++) SGW_DS_READ128(ov[j], a2, j * 1024);
        const int p0 = tile * 256 + hf * 128 + wid * 16 + (elane >> 5);
        char *dst = yb + (size_t)p0 * ROWB;   // row (wid*16 + 2j + (lane>>5)) of the half; its swizzle key is 2j + (lane>>5)
        SGW_LGKM0();
        if (tile * 256 + hf * 128 + 128 <= M) {   // whole half inside the tensor (wave-uniform)
#pragma unroll
            for (int j = 0; j < 8; j++)
                *reinterpret_cast<intx4 *>(dst + j * 2 * ROWB + (((elane & 31) ^ (j * 2 + (elane >> 5))) << 4)) = ov[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (p0 + j * 2 < M)
                    *reinterpret_cast<intx4 *>(dst + j * 2 * ROWB + (((elane & 31) ^ (j * 2 + (elane >> 5))) << 4)) = ov[j];
        }
    }
#ifdef SGO_CONV8_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long st3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        long long *o = stamps + ((size_t)tile * 8 + wid) * 6;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = __builtin_amdgcn_s_memrealtime();
        o[5] = rt0;
    }
#endif
}

// x: [n][h][w][256] fp16, wgt: [256][3][3][256] fp16, bias: fp16[256], skip (may be null) / y: [n][h][w][256] fp16.
// Requires w <= 19 and n*h*w*512 < 2^31 (the caller slices larger batches).
// tile order of the launches: 1 = XCD-contiguous (default), 0 = identity (A/B measurements)
static inline int &tile_order() {
    static int mode = 1;
    return mode;
}

static inline int launch(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y,
                         hipStream_t st
#ifdef SGO_CONV8_STAMPS
                         , long long *stamps
#endif
                         ) {
    const long M = (long)n * h * w;
    if (M <= 0 || M * ROWB >= (1L << 31) || w > MAXW || w < 1 || h < 1) return -1;
    // the pixel -> (sample, y, x) split uses magic-number division, exact only while p * (h*w) < 2^32 for every pixel
    // index the kernel forms (p < M + 256)
    if ((unsigned long long)(M + 256) * (unsigned long long)(h * w) >= (1ULL << 32)) return -1;
    const int tiles = (int)((M + 255) / 256);
    const int xq = tile_order() ? tiles / 8 : -1, xr = tiles % 8;
    const unsigned mhw = (unsigned)(((1ULL << 32) + (unsigned)(h * w) - 1) / (unsigned)(h * w)), mw = (unsigned)(((1ULL << 32) + (unsigned)w - 1) / (unsigned)w);
#ifdef SGO_CONV8_STAMPS
#define SGW_ARGS (const char *)x, (const char *)wgt, (const _Float16 *)bias, (const char *)skip, (char *)y, (int)M, h, w, mhw, mw, xq, xr, stamps
#else
#define SGW_ARGS (const char *)x, (const char *)wgt, (const _Float16 *)bias, (const char *)skip, (char *)y, (int)M, h, w, mhw, mw, xq, xr
#endif
    if (skip) hipLaunchKernelGGL(k_conv8w<true>, dim3(tiles), dim3(512), 0, st, SGW_ARGS);
    else hipLaunchKernelGGL(k_conv8w<false>, dim3(tiles), dim3(512), 0, st, SGW_ARGS);
#undef SGW_ARGS
    return 0;
}

}  // namespace sgo_conv8w

// the kernel's working macros stay private to this header
#undef SGW_ARGS
#undef SGW_DS_READ128
#undef SGW_DS_READ64
#undef SGW_DS_WRITE64
#undef SGW_GLDS
#undef SGW_GLDS_LOOP
#undef SGW_LDS16
#undef SGW_LGKM0
#undef SGW_MFMA
#undef SGW_NOEXTRA
#undef SGW_PRIO
#undef SGW_READ_A
#undef SGW_READ_B
#undef SGW_SHIFT
#undef SGW_SKIP_LO
#undef SGW_STAGE_B
#undef SGW_STAGE_BK
#undef SGW_STAGE_SKIP
#undef SGW_STAGE_W
#undef SGW_STAGE_WK
#undef SGW_SYNC_IN
#undef SGW_SYNC_OUT
#undef SGW_TILE_U
#undef SGW_VMWAIT
#undef SGW_WAIT_B
#undef SGW_WP_COND
