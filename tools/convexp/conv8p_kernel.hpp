// sgo_conv8p.hpp -- hand-written MFMA implicit-GEMM kernel for the residual tower's 3x3 / 256 -> 256 'same'
// convolution with bias (+ skip) + ReLU fused (model.py:37-46 of the reference: Conv2D -> BatchNorm (folded) -> [Add]
// -> ReLU).  NHWC fp16 in/out, weights [K][3][3][C] fp16, fp32 accumulate.  gfx950 only.
//
// GEMM view: M = n*h*w output pixels, N = 256 output channels, K = 9 taps x 256 input channels = 36 K-tiles of 64,
// ordered channel-chunk-major (K-tile kt = chunk kt/9, tap kt%9): the nine taps of one 64-channel chunk re-read the same
// 128-B slice of the tile's pixel window back to back, so eight of nine reads hit the XCD's L2.
// One 512-thread workgroup (8 waves = 2 pixel groups x 4 channel groups) per 256 pixels x 256 channels; LDS holds two
// K-tile buffers of 64 KiB: pixel rows [lo 128 | hi 128] x 128 B and channel rows [lo 128 | hi 128] x 128 B.
//
// * Staging is LDS-DMA only (global_load_lds_dwordx4): one wave instruction fills 8 rows x 128 B, whole 128-B lines of
//   8 pixels (or 8 output channels).  The im2col gather is the per-lane SOURCE address: pixel p of tap (dy,dx) reads
//   x[p + (dy-1)*w + (dx-1)], and a tap that falls off the board reads a zero line instead (per-row 9-bit masks).
// * The LDS image is XOR-swizzled on the source side (16-B chunk c of row r sits at chunk c ^ ((r >> 1) & 7)), which
//   makes every ds_read_b128 fragment read conflict-free for the b128 lane groups of MI355X_MICROARCH.md (LDS table).
// * Schedule: 4 phases per K-tile, each { fragment reads | one 16-KiB stage | counted vmcnt | barrier | 16 MFMA |
//   barrier }; the two pixel groups run one barrier apart, so on every SIMD one wave issues MFMAs while its partner
//   reads LDS and issues DMA.  Stages are issued 4 phases before the wait that retires them (vmcnt(8), never 0 in the
//   loop), and every region is re-staged at least two barrier intervals after its last read:
//        phase 1: read pixel-lo + chan-lo | stage chan-hi [t+1] | MFMA (lo,lo)
//        phase 2: read chan-hi            | stage pixel-hi[t+1] | MFMA (lo,hi)
//        phase 3: read pixel-hi           | stage pixel-lo[t+2] | MFMA (hi,hi)
//        phase 4: --                      | stage chan-lo [t+2] | MFMA (hi,lo)
// * MFMA: v_mfma_f32_16x16x32_f16 with the WEIGHTS as the row operand, so a lane ends up with 4 consecutive output
//   channels of one pixel and the epilogue stores 8-byte pieces (bias, skip, ReLU in registers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgo_conv8p {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define SGO_AS1 __attribute__((address_space(1)))
#define SGO_AS3 __attribute__((address_space(3)))

constexpr int CIN = 256, COUT = 256, NTILE = 36;
constexpr int ROWB = CIN * 2;        // bytes per pixel row of x / y
constexpr int WROWB = 9 * CIN * 2;   // bytes per output channel of the weights

typedef int intx2 __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));
// LDS accesses the compiler must not order against in-flight LDS-DMA (it would drain vmcnt to 0 before each of its own
// ds_read once a DMA is pending): issued as asm, waited for by hand (SGO_LGKM0 = lgkmcnt(0) + a scheduling fence).
#define SGO_DS_READ64(dst, addr, OFF) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define SGO_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define SGO_DS_WRITE64(addr, val, OFF) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(OFF) : "memory")
#define SGO_LGKM0()                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)
#define SGO_VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool HAS_SKIP>
__global__ __launch_bounds__(512) void k_conv8p(const char *__restrict__ xb, const char *__restrict__ wb,
                                                 const _Float16 *__restrict__ bias, const char *__restrict__ skipb,
                                                 char *__restrict__ yb, const char *__restrict__ zb, int M, int H, int W
#ifdef SGO_CONV8P_STAMPS
                                                 , long long *stamps
#endif
                                                 ) {
#ifdef SGO_CONV8P_STAMPS
    const long long st0 = __builtin_amdgcn_s_memtime();
#endif
    __shared__ __attribute__((aligned(1024))) char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tile = blockIdx.x;
    const int HW = H * W;

    // ---- staging assignment: instruction i of this wave fills rows (wid*2+i)*8 + (lane>>3) of a 128-row granule, 16-B
    //      chunk (lane&7) ^ swizzle.  Only the offsets of (granule 0, i = 0) are kept; the other three differ by
    //      constants and one XOR (the row part of every offset has zero low 9 bits): (o ^ 64*i) + i*8*rowbytes +
    //      g*128*rowbytes.
    int amask[2];     // 2 x 9 tap-validity bits per granule
    int aoff00, boff00;
    {
        const int row0 = wid * 16 + (lane >> 3), c0 = (lane & 7) ^ (lane >> 4);
        aoff00 = (tile * 256 + row0) * ROWB + c0 * 16;
        boff00 = row0 * WROWB + c0 * 16;
    }
#pragma unroll
    for (int g = 0; g < 2; g++) {
        amask[g] = 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = (wid * 2 + i) * 8 + (lane >> 3);
            const int p = tile * 256 + g * 128 + row;
            const int q = p % HW, yy = q / W, xx = q - yy * W;
            int m = 0;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
                if (p < M && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) m |= 1 << t;
            }
            amask[g] |= m << (9 * i);
        }
    }
    // ---- fragment read offsets (bytes inside a 16-KiB granule)
    const int swz = (lane >> 1) & 7;
    const int fragc = (((lane >> 4) ^ swz) << 4);
    const int rdA0 = (wr * 64 + (lane & 15)) * 128 + fragc, rdA1 = rdA0 ^ 64;
    const int rdB0 = (wc * 32 + (lane & 15)) * 128 + fragc, rdB1 = rdB0 ^ 64;

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

#ifdef SGO_C8_NODMA
#define SGO_GLDS_LOOP(src, ldsoff) asm volatile("" ::"v"(src))
#else
#define SGO_GLDS_LOOP(src, ldsoff) SGO_GLDS(src, ldsoff)
#endif
#define SGO_GLDS(src, ldsoff) \
    __builtin_amdgcn_global_load_lds((const SGO_AS1 void *)(src), (SGO_AS3 void *)((SGO_AS3 char *)smem + (ldsoff)), 16, 0, 0)

// stage the pixel granule G (0 lo, 1 hi) of K-tile ts into buffer BUF
#define SGO_STAGE_A(BUF, G, ts)                                                                       \
    do {                                                                                              \
        const int cc_ = ((ts) * 57) >> 9, tap_ = (ts) - 9 * cc_;                                      \
        const int dy_ = (tap_ * 11) >> 5, dx_ = tap_ - 3 * dy_;                                       \
        const int toff_ = ((dy_ - 1) * W + (dx_ - 1)) * ROWB + cc_ * 128;                             \
        int ao_ = aoff00;                                                                             \
        asm volatile("" : "+v"(ao_));                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; i_++) {                                            \
            const bool ok_ = (amask[G] >> (9 * i_ + tap_)) & 1;                                       \
            const char *src_ = ok_ ? xb + (unsigned)((ao_ ^ (i_ * 64)) + (i_ * 8 + (G) * 128) * ROWB + toff_) : zb; \
            SGO_GLDS_LOOP(src_, (BUF) * 65536 + (G) * 16384 + (wid * 2 + i_) * 1024);                      \
        }                                                                                             \
    } while (0)
// stage the channel granule G of K-tile ts
#define SGO_STAGE_B(BUF, G, ts)                                                                       \
    do {                                                                                              \
        const int kc_ = ((ts) * 57) >> 9, koff_ = ((ts) - 9 * kc_) * (CIN * 2) + kc_ * 128;          \
        int bo_ = boff00;                                                                             \
        asm volatile("" : "+v"(bo_));                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; i_++) {                                            \
            const char *src_ = wb + (unsigned)((bo_ ^ (i_ * 64)) + (i_ * 8 + (G) * 128) * WROWB + koff_); \
            SGO_GLDS_LOOP(src_, (BUF) * 65536 + 32768 + (G) * 16384 + (wid * 2 + i_) * 1024);              \
        }                                                                                             \
    } while (0)
#define SGO_LDS16(off) (*reinterpret_cast<const half8 *>(smem + (off)))
#ifdef SGO_C8_NOREAD
#define SGO_LDS16_LOOP(off) (__builtin_bit_cast(half8, intx4{rdA0, rdB0, (int)(off), rdA1}))
#else
#define SGO_LDS16_LOOP(off) SGO_LDS16(off)
#endif
#define SGO_READ_A(BUF, G)                                                                            \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) {                                             \
        pa[mt_][0] = SGO_LDS16_LOOP((BUF) * 65536 + (G) * 16384 + mt_ * 2048 + rdA0);                      \
        pa[mt_][1] = SGO_LDS16_LOOP((BUF) * 65536 + (G) * 16384 + mt_ * 2048 + rdA1);                      \
    }
#define SGO_READ_B(BUF, G, dst)                                                                       \
    _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) {                                             \
        dst[nt_][0] = SGO_LDS16_LOOP((BUF) * 65536 + 32768 + (G) * 16384 + nt_ * 2048 + rdB0);             \
        dst[nt_][1] = SGO_LDS16_LOOP((BUF) * 65536 + 32768 + (G) * 16384 + nt_ * 2048 + rdB1);             \
    }
#define SGO_SYNC_IN()                                 \
    __builtin_amdgcn_s_barrier();                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                \
    __builtin_amdgcn_s_setprio(1)
#ifdef SGO_C8_NOMFMA   // ablation: fragments kept live, no matrix work
#define SGO_MFMA(QM, QN, wfrag)                                                                        \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) {                                              \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) asm volatile("" ::"v"(pa[mt_][ks_]));      \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) asm volatile("" ::"v"(wfrag[nt_][ks_]));   \
    }
#else
#define SGO_MFMA(QM, QN, wfrag)                                                                        \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) acc[QM][QN][mt_][nt_] =                    \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[nt_][ks_], pa[mt_][ks_], acc[QM][QN][mt_][nt_], 0, 0, 0)
#endif
#define SGO_SYNC_OUT()                 \
    __builtin_amdgcn_s_setprio(0);     \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier()

// one K-tile t held in buffer BUF; S12 / S34: whether K-tiles t+1 / t+2 exist; W1, W2, W4: vmcnt of phases 1, 2, 4
#define SGO_TILE(BUF, t, S12, S34, W1, W2, W4, EXTRA)   \
    do {                                                \
        SGO_READ_B(BUF, 0, wlo);                        \
        __builtin_amdgcn_sched_barrier(0);              \
        SGO_READ_A(BUF, 0);                             \
        if (S12) SGO_STAGE_B((BUF) ^ 1, 1, (t) + 1);    \
        EXTRA(0);                                       \
        SGO_VMWAIT(W1);                                 \
        SGO_SYNC_IN();                                  \
        SGO_MFMA(0, 0, wlo);                            \
        SGO_SYNC_OUT();                                 \
        SGO_READ_B(BUF, 1, whi);                        \
        if (S12) SGO_STAGE_A((BUF) ^ 1, 1, (t) + 1);    \
        EXTRA(1);                                       \
        SGO_VMWAIT(W2);                                 \
        SGO_SYNC_IN();                                  \
        SGO_MFMA(0, 1, whi);                            \
        SGO_SYNC_OUT();                                 \
        SGO_READ_A(BUF, 1);                             \
        if (S34) SGO_STAGE_A(BUF, 0, (t) + 2);          \
        EXTRA(2);                                       \
        SGO_SYNC_IN();                                  \
        SGO_MFMA(1, 1, whi);                            \
        SGO_SYNC_OUT();                                 \
        if (S34) SGO_STAGE_B(BUF, 0, (t) + 2);          \
        EXTRA(3);                                       \
        SGO_VMWAIT(W4);                                 \
        SGO_SYNC_IN();                                  \
        SGO_MFMA(1, 0, wlo);                            \
        SGO_SYNC_OUT();                                 \
    } while (0)

    // ---- prologue: K-tile 0 complete, pixel-lo and chan-lo of K-tile 1 in flight
    SGO_STAGE_A(0, 0, 0);
    SGO_STAGE_B(0, 0, 0);
    SGO_STAGE_B(0, 1, 0);
    SGO_STAGE_A(0, 1, 0);
    SGO_STAGE_A(1, 0, 1);
    SGO_STAGE_B(1, 0, 1);
    SGO_VMWAIT(4);
    __builtin_amdgcn_s_barrier();
#ifndef SGO_C8_NOSTAGGER
    if (wr == 1) __builtin_amdgcn_s_barrier();   // the hi pixel group runs one barrier behind
#endif
#ifdef SGO_CONV8P_STAMPS
    const long long st1 = __builtin_amdgcn_s_memtime();
#endif

    // skip rows of the tile's half HF (128 pixels x 512 B) into LDS buffer HF by DMA: instruction j of this wave fills
    // rows (wid*8+j)*2 + (lane>>5); 16-B chunk c of row r sits at chunk c ^ (r & 15)
#define SGO_STAGE_SKIP(HF, j)                                                                          \
    do {                                                                                               \
        const int r_ = (wid * 8 + (j)) * 2 + (elane >> 5);                                             \
        int p_ = tile * 256 + (HF) * 128 + r_;                                                         \
        p_ = p_ < M ? p_ : M - 1;                                                                      \
        SGO_GLDS(skipb + (unsigned)(p_ * ROWB + (((elane & 31) ^ (r_ & 15)) << 4)), (HF) * 65536 + (wid * 8 + (j)) * 1024); \
    } while (0)
#define SGO_NOEXTRA(ph) do { } while (0)
#define SGO_SKIP_LO(ph) do { SGO_STAGE_SKIP(0, 2 * (ph)); SGO_STAGE_SKIP(0, 2 * (ph) + 1); } while (0)
#define SGO_NOWAIT 63

    for (int t = 0; t < NTILE - 2; t += 2) {
        SGO_TILE(0, t, true, true, 8, 8, 8, SGO_NOEXTRA);
        SGO_TILE(1, t + 1, true, true, 8, 8, 8, SGO_NOEXTRA);
    }
    SGO_TILE(0, NTILE - 2, true, false, 8, 8, 4, SGO_NOEXTRA);
    int elane = lane;   // opaque copy: keeps the epilogue's address arithmetic from being hoisted above the main loop
    asm volatile("" : "+v"(elane));
    // the last K-tile has no successor to stage: its DMA slots carry the lo half of the skip tile into buffer 0 (idle
    // since phase 3 of K-tile 34)
    if constexpr (HAS_SKIP) SGO_TILE(1, NTILE - 1, false, false, 4, 4, 63, SGO_SKIP_LO);
    else SGO_TILE(1, NTILE - 1, false, false, 2, 0, 63, SGO_NOEXTRA);
#ifndef SGO_C8_NOSTAGGER
    if (wr == 0) __builtin_amdgcn_s_barrier();
#endif
#ifdef SGO_CONV8P_STAMPS
    const long long st2 = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue, one half (128 pixels) at a time through the now idle LDS buffer of that half:
    //   (skip rows arrive by DMA) -> every elane adds bias (+ skip) to its 4-channel pieces, ReLU, writes them back in
    //   place -> barrier -> whole 512-B rows leave as 16 B per elane.
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
        for (int j = 0; j < 8; j++) SGO_STAGE_SKIP(1, j);
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
                    SGO_DS_READ64(sk[mt][qn][0], a0, mt * 8192 + qn * 256);
                    SGO_DS_READ64(sk[mt][qn][1], a1, mt * 8192 + qn * 256);
                }
            SGO_LGKM0();
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
                    if (nt == 0) SGO_DS_WRITE64(a0, oi, mt * 8192 + qn * 256);
                    else SGO_DS_WRITE64(a1, oi, mt * 8192 + qn * 256);
                }
        SGO_LGKM0();
        __builtin_amdgcn_s_barrier();
        intx4 ov[8];
        const int a2 = hf * 65536 + wid * 8192 + elane * 16;
#pragma unroll
        for (int j = 0; j < 8; j++) SGO_DS_READ128(ov[j], a2, j * 1024);
        const int p0 = tile * 256 + hf * 128 + wid * 16 + (elane >> 5);
        char *dst = yb + (size_t)p0 * ROWB;   // row (wid*16 + 2j + (lane>>5)) of the half; its swizzle key is 2j + (lane>>5)
        SGO_LGKM0();
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
#ifdef SGO_CONV8P_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long st3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        long long *o = stamps + ((size_t)tile * 8 + wid) * 6;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = __builtin_amdgcn_s_memrealtime();
        o[5] = 0;
    }
#endif
}

// x: [n][h][w][256] fp16, wgt: [256][3][3][256] fp16, bias: fp16[256], skip (may be null) / y: [n][h][w][256] fp16,
// zeros: at least 16 bytes of device zeros.  Requires n*h*w*512 < 2^31 (the caller slices larger batches).
static inline int launch(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y,
                         const void *zeros, hipStream_t st
#ifdef SGO_CONV8P_STAMPS
                         , long long *stamps
#endif
                         ) {
    const long M = (long)n * h * w;
    if (M <= 0 || M * ROWB >= (1L << 31)) return -1;
    const int tiles = (int)((M + 255) / 256);
#ifdef SGO_CONV8P_STAMPS
#define SGO_C8_ARGS (const char *)x, (const char *)wgt, (const _Float16 *)bias, (const char *)skip, (char *)y, (const char *)zeros, (int)M, h, w, stamps
#else
#define SGO_C8_ARGS (const char *)x, (const char *)wgt, (const _Float16 *)bias, (const char *)skip, (char *)y, (const char *)zeros, (int)M, h, w
#endif
    if (skip) hipLaunchKernelGGL(k_conv8p<true>, dim3(tiles), dim3(512), 0, st, SGO_C8_ARGS);
    else hipLaunchKernelGGL(k_conv8p<false>, dim3(tiles), dim3(512), 0, st, SGO_C8_ARGS);
#undef SGO_C8_ARGS
    return 0;
}

}  // namespace sgo_conv8p
