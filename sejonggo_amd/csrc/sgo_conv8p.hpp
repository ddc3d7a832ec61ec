// sgo_conv8p.hpp -- hand-written MFMA implicit-GEMM kernel for the residual tower's 3x3 / 256 -> 256 'same'
// convolution with bias (+ skip) + ReLU fused (model.py:37-46 of the reference: Conv2D -> BatchNorm (folded) -> [Add]
// -> ReLU).  NHWC fp16 in/out, weights [K][3][3][C] fp16, fp32 accumulate.  gfx950 only.
//
// GEMM view: M = n*h*w output pixels, N = 256 output channels, K = 9 taps x 256 input channels = 36 K-tiles of 64.
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

#define SGO_VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ __launch_bounds__(512) void k_conv8p(const char *__restrict__ xb, const char *__restrict__ wb,
                                                 const _Float16 *__restrict__ bias, const char *__restrict__ skipb,
                                                 char *__restrict__ yb, const char *__restrict__ zb, int M, int H, int W) {
    __shared__ __attribute__((aligned(1024))) char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tile = blockIdx.x;
    const int HW = H * W;

    // ---- staging assignment: instruction i of this wave fills rows (wid*2+i)*8 + (lane>>3) of a 128-row granule
    int aoff[2][2];   // byte offset of the lane's 16-B chunk in x, tap (1,1)
    int amask[2];     // 2 x 9 validity bits
    int boff[2][2];   // byte offset in the weights, K-tile 0
#pragma unroll
    for (int g = 0; g < 2; g++) {
        amask[g] = 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = (wid * 2 + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((i << 2) | (lane >> 4));
            const int p = tile * 256 + g * 128 + row;
            const int q = p % HW, yy = q / W, xx = q - yy * W;
            int m = 0;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
                if (p < M && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) m |= 1 << t;
            }
            amask[g] |= m << (9 * i);
            aoff[g][i] = p * ROWB + c * 16;
            boff[g][i] = (g * 128 + row) * WROWB + c * 16;
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

#define SGO_GLDS(src, ldsoff) \
    __builtin_amdgcn_global_load_lds((const SGO_AS1 void *)(src), (SGO_AS3 void *)((SGO_AS3 char *)smem + (ldsoff)), 16, 0, 0)

// stage the pixel granule G (0 lo, 1 hi) of K-tile ts into buffer BUF
#define SGO_STAGE_A(BUF, G, ts)                                                                       \
    do {                                                                                              \
        const int tap_ = (ts) >> 2, cc_ = (ts) & 3;                                                   \
        const int dy_ = (tap_ * 11) >> 5, dx_ = tap_ - 3 * dy_;                                       \
        const int toff_ = ((dy_ - 1) * W + (dx_ - 1)) * ROWB + cc_ * 128;                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; i_++) {                                            \
            const bool ok_ = (amask[G] >> (9 * i_ + tap_)) & 1;                                       \
            const char *src_ = ok_ ? xb + (unsigned)(aoff[G][i_] + toff_) : zb;                       \
            SGO_GLDS(src_, (BUF) * 65536 + (G) * 16384 + (wid * 2 + i_) * 1024);                      \
        }                                                                                             \
    } while (0)
// stage the channel granule G of K-tile ts
#define SGO_STAGE_B(BUF, G, ts)                                                                       \
    do {                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; i_++) {                                            \
            const char *src_ = wb + (unsigned)(boff[G][i_] + (ts) * 128);                             \
            SGO_GLDS(src_, (BUF) * 65536 + 32768 + (G) * 16384 + (wid * 2 + i_) * 1024);              \
        }                                                                                             \
    } while (0)
#define SGO_LDS16(off) (*reinterpret_cast<const half8 *>(smem + (off)))
#define SGO_READ_A(BUF, G)                                                                            \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) {                                             \
        pa[mt_][0] = SGO_LDS16((BUF) * 65536 + (G) * 16384 + mt_ * 2048 + rdA0);                      \
        pa[mt_][1] = SGO_LDS16((BUF) * 65536 + (G) * 16384 + mt_ * 2048 + rdA1);                      \
    }
#define SGO_READ_B(BUF, G, dst)                                                                       \
    _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) {                                             \
        dst[nt_][0] = SGO_LDS16((BUF) * 65536 + 32768 + (G) * 16384 + nt_ * 2048 + rdB0);             \
        dst[nt_][1] = SGO_LDS16((BUF) * 65536 + 32768 + (G) * 16384 + nt_ * 2048 + rdB1);             \
    }
#define SGO_SYNC_IN()                                 \
    __builtin_amdgcn_s_barrier();                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                \
    __builtin_amdgcn_s_setprio(1)
#define SGO_MFMA(QM, QN, wfrag)                                                                        \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++) _Pragma("unroll") for (int mt_ = 0; mt_ < 4; mt_++) \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; nt_++) acc[QM][QN][mt_][nt_] =                    \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag[nt_][ks_], pa[mt_][ks_], acc[QM][QN][mt_][nt_], 0, 0, 0)
#define SGO_SYNC_OUT()                 \
    __builtin_amdgcn_s_setprio(0);     \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier()

// one K-tile t held in buffer BUF; S12 / S34: whether K-tiles t+1 / t+2 exist; W1, W2, W4: vmcnt of phases 1, 2, 4
#define SGO_TILE(BUF, t, S12, S34, W1, W2, W4)          \
    do {                                                \
        SGO_READ_B(BUF, 0, wlo);                        \
        __builtin_amdgcn_sched_barrier(0);              \
        SGO_READ_A(BUF, 0);                             \
        if (S12) SGO_STAGE_B((BUF) ^ 1, 1, (t) + 1);    \
        SGO_VMWAIT(W1);                                 \
        SGO_SYNC_IN();                                  \
        SGO_MFMA(0, 0, wlo);                            \
        SGO_SYNC_OUT();                                 \
        SGO_READ_B(BUF, 1, whi);                        \
        if (S12) SGO_STAGE_A((BUF) ^ 1, 1, (t) + 1);    \
        SGO_VMWAIT(W2);                                 \
        SGO_SYNC_IN();                                  \
        SGO_MFMA(0, 1, whi);                            \
        SGO_SYNC_OUT();                                 \
        SGO_READ_A(BUF, 1);                             \
        if (S34) SGO_STAGE_A(BUF, 0, (t) + 2);          \
        SGO_SYNC_IN();                                  \
        SGO_MFMA(1, 1, whi);                            \
        SGO_SYNC_OUT();                                 \
        if (S34) SGO_STAGE_B(BUF, 0, (t) + 2);          \
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
    if (wr == 1) __builtin_amdgcn_s_barrier();   // the hi pixel group runs one barrier behind

    for (int t = 0; t < NTILE - 2; t += 2) {
        SGO_TILE(0, t, true, true, 8, 8, 8);
        SGO_TILE(1, t + 1, true, true, 8, 8, 8);
    }
    SGO_TILE(0, NTILE - 2, true, false, 8, 8, 4);
    SGO_TILE(1, NTILE - 1, false, false, 2, 0, 0);
    if (wr == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue: bias (+ skip) + ReLU, 8-byte pieces: lane holds channels cb + 4*(lane>>4) .. +3 of pixel lane&15
    const int chl = (lane >> 4) * 4;
    half4 bv[2][2];
#pragma unroll
    for (int qn = 0; qn < 2; qn++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) bv[qn][nt] = *reinterpret_cast<const half4 *>(bias + qn * 128 + wc * 32 + nt * 16 + chl);
#pragma unroll
    for (int qm = 0; qm < 2; qm++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const int p = tile * 256 + qm * 128 + wr * 64 + mt * 16 + (lane & 15);
            if (p < M) {
                const size_t rowoff = (size_t)p * ROWB;
#pragma unroll
                for (int qn = 0; qn < 2; qn++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) {
                        const int ch = qn * 128 + wc * 32 + nt * 16 + chl;
                        floatx4 v = acc[qm][qn][mt][nt];
                        half4 o;
                        if (skipb) {
                            const half4 s = *reinterpret_cast<const half4 *>(skipb + rowoff + ch * 2);
#pragma unroll
                            for (int j = 0; j < 4; j++) v[j] += (float)s[j];
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float f = v[j] + (float)bv[qn][nt][j];
                            o[j] = (_Float16)(f > 0.f ? f : 0.f);
                        }
                        *reinterpret_cast<half4 *>(yb + rowoff + ch * 2) = o;
                    }
            }
        }
}

// x: [n][h][w][256] fp16, wgt: [256][3][3][256] fp16, bias: fp16[256], skip (may be null) / y: [n][h][w][256] fp16,
// zeros: at least 16 bytes of device zeros.  Requires n*h*w*512 < 2^31 (the caller slices larger batches).
static inline int launch(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y,
                         const void *zeros, hipStream_t st) {
    const long M = (long)n * h * w;
    if (M <= 0 || M * ROWB >= (1L << 31)) return -1;
    const int tiles = (int)((M + 255) / 256);
    hipLaunchKernelGGL(k_conv8p, dim3(tiles), dim3(512), 0, st, (const char *)x, (const char *)wgt, (const _Float16 *)bias,
                       (const char *)skip, (char *)y, (const char *)zeros, (int)M, h, w);
    return 0;
}

}  // namespace sgo_conv8p
