// sgo_stem_packed.hpp -- the STEM convolution of the resident policy/value net straight from PACKED POSITION RECORDS
// (model.py:57-60 of the reference: Conv2D(256, 3x3, padding omitted => 'valid') on the 17-plane board tensor, BatchNorm
// folded, ReLU; the symmetry of random_symmetry_predict, symmetry.py:127-132, applied while the planes are expanded).
//
// Why: the network input of a leaf is 16 bit-planes + one colour bit -- a 768-byte record at 19x19 that board_advance has
// just written.  Presenting it to a convolution as fp16 NHWC rows costs 23 KB per leaf (17 planes padded to 32 channels),
// written by one kernel and read back by the next: 12x the record, pure expansion traffic (round 2: 189 MB per 8 192-leaf
// launch).  Here the expansion happens in LDS: a workgroup reads the records its pixels touch (<= 2 per 256-pixel tile at
// 19x19), turns them into [point][16 planes] fp16 there and feeds the MFMAs from that.  HBM sees 768 B in and the 256-channel
// activations out, nothing else.
//
// Math: the colour plane (plane 16 = +-1 over the whole board) under a 'valid' convolution contributes the constant
// c * sum_taps w[k][16][tap] to every output pixel, so it is folded into the bias per position (wc[k], fp32); what is
// left is K = 9 taps x 16 stone planes.  One v_mfma_f32_16x16x32_f16 covers TWO taps (lane groups 0,1: planes 0-7 / 8-15 of
// tap 2j; groups 2,3: the same of tap 2j+1), five K-steps per output tile, the tenth tap's weights being zero.
//
// GEMM view and decomposition are the old k_stem's (sgo_stem.hpp): M = n*(S-2)^2 output pixels, N = 256 channels, one
// PERSISTENT 512-thread workgroup per CU walking 256-pixel tiles, 8 waves = 2 pixel halves x 4 channel groups, weights
// resident in LDS (256 rows x 320 B, pitch 336 B), bias + colour term + ReLU in registers, 16-byte stores.
// Per tile: (A) the raw records of the tile's positions, prefetched into registers during the previous tile's MFMAs, go to
// LDS; (B) every thread expands ~1.4 points (16 plane bits -> 16 halves, symmetry on the gather side); (C) MFMAs + stores.
// gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgo_stemp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));

constexpr int COUT = 256, TAPS = 10, KROW = TAPS * 16 * 2;   // bytes of one output channel's weights: [10 taps][16 planes] fp16
constexpr int WPAD = KROW + 16;                              // LDS pitch of a weight row (336 B = 21 x 16: odd in 16-B slots)
constexpr int W_BYTES = COUT * WPAD;                         // 86 016 B
constexpr int PROW = 32, YROW = COUT * 2;                    // bytes per expanded point (16 halves) / per output pixel

template <int S>
struct Cfg {
    static constexpr int N = S * S, NW = (N + 31) / 32, RW = 16 * NW;
    static constexpr int HO = S - 2, HWO = HO * HO;
    static constexpr int NPOS = (255 + HWO - 1) / HWO + 1;      // positions a 256-pixel tile can touch
    static constexpr int REC_WORDS = NPOS * RW;                 // <= 512: one word per thread
    static constexpr int IN_BYTES = NPOS * N * PROW;
    static constexpr uint32_t META_BIT = 0x80000000u;
    static_assert(REC_WORDS <= 512, "one record word per thread");
};

// transformed[i][j] = source[si][sj], symmetry.py:45-114 (same map as sgo_rules.hip sym_src)
__device__ __forceinline__ int sym_point(int S, int k, int i, int j) {
    int si = i, sj = j;
    switch (k) {
    case 1: si = j; sj = i; break;
    case 2: sj = S - 1 - j; break;
    case 3: si = S - 1 - i; break;
    case 4: si = j; sj = S - 1 - i; break;
    case 5: si = S - 1 - i; sj = S - 1 - j; break;
    case 6: si = S - 1 - j; sj = i; break;
    case 7: si = S - 1 - j; sj = S - 1 - i; break;
    default: break;
    }
    return si * S + sj;
}

template <int S>
__global__ __launch_bounds__(512) void k_stem_packed(const uint32_t *__restrict__ recs, const int32_t *__restrict__ idx, int n, int k_imm,
                                                      const int32_t *__restrict__ k_dev, const char *__restrict__ wb,
                                                      const _Float16 *__restrict__ bias, const float *__restrict__ wcol,
                                                      char *__restrict__ yb, int tiles) {
    using C = Cfg<S>;
    __shared__ __attribute__((aligned(1024))) char sW[W_BYTES];
    __shared__ __attribute__((aligned(16))) char sIn[C::IN_BYTES];
    __shared__ uint32_t sRec[C::REC_WORDS];
    __shared__ float sCol[C::NPOS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3, g = lane >> 4;
    const int M = n * C::HWO;
    const int ksym = k_dev ? (*k_dev & 7) : k_imm;

    // ---- weights -> LDS, once per workgroup
    for (int c = tid; c < COUT * (KROW / 16); c += 512) {
        const int row = c / (KROW / 16), col = c - row * (KROW / 16);
        *reinterpret_cast<intx4 *>(sW + row * WPAD + col * 16) = *reinterpret_cast<const intx4 *>(wb + (size_t)row * KROW + col * 16);
    }
    // MFMA row r of channel tile nt is output channel wc*64 + (nt>>1)*32 + (r>>2)*8 + (nt&1)*4 + (r&3) (see sgo_stem.hpp: a lane's
    // rows of tiles 2j and 2j+1 are then eight consecutive channels = one 16-byte store)
    const int arow = wc * 64 + ((lane & 15) >> 2) * 8 + (lane & 3);
    const int wrow = arow * WPAD + g * 16;
    // the lane's tap of K-step j is 2j + (g >> 1) (tap 9 does not exist: its weights are zero, it re-reads tap 8's point)
    int tapoff[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        int t = 2 * j + (g >> 1);
        t = t > 8 ? 8 : t;
        tapoff[j] = ((t / 3) * S + t % 3) * PROW + (g & 1) * 16;
    }

    // raw record words of a tile's positions: thread t holds word (t % RW) of position s0 + t / RW
    auto fetch = [&](int tile) -> uint32_t {
        const int s0 = (tile * 256) / C::HWO;
        const int ps = tid / C::RW, w = tid - ps * C::RW;
        const int s = s0 + ps;
        if (tid < C::REC_WORDS && tile < tiles && s < n) return recs[(size_t)(idx ? idx[s] : s) * C::RW + w];
        return 0u;
    };
    uint32_t pre = fetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int s0 = (tile * 256) / C::HWO;
        if (tid < C::REC_WORDS) sRec[tid] = pre;
        __syncthreads();                                    // (A) records in LDS; every wave has left the previous tile's MFMAs
        pre = fetch(tile + gridDim.x);                      // the next tile's records travel under this tile's work
        for (int i = tid; i < C::NPOS * C::N; i += 512) {   // (B) expansion: one point per thread and pass
            const int ps = i / C::N, pt = i - ps * C::N;
            const uint32_t *rec = sRec + ps * C::RW;
            const int flip = (rec[C::NW - 1] & C::META_BIT) ? 1 : 0;     // network planes are relative to the side to move
            const int sp = sym_point(S, ksym, pt / S, pt % S);
            half8 lo, hi;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                lo[c] = (_Float16)(float)((rec[(c ^ flip) * C::NW + (sp >> 5)] >> (sp & 31)) & 1u);
                hi[c] = (_Float16)(float)((rec[((c + 8) ^ flip) * C::NW + (sp >> 5)] >> (sp & 31)) & 1u);
            }
            *reinterpret_cast<half8 *>(sIn + i * PROW) = lo;
            *reinterpret_cast<half8 *>(sIn + i * PROW + 16) = hi;
            if (pt == 0) sCol[ps] = flip ? -1.0f : 1.0f;
        }
        __syncthreads();                                    // (B) done

        // (C) the lane's 8 pixel columns: byte offset of the pixel's top-left input point inside sIn
        int q0[8], psl[8];
#pragma unroll
        for (int mt = 0; mt < 8; mt++) {
            int p = tile * 256 + wr * 128 + mt * 16 + (lane & 15);
            p = p < M ? p : M - 1;
            const int s = p / C::HWO, r = p - s * C::HWO, oy = r / C::HO, ox = r - oy * C::HO;
            psl[mt] = s - s0;
            q0[mt] = (psl[mt] * C::N + oy * S + ox) * PROW;
        }
        floatx4 acc[8][4];
#pragma unroll
        for (int mt = 0; mt < 8; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 5; j++) {
            half8 wf[4], pf[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
                wf[nt] = *reinterpret_cast<const half8 *>(sW + wrow + ((nt >> 1) * 32 + (nt & 1) * 4) * WPAD + j * 64);
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int mt = 0; mt < 4; mt++) pf[mt] = *reinterpret_cast<const half8 *>(sIn + q0[4 * h + mt] + tapoff[j]);
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int nt = 0; nt < 4; nt++)
                        acc[4 * h + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], pf[mt], acc[4 * h + mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);   // keeps hipcc from hoisting every K-step's fragment loads (it spills the accumulators otherwise)
            }
        }
        // epilogue from the registers: bias + colour term of the pixel's position + ReLU, 16 B (8 channels) per lane and tile pair
        int eo = wc * 64 + g * 8;
        asm volatile("" : "+v"(eo));    // opaque per tile: the bias / colour-weight loads below stay HERE instead of being hoisted
                                        // out of the tile loop, where they would sit in 24 registers under the accumulators
#pragma unroll
        for (int mt = 0; mt < 8; mt++) {
            const int p = tile * 256 + wr * 128 + mt * 16 + (lane & 15);
            const float col = sCol[psl[mt]];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // bias and colour weights of the lane's 8 channels: re-read per use (L1 hits) instead of 24 registers held
                // across the MFMA loop
                const half8 bv = *reinterpret_cast<const half8 *>(bias + eo + j * 32);
                const floatx4 c0 = *reinterpret_cast<const floatx4 *>(wcol + eo + j * 32);
                const floatx4 c1 = *reinterpret_cast<const floatx4 *>(wcol + eo + j * 32 + 4);
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float f = acc[mt][2 * j + (e >> 2)][e & 3] + (float)bv[e] + col * (e < 4 ? c0[e & 3] : c1[e & 3]);
                    o[e] = (_Float16)(f > 0.f ? f : 0.f);
                }
                if (p < M) *reinterpret_cast<half8 *>(yb + (size_t)p * YROW + (eo + j * 32) * 2) = o;
            }
        }
    }
}

// recs: packed position records (RW words each; idx[i] selects the record of row i, null = dense), w10: [256][10][16] fp16,
// bias fp16[256], wcol float[256], y: [n][S-2][S-2][256] fp16.  k_dev (device int, optional) overrides the immediate symmetry k.
template <int S>
static inline int launch(int n, const uint32_t *recs, const int32_t *idx, int k, const int32_t *k_dev, const void *w10, const void *bias,
                         const float *wcol, void *y, hipStream_t st) {
    if (n <= 0 || k < 0 || k > 7) return -1;
    const long M = (long)n * Cfg<S>::HWO;
    if (M + 256 >= (1L << 31)) return -1;
    const int tiles = (int)((M + 255) / 256);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    hipLaunchKernelGGL(k_stem_packed<S>, dim3(tiles < n_cu ? tiles : n_cu), dim3(512), 0, st, recs, idx, n, k, k_dev, (const char *)w10,
                       (const _Float16 *)bias, wcol, (char *)y, tiles);
    return 0;
}

}  // namespace sgo_stemp
