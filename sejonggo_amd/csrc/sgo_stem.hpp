// sgo_stem.hpp -- hand-written MFMA kernel for the STEM convolution of the resident policy/value net: 3x3, 'valid'
// (model.py:58-59 of the reference omits `padding`, so the 19x19 board becomes a 17x17 tower), 17 input planes presented as
// 32 fp16 channels (the network-input row written by k_nn_pack, layout 2: channels 17..31 zero) -> 256
// channels, bias (BatchNorm folded) + ReLU fused.  NHWC fp16 in / out, weights [256][3][3][32] fp16, fp32 accumulate.
// gfx950 only.
//
// GEMM view: M = n*(h-2)*(w-2) output pixels, N = 256, K = 9 taps x 32 channels = 9 K-steps of one
// v_mfma_f32_16x16x32_f16 each.  One 512-thread workgroup (8 waves = 2 pixel halves x 4 channel groups, 128 pixels x 64
// channels per wave) per 256 pixels x 256 channels -- the same decomposition as the tower kernel (sgo_conv8w.hpp), but the
// roles of the memories differ because K is tiny:
//   * WEIGHTS: all 144 KiB of them live in LDS for the whole workgroup (rows padded 576 -> 592 B so that the sixteen rows a
//     ds_read_b128 lane group touches fall into sixteen different 16-byte bank slots); staged once, read 9 x 4 fragments.
//   * PIXELS: the 64-byte input rows (32 channels) are read straight from global memory as MFMA fragments, 16 B per lane:
//     a workgroup's 256 pixels touch ~300 input rows = 19 KiB, which the CU's L1 holds, so the 9-fold tap reuse and the
//     4-fold reuse across channel groups are L1 hits.  A 'valid' convolution has no off-board taps: no masks.  The next
//     tap's fragments are loaded while the current tap's MFMAs run.
//   * OUTPUT: bias + ReLU in registers and 16-byte stores straight from them: the MFMA rows of two neighbouring channel
//     tiles are assigned to output channels so that a lane's 4 + 4 results are eight consecutive channels.
//   * PERSISTENT workgroups (one per CU): the weights are staged once and the workgroup walks tiles b, b + grid, ...; the
//     stores of one tile drain under the loads and MFMAs of the next.
// The work is small (0.35 TFLOP per 8 192-position batch against 1.2 GB written): the kernel matters because it removes the
// last library convolution from the product path, not because of its share of a step (< 1 %).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgo_stem {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx2 __attribute__((ext_vector_type(2)));

constexpr int CIN = 32, COUT = 256, KROW = 9 * CIN * 2;   // bytes of one output channel's weights
constexpr int WPAD = KROW + 16;                           // LDS row pitch of the weights (592 B)
constexpr int LDS_BYTES = COUT * WPAD;                    // 151 552 B
constexpr int XROW = CIN * 2, YROW = COUT * 2;

__global__ __launch_bounds__(512) void k_stem(const char *__restrict__ xb, const char *__restrict__ wb,
                                               const _Float16 *__restrict__ bias, char *__restrict__ yb, int M, int H, int W, int tiles) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3;
    const int HO = H - 2, WO = W - 2, HWO = HO * WO;

    // ---- weights -> LDS, once per workgroup (16 B per thread and step; 9 216 chunks); the workgroup then walks its tiles
    for (int c = tid; c < COUT * (KROW / 16); c += 512) {
        const int row = c / (KROW / 16), col = c - row * (KROW / 16);
        *reinterpret_cast<intx4 *>(smem + row * WPAD + col * 16) = *reinterpret_cast<const intx4 *>(wb + (size_t)row * KROW + col * 16);
    }
    // MFMA row r of channel tile nt (a lane ends up with rows 4g..4g+3, g = lane >> 4, of every tile it accumulates) is output
    // channel wc*64 + (nt>>1)*32 + (r>>2)*8 + (nt&1)*4 + (r&3): the lane's rows of tiles 2j and 2j+1 are then EIGHT consecutive
    // channels, i.e. one 16-byte store per pixel and tile pair instead of two 8-byte ones
    const int arow = wc * 64 + ((lane & 15) >> 2) * 8 + (lane & 3);
    const int wrow = arow * WPAD + (lane >> 4) * 16;
    // bias of the lane's 16 channels: pair j covers channels wc*64 + j*32 + g*8 .. +7
    half8 bv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) bv[j] = *reinterpret_cast<const half8 *>(bias + wc * 64 + j * 32 + (lane >> 4) * 8);
    __syncthreads();

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        // the lane's 8 pixel columns (mt = 0..7): byte offset of input pixel (s, oy, ox); tap (dy, dx) adds (dy*W+dx)*64
        int q0[8];
#pragma unroll
        for (int mt = 0; mt < 8; mt++) {
            int p = tile * 256 + wr * 128 + mt * 16 + (lane & 15);
            p = p < M ? p : M - 1;
            const int s = p / HWO, r = p - s * HWO, oy = r / WO, ox = r - oy * WO;
            q0[mt] = ((s * H + oy) * W + ox) * XROW + (lane >> 4) * 16;
        }
        floatx4 acc[8][4];
#pragma unroll
        for (int mt = 0; mt < 8; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
        // pixel fragments: 8 of 16 B per tap, in two groups of 4 so that the second group's loads (and the next tap's first
        // group) are in flight under the MFMAs of the group before; the other wave of the SIMD covers the rest of the latency
        half8 pa[4], pc[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) pa[mt] = *reinterpret_cast<const half8 *>(xb + (unsigned)q0[mt]);
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int sh = ((t / 3) * W + t % 3) * XROW;
#pragma unroll
            for (int mt = 0; mt < 4; mt++) pc[mt] = *reinterpret_cast<const half8 *>(xb + (unsigned)(q0[4 + mt] + sh));
            half8 wf[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
                wf[nt] = *reinterpret_cast<const half8 *>(smem + wrow + ((nt >> 1) * 32 + (nt & 1) * 4) * WPAD + t * 64);
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], pa[mt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) {
                const int sn = (((t + 1) / 3) * W + (t + 1) % 3) * XROW;
#pragma unroll
                for (int mt = 0; mt < 4; mt++) pa[mt] = *reinterpret_cast<const half8 *>(xb + (unsigned)(q0[mt] + sn));
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
                    acc[4 + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], pc[mt], acc[4 + mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // keeps hipcc from hoisting several taps' loads (it spills the accumulators otherwise)
        }
        // ---- epilogue straight from the registers: bias + ReLU, 16 B (8 channels) per lane, pixel and tile pair; the four
        //      lane groups of a pixel write 64 contiguous bytes, the two pairs complete the 128-byte line
#pragma unroll
        for (int mt = 0; mt < 8; mt++) {
            const int p = tile * 256 + wr * 128 + mt * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float f = acc[mt][2 * j + (e >> 2)][e & 3] + (float)bv[j][e];
                    o[e] = (_Float16)(f > 0.f ? f : 0.f);
                }
                if (p < M) *reinterpret_cast<half8 *>(yb + (unsigned)(p * YROW + (wc * 64 + j * 32 + (lane >> 4) * 8) * 2)) = o;
            }
        }
    }
}

// x: [n][h][w][32] fp16, wgt: [256][3][3][32] fp16, bias fp16[256], y: [n][h-2][w-2][256] fp16.
static inline int launch(int n, int h, int w, const void *x, const void *wgt, const void *bias, void *y, hipStream_t st) {
    if (n <= 0 || h < 3 || w < 3) return -1;
    const long M = (long)n * (h - 2) * (w - 2);
    if (M * YROW >= (1L << 31) || (long)n * h * w * XROW >= (1L << 31)) return -1;
    const int tiles = (int)((M + 255) / 256);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    // one workgroup per CU (its 148 KiB of LDS hold the weights for all the tiles it walks)
    hipLaunchKernelGGL(k_stem, dim3(tiles < n_cu ? tiles : n_cu), dim3(512), 0, st, (const char *)x, (const char *)wgt,
                       (const _Float16 *)bias, (char *)y, (int)M, h, w, tiles);
    return 0;
}

}  // namespace sgo_stem
