// sgo_conv.hip -- the residual tower's 3x3 convolution with the bias / skip / ReLU epilogue fused into the GEMM's
// output stage, for the resident policy/value net (model.py:37-46 of the reference: Conv2D -> BatchNorm (folded) ->
// [Add] -> ReLU).
//
// Two back ends behind sgo_conv3x3_bias_act_dev:
//   * sgo_conv8w.hpp -- the hand-written CDNA4 kernel for the tower shape (256 -> 256 channels, 'same' padding, board
//     width <= 19): sgo_conv3x3_tower_dev.  This is what the 20-block tower runs on.
//   * the generic fall-through for every other shape (the 32 -> 256 'valid' stem, other channel counts): an
//     implicit-GEMM main loop from AMD's composable_kernel xdlops (MFMA) grouped-convolution template,
// instantiated here with a tile found by sweeping on gfx950 (tools/ckexp: 256 threads, 128 pixels x 256 channels,
// 32x32 MFMA, 2x4 tiles per wave, K step 64 -- the library's own instances all use a K step of 32: 3.17 ms vs
// 3.54 ms per 8192x256x17x17 convolution for the best of those), and with OUR epilogue functors, so that the separate
// bias/skip/ReLU pass over the 1.2 GB activation tensor disappears.  NHWC fp16 in, fp16 out, fp32 accumulate.
#include <array>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/convolution_forward_specialization.hpp"
#include "ck/tensor_operation/gpu/device/gemm_specialization.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"

#include "sgo_common.hpp"
#include "sgo_conv8w.hpp"

namespace {

using F16 = ck::half_t;
using F32 = float;
template <ck::index_t... Is>
using S = ck::Sequence<Is...>;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
namespace lay = ck::tensor_layout::convolution;

// e = relu(conv + bias[k])
struct BiasRelu {
    template <typename E, typename C, typename D0>
    __host__ __device__ constexpr void operator()(E &e, const C &c, const D0 &bias) const {
        const float x = ck::type_convert<float>(c) + ck::type_convert<float>(bias);
        e = ck::type_convert<E>(x > 0.f ? x : 0.f);
    }
};
// e = relu(conv + bias[k] + skip)
struct BiasAddRelu {
    template <typename E, typename C, typename D0, typename D1>
    __host__ __device__ constexpr void operator()(E &e, const C &c, const D0 &bias, const D1 &skip) const {
        const float x = ck::type_convert<float>(c) + ck::type_convert<float>(bias) + ck::type_convert<float>(skip);
        e = ck::type_convert<E>(x > 0.f ? x : 0.f);
    }
};

template <typename DsLayout, typename DsTypes, typename Op>
using Conv = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<
    2, lay::NHWGC, lay::GKYXC, DsLayout, lay::NHWGK, F16, F16, F32, F16, DsTypes, F16, PassThrough, PassThrough, Op,
    ck::tensor_operation::device::ConvolutionForwardSpecialization::Default,
    ck::tensor_operation::device::GemmSpecialization::MNKPadding, 1, 256, 128, 256, 64, 8, 8, 32, 32, 2, 4, S<8, 32, 1>,
    S<1, 0, 2>, S<1, 0, 2>, 2, 8, 8, 1, S<8, 32, 1>, S<1, 0, 2>, S<1, 0, 2>, 2, 8, 8, 1, 1, 1, S<1, 32, 1, 8>, 8>;

using ConvBias = Conv<ck::Tuple<lay::G_K>, ck::Tuple<F16>, BiasRelu>;
using ConvBiasSkip = Conv<ck::Tuple<lay::G_K, lay::NHWGK>, ck::Tuple<F16, F16>, BiasAddRelu>;

using Arr5 = std::array<ck::index_t, 5>;

}  // namespace

namespace {
int g_conv_backend = 0;   // 0: hand-written kernel where the shape fits, 1: generic path only
long g_tower_slice_cap = 0;   // > 0: samples per launch of the tower kernel are capped (tests of the slice loop)
}

extern "C" long sgo_conv_tower_slice_cap(long cap) {
    const long old = g_tower_slice_cap;
    if (cap >= 0) g_tower_slice_cap = cap;
    return old;
}

extern "C" int sgo_conv_backend(int mode) {
    const int old = g_conv_backend;
    if (mode == 0 || mode == 1) g_conv_backend = mode;
    return old;
}

extern "C" int sgo_conv_tile_order(int mode) {
    const int old = sgo_conv8w::tile_order();
    if (mode == 0 || mode == 1) sgo_conv8w::tile_order() = mode;
    return old;
}

extern "C" int sgo_conv3x3_tower_dev(int n, int h, int w, const void *d_x, const void *d_w, const void *d_bias,
                                     const void *d_skip, void *d_y, void *stream) {
    using namespace sgo;
    if (n <= 0 || h <= 0 || w <= 0 || w > sgo_conv8w::MAXW || !d_x || !d_w || !d_bias || !d_y) {
        set_error("sgo_conv3x3_tower_dev: bad argument (256 -> 256 channels, pad 1, board width <= 19)");
        return SGO_ERR_ARG;
    }
    if ((((uintptr_t)d_x | (uintptr_t)d_w | (uintptr_t)d_y | (uintptr_t)d_skip) & 15) || ((uintptr_t)d_bias & 7)) {
        set_error("sgo_conv3x3_tower_dev: x, w, skip, y must be 16-byte aligned (bias 8-byte): the kernel moves 16 B per lane");
        return SGO_ERR_ARG;
    }
    // the kernel addresses pixels with 32-bit byte offsets: batches beyond 2^31 bytes per tensor run in slices
    const long per = (long)h * w * sgo_conv8w::ROWB;
    long max_n = ((1L << 31) - 1) / per;
    // ... and splits pixel indices by magic-number division, exact while (M + 256) * (h*w) < 2^32
    const long max_n_div = (long)(((1ULL << 32) - 1) / (unsigned long long)(h * w) - 256) / ((long)h * w);
    if (max_n_div < max_n) max_n = max_n_div;
    if (g_tower_slice_cap > 0 && g_tower_slice_cap < max_n) max_n = g_tower_slice_cap;   // test hook: exercise the slice loop
    if (max_n > 256) max_n -= max_n % 256;
    if (max_n < 1) { set_error("sgo_conv3x3_tower_dev: one sample exceeds the addressable range"); return SGO_ERR_ARG; }
    for (long n0 = 0; n0 < n; n0 += max_n) {
        const int nn = (int)((n - n0 < max_n) ? (n - n0) : max_n);
        const char *s0 = d_skip ? (const char *)d_skip + n0 * per : nullptr;
        if (sgo_conv8w::launch(nn, h, w, (const char *)d_x + n0 * per, d_w, d_bias, s0, (char *)d_y + n0 * per,
                               (hipStream_t)stream) != 0) {
            set_error("sgo_conv3x3_tower_dev: launch rejected");
            return SGO_ERR_ARG;
        }
    }
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}

extern "C" int sgo_conv3x3_bias_act_dev(int n, int h, int w, int c, int k, int pad, const void *d_x, const void *d_w,
                                        const void *d_bias, const void *d_skip, void *d_y, void *stream) {
    using namespace sgo;
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || pad < 0 || pad > 1 || c % 8 || k % 8 || !d_x || !d_w || !d_bias || !d_y) {
        set_error("sgo_conv3x3_bias_act_dev: bad argument (channels must be multiples of 8)");
        return SGO_ERR_ARG;
    }
    if (g_conv_backend == 0 && c == sgo_conv8w::CIN && k == sgo_conv8w::COUT && pad == 1 && w <= sgo_conv8w::MAXW)
        return sgo_conv3x3_tower_dev(n, h, w, d_x, d_w, d_bias, d_skip, d_y, stream);
    const int ho = h + 2 * pad - 2, wo = w + 2 * pad - 2;
    if (ho <= 0 || wo <= 0) { set_error("sgo_conv3x3_bias_act_dev: empty output"); return SGO_ERR_ARG; }
    // the instance addresses tensors with 32-bit element offsets: run batches whose tensors exceed 2^31 bytes in slices
    const long per_in = (long)h * w * c * 2, per_out = (long)ho * wo * k * 2;
    long max_n = ((1L << 31) - 1) / (per_in > per_out ? per_in : per_out);
    if (max_n > 256) max_n -= max_n % 256;
    if (max_n < 1) { set_error("sgo_conv3x3_bias_act_dev: one sample exceeds the addressable range"); return SGO_ERR_ARG; }
    const std::array<ck::index_t, 2> ones{1, 1}, pads{pad, pad};
    const StreamConfig cfg{(hipStream_t)stream, false};
    for (long n0 = 0; n0 < n; n0 += max_n) {
        const int nn = (int)((n - n0 < max_n) ? (n - n0) : max_n);
        const char *x0 = (const char *)d_x + n0 * per_in;
        const char *s0 = d_skip ? (const char *)d_skip + n0 * per_out : nullptr;
        char *y0 = (char *)d_y + n0 * per_out;
        // [G, N, C, Hi, Wi] lengths with NHWGC strides (G = 1)
        const Arr5 a_len{1, nn, c, h, w}, a_str{c, h * w * c, 1, w * c, c};
        const Arr5 b_len{1, k, c, 3, 3}, b_str{k * 9 * c, 9 * c, 1, 3 * c, c};
        const Arr5 e_len{1, nn, k, ho, wo}, e_str{k, ho * wo * k, 1, wo * k, k};
        const Arr5 bias_str{k, 0, 1, 0, 0};
        if (s0) {
            ConvBiasSkip op;
            auto arg = op.MakeArgument(x0, d_w, std::array<const void *, 2>{d_bias, s0}, y0, a_len, a_str, b_len, b_str,
                                       std::array<Arr5, 2>{e_len, e_len}, std::array<Arr5, 2>{bias_str, e_str}, e_len, e_str, ones,
                                       ones, pads, pads, PassThrough{}, PassThrough{}, BiasAddRelu{});
            if (!op.IsSupportedArgument(arg)) { set_error("sgo_conv3x3_bias_act_dev: shape not supported by the instance"); return SGO_ERR_ARG; }
            op.MakeInvoker().Run(arg, cfg);
        } else {
            ConvBias op;
            auto arg = op.MakeArgument(x0, d_w, std::array<const void *, 1>{d_bias}, y0, a_len, a_str, b_len, b_str,
                                       std::array<Arr5, 1>{e_len}, std::array<Arr5, 1>{bias_str}, e_len, e_str, ones, ones, pads,
                                       pads, PassThrough{}, PassThrough{}, BiasRelu{});
            if (!op.IsSupportedArgument(arg)) { set_error("sgo_conv3x3_bias_act_dev: shape not supported by the instance"); return SGO_ERR_ARG; }
            op.MakeInvoker().Run(arg, cfg);
        }
    }
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}
