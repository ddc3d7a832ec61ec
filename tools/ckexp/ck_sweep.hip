// Experiment: time composable_kernel grouped-conv tile variants (one per -DVARIANT=n build) with the fused
// bias+skip+ReLU epilogue on the tower shape.  Built and run by tools/ckexp/run.py; not part of the product library.
#include <array>
#include <cstdio>
#include <hip/hip_runtime.h>
#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/convolution_forward_specialization.hpp"
#include "ck/tensor_operation/gpu/device/gemm_specialization.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle_v3.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"

using F16 = ck::half_t;
using F32 = float;
template <ck::index_t... Is> using S = ck::Sequence<Is...>;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
namespace lay = ck::tensor_layout::convolution;
using namespace ck::tensor_operation::device;
using ck::BlockGemmPipelineScheduler;
using ck::BlockGemmPipelineVersion;

struct BiasAddRelu {
    template <typename E, typename C, typename D0, typename D1>
    __host__ __device__ constexpr void operator()(E &e, const C &c, const D0 &bias, const D1 &skip) const {
        const float x = ck::type_convert<float>(c) + ck::type_convert<float>(bias) + ck::type_convert<float>(skip);
        e = ck::type_convert<E>(x > 0.f ? x : 0.f);
    }
};
using DsL = ck::Tuple<lay::G_K, lay::NHWGK>;
using DsT = ck::Tuple<F16, F16>;
#define COMMON 2, lay::NHWGC, lay::GKYXC, DsL, lay::NHWGK, F16, F16, F32, F16, DsT, F16, PassThrough, PassThrough, BiasAddRelu, ConvolutionForwardSpecialization::Default, GemmSpecialization::MNKPadding

#define TAIL S<1, 0, 2>, S<1, 0, 2>, 2, 8, 8, 1
#if VARIANT == 0
using VX = DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<COMMON, 1, 256, 256, 128, 64, 8, 8, 32, 32, 4, 2, S<8, 32, 1>, TAIL, S<8, 32, 1>, TAIL, 1, 1, S<1, 32, 1, 8>, 8>;
#define VNAME "256thr 256x128 K64"
#elif VARIANT == 1
using VX = DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<COMMON, 1, 256, 128, 256, 64, 8, 8, 32, 32, 2, 4, S<8, 32, 1>, TAIL, S<8, 32, 1>, TAIL, 1, 1, S<1, 32, 1, 8>, 8>;
#define VNAME "256thr 128x256 K64 (product)"
#elif VARIANT == 2
using VX = DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<COMMON, 1, 256, 128, 128, 64, 8, 8, 32, 32, 2, 2, S<8, 32, 1>, TAIL, S<8, 32, 1>, TAIL, 1, 1, S<1, 32, 1, 8>, 8>;
#define VNAME "256thr 128x128 K64"
#elif VARIANT == 3
using VX = DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<COMMON, 1, 256, 256, 128, 128, 8, 8, 32, 32, 4, 2, S<16, 16, 1>, TAIL, S<16, 16, 1>, TAIL, 1, 1, S<1, 32, 1, 8>, 8>;
#define VNAME "256thr 256x128 K128"
#elif VARIANT == 4
using VX = DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<COMMON, 1, 128, 128, 128, 64, 8, 8, 32, 32, 4, 2, S<8, 16, 1>, TAIL, S<8, 16, 1>, TAIL, 1, 1, S<1, 16, 1, 8>, 8>;
#define VNAME "128thr 128x128 K64"
#elif VARIANT == 5
using VX = DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<COMMON, 1, 256, 256, 128, 96, 8, 8, 32, 32, 4, 2, S<4, 64, 1>, TAIL, S<4, 64, 1>, TAIL, 1, 1, S<1, 32, 1, 8>, 8>;
#define VNAME "256thr 256x128 K96"
#elif VARIANT == 6
using VX = DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle<COMMON, 1, 256, 256, 64, 64, 8, 8, 32, 32, 2, 2, S<8, 32, 1>, TAIL, S<8, 32, 1>, TAIL, 1, 1, S<1, 32, 1, 8>, 8>;
#define VNAME "256thr 256x64 K64"
#endif
using Arr5 = std::array<ck::index_t, 5>;

template <typename Op>
static float run_one(const char *name, int n, int h, int c, int k, const void *x, const void *w, const void *b, const void *skip,
                     void *y, int iters) {
    const Arr5 a_len{1, n, c, h, h}, a_str{c, h * h * c, 1, h * c, c};
    const Arr5 b_len{1, k, c, 3, 3}, b_str{k * 9 * c, 9 * c, 1, 3 * c, c};
    const Arr5 e_len{1, n, k, h, h}, e_str{k, h * h * k, 1, h * k, k};
    const Arr5 bias_str{k, 0, 1, 0, 0};
    const std::array<ck::index_t, 2> ones{1, 1}, pads{1, 1};
    Op op;
    auto arg = op.MakeArgument(x, w, std::array<const void *, 2>{b, skip}, y, a_len, a_str, b_len, b_str,
                               std::array<Arr5, 2>{e_len, e_len}, std::array<Arr5, 2>{bias_str, e_str}, e_len, e_str, ones, ones,
                               pads, pads, PassThrough{}, PassThrough{}, BiasAddRelu{});
    if (!op.IsSupportedArgument(arg)) { printf("%s: not supported\n", name); return -1; }
    auto inv = op.MakeInvoker();
    StreamConfig cfg{nullptr, false};
    for (int i = 0; i < 3; i++) inv.Run(arg, cfg);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; i++) inv.Run(arg, cfg);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    printf("%s: %.3f ms  %.0f TFLOP/s\n", name, ms, 2.0 * n * h * h * 9.0 * c * k / (ms * 1e-3) / 1e12);
    fflush(stdout);
    return ms;
}

extern "C" int ck_sweep(int n, int h, int c, int k, const void *x, const void *w, const void *b, const void *skip, void *y, int iters) {
    run_one<VX>(VNAME, n, h, c, k, x, w, b, skip, y, iters);
    return 0;
}
