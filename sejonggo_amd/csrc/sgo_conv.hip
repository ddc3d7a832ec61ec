// sgo_conv.hip -- the 3x3 convolutions of the resident policy/value net with the bias / skip / ReLU epilogue fused
// into the kernel (model.py:37-46, :57-60 of the reference: Conv2D -> BatchNorm (folded) -> [Add] -> ReLU).  Both shapes
// the reference's topology needs are hand-written CDNA4 kernels:
//   * sgo_conv8w.hpp -- the residual tower (256 -> 256 channels, 'same' padding, board width <= 19): sgo_conv3x3_tower_dev,
//     40 of the 41 convolutions of a forward pass and 98 % of a self-play step's GPU time;
//   * sgo_stem.hpp   -- the stem (17 planes presented as 32 channels -> 256, 'valid'): sgo_conv3x3_stem_dev.
// sgo_conv3x3_bias_act_dev dispatches on the shape and reports SGO_ERR_UNSUPPORTED for anything else (other channel
// counts: the host then runs that layer through the framework's convolution + sgo_bias_act_dev, net.FusedInferenceNet).
// No library GEMM / convolution code is linked into libsgo_hip.so.
#include "sgo_common.hpp"
#include "sgo_conv8w.hpp"
#include "sgo_conv4w.hpp"
#include "sgo_conv4r.hpp"
#include "sgo_stem.hpp"
#include "sgo_stem_packed.hpp"

namespace {
int g_tower_kernel = 1;       // 1: k_conv4w (two 256-thread workgroups per CU; default, +2-3 %), 0: k_conv8w (one 512-thread workgroup per CU)
int g_packed_variant = 1;     // schedule variant of k_conv4r (A/B builds; 1 = product)
long g_tower_slice_cap = 0;   // > 0: samples per launch of the tower kernel are capped (tests of the slice loop)
}

extern "C" long sgo_conv_tower_slice_cap(long cap) {
    const long old = g_tower_slice_cap;
    if (cap >= 0) g_tower_slice_cap = cap;
    return old;
}

extern "C" int sgo_conv_tower_kernel(int mode) {
    const int old = g_tower_kernel;
    if (mode == 0 || mode == 1 || (mode >= 16 && mode < 16 + 4096)) g_tower_kernel = mode;   // 16 + v: k_conv4w schedule variant v (A/B builds)
    return old;
}

extern "C" int sgo_conv_tile_order(int mode) {
    const int old = sgo_conv8w::tile_order();
    if (mode == 0 || mode == 1) sgo_conv8w::tile_order() = mode;
    return old;
}

namespace {
// The slice loop both tower entry points share.  packed: d_w is a fragment-order filter bank (sgo_conv3x3_tower_prepack_dev) and
// the launch goes to k_conv4r; otherwise d_w is OHWI and the launch goes to k_conv4w / k_conv8w (sgo_conv_tower_kernel).
int tower_launch(const char *who, bool packed, int n, int h, int w, const void *d_x, const void *d_w, const void *d_bias,
                 const void *d_skip, void *d_y, void *stream) {
    using namespace sgo;
    if (n <= 0 || h <= 0 || w <= 0 || w > sgo_conv8w::MAXW || !d_x || !d_w || !d_bias || !d_y) {
        set_error(std::string(who) + ": bad argument (256 -> 256 channels, pad 1, board width <= 19)");
        return SGO_ERR_ARG;
    }
    if ((((uintptr_t)d_x | (uintptr_t)d_w | (uintptr_t)d_y | (uintptr_t)d_skip) & 15) || ((uintptr_t)d_bias & 7)) {
        set_error(std::string(who) + ": x, w, skip, y must be 16-byte aligned (bias 8-byte): the kernel moves 16 B per lane");
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
    if (max_n < 1) { set_error(std::string(who) + ": one sample exceeds the addressable range"); return SGO_ERR_ARG; }
    for (long n0 = 0; n0 < n; n0 += max_n) {
        const int nn = (int)((n - n0 < max_n) ? (n - n0) : max_n);
        const char *s0 = d_skip ? (const char *)d_skip + n0 * per : nullptr;
        const char *x0 = (const char *)d_x + n0 * per;
        char *y0 = (char *)d_y + n0 * per;
        int rc;
        if (packed) rc = sgo_conv4r::launch(nn, h, w, x0, d_w, d_bias, s0, y0, (hipStream_t)stream, g_packed_variant);
        else if (g_tower_kernel >= 1)
            rc = sgo_conv4w::launch(nn, h, w, x0, d_w, d_bias, s0, y0, (hipStream_t)stream, g_tower_kernel >= 16 ? g_tower_kernel - 16 : 7);
        else rc = sgo_conv8w::launch(nn, h, w, x0, d_w, d_bias, s0, y0, (hipStream_t)stream);
        if (rc != 0) {
            set_error(std::string(who) + ": launch rejected");
            return SGO_ERR_ARG;
        }
    }
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}
}  // namespace

extern "C" int sgo_conv3x3_tower_dev(int n, int h, int w, const void *d_x, const void *d_w, const void *d_bias,
                                     const void *d_skip, void *d_y, void *stream) {
    return tower_launch("sgo_conv3x3_tower_dev", false, n, h, w, d_x, d_w, d_bias, d_skip, d_y, stream);
}

extern "C" long sgo_conv3x3_tower_packed_bytes(void) { return sgo_conv4r::PACKED_BYTES; }

extern "C" int sgo_conv3x3_tower_prepack_dev(const void *d_w, void *d_wp, void *stream) {
    using namespace sgo;
    if (!d_w || !d_wp || (((uintptr_t)d_w | (uintptr_t)d_wp) & 15)) {
        set_error("sgo_conv3x3_tower_prepack_dev: w [256][3][3][256] fp16 and wp (sgo_conv3x3_tower_packed_bytes()) must be 16-byte aligned device pointers");
        return SGO_ERR_ARG;
    }
    sgo_conv4r::prepack(d_w, d_wp, (hipStream_t)stream);
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}

extern "C" int sgo_conv3x3_tower_packed_dev(int n, int h, int w, const void *d_x, const void *d_wp, const void *d_bias,
                                            const void *d_skip, void *d_y, void *stream) {
    return tower_launch("sgo_conv3x3_tower_packed_dev", true, n, h, w, d_x, d_wp, d_bias, d_skip, d_y, stream);
}

extern "C" int sgo_conv_packed_variant(int v) {
    const int old = g_packed_variant;
    if (v >= 0) g_packed_variant = v;
    return old;
}

extern "C" int sgo_conv3x3_stem_dev(int n, int h, int w, const void *d_x, const void *d_w, const void *d_bias, void *d_y,
                                    void *stream) {
    using namespace sgo;
    if (n <= 0 || h < 3 || w < 3 || !d_x || !d_w || !d_bias || !d_y) {
        set_error("sgo_conv3x3_stem_dev: bad argument (32 -> 256 channels, no padding, h, w >= 3)");
        return SGO_ERR_ARG;
    }
    if ((((uintptr_t)d_x | (uintptr_t)d_w | (uintptr_t)d_y) & 15) || ((uintptr_t)d_bias & 7)) {
        set_error("sgo_conv3x3_stem_dev: x, w, y must be 16-byte aligned (bias 8-byte)");
        return SGO_ERR_ARG;
    }
    // 32-bit byte offsets inside the kernel: larger batches run in slices
    const long per_in = (long)h * w * sgo_stem::XROW, per_out = (long)(h - 2) * (w - 2) * sgo_stem::YROW;
    long max_n = ((1L << 31) - 1) / (per_in > per_out ? per_in : per_out);
    if (g_tower_slice_cap > 0 && g_tower_slice_cap < max_n) max_n = g_tower_slice_cap;
    if (max_n < 1) { set_error("sgo_conv3x3_stem_dev: one sample exceeds the addressable range"); return SGO_ERR_ARG; }
    for (long n0 = 0; n0 < n; n0 += max_n) {
        const int nn = (int)((n - n0 < max_n) ? (n - n0) : max_n);
        if (sgo_stem::launch(nn, h, w, (const char *)d_x + n0 * per_in, d_w, d_bias, (char *)d_y + n0 * per_out,
                             (hipStream_t)stream) != 0) {
            set_error("sgo_conv3x3_stem_dev: launch rejected");
            return SGO_ERR_ARG;
        }
    }
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}

extern "C" int sgo_stem_packed_dev(int S, int n, const uint32_t *d_records, const int32_t *d_index, int sym_k, const int32_t *d_sym_k,
                                   const void *d_w10, const void *d_bias, const float *d_wcol, void *d_y, void *stream) {
    using namespace sgo;
    if (n <= 0 || !d_records || !d_w10 || !d_bias || !d_wcol || !d_y || sym_k < 0 || sym_k > 7) {
        set_error("sgo_stem_packed_dev: bad argument");
        return SGO_ERR_ARG;
    }
    if ((((uintptr_t)d_w10 | (uintptr_t)d_y | (uintptr_t)d_bias) & 15) || ((uintptr_t)d_records & 3) || ((uintptr_t)d_wcol & 3)) {
        set_error("sgo_stem_packed_dev: w10, bias, y must be 16-byte aligned");
        return SGO_ERR_ARG;
    }
    int rc = 0;
    SGO_DISPATCH(S, rc = sgo_stemp::launch<kS>(n, d_records, d_index, sym_k, d_sym_k, d_w10, d_bias, d_wcol, d_y, (hipStream_t)stream));
    if (rc != 0) { set_error("sgo_stem_packed_dev: launch rejected (batch too large for 32-bit pixel indices)"); return SGO_ERR_ARG; }
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}

extern "C" int sgo_conv3x3_bias_act_dev(int n, int h, int w, int c, int k, int pad, const void *d_x, const void *d_w,
                                        const void *d_bias, const void *d_skip, void *d_y, void *stream) {
    using namespace sgo;
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || pad < 0 || pad > 1 || !d_x || !d_w || !d_bias || !d_y) {
        set_error("sgo_conv3x3_bias_act_dev: bad argument");
        return SGO_ERR_ARG;
    }
    if (c == sgo_conv8w::CIN && k == sgo_conv8w::COUT && pad == 1 && w <= sgo_conv8w::MAXW)
        return sgo_conv3x3_tower_dev(n, h, w, d_x, d_w, d_bias, d_skip, d_y, stream);
    if (c == sgo_stem::CIN && k == sgo_stem::COUT && pad == 0 && !d_skip && h >= 3 && w >= 3)
        return sgo_conv3x3_stem_dev(n, h, w, d_x, d_w, d_bias, d_y, stream);
    set_error("sgo_conv3x3_bias_act_dev: no hand-written kernel for this shape (tower: 256 -> 256, pad 1, w <= 19; stem: 32 -> 256, pad 0)");
    return SGO_ERR_UNSUPPORTED;
}
