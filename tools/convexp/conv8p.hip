// conv8p.hip -- experiment harness copy of the hand-written tower convolution (see sejonggo_amd/csrc/sgo_conv8w.hpp
// for the product version and the design notes).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared conv8p.hip -o libconv8p.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#ifdef USE_CONV8W
#include "../../sejonggo_amd/csrc/sgo_conv8w.hpp"
namespace sgo_conv8p {
#ifdef SGO_CONV8_STAMPS
static inline int launch(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y, const void *, hipStream_t st, long long *stamps) {
    return sgo_conv8w::launch(n, h, w, x, wgt, bias, skip, y, st, stamps);
}
#else
static inline int launch(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y, const void *, hipStream_t st) {
    return sgo_conv8w::launch(n, h, w, x, wgt, bias, skip, y, st);
}
#endif
}
#else
#include "conv8p_kernel.hpp"
#endif

extern "C" int conv8p_run(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y,
                          const void *zeros, int iters, float *ms_out) {
    hipStream_t st = 0;
#if defined(SGO_CONV8P_STAMPS) || defined(SGO_CONV8_STAMPS)
    return -9;
#else
    int rc = sgo_conv8p::launch(n, h, w, x, wgt, bias, skip, y, zeros, st);
    if (rc) return rc;
    if (hipStreamSynchronize(st) != hipSuccess) return -2;
    if (iters > 0) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, st);
        for (int i = 0; i < iters; i++) sgo_conv8p::launch(n, h, w, x, wgt, bias, skip, y, zeros, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / iters;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
#endif
}

#if defined(SGO_CONV8P_STAMPS) || defined(SGO_CONV8_STAMPS)
extern "C" int conv8p_stamps(int n, int h, int w, const void *x, const void *wgt, const void *bias, const void *skip, void *y,
                             const void *zeros, long long *stamps, int warm) {
    for (int i = 0; i < warm; i++) sgo_conv8p::launch(n, h, w, x, wgt, bias, skip, y, zeros, 0, stamps);
    sgo_conv8p::launch(n, h, w, x, wgt, bias, skip, y, zeros, 0, stamps);
    return hipStreamSynchronize(0) == hipSuccess ? 0 : -2;
}
#endif
