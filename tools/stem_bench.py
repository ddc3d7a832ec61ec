#!/usr/bin/env python3
"""Times the hand-written stem kernel (sgo_conv3x3_stem_dev) against the framework's convolution on the engine's batch."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from sejonggo_amd import _lib as L


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    lib = L.require_gpu()
    torch.backends.cudnn.benchmark = True
    x = torch.zeros(n, 19, 19, 32, device="cuda", dtype=torch.float16)
    x[..., :16] = (torch.rand(n, 19, 19, 16, device="cuda") < 0.3).half()
    x[..., 16] = 1.0
    w = (torch.randn(256, 3, 3, 32, device="cuda") * 0.1).half()
    b = torch.randn(256, device="cuda").half()
    y = torch.empty(n, 17, 17, 256, device="cuda", dtype=torch.float16)
    st = L.stream_ptr()
    xn, wn = x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2)

    def ours():
        L.check(lib.sgo_conv3x3_stem_dev(n, 19, 19, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), st))

    def torchconv():
        return F.conv2d(xn, wn, None)

    for name, fn in (("sgo_conv3x3_stem_dev", ours), ("torch conv2d (no bias / relu)", torchconv)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-32s %.3f ms  (%.0f GB/s of output, %.0f TFLOP/s)" % (name, ms, n * 289 * 512 / ms / 1e6, 2.0 * n * 289 * 288 * 256 / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
