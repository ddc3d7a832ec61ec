#!/usr/bin/env python3
"""One tower-convolution kernel, launched back to back: the subject of a rocprofv3 --pmc pass (tools/pmc_conv4r.sh).
usage: conv_one.py <4w|4r> [variant] [n=8192] [iters=8]      data: ReLU(random) activations + skip, what the tower sees."""
import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from sejonggo_amd import _lib as L


def main():
    kern = sys.argv[1]
    var = int(sys.argv[2]) if len(sys.argv) > 2 else (1 if kern == "4r" else 7)
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    lib = L.require_gpu()
    h = w = 17
    torch.manual_seed(0)
    x = torch.relu(torch.randn(n, h, w, 256, device="cuda", dtype=torch.float16) * 0.5)
    skip = torch.relu(torch.randn(n, h, w, 256, device="cuda", dtype=torch.float16) * 0.5)
    wt = (torch.randn(256, 3, 3, 256, device="cuda", dtype=torch.float16) * 0.03)
    b = torch.randn(256, device="cuda", dtype=torch.float16) * 0.1
    y = torch.empty_like(x)
    st = L.stream_ptr()
    wp = torch.empty(lib.sgo_conv3x3_tower_packed_bytes(), device="cuda", dtype=torch.uint8)
    L.check(lib.sgo_conv3x3_tower_prepack_dev(L.ptr(wt), L.ptr(wp), st))
    if kern == "4r":
        lib.sgo_conv_packed_variant(var)
    else:
        lib.sgo_conv_tower_kernel(16 + var if var != 7 else 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(iters + 2):
        if it == 2:
            e0.record()
        if kern == "4r":
            L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, w, L.ptr(x), L.ptr(wp), L.ptr(b), L.ptr(skip), L.ptr(y), st))
        else:
            L.check(lib.sgo_conv3x3_tower_dev(n, h, w, L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(skip), L.ptr(y), st))
    e1.record()
    torch.cuda.synchronize()
    print("%s var %d: %.4f ms per launch" % (kern, var, e0.elapsed_time(e1) / iters), flush=True)


if __name__ == "__main__":
    main()
