#!/usr/bin/env python3
"""Exploration: which PyTorch-ROCm convolution configuration runs the tower's 3x3/256->256 conv fastest."""
import sys
import time

import torch
import torch.nn.functional as F


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    H = 17
    fl = 2 * B * H * H * 9 * 256 * 256
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for dt in (torch.float16, torch.bfloat16):
            for cl in (True, False):
                x = torch.randn(B, 256, H, H, device="cuda", dtype=dt)
                w = torch.randn(256, 256, 3, 3, device="cuda", dtype=dt) * 0.02
                if cl:
                    x = x.to(memory_format=torch.channels_last)
                    w = w.to(memory_format=torch.channels_last)
                try:
                    s = t(lambda: F.conv2d(x, w, None, padding=1))
                    print("conv3x3 B=%d dt=%s channels_last=%s benchmark=%s: %.3f ms  %.0f TFLOP/s" % (
                        B, str(dt)[6:], cl, bench, s * 1e3, fl / s / 1e12), flush=True)
                except Exception as e:
                    print("failed", dt, cl, bench, repr(e)[:100], flush=True)
    torch.backends.cudnn.benchmark = False
    # stem: 17 vs 24 vs 32 input channels, valid padding, on 19x19
    for cin in (17, 24, 32):
        x = torch.randn(B, cin, 19, 19, device="cuda", dtype=torch.float16).to(memory_format=torch.channels_last)
        w = (torch.randn(256, cin, 3, 3, device="cuda", dtype=torch.float16) * 0.02).to(memory_format=torch.channels_last)
        s = t(lambda: F.conv2d(x, w, None))
        print("stem cin=%d: %.3f ms" % (cin, s * 1e3), flush=True)
    # heads: conv1x1 256->2 vs linear on the NHWC view ; 256->4 merged
    x = torch.randn(B, 256, H, H, device="cuda", dtype=torch.float16).to(memory_format=torch.channels_last)
    w = torch.randn(2, 256, 1, 1, device="cuda", dtype=torch.float16)
    s = t(lambda: F.conv2d(x, w, None))
    print("head conv1x1 256->2: %.3f ms" % (s * 1e3), flush=True)
    xl = x.permute(0, 2, 3, 1).reshape(-1, 256)
    wl = torch.randn(4, 256, device="cuda", dtype=torch.float16)
    s = t(lambda: F.linear(xl, wl))
    print("head linear 256->4: %.3f ms" % (s * 1e3), flush=True)
    wl8 = torch.randn(8, 256, device="cuda", dtype=torch.float16)
    s = t(lambda: F.linear(xl, wl8))
    print("head linear 256->8: %.3f ms" % (s * 1e3), flush=True)
    # elementwise passes
    y = torch.randn_like(x)
    b = torch.randn(256, device="cuda", dtype=torch.float16)
    s = t(lambda: torch.relu_(x + y))
    print("add+relu (2 kernels): %.3f ms" % (s * 1e3), flush=True)
    s = t(lambda: x.add_(b.view(1, -1, 1, 1)))
    print("bias add in place: %.3f ms" % (s * 1e3), flush=True)
    # GEMM reference point: hipBLASLt on an im2col-shaped problem M=B*289, K=2304, N=256
    M = B * H * H
    a = torch.randn(M, 2304, device="cuda", dtype=torch.float16)
    wg = torch.randn(2304, 256, device="cuda", dtype=torch.float16)
    s = t(lambda: a @ wg)
    print("gemm M=%d K=2304 N=256: %.3f ms %.0f TFLOP/s" % (M, s * 1e3, 2 * M * 2304 * 256 / s / 1e12), flush=True)


if __name__ == "__main__":
    main()
