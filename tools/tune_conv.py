#!/usr/bin/env python3
"""Experiment: does MIOpen's exhaustive search find a faster solver for the tower convolution?"""
import os, sys, time
import torch
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.backends.cudnn.benchmark = True
x = torch.randn(B, 256, 17, 17, device="cuda", dtype=torch.float16).to(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, device="cuda", dtype=torch.float16) * 0.02).to(memory_format=torch.channels_last)
t0 = time.time()
y = F.conv2d(x, w, None, padding=1)
torch.cuda.synchronize()
print("first call (find/tune) %.1f s" % (time.time() - t0), flush=True)
for _ in range(3):
    F.conv2d(x, w, None, padding=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    F.conv2d(x, w, None, padding=1)
torch.cuda.synchronize()
s = (time.perf_counter() - t0) / 10
print("B=%d tuned conv: %.3f ms %.0f TFLOP/s" % (B, s * 1e3, 2 * B * 289 * 9 * 256 * 256 / s / 1e12), flush=True)
