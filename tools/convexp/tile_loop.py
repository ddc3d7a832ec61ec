#!/usr/bin/env python3
"""Bare LDS-fed MFMA loops at the tower's matrix work: what the per-wave register tile is worth (tile_loop.hip).
usage: tile_loop.py [iters=20]     arms interleaved; prints ms and TFLOP/s per mode."""
import ctypes as C
import os
import sys
import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libtile_loop.so"))
lib.tile_loop_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
# 61 x 16 KB of fragments: the first 8 KB of each block activations (post-ReLU), the rest weights
bank = torch.empty(61, 16, 512, dtype=torch.float16, device="cuda")
bank[:, :8] = torch.relu(torch.randn(61, 8, 512, device="cuda") * 0.5).half()
bank[:, 8:] = (torch.randn(61, 8, 512, device="cuda") * 0.03).half()
out = torch.zeros(512 * 256, dtype=torch.float32, device="cuda")
steps = 2600                      # x 32 (or 64) MFMAs per wave and step: 2.79 TFLOP per launch in every mode
flops = {0: 512 * 4 * steps * 32 * 16384.0, 2: 512 * 4 * steps * 32 * 16384.0, 1: 256 * 4 * steps * 64 * 16384.0}
names = {0: "128 x 64 per wave, 2 waves / SIMD, reads then burst (the tower kernels' shape)",
         2: "128 x 64 per wave, 2 waves / SIMD, next step's reads between the MFMAs",
         1: "128 x 128 per wave, 1 wave / SIMD, next step's reads between the MFMAs"}
tot = {0: 0.0, 1: 0.0, 2: 0.0}
ms = C.c_float(0)
for rep in range(4):
    for mode in (0, 2, 1):
        rc = lib.tile_loop_run(mode, steps, iters, bank.data_ptr(), out.data_ptr(), C.byref(ms))
        assert rc == 0, rc
        if rep:
            tot[mode] += ms.value
for mode in (0, 2, 1):
    t = tot[mode] / 3
    print("mode %d  %-82s %.4f ms  %5.0f TFLOP/s  (%.3f of 2.5 PFLOP/s)" % (mode, names[mode], t, flops[mode] / t / 1e9, flops[mode] / t / 1e9 / 2500), flush=True)
print("checksum", float(out.sum()))
