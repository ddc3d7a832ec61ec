#!/usr/bin/env python3
"""Where a workgroup of the tower convolution spends its cycles (diagnostic build with s_memtime stamps)."""
import ctypes as C, os, sys
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libconv8p_stamps.so"))
P = C.c_void_p
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
h = wd = 17
x = (torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
b = torch.randn(256, device="cuda").half()
s = torch.randn(n, h, wd, 256, device="cuda").half()
y = torch.empty_like(s)
z = torch.zeros(64, device="cuda", dtype=torch.float16)
tiles = (n * h * wd + 255) // 256
st = torch.zeros(tiles * 8 * 6, device="cuda", dtype=torch.int64)
rc = lib.conv8p_stamps(n, h, wd, P(x.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(s.data_ptr()), P(y.data_ptr()), P(z.data_ptr()),
                       P(st.data_ptr()), 30)
assert rc == 0
a = st.cpu().numpy().reshape(tiles, 8, 6)
pro, main, epi = a[:, :, 1] - a[:, :, 0], a[:, :, 2] - a[:, :, 1], a[:, :, 3] - a[:, :, 2]
tot = a[:, :, 3] - a[:, :, 0]
for name, v in (("prologue", pro), ("main loop", main), ("epilogue", epi), ("total", tot)):
    print("%-10s cycles: median %8.0f  p10 %8.0f  p90 %8.0f   (wave 0: %8.0f, wave 7: %8.0f)" % (
        name, np.median(v), np.percentile(v, 10), np.percentile(v, 90), np.median(v[:, 0]), np.median(v[:, 7])))
print("ideal main loop at 16 cyc/MFMA, 2 waves/SIMD: %d cycles" % (36 * 64 * 16 * 2))
clk = (a[:, :, 3] - a[:, :, 0]) / np.maximum(a[:, :, 4] - a[:, :, 5], 1) * 100.0   # s_memtime cycles per 100 MHz realtime tick
print("in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz): median %.0f MHz, p10 %.0f, p90 %.0f" % (
    np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
