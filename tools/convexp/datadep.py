#!/usr/bin/env python3
"""Data dependence of the tower convolution's speed (DVFS): the same launch on random, ReLU-sparse, small-integer and zero
operands.  Cycles per MFMA do not depend on the data; the clock the chip holds does."""
import ctypes as C, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libconv8p_w.so"))
P = C.c_void_p
n, h, wd = 8192, 17, 17
torch.manual_seed(0)
w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
b = torch.randn(256, device="cuda").half()
z = torch.zeros(64, device="cuda", dtype=torch.float16)
y = torch.empty(n, h, wd, 256, device="cuda", dtype=torch.float16)
cases = {
    "random N(0, 0.5)": ((torch.randn(n, h, wd, 256, device="cuda") * 0.5).half(), w),
    "ReLU(random) (what the tower sees)": (torch.relu(torch.randn(n, h, wd, 256, device="cuda") * 0.5).half(), w),
    "small integers": (torch.randint(-2, 3, (n, h, wd, 256), device="cuda").half(), torch.randint(-1, 2, (256, 3, 3, 256), device="cuda").half()),
    "zeros": (torch.zeros(n, h, wd, 256, device="cuda", dtype=torch.float16), torch.zeros_like(w)),
}
fl = 2.0 * n * h * wd * 9 * 256 * 256
for rnd in range(2):
    for name, (x, ww) in cases.items():
        ms = C.c_float(0)
        rc = lib.conv8p_run(n, h, wd, P(x.data_ptr()), P(ww.data_ptr()), P(b.data_ptr()), P(x.data_ptr()), P(y.data_ptr()), P(z.data_ptr()), 20, C.byref(ms))
        assert rc == 0
        print("round %d  %-36s %.3f ms  %.0f TFLOP/s" % (rnd, name, ms.value, fl / ms.value / 1e9), flush=True)
