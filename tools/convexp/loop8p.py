#!/usr/bin/env python3
"""Launch the tower convolution a few times (target of rocprofv3 runs)."""
import ctypes as C, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "lib%s.so" % os.environ.get("CONV8_MAIN", "conv8p_w")))
P = C.c_void_p
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
h = wd = 17
torch.manual_seed(0)
x = (torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
b = torch.randn(256, device="cuda").half()
s = torch.randn(n, h, wd, 256, device="cuda").half()
y = torch.empty_like(s)
z = torch.zeros(64, device="cuda", dtype=torch.float16)
ms = C.c_float(0)
rc = lib.conv8p_run(n, h, wd, P(x.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(s.data_ptr()), P(y.data_ptr()), P(z.data_ptr()), 10, C.byref(ms))
print("rc", rc, "ms", ms.value)
