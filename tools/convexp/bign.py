#!/usr/bin/env python3
"""Full-tensor check of the tower convolution at batch sizes near the 2^31-byte slicing limit."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from sejonggo_amd import _lib as L
lib = L.load()
L.require_gpu()
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for n in [int(a) for a in sys.argv[1:]] or [9000, 12000, 14336, 15000]:
    h = 17
    x = (torch.randn(n, h, h, 256, device="cuda") * 0.5).half()
    w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
    b = torch.randn(256, device="cuda").half()
    s = torch.randn(n, h, h, 256, device="cuda").half()
    y = torch.empty_like(s)
    L.check(lib.sgo_conv3x3_tower_dev(n, h, h, x.data_ptr(), w.data_ptr(), b.data_ptr(), s.data_ptr(), y.data_ptr(), st))
    torch.cuda.synchronize()
    bad_imgs = []
    worst = 0.0
    wr = w.float().permute(0, 3, 1, 2)
    for o in range(0, n, 1024):
        ref = torch.relu(F.conv2d(x[o:o + 1024].float().permute(0, 3, 1, 2), wr, b.float(), padding=1).permute(0, 2, 3, 1) + s[o:o + 1024].float())
        err = (y[o:o + 1024].float() - ref).abs()
        worst = max(worst, float(err.max()))
        bad = (err > 2e-3 * ref.abs() + 2e-3)
        if bool(bad.any()):
            idx = bad.nonzero()
            imgs = torch.unique(idx[:, 0]) + o
            bad_imgs += imgs.tolist()
            if len(bad_imgs) < 40:
                i0 = idx[0].tolist()
                print("  first bad in chunk:", [i0[0] + o] + i0[1:], "n bad", int(bad.sum()), "pixels rows", torch.unique(idx[:, 1]).tolist()[:10], "cols", torch.unique(idx[:, 2]).tolist()[:10], "chans", torch.unique(idx[:, 3]).tolist()[:8])
    print("n=%d: worst abs err %.4g, bad images %d %s" % (n, worst, len(bad_imgs), bad_imgs[:12]), flush=True)
