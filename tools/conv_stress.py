#!/usr/bin/env python3
"""Stress screen of the tower convolution kernels (libsgo_hip.so) on an MI355X: random shapes (h, w in 1..19, ragged n, with and
without skip), k_conv4w against k_conv8w BIT FOR BIT (same per-wave MFMA tile and K order), every shape launched several times
(run-to-run determinism: a missed wait shows as a changing result), and a float32 torch reference on a subset.
usage: conv_stress.py [shapes=150] [seed=0]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from sejonggo_amd import _lib as L


def main():
    shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(seed)
    st = L.stream_ptr()
    wt = (torch.randn(256, 3, 3, 256, device="cuda", dtype=torch.float16) * 0.03)
    b = torch.randn(256, device="cuda", dtype=torch.float16) * 0.1
    bad = 0
    for i in range(shapes):
        h = int(torch.randint(1, 20, (1,), generator=g))
        w = int(torch.randint(1, 20, (1,), generator=g))
        n = int(torch.randint(1, max(2, 600000 // (h * w) // 4), (1,), generator=g))
        use_skip = bool(torch.randint(0, 2, (1,), generator=g))
        x = torch.relu(torch.randn(n, h, w, 256, device="cuda", dtype=torch.float16) * 0.5)
        skip = torch.relu(torch.randn(n, h, w, 256, device="cuda", dtype=torch.float16) * 0.5) if use_skip else None
        outs = []
        for kern in (0, 1, 1, 0, 1):
            lib.sgo_conv_tower_kernel(kern)
            y = torch.full_like(x, float("nan"))
            L.check(lib.sgo_conv3x3_tower_dev(n, h, w, L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(skip) if use_skip else None, L.ptr(y), st))
            torch.cuda.synchronize()
            outs.append(y)
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        ok_ref = True
        if i % 10 == 0:
            ref = F.conv2d(x.permute(0, 3, 1, 2).float(), wt.permute(0, 3, 1, 2).float(), b.float(), padding=1)
            if use_skip:
                ref = ref + skip.permute(0, 3, 1, 2).float()
            ref = torch.relu(ref).permute(0, 2, 3, 1)
            err = (outs[0].float() - ref).abs().max().item()
            ok_ref = err <= 2e-2 * max(1.0, ref.abs().max().item())
        if not (same and ok_ref):
            bad += 1
            print("MISMATCH shape n=%d h=%d w=%d skip=%s same=%s ref_ok=%s" % (n, h, w, use_skip, same, ok_ref), flush=True)
        if i % 25 == 0:
            print("shape %d/%d: n=%d h=%d w=%d skip=%s ok" % (i, shapes, n, h, w, use_skip), flush=True)
    lib.sgo_conv_tower_kernel(1)
    print("conv_stress: %d shapes, %d bad" % (shapes, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
