#!/usr/bin/env python3
"""A/B of runtime knobs of the tower convolution (libsgo_hip.so) in ONE process, arms interleaved launch by launch so
that clock / temperature drift hits both alike (MI355X_MICROARCH.md 'DVFS give-back': never compare separate runs).
usage: conv_ab.py [n=8192] [iters=40]      data: ReLU(random) activations + skip, what the tower sees."""
import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from sejonggo_amd import _lib as L


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    lib = L.require_gpu()
    h = w = 17
    torch.manual_seed(0)
    x = torch.relu(torch.randn(n, h, w, 256, device="cuda", dtype=torch.float16) * 0.5)
    skip = torch.relu(torch.randn(n, h, w, 256, device="cuda", dtype=torch.float16) * 0.5)
    wt = (torch.randn(256, 3, 3, 256, device="cuda", dtype=torch.float16) * 0.03)
    b = torch.randn(256, device="cuda", dtype=torch.float16) * 0.1
    y = torch.empty_like(x)
    st = L.stream_ptr()
    fl = 2.0 * n * h * w * 9 * 256 * 256

    wp = torch.empty(lib.sgo_conv3x3_tower_packed_bytes(), device="cuda", dtype=torch.uint8)
    L.check(lib.sgo_conv3x3_tower_prepack_dev(L.ptr(wt), L.ptr(wp), st))
    packed = [False]

    def run():
        if packed[0]:
            L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, w, L.ptr(x), L.ptr(wp), L.ptr(b), L.ptr(skip), L.ptr(y), st))
        else:
            L.check(lib.sgo_conv3x3_tower_dev(n, h, w, L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(skip), L.ptr(y), st))

    side = torch.cuda.Stream()
    half = (n // 2) * h * w * 256 * 2            # bytes of the first half of x / skip / y

    def run_mix(k0, k1):
        """first half of the batch on the current stream with kernel k0, second half on a side stream with k1 ('4w' / '4r'): the
        two kernels' workgroups share the CUs"""
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        for (k, s_, off, nn) in ((k0, cur, 0, n // 2), (k1, side, half, n - n // 2)):
            with torch.cuda.stream(s_):
                sp = s_.cuda_stream
                if k == "4r":
                    L.check(lib.sgo_conv3x3_tower_packed_dev(nn, h, w, x.data_ptr() + off, L.ptr(wp), L.ptr(b), skip.data_ptr() + off, y.data_ptr() + off, sp))
                else:
                    L.check(lib.sgo_conv3x3_tower_dev(nn, h, w, x.data_ptr() + off, L.ptr(wt), L.ptr(b), skip.data_ptr() + off, y.data_ptr() + off, sp))
        cur.wait_stream(side)

    mix = [None]
    _run1 = run

    def run():
        if mix[0]:
            run_mix(*mix[0])
        else:
            _run1()

    def select(mode):
        mix[0] = None
        if isinstance(mode, tuple):
            lib.sgo_conv_tower_kernel(1)
            lib.sgo_conv_packed_variant(1)
            mix[0] = mode
            return
        packed[0] = mode < 0
        if mode < 0:                              # k_conv4r (register-fed weights), schedule variant -mode - 1
            lib.sgo_conv_packed_variant(-mode - 1)
        elif mode < 2:
            lib.sgo_conv_tile_order(mode & 1)
            lib.sgo_conv_tower_kernel(0)
        else:
            lib.sgo_conv_tower_kernel(mode)

    arms = [("k_conv8w, identity tile order", 0), ("k_conv8w, XCD-contiguous tiles", 1), ("k_conv4w (2 workgroups / CU)", 16 + 7),
            ("k_conv4r (weights L2 -> registers)", -2)]
    if os.environ.get("SGO_AB_MIX"):               # two half-batches on two streams: same kernel twice, and one of each
        arms += [("two streams: 4w + 4w", ("4w", "4w")), ("two streams: 4r + 4r", ("4r", "4r")), ("two streams: 4w + 4r", ("4w", "4r"))]
    if os.environ.get("SGO_AB_PACKED"):            # k_conv4r schedule variants (library built with -DSGO_CONV4W_VARIANTS)
        arms += [("k_conv4r var %d" % int(v), -int(v) - 1) for v in os.environ["SGO_AB_PACKED"].split(",") if int(v) != 1]
    if os.environ.get("SGO_AB_VARIANTS"):          # library built with -DSGO_CONV4W_VARIANTS
        arms += [("k_conv4w var %d" % v, 16 + v) for v in [int(a) for a in os.environ["SGO_AB_VARIANTS"].split(",") if a.isdigit() and int(a) != 1] or (0, 4, 5, 6)]
    outs = {}
    for name, mode in arms:
        select(mode)
        run()
        torch.cuda.synchronize()
        outs[name] = y.clone()
    # ablation arms (variant bits >= 16: no MFMAs / no fragment reads / no weight staging / no barriers) time a kernel whose
    # results are wrong by construction
    ablation = lambda mode: not isinstance(mode, tuple) and ((mode >= 16 and ((mode - 16) & (16 | 32 | 64 | 128 | 2048)) != 0) or (mode < 0 and ((-mode - 1) & (4 | 32 | 64 | 128)) != 0))
    assert all(torch.equal(outs[arms[0][0]], o) for (name, mode), o in zip(arms, outs.values()) if not ablation(mode)), "an arm changed the result"
    for _ in range(10):
        run()
    ms = {name: 0.0 for name, _ in arms}
    evs = []
    for it in range(iters):
        for name, mode in arms:
            select(mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            evs.append((name, e0, e1))
    torch.cuda.synchronize()
    for name, e0, e1 in evs:
        ms[name] += e0.elapsed_time(e1)
    for name, _ in arms:
        t = ms[name] / iters
        print("%-34s %.4f ms  %.0f TFLOP/s" % (name, t, fl / t / 1e9), flush=True)
    lib.sgo_conv_tile_order(1)
    lib.sgo_conv_tower_kernel(1)
    lib.sgo_conv_packed_variant(1)


if __name__ == "__main__":
    main()
