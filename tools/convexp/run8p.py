#!/usr/bin/env python3
"""Correctness + timing harness for the hand-written tower convolution (tools/convexp/conv8p.hip) on an MI355X.

  python tools/convexp/run8p.py [n ...]

For every batch size: compares against torch (fp32 convolution of the same fp16 inputs), checks run-to-run bitwise
determinism (a race screen), and times it interleaved with the composable_kernel-based sgo_conv3x3_bias_act_dev.
"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
import glob
LIBS = {os.path.basename(f)[3:-3]: C.CDLL(f) for f in sorted(glob.glob(os.path.join(here, "libconv8p*.so"))) if "stamps" not in f}
MAIN = os.environ.get("CONV8_MAIN", "conv8p")
lib = LIBS[MAIN]
P = C.c_void_p


def run8p(x, w, b, s, y, zeros, h, wd, iters=0, lib=None):
    ms = C.c_float(0)
    rc = (lib or LIBS[MAIN]).conv8p_run(x.shape[0], h, wd, P(x.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(s.data_ptr()) if s is not None else None,
                        P(y.data_ptr()), P(zeros.data_ptr()), iters, C.byref(ms))
    assert rc == 0, rc
    return ms.value


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [3, 64, 1024, 8192]
    torch.manual_seed(0)
    c = k = 256
    zeros = torch.zeros(64, device="cuda", dtype=torch.float16)
    try:
        from sejonggo_amd import _lib
        ck = _lib.load()
    except Exception as e:  # noqa: BLE001
        print("ck lib unavailable:", e)
        ck = None
    for n in sizes:
        for (h, wd) in ((17, 17), (7, 7)) if n <= 64 else ((17, 17),):
            x = (torch.randn(n, h, wd, c, device="cuda") * 0.5).half()
            w = (torch.randn(k, 3, 3, c, device="cuda") * 0.03).half()
            b = torch.randn(k, device="cuda").half()
            s = torch.randn(n, h, wd, k, device="cuda").half()
            y = torch.full((n, h, wd, k), 7.0, device="cuda", dtype=torch.float16)
            for skip in (s, None):
                run8p(x, w, b, skip, y, zeros, h, wd)
                torch.cuda.synchronize()
                # reference on a slice (fp32 math of the same fp16 values)
                m = min(n, 96)
                idx = torch.cat([torch.arange(0, m // 2), torch.arange(n - (m - m // 2), n)]).unique().cuda()
                xr = x[idx].float().permute(0, 3, 1, 2)
                wr = w.float().permute(0, 3, 1, 2)
                ref = F.conv2d(xr, wr, b.float(), padding=1).permute(0, 2, 3, 1)
                if skip is not None:
                    ref = ref + s[idx].float()
                ref = torch.relu(ref)
                got = y[idx].float()
                err = (got - ref).abs()
                tol = 2e-3 * ref.abs() + 2e-3
                bad = int((err > tol).sum())
                print("n=%d %dx%d skip=%s: max abs err %.4g, out of tolerance %d / %d" % (
                    n, h, wd, skip is not None, float(err.max()), bad, ref.numel()), flush=True)
                # race screen: repeated launches must be bit-identical
                y0 = y.clone()
                diff = 0
                for _ in range(5):
                    y.fill_(3.0)
                    run8p(x, w, b, skip, y, zeros, h, wd)
                    torch.cuda.synchronize()
                    diff += int((y != y0).sum())
                print("   determinism: %d differing elements over 5 reruns" % diff, flush=True)
            if n >= 1024:
                fl = 2.0 * n * h * wd * 9 * c * k
                for rnd in range(3):
                    line = "   round %d:" % rnd
                    for name, l in LIBS.items():
                        if l is not lib:
                            ms = run8p(x, w, b, s, y, zeros, h, wd, iters=20, lib=l)
                            line += " %s %.3f ms %.0f TF |" % (name, ms, fl / ms / 1e9)
                    ms = run8p(x, w, b, s, y, zeros, h, wd, iters=20)
                    line += " %s %.3f ms %.0f TFLOP/s" % (MAIN, ms, fl / ms / 1e9)
                    if ck is not None:
                        xs = x
                        st = torch.cuda.current_stream().cuda_stream
                        e0 = torch.cuda.Event(enable_timing=True)
                        e1 = torch.cuda.Event(enable_timing=True)
                        y2 = torch.empty_like(y)
                        ck.sgo_conv3x3_bias_act_dev(n, h, wd, c, k, 1, P(xs.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(s.data_ptr()),
                                                    P(y2.data_ptr()), P(st))
                        e0.record()
                        for _ in range(20):
                            ck.sgo_conv3x3_bias_act_dev(n, h, wd, c, k, 1, P(xs.data_ptr()), P(w.data_ptr()), P(b.data_ptr()),
                                                        P(s.data_ptr()), P(y2.data_ptr()), P(st))
                        e1.record()
                        torch.cuda.synchronize()
                        ms2 = e0.elapsed_time(e1) / 20
                        line += " | ck %.3f ms %.0f TFLOP/s | max |8p - ck| %.4g" % (ms2, fl / ms2 / 1e9, float((y.float() - y2.float()).abs().max()))
                    print(line, flush=True)


if __name__ == "__main__":
    main()
