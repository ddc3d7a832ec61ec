#!/usr/bin/env python3
"""Tune MIOpen for the convolution shapes of the resident net and store the result in the in-tree user database
(sejonggo_amd/miopen_db), which sejonggo_amd/__init__.py points MIOPEN_USER_DB_PATH at.

Run on an MI355X box:   MIOPEN_FIND_ENFORCE=3 python tools/tune_miopen.py --out gpurun_out/miopen_db
then copy the *.udb.txt / *.ufdb.txt files into sejonggo_amd/miopen_db/ and commit them (they are a few hundred
bytes of text: solver ids + tuning parameters per problem key)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--configs", default="19:20:256:1024:8,9:4:256:256:8")   # size:blocks:channels:games:energy
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    os.environ["MIOPEN_USER_DB_PATH"] = os.path.abspath(a.out)
    os.environ.setdefault("MIOPEN_FIND_ENFORCE", "3")
    os.environ.setdefault("MIOPEN_FIND_MODE", "1")
    import torch
    from sejonggo_amd.net import build_fused_net
    for cfg in a.configs.split(","):
        S, nb, ch, G, E = (int(v) for v in cfg.split(":"))
        net, _ = build_fused_net(S, min(nb, 2), ch)   # two blocks hit every distinct conv shape
        for n in (G * E, G):
            x = torch.zeros((n, S, S, 32), dtype=torch.float16, device="cuda")
            t0 = time.time()
            net.predict_on_batch(x)
            torch.cuda.synchronize()
            print("tuned size=%d batch=%d in %.1f s" % (S, n, time.time() - t0), flush=True)
            t0 = time.perf_counter()
            for _ in range(5):
                net.predict_on_batch(x)
            torch.cuda.synchronize()
            print("   forward (2 blocks) %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
    print(os.listdir(a.out))


if __name__ == "__main__":
    main()
