#!/usr/bin/env python3
"""Saturated micro-benchmark of the board_advance kernel (sgo_advance_legal_dev) on synthetic inputs as
SURVEY.md §8d prescribes: positions sampled from seeded random legal playouts at plies {0, 30, 120, 250}
(generated on the GPU with the same kernel), moves uniform over the reference-legal set.
Prints one JSON line per ply: leaves/s, achieved algorithmic GB/s (1834 B per leaf at 19x19) and the
fraction of the 8 TB/s HBM peak."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=19)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--plies", default="0,30,120,250")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import torch
    from sejonggo_amd import _lib as L
    lib = L.require_gpu()
    S, n = args.size, args.n
    A, NW, RW = S * S + 1, lib.sgo_plane_words(S), lib.sgo_packed_words(S)
    algo = {19: 1834, 9: 434}.get(S, 2 * (16 * NW * 4 + 8) + A)
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    cur = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    nxt = torch.zeros_like(cur)
    legal = torch.full((n, NW), -1, dtype=torch.int32, device="cuda")
    legal[:, NW - 1] = (1 << ((A - 1) % 32 + 1)) - 1
    shifts = torch.arange(32, device="cuda", dtype=torch.int32)
    st = L.stream_ptr()
    want = sorted(int(p) for p in args.plies.split(","))
    ply = 0
    chunk = 1 << 16

    def sample_moves():
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        for o in range(0, n, chunk):
            lg = legal[o:o + chunk]
            bits = ((lg.unsqueeze(-1) >> shifts) & 1).reshape(lg.shape[0], NW * 32)[:, :A].float()
            bits[:, A - 1] = 0.01
            out[o:o + chunk] = torch.multinomial(bits, 1, generator=g).reshape(-1).to(torch.int32)
        return out

    for target in want:
        while ply < target:
            mv = sample_moves()
            L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(mv), None, L.ptr(nxt), None, L.ptr(legal), None, st))
            cur, nxt = nxt, cur
            ply += 1
        mv = sample_moves()
        occ = float(((cur[:4096, :2 * NW].unsqueeze(-1) >> shifts) & 1).sum()) / (4096 * S * S)
        for _ in range(3):
            L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(mv), None, L.ptr(nxt), None, L.ptr(legal), None, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(mv), None, L.ptr(nxt), None, L.ptr(legal), None, st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        gbs = algo * n / (ms * 1e-3) / 1e9
        print(json.dumps({"kernel": "board_advance split (k_history_shift + k_advance_planes), out of place", "size": S,
                          "n": n, "ply": target, "occupancy": round(occ, 3), "ms": ms, "leaves_per_s": n / (ms * 1e-3),
                          "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / 8000.0}), flush=True)
        # fused per-lane kernel (the in-place-safe form): run it in place on a scratch copy
        tmp = cur.clone()
        e0.record()
        for _ in range(args.iters):
            tmp.copy_(cur)
        e1.record()
        torch.cuda.synchronize()
        ms_copy = e0.elapsed_time(e1) / args.iters
        e0.record()
        for _ in range(args.iters):
            tmp.copy_(cur)
            L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(tmp), None, L.ptr(mv), None, L.ptr(tmp), None, L.ptr(legal), None, st))
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / args.iters - ms_copy
        print(json.dumps({"kernel": "k_advance_legal fused, in place", "size": S, "n": n, "ply": target, "ms": ms2,
                          "algorithmic_GBps": algo * n / (ms2 * 1e-3) / 1e9, "frac_of_8TBps": algo * n / (ms2 * 1e-3) / 1e9 / 8000.0}),
              flush=True)


if __name__ == "__main__":
    main()
