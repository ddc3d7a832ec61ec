#!/usr/bin/env python3
"""Full-length self-play stress: plays whole games (not just the first plies the bench times) through the worker
body, restarting slots, and reports game lengths, results, tree-pool high-water behaviour (capacity errors are loud)."""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=9)
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--games", type=int, default=64)
    ap.add_argument("--resident", type=int, default=32)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--stop", type=int, default=30)
    a = ap.parse_args()
    import numpy as np
    from sejonggo_amd import predicting_queue_worker as pq
    from sejonggo_amd.conf import conf
    from sejonggo_amd.net import build_fused_net
    from sejonggo_amd.selfplay_worker import run_selfplay
    d = tempfile.mkdtemp(prefix="sgo_stress_")
    conf.update({'SIZE': a.size, 'MCTS_SIMULATIONS': a.sims, 'ENERGY': 8, 'STOP_EXPLORATION': a.stop, 'SELF_PLAY_DIR': d,
                 'GAMES_PER_GPU': a.resident})
    fnet, _ = build_fused_net(a.size, a.blocks, a.channels, name="stress")
    pq.set_model_factory(lambda kind: fnet)
    lens, results = [], []
    t0 = time.time()
    played = run_selfplay(0, "BEST_SYM", n_games=a.games, games_per_gpu=a.resident,
                          on_game=lambda g, gd: (lens.append(len(gd['moves'])), results.append(gd['result'] + " " + gd['end_reason'])))
    dt = time.time() - t0
    lens = np.array(lens)
    print("played %d games in %.1f s: %d positions, %.1f positions/s; length min/mean/max %d/%.1f/%d" % (
        played, dt, lens.sum(), lens.sum() / dt, lens.min(), lens.mean(), lens.max()))
    print("end reasons:", {k: sum(1 for r in results if r.endswith(k)) for k in ("BOTH_PASSED", "PLAYED ALL MOVES", "resign")})
    print("sample results:", results[:6])


if __name__ == "__main__":
    main()
