#!/usr/bin/env python3
"""End-to-end determinism of the self-play path on the production net form: the same seeds must give the same games bit for
bit (moves, root values, policy targets), twice in one process.  Every kernel on the path is deterministic by design
(no atomics in the tree, fixed accumulation order in the convolution), so any difference is a race."""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=19)
    ap.add_argument("--games", type=int, default=64)
    ap.add_argument("--sims", type=int, default=80)
    ap.add_argument("--moves", type=int, default=40)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--packed-tower", type=int, default=0, help="1: a third run with the tower on k_conv4r (net.use_packed_tower): same games bit for bit")
    a = ap.parse_args()
    import numpy as np
    import random
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.net import build_fused_net
    net, _ = build_fused_net(a.size, a.blocks, 256, name="det", seed=3, device="cuda")
    digests = []
    for rep in range(3 if a.packed_tower else 2):
        if rep == 2:
            assert net.use_packed_tower(True)
        random.seed(11)                      # the per-batch symmetry choice
        eng = SelfPlayEngine(net, size=a.size, n_games=a.games, sims=a.sims, energy=8, stop_exploration=30, num_moves=a.moves,
                             symmetry="random1", seed=5)
        eng.start_games(np.arange(a.games))
        games = eng.run()
        h = hashlib.sha1()
        n_moves = 0
        for g in games:
            for mv in g["moves"]:
                h.update(np.asarray(mv["move"], np.int32).tobytes())
                h.update(np.asarray(mv["value"], np.float32).tobytes())
                h.update(np.ascontiguousarray(mv["policy"]).tobytes())
                n_moves += 1
        digests.append(h.hexdigest())
        print("run %d: %d games, %d moves, digest %s" % (rep, len(games), n_moves, digests[-1]), flush=True)
        eng.close()
    assert digests[0] == digests[1], "self-play is not reproducible"
    if a.packed_tower:
        assert digests[2] == digests[0], "the packed tower route (k_conv4r) plays different games"
        print("packed tower route: identical games")
    print("deterministic")


if __name__ == "__main__":
    main()
