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
    ap.add_argument("--blocks-per-game", type=int, default=0, help="tree blocks per game (0 = the engine's default)")
    ap.add_argument("--shared-blocks", type=int, default=0, help="tree blocks shared by all games (0 = the engine's default rule)")
    ap.add_argument("--policy-gain", type=float, default=1.0,
                    help="multiply the policy head's last layer: > 1 makes a random-init net's priors peaky, like a trained net's "
                         "(concentrated search -> the kept subtree holds most of the tree -> the block pool is stressed)")
    ap.add_argument("--max-plies", type=int, default=0, help="cap on a game's length (0 = the rules' 2 * S * S)")
    ap.add_argument("--json", default="", help="write the summary as JSON to this path")
    ap.add_argument("--energy", type=int, default=8)
    ap.add_argument("--no-resign", type=int, default=0, help="1: RESIGNATION_PERCENT = 1 (every game runs to its natural end)")
    ap.add_argument("--packed-tower", type=int, default=0, help="1: the tower on k_conv4r (net.use_packed_tower)")
    a = ap.parse_args()
    import numpy as np
    from sejonggo_amd import predicting_queue_worker as pq
    from sejonggo_amd.conf import conf
    from sejonggo_amd.net import build_fused_net
    from sejonggo_amd.selfplay_worker import run_selfplay
    d = tempfile.mkdtemp(prefix="sgo_stress_")
    conf.update({'SIZE': a.size, 'MCTS_SIMULATIONS': a.sims, 'ENERGY': a.energy, 'STOP_EXPLORATION': a.stop, 'SELF_PLAY_DIR': d,
                 'GAMES_PER_GPU': a.resident})
    if a.no_resign:
        conf['RESIGNATION_PERCENT'] = 1.0
    if a.policy_gain != 1.0:
        import torch
        from sejonggo_amd.net import FusedInferenceNet, PolicyValueNet
        torch.manual_seed(0)
        plain = PolicyValueNet(a.size, a.blocks, a.channels, name="stress")
        for mod in plain.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
        plain.eval()
        with torch.no_grad():
            plain.p_fc.weight.mul_(a.policy_gain)
        fnet = FusedInferenceNet(plain, torch.float16, "cuda")
        fnet.name = "stress"
    else:
        fnet, _ = build_fused_net(a.size, a.blocks, a.channels, name="stress")
    if a.packed_tower:
        assert fnet.use_packed_tower(True), "the packed tower route needs the reference's 256-channel shape"
    pq.set_model_factory(lambda kind: fnet)
    lens, results = [], []
    probe = {}
    t0 = time.time()
    stats = {}
    last = [t0]

    def on_game(g, gd):
        lens.append(len(gd['moves']))
        probe.setdefault("hw", []).append(gd.get('blocks_high_water', 0))
        results.append(gd['result'] + " " + gd['end_reason'])
        if time.time() - last[0] > 30:          # a progress line at least every 30 s (gpurun takes a silent run for hung)
            last[0] = time.time()
            print("  %d games done after %.0f s" % (len(lens), time.time() - t0), flush=True)

    import threading
    stop = threading.Event()

    def heartbeat():
        while not stop.wait(45):
            print("  ... %.0f s, %d games done" % (time.time() - t0, len(lens)), flush=True)

    # diagnostic library (-DSGO_KSEARCH_PROFILE): cycles per phase of k_search, read every few seconds through the engine's context
    import ctypes as C
    from sejonggo_amd import engine as eng_mod
    real_engine = eng_mod.SelfPlayEngine

    class Probed(real_engine):
        def __init__(self, *aa, **kk):
            real_engine.__init__(self, *aa, **kk)
            probe["eng"] = self
            probe["cap"] = int(self.lib.sgo_blocks_per_game(self.ctx))
            probe["pool0"] = self.pool_info()

        def close(self):
            try:
                probe["pool1"] = self.pool_info()
            except Exception:
                pass
            real_engine.close(self)
            probe.setdefault("hw", [])

        def step(self):
            st = real_engine.step(self)
            n = probe["n"] = probe.get("n", 0) + 1
            if n % 2550 == 0:                    # every 50 moves
                out = (C.c_ulonglong * 8)()
                self.lib.sgo_debug_counters(self.ctx, out, 8)
                cur = [int(v) for v in out]
                prev = probe.get("prev", [0] * 8)
                d = [a - b for a, b in zip(cur, prev)]
                probe["prev"] = cur
                if any(d):
                    tot = float(sum(d[i] for i in (0, 2, 3, 7))) or 1.0
                    print("  k_search cycles, moves %d-%d: consume %.0f%%, select %.0f%%, round back-propagation %.0f%%, move step + exit %.0f%%; "
                          "%.0f cycles per wave-call" % (n // 51 - 50, n // 51, 100 * d[0] / tot, 100 * d[2] / tot, 100 * d[7] / tot,
                                                        100 * d[3] / tot, tot / max(d[4], 1)), flush=True)
            return st

    eng_mod.SelfPlayEngine = Probed
    import sejonggo_amd.selfplay_worker as sw_mod

    hb = threading.Thread(target=heartbeat, daemon=True)
    hb.start()
    played = run_selfplay(0, "BEST_SYM", n_games=a.games, games_per_gpu=a.resident, on_game=on_game, stats=stats,
                          engine_kwargs=dict({'blocks_per_game': a.blocks_per_game, 'shared_blocks': a.shared_blocks}, **({'num_moves': a.max_plies} if a.max_plies else {})))
    stop.set()
    dt = time.time() - t0
    print("host seconds: stepping %.1f, turnover %.1f, waiting for writers at the end %.1f" % (
        stats.get("step", 0), stats.get("turnover", 0), stats.get("writer_wait", 0)))
    print("engine steps %d; games started but not written (failed slots, and zero-move resignations when resigning is on): %d of %d" % (
        stats.get("steps", -1), a.games - played if a.games >= played else 0, a.games))
    lens = np.array(lens)
    print("played %d games in %.1f s: %d positions, %.1f positions/s; length min/mean/max %d/%.1f/%d" % (
        played, dt, lens.sum(), lens.sum() / dt, lens.min(), lens.mean(), lens.max()))
    allhw = np.array(probe.get("hw", [0]) or [0])
    print("private tree blocks per game: %d; high-water mark of finished games: max %d, p99 %d, median %d" % (
        probe.get("cap", -1), allhw.max(), int(np.percentile(allhw, 99)), int(np.median(allhw))))
    print("pool at the start:", probe.get("pool0"), "at the end:", probe.get("pool1"))
    if a.json:
        import json
        json.dump({"config": vars(a), "games_played": int(played), "games_started": a.games, "discarded": int(max(0, a.games - played)),
                   "seconds": dt, "positions": int(lens.sum()), "positions_per_sec": float(lens.sum() / dt),
                   "game_length": {"min": int(lens.min()), "mean": float(lens.mean()), "max": int(lens.max())},
                   "blocks_per_game": probe.get("cap"), "pool": probe.get("pool0"), "pool_at_end": probe.get("pool1"), "pool_high_water": {"max": int(allhw.max()), "p99": int(np.percentile(allhw, 99)),
                                                                           "median": int(np.median(allhw))},
                   "host_seconds": {k: stats.get(k) for k in ("step", "turnover", "writer_wait")}, "engine_steps": stats.get("steps"),
                   "end_reasons": {k: sum(1 for r in results if r.endswith(k)) for k in ("BOTH_PASSED", "PLAYED ALL MOVES", "resign")}},
                  open(a.json, "w"), indent=1)
    print("end reasons:", {k: sum(1 for r in results if r.endswith(k)) for k in ("BOTH_PASSED", "PLAYED ALL MOVES", "resign")})
    print("sample results:", results[:6])


if __name__ == "__main__":
    main()
