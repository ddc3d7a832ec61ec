"""Self-play on the N GPUs of one node with the finished games' (s, pi, z) tuples gathered to rank 0 over RCCL
(BASELINE.json north_star; SURVEY.md §8e: the reference ships game files between machines by scp, scpy.py:68-76).

    python -m sejonggo_amd.dist_selfplay --gpus 8                       # starts the 8 ranks itself
    torchrun --nproc-per-node 8 -m sejonggo_amd.dist_selfplay --gpus 8  # or as one rank of an external launcher

One process per GPU.  Game numbers shard statically (game g -> rank g mod N), every rank keeps conf['GAMES_PER_GPU'] games
resident on its own engine with its own replica of the best model (weights broadcast from rank 0, checksums compared), and
there is NO exchange during search.  Every `sync_every` engine steps all ranks meet in one variable-length gather
(distributed.TupleGather: counts all_gather + padded byte gather, pipelined on a side stream; 7 concurrent point-to-point transfers into rank 0 on the
xGMI mesh) carrying the tuples of the games that finished since the last meeting, plus one all_reduce that tells everybody
whether anyone still plays.  Rank 0 turns the tuples back into the reference's files
(SELF_PLAY_DIR/<model>/game_%05d/move_%03d/sample.h5, sgfsave.py:49-79) on its writer threads; the other ranks never touch
the self-play directory except to skip game numbers that already exist (the reference's resume rule, selfplay_worker.py:83-90).

conf overrides for the ranks travel in the SGO_CONF_JSON environment variable (a JSON object)."""
import argparse
import json
import os
import sys

import numpy as np


def _tuples_of(game_no, gd, rank, size):
    """A finished game as tuple records (distributed.tuple_dtype): packed state, prior vector, outcome z per move."""
    from .distributed import tuple_dtype
    from .sgfsave import value_target
    moves = gd['moves']
    t = np.zeros(len(moves), dtype=tuple_dtype(size))
    for i, mv in enumerate(moves):
        t[i]["rank"], t[i]["game"], t[i]["game_seq"], t[i]["move_n"] = rank, game_no, mv.get('game_seq', 0), mv['move_n']
        t[i]["action"], t[i]["player"], t[i]["value"] = mv['action'], mv['player'], mv['value']
        t[i]["z"] = value_target(gd['winner'], mv['player'], mv['move_n'])
        t[i]["state"] = mv['packed']
        t[i]["pi"] = mv['policy']
    return t


def _write_games(tuples, model_name, size, pool, pending):
    """rank 0: tuples -> sample files, one job per game on the writer threads."""
    from .engine import unpack_positions
    from .sgfsave import _make_move_dir, _write_sample_arrays
    from .conf import conf
    if tuples is None or len(tuples) == 0:
        return 0
    order = np.lexsort((tuples["move_n"], tuples["game"]))
    tuples = tuples[order]
    games, starts = np.unique(tuples["game"], return_index=True)
    bounds = list(starts) + [len(tuples)]

    def job(rows):
        boards = unpack_positions(rows["state"], size)
        g = int(rows["game"][0])
        for i in range(len(rows)):
            directory, g = _make_move_dir(conf['SELF_PLAY_DIR'], model_name, "game_%05d", g, int(rows["move_n"][i]))
            _write_sample_arrays(directory, boards[i:i + 1].astype(np.float32), rows["pi"][i].astype(np.float32),
                                 np.array(rows["z"][i], dtype=np.float32))
        return len(rows)

    for k in range(len(games)):
        pending.append(pool.submit(job, tuples[bounds[k]:bounds[k + 1]].copy()))
    return len(games)


def run_rank(backend="nccl", sync_every=4, max_steps=None):
    """One rank.  Returns (games played by this rank, positions written by rank 0 or None)."""
    import torch
    import torch.distributed as dist
    from concurrent.futures import ThreadPoolExecutor
    from .conf import conf
    from .distributed import TupleGather, broadcast_net, init_from_env, shard_games, tuple_dtype
    from .engine import SelfPlayEngine
    from .predicting_queue_worker import get_model, init_predicting_workers, put_name_request
    from .selfplay_worker import GameScheduler
    over = os.environ.get("SGO_CONF_JSON")
    if over:
        conf.update(json.loads(over))
    rank, world, local = init_from_env(backend)
    if local is None:
        raise SystemExit("dist_selfplay: no HIP device visible; the hot path has no CPU fallback")
    init_predicting_workers([local])
    net = get_model("BEST_SYM", local)
    model_name = put_name_request("BEST_SYM")
    info = broadcast_net(net)
    if not info["identical"]:
        raise SystemExit("dist_selfplay: weight replicas differ after the broadcast")
    S = conf['SIZE']
    mine = shard_games(conf['N_GAMES'], world, rank)
    G = max(1, min(conf['GAMES_PER_GPU'], len(mine)))
    sched = GameScheduler(conf['SELF_PLAY_DIR'], model_name, 0, conf['RESIGNATION_PERCENT'], conf['RESIGNATION_ALLOWED_ERROR'])
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=conf['MCTS_SIMULATIONS'], energy=conf['ENERGY'],
                         stop_exploration=conf['STOP_EXPLORATION'], komi=conf['KOMI'], self_play=True,
                         symmetry=conf.get('SYMMETRY_MODE', 'random1'), device=local, seed=1000 + rank, raise_on_error=False,
                         num_moves=conf.get('NUM_MOVES'))
    todo = list(mine)
    slot_game, slot_resign = {}, {}

    def fill(slots):
        start, res = [], []
        for s in slots:
            g = None
            while todo:
                cand = todo.pop(0)
                if not os.path.isdir(os.path.join(conf['SELF_PLAY_DIR'], model_name, "game_%05d" % cand)):   # resume rule
                    g = cand
                    break
            if g is None:
                continue
            r = sched.pick_resign()
            slot_game[s], slot_resign[s] = g, r
            start.append(s); res.append(r)
        if start:
            eng.start_games(start, resign=res)
        return len(start)

    pool = ThreadPoolExecutor(max_workers=max(1, int(conf.get('WRITER_THREADS', 2)))) if rank == 0 else None
    pending, outbox = [], []
    played = written = steps = 0
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    exchange = TupleGather(tuple_dtype(S), device=dev)     # three-stage pipeline: the host never waits for a collective it has just issued
    try:
        active = fill(range(G))
        idle = G - active
        while True:
            for _ in range(sync_every):
                if active == 0:
                    break
                st = eng.step()
                steps += 1
                if st.n_records >= G:
                    eng.drain()
                if st.n_done > idle or (st.error and st.error_game in slot_game):
                    eng.drain()
                    res = eng.results()
                    free = []
                    for s in list(slot_game):
                        if res[s]["done"] == 0:
                            continue
                        g = slot_game.pop(s)
                        r = slot_resign.pop(s)
                        free.append(s)
                        active -= 1
                        if res[s]["done"] < 0:
                            print("rank %d: slot %d (game %d) failed with engine error %d; game dropped" % (rank, s, g, res[s]["done"]),
                                  file=sys.stderr)
                            eng.records[s] = []
                            continue
                        gd = eng.game_data(s, res[s], model_name)
                        eng.records[s] = []
                        sched.finished(gd, r)
                        if gd['moves']:
                            outbox.append(_tuples_of(g, gd, rank, S))
                            played += 1
                    refilled = fill(free)
                    active += refilled
                    idle += len(free) - refilled
                if max_steps is not None and steps >= max_steps:
                    active = 0
            # the meeting: finished games to rank 0, and does anybody still play?
            batch = np.concatenate(outbox) if outbox else np.zeros(0, dtype=tuple_dtype(S))
            outbox = []
            for got in exchange.submit(batch):               # batches of EARLIER meetings whose gather has completed
                if rank == 0:
                    _write_games(got, model_name, S, pool, pending)
            flag = torch.tensor([1 if active > 0 else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) == 0:
                break
        for got in exchange.flush():
            if rank == 0:
                _write_games(got, model_name, S, pool, pending)
    finally:
        eng.close()
        if pool is not None:
            for f in pending:
                written += f.result()
            pool.shutdown(wait=True)
    dist.barrier()
    dist.destroy_process_group()
    return played, (written if rank == 0 else None)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--sync-every", type=int, default=4, help="engine steps between two gathers")
    ap.add_argument("--max-steps", type=int, default=None)
    a = ap.parse_args(argv)
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        import torch
        from .distributed import launch_ranks, free_port
        ndev = torch.cuda.device_count()            # enumerates only; the launcher never initialises a GPU
        if ndev < 1 or (a.backend == "nccl" and ndev < a.gpus):
            print("dist_selfplay: --gpus %d needs %d HIP devices, %d visible" % (a.gpus, a.gpus, ndev), file=sys.stderr)
            return 2
        if a.gpus == 1:
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        else:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
            return launch_ranks(["-m", "sejonggo_amd.dist_selfplay"] + list(sys.argv[1:] if argv is None else argv), a.gpus, env=env)
    played, written = run_rank(a.backend, a.sync_every, a.max_steps)
    # one write, flushed: the ranks share the launcher's stdout
    sys.stdout.write("rank %s: %d games played%s\n" % (os.environ.get("RANK"), played, "" if written is None else ", %d positions written" % written))
    sys.stdout.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
