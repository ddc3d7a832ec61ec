#!/usr/bin/env python3
"""Headline benchmark: self-play positions/sec (19x19, 400 sims/move) on N MI355X.

Contract (one JSON line on rank 0): see the task statement.  A "step" = every resident game advances
by ONE move: 1 root evaluation + sims/energy search rounds of `energy` leaves each, i.e. (1 + 400)
network evaluations, 400 board_advance leaf positions and one move record per game.
  value     = n_gpus * games_per_gpu * steps / seconds   (whole job, positions/sec)
  roofline  = the step's dominant kernel (98 % of the GPU time): the hand-written tower convolution (k_conv4w; k_conv8w with --tower-kernel 0),
              bound "mfma": FLOPs of its launches / their durations, every launch of the timed region bracketed by
              HIP events on the launch stream (both batch sizes; "largest_batch" = the games x energy batches alone)
  roofline_board_advance = the board_advance kernel in situ, bound "hbm": algorithmic bytes per launch (1834 B per
              leaf at 19x19) / its average duration, HIP events on the launch stream inside sgo_step;
              roofline_saturated = the same kernel family on a chip-filling dense batch
  cpu_baseline = the oracle (C restatement of the reference's rules + tree) driving the same net on torch
              CPU fp32, on a bounded sample, host cores stated ("kind": "port")
Multi-GPU: one process per GPU, games sharded statically, no collective in the search; the weights are broadcast
from rank 0 once (RCCL) and the per-step move records are gathered to rank 0 over RCCL inside the timed region
(weak scaling).  Launch: `python bench.py --gpus N` from a bare shell starts the N ranks itself (fresh interpreters,
the launcher never touches a GPU; fewer than N visible devices is an error); under torch.distributed.run (RANK /
WORLD_SIZE in the environment) the process is one rank.  N = 1 runs the same code with a one-rank process group.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def _algo_bytes(size):
    """packed 2-bit x 8-history state (8*ceil(2N/8) + 8 B meta) read + written, plus the u8 legal mask."""
    n = size * size
    rec = 8 * ((2 * n + 7) // 8) + 8
    return 2 * rec + n + 1


ALGO_BYTES = {s: _algo_bytes(s) for s in (5, 7, 9, 13, 19)}   # 19 -> 1834, 9 -> 434


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=19)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--energy", type=int, default=8)
    ap.add_argument("--games", type=int, default=1024, help="concurrent games per GPU")
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--symmetry", default="random1", choices=["random1", "avg8", "identity"])
    ap.add_argument("--net", default="resnet", choices=["resnet", "uniform", "hash"])
    ap.add_argument("--split-streams", type=int, default=0, help="evaluate the net as two half batches on two streams")
    ap.add_argument("--tower-kernel", type=int, default=-1, help="0: k_conv8w, 1: k_conv4w, 2: k_conv4r (filter banks in fragment order, weights L2 -> registers), -1: the library's default")
    ap.add_argument("--tail-split", type=int, default=0, help="0: evaluate every batch as one launch chain (A/B of net.FusedInferenceNet._tail_split)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the N>1 path on fewer GPUs than ranks (ranks share devices, CPU tensors in the collectives)")
    ap.add_argument("--plain-net", type=int, default=0, help="1: plain PyTorch module instead of the fused inference net")
    ap.add_argument("--saturated", type=int, default=1, help="also time board_advance on a chip-filling dense batch")
    ap.add_argument("--steady-state", type=int, default=1, help="also measure run_selfplay-style operation with slot turnover")
    ap.add_argument("--steady-moves", type=int, default=5, help="num_moves cap of the steady-state games (short, so slots turn over)")
    ap.add_argument("--steady-generations", type=int, default=2, help="games per slot in the steady-state leg")
    ap.add_argument("--writer-processes", type=int, default=0, help="steady-state leg: conf['WRITER_PROCESSES'] (0 = writer threads)")
    ap.add_argument("--graph", type=int, default=0, help="1: every round is one captured launch chain (hipGraph): stem -> tower -> heads -> "
                    "k_search -> k_compact -> board_advance; no per-launch event timing in this mode")
    ap.add_argument("--halves", type=int, default=1, choices=[1, 2], help="2: the games run as two half-populations alternating on two "
                    "streams (engine.DualEngine; implies --graph 1)")
    ap.add_argument("--gather-stream", default="side", choices=["side", "step"],
                    help="side: the gather's copies and collectives run on a side stream beside the step's kernels; step: ordered into the "
                         "stepping stream (between two steps)")
    ap.add_argument("--gather-collective", default="gather", choices=["gather", "all_gather"])
    ap.add_argument("--gather-flush", type=int, default=0, help="1: complete every step's gather before the next step (blocking form, A/B)")
    ap.add_argument("--gather", type=int, default=1, help="0: skip the per-step tuple gather (A/B of its cost on small configurations)")
    ap.add_argument("--avg8-leg", type=int, default=1, help="headline configuration only: one warm-up + one step with 8-fold symmetry "
                    "averaging (BASELINE config 3 as written), reported as config3_avg8")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-games", type=int, default=32, help="concurrent games of the CPU baseline (the reference's N_GAME_PROCESS, conf.py:30)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-clock budget of the CPU baseline sample")
    ap.add_argument("--cpu-moves", type=int, default=1)
    return ap.parse_args()


def cpu_baseline(args, size, sims, energy, n_blocks, channels, symmetry):
    """Oracle (port) + the same net on torch CPU fp32; bounded sample; returns dict."""
    import numpy as np
    import torch
    from oracle import oracle as ora
    from sejonggo_amd.net import PolicyValueNet
    from sejonggo_amd.stub_nets import make_stub
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)   # the GPU box grants a 16-CPU share per GPU; more threads only oversubscribe it
    torch.set_num_threads(cores)
    if args.net == "resnet":
        torch.manual_seed(0)
        net = PolicyValueNet(size, n_blocks, channels, name="cpu").eval().fused(torch.float32)
        net = net.to(memory_format=torch.channels_last)
    else:
        net = make_stub(args.net, size)
    from concurrent.futures import ThreadPoolExecutor
    ng, nm = args.cpu_games, args.cpu_moves
    pool = ThreadPoolExecutor(max_workers=cores)      # the oracle's C calls release the GIL: games spread over the cores
    rng = np.random.RandomState(0)
    games = [ora.Game(size, sims, energy, 30, nm, uniforms=rng.random_sample(nm + 1),
                      noises=rng.dirichlet([0.03] * (size * size + 1), size=1)) for _ in range(ng)]
    ks = list(range(8)) if symmetry == "avg8" else [0]
    evals_per_position = (sims // energy) * energy + 1
    budget = args.cpu_seconds
    net.predict_on_batch(np.zeros((ng * energy, size, size, 17), np.float32))   # thread-pool / allocator warm-up, untimed
    t0 = time.time()
    c0 = os.times()
    evals = 0
    ticks = 0
    while any(g.phase != ora.PH_DONE for g in games) and time.time() - t0 < budget:
        live = [g for g in games if g.phase != ora.PH_DONE]
        pend = list(zip(live, pool.map(lambda g: g.pending().copy(), live)))
        boards = np.concatenate([b for _, b in pend])
        pol, val = None, None
        for k in ks:
            xb = ora.sym_board(k, boards) if k else boards
            p, v = net.predict_on_batch(xb.astype(np.float32))
            p = p.numpy() if torch.is_tensor(p) else p
            v = v.numpy() if torch.is_tensor(v) else v
            if k:
                p = ora.sym_policy_inverse(size, k, p)
            pol = p if pol is None else pol + p
            val = v if val is None else val + v
        pol = (pol / len(ks)).astype(np.float32)
        val = (val / len(ks)).astype(np.float32)
        evals += len(boards)
        ticks += 1
        o, jobs = 0, []
        for g, b in pend:
            jobs.append((g, pol[o:o + len(b)], val[o:o + len(b)]))
            o += len(b)
        list(pool.map(lambda j: j[0].submit(j[1], j[2]), jobs))
    dt = time.time() - t0
    pool.shutdown()
    c1 = os.times()
    busy = (c1.user - c0.user) + (c1.system - c0.system)
    used = max(1, int(round(busy / max(dt, 1e-9))))   # cores actually kept busy by this process during the sample
    # positions = completed moves + the completed fraction of the moves in flight (every position costs
    # evals_per_position network evaluations, which is where the CPU time goes)
    positions = evals / float(evals_per_position)
    return {"value": positions / dt, "unit": "positions/sec", "cores": used, "kind": "port",
            "sample": "the reference's topology: %d concurrent games x %d leaves in flight each (conf.py:30-31), first %d rounds "
                      "(%d net evals = %.2f positions' worth) in a %.0f s budget; oracle C rules+tree, games spread over %d host "
                      "threads, + the same net on torch CPU fp32 (torch threads %d; measured CPU time / wall = %.1f cores), %.1f s"
                      % (ng, energy, ticks, evals, positions, budget, cores, torch.get_num_threads(), busy / max(dt, 1e-9), dt),
            # the Python reference itself cannot travel to the GPU box; its own speed as measured in the build container
            # (8-vCPU Xeon 2.1 GHz, uniform stub net = NO network cost, one game process)
            "reference_python": {"sync_path_19x19_400sims": 0.55, "async_path_9x9_48sims_pool8": 17.5, "sync_path_9x9_48sims": 25.3,
                                 "unit": "positions/sec", "measured": "build container, not this box; network evaluation excluded (stub)",
                                 "source": "BASELINE.md section 2"}}


def steady_state(args, net, S, G, sims, E, device, resident_value):
    """The shipped worker body (selfplay_worker.run_selfplay) on the bench's own configuration, with games capped at
    --steady-moves plies so that every slot finishes and is restarted --steady-generations times inside the window:
    game-result read-back, record drain, batched restarts (one H2D copy per step), and the sample files of every
    position written by the writer threads (sample.h5 + the .npz twin, the reference's directory layout).  The clock
    runs from the first restart batch to the last file on disk."""
    import shutil
    import tempfile
    from sejonggo_amd import predicting_queue_worker as pq
    from sejonggo_amd.conf import conf
    from sejonggo_amd.selfplay_worker import run_selfplay
    keep = dict(conf)
    tmp = tempfile.mkdtemp(prefix="sgo_steady_")
    gens, cap = args.steady_generations, args.steady_moves
    stats = {}
    try:
        conf.update(SIZE=S, MCTS_SIMULATIONS=sims, ENERGY=E, GAMES_PER_GPU=G, N_GAMES=G * gens, SELF_PLAY_DIR=tmp,
                    STOP_EXPLORATION=30, SYMMETRY_MODE=args.symmetry,
                    WRITER_PROCESSES=args.writer_processes,
                    RESIGNATION_PERCENT=1.0)    # never resign: every game plays its `cap` plies (a random net resigns at move 0)
        pq.set_model_factory(lambda kind: net)
        played = run_selfplay(device, "BEST_SYM", n_games=G * gens, games_per_gpu=G, engine_kwargs={"num_moves": cap, "seed": 4321},
                              stats=stats)
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([device])
        conf.clear()
        conf.update(keep)
        shutil.rmtree(tmp, ignore_errors=True)
    value = stats["moves"] / stats["loop_s"]
    return {"value": value, "unit": "positions/sec", "vs_resident": value / resident_value if resident_value else None,
            "games": played, "positions": stats["moves"], "seconds": stats["loop_s"], "engine_steps": stats["steps"],
            "num_moves_cap": cap, "generations": gens, "sample_files": stats["files"],
            "host_seconds": {"stepping (engine + net, incl. per-step record drain)": stats["step"],
                             "turnover on the stepping thread (results, game_data, restart batch)": stats["turnover"],
                             "waiting for the writer threads after the last step": stats["writer_wait"]},
            "writer_threads": int(keep.get('WRITER_THREADS', 2)), "writer_processes": args.writer_processes,
            "note": "games start from the empty board and are capped at %d plies, so slots turn over every %d steps -- far more "
                    "often than full-length games would (~300 plies); the network cost per position is the same (%d evaluations)"
                    % (cap, cap, (sims // E) * E + 1)}


def avg8_leg(net, S, G, sims, E, device, steps=1, warmup=1):
    """BASELINE config 3 as written -- "+ 8-fold symmetry averaging" (a build extension: the reference applies ONE random
    symmetry per batch, symmetry.py:127-132, which is what the headline `value` runs): every evaluation list goes through the
    net under all 8 symmetries, the policies are inverse-permuted and averaged in float32.  Same games, sims and net."""
    import numpy as np
    import torch
    from sejonggo_amd.engine import SelfPlayEngine
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=30, symmetry="avg8", device=device, seed=4242)
    eng.start_games(np.arange(G))

    def one_step():
        target = eng.status.total_moves + G
        while eng.status.total_moves < target:
            if eng.step().n_active < G:
                raise RuntimeError("a game ended inside the avg8 leg")
        eng.drain()
        for s in range(G):
            eng.records[s] = []

    try:
        for _ in range(warmup):
            one_step()
        timed = hasattr(net, "conv_events")
        if timed:
            net.conv_events, net.side_flops, net.conv_event_stride = [], 0.0, 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = {"value": G * steps / dt, "unit": "positions/sec", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
               "symmetry": "avg8", "net_evals_per_position": 8 * ((sims // E) * E + 1)}
        if timed:
            evs, net.conv_events = net.conv_events, None
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
            fl = sum(f for _, _, f in evs)
            if ms > 0:
                out["roofline"] = {"bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                   "frac": fl / (ms * 1e-3) / 1e12 / 2500.0, "launches": len(evs), "share_of_step_time": ms * 1e-3 / dt}
        return out
    finally:
        eng.close()


def saturated_advance(S, n=1 << 18, ply=60, iters=10):
    """board_advance through the dense C-ABI entry point on a batch large enough to fill the chip (the in-situ
    launch holds only games*energy leaves).  Positions come from seeded random legal playouts on the GPU."""
    import torch
    from sejonggo_amd import _lib as L
    lib = L.require_gpu()
    A, NW, RW = S * S + 1, lib.sgo_plane_words(S), lib.sgo_packed_words(S)
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    cur = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    nxt = torch.zeros_like(cur)
    legal = torch.full((n, NW), -1, dtype=torch.int32, device="cuda")
    legal[:, NW - 1] = (1 << ((A - 1) % 32 + 1)) - 1
    shifts = torch.arange(32, device="cuda", dtype=torch.int32)
    st = L.stream_ptr()

    def moves():
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        for o in range(0, n, 1 << 16):
            lg = legal[o:o + (1 << 16)]
            bits = ((lg.unsqueeze(-1) >> shifts) & 1).reshape(lg.shape[0], NW * 32)[:, :A].float()
            bits[:, A - 1] = 0.01
            out[o:o + (1 << 16)] = torch.multinomial(bits, 1, generator=g).reshape(-1).to(torch.int32)
        return out

    for _ in range(ply):
        mv = moves()
        L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(mv), None, L.ptr(nxt), None, L.ptr(legal), None, st))
        cur, nxt = nxt, cur
    mv = moves()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(mv), None, L.ptr(nxt), None, L.ptr(legal), None, st))
    e0.record()
    for _ in range(iters):
        L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(mv), None, L.ptr(nxt), None, L.ptr(legal), None, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    ach = ALGO_BYTES.get(S, 0) * n / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "board_advance, dense batch through sgo_advance_legal_dev (k_history_shift + k_advance_planes)", "achieved": ach,
            "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "positions_per_launch": n, "avg_launch_ms": ms,
            "inputs": "seeded random legal playouts, ply %d" % ply}


def launcher(args):
    """Bare-shell entry: start the ranks.  Nothing here initialises a GPU (device_count() only enumerates)."""
    import torch
    from sejonggo_amd.distributed import launch_ranks, free_port
    n = args.gpus
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and ndev < n:
        print("bench.py: --gpus %d needs %d HIP devices, %d visible (use --backend gloo to rehearse the N-rank path on "
              "fewer devices)" % (n, n, ndev), file=sys.stderr)
        return 2
    if ndev < 1:
        print("bench.py: no HIP device visible; the hot path has no CPU fallback", file=sys.stderr)
        return 2
    if n == 1:
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        run_rank(args)
        return 0
    return launch_ranks([os.path.abspath(__file__)] + sys.argv[1:], n)


def main():
    args = parse()
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        sys.exit(launcher(args))
    if int(os.environ["WORLD_SIZE"]) != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
        sys.exit(2)
    run_rank(args)


def run_rank(args):
    # stdout carries the result line and nothing else: libraries that print there from C (RCCL's version banner at communicator
    # start) are sent to stderr by pointing fd 1 at fd 2 for the life of the rank; the line goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    from sejonggo_amd.distributed import init_from_env, broadcast_net
    rank, world, local = init_from_env(args.backend)
    if local is None:
        raise SystemExit("bench.py: no HIP device visible; the hot path has no CPU fallback")
    from sejonggo_amd.engine import SelfPlayEngine, DualEngine
    from sejonggo_amd.net import build_net, build_fused_net
    from sejonggo_amd.stub_nets import make_stub
    from sejonggo_amd.distributed import tuple_dtype, TupleGather, device_identities
    S, G, sims, E = args.size, args.games, args.sims, args.energy
    if args.tower_kernel in (0, 1):
        from sejonggo_amd import _lib as _L
        _L.load().sgo_conv_tower_kernel(args.tower_kernel)
    if args.net == "resnet":
        if args.plain_net:
            net = build_net(S, args.blocks, args.channels, name="bench_%db" % args.blocks, seed=0, device="cuda")
        else:
            net, _ = build_fused_net(S, args.blocks, args.channels, name="bench_%db" % args.blocks, seed=0, device="cuda")
            net.split_streams = bool(args.split_streams)
            net.tail_split = bool(args.tail_split)
            if args.tower_kernel == 2:
                net.use_packed_tower(True)
    else:
        net = make_stub(args.net, S)
    # replicas take their weights from rank 0 (one RCCL broadcast), checked identical by checksum
    bcast = broadcast_net(net) if args.net == "resnet" else None
    if bcast is not None and not bcast["identical"]:
        raise SystemExit("bench.py: weight replicas differ after the broadcast")
    if args.halves == 2:
        eng = DualEngine(net, n_games=G, size=S, sims=sims, energy=E, stop_exploration=30, symmetry=args.symmetry, device=local,
                         seed=1234 + rank)
    else:
        eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=30, symmetry=args.symmetry,
                             device=local, seed=1234 + rank, graph=bool(args.graph))
    eng.start_games(np.arange(G))
    tdt = tuple_dtype(S)
    exchange = TupleGather(tdt, side_stream=(args.gather_stream == "side"), collective=args.gather_collective)          # three-stage pipeline, recycled pinned staging
    gathered = [0]
    rccl = device_identities()           # which physical device every rank computes on (N ranks must show N distinct devices)

    restarts = [0]
    host_s = {"drain": 0.0, "tuples": 0.0, "gather_submit": 0.0}     # host seconds per activity besides stepping (timed region and warm-up)

    def one_step():
        """G more positions are produced (every resident game advances one move; a game that ends on the way -- two passes in a
        row happen with a random-init net on small boards -- is restarted at once, like the worker body does); then this step's
        records are gathered to rank 0"""
        target = eng.status.total_moves + G
        while eng.status.total_moves < target:
            st = eng.step()
            if st.error:
                raise RuntimeError("engine error %d in slot %d" % (st.error, st.error_game))
            if st.n_active < G:
                res = eng.results()
                again = [s for s in range(G) if res[s]["done"] == 1]
                if again:
                    eng.drain()
                    eng.start_games(again)
                    restarts[0] += len(again)
        h0 = time.perf_counter()
        n = eng.drain()
        h1 = time.perf_counter()
        recs = np.zeros(n + G, dtype=tdt)
        k = 0
        for s in range(G):
            for mv in eng.records[s]:
                r = recs[k]
                r["rank"] = rank; r["game"] = s; r["move_n"] = mv["move_n"]
                r["player"] = mv["player"]; r["value"] = mv["value"]; r["z"] = np.nan
                r["pi"] = mv["policy"]; r["state"] = mv["packed"]; r["action"] = mv["action"]
                r["game_seq"] = mv["game_seq"]
                k += 1
            eng.records[s] = []
        h2 = time.perf_counter()
        if args.gather:
            for got in exchange.submit(recs[:k]) + (exchange.flush() if args.gather_flush else []):
                gathered[0] += 0 if got is None else len(got)
        h3 = time.perf_counter()
        host_s["drain"] += h1 - h0; host_s["tuples"] += h2 - h1; host_s["gather_submit"] += h3 - h2

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    eng.advance_timing()
    captured = bool(getattr(eng, "graph", False))      # launches inside a graph carry no event timestamps
    time_convs = args.net == "resnet" and not args.plain_net and hasattr(net, "conv_events") and not captured
    if time_convs:
        net.conv_events = []
        net.side_flops = 0.0
        # launches shorter than ~0.5 ms are sampled (1 in 16): two event calls per launch would make the host the bottleneck
        net.conv_event_stride = 1 if G * E * (S - 2) * (S - 2) >= (1 << 20) else 16
    evals0 = eng.status.total_evals
    moves0 = eng.status.total_moves
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    for got in exchange.flush():             # the last two steps' tuples: the timed region ends with every tuple on rank 0
        gathered[0] += 0 if got is None else len(got)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    adv_ms, adv_n, adv_pos = eng.advance_timing()
    evals = eng.status.total_evals - evals0
    moves = eng.status.total_moves - moves0        # >= G * steps (the last round of a step may carry a few games further)
    conv_ms, conv_fl, conv_n, conv_big_ms, conv_big_n, conv_big_fl = 0.0, 0.0, 0, 0.0, 0, 0.0
    if time_convs:
        evs, net.conv_events = net.conv_events, None
        big = max((f for _, _, f in evs), default=0.0)
        for e0, e1, f in evs:
            ms = e0.elapsed_time(e1)
            conv_ms += ms; conv_fl += f; conv_n += 1
            if f == big:
                conv_big_ms += ms; conv_big_n += 1; conv_big_fl = f
        # tail launches of split batches ran on the side stream INSIDE the bracketed launches' wall time: their FLOPs count,
        # their time is already in conv_ms (net.FusedInferenceNet._tail_split)
        conv_side_fl = net.side_flops * (1.0 / net.conv_event_stride)
        conv_fl += conv_side_fl
    out = None
    if rank == 0:
        positions = world * G * args.steps
        per_launch = adv_pos / max(adv_n, 1)
        avg_ms = adv_ms / max(adv_n, 1)
        # record read + record written + legal set.  No network-input row exists any more: the stem kernel reads
        # the 768-byte records board_advance wrote (sgo_stem_packed_dev), so nothing else leaves this kernel
        per_leaf = ALGO_BYTES.get(S, 0)
        algo = per_leaf * per_launch
        achieved = algo / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        flops = net.flops_per_eval() if hasattr(net, "flops_per_eval") else 0
        sym_mult = 8 if args.symmetry == "avg8" else 1
        out = {
            "metric": "self-play positions/sec (19x19, 400 sims/move)" if (S, sims) == (19, 400) else
                      "self-play positions/sec (%dx%d, %d sims/move)" % (S, S, sims),
            "value": positions / dt, "unit": "positions/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 bitboards + f32 tree statistics (net: fp16)",
            "data": "synthetic: self-generated positions from the empty board, random-init weights",
            "config": {"workload": "%dx%d board, %d sims/move (energy %d), %d concurrent games per GPU, %s, symmetry=%s"
                                   % (S, S, sims, E, G, ("random-init %d-block/%d-filter resnet fp16" % (args.blocks, args.channels))
                                      if args.net == "resnet" else args.net + " stub net", args.symmetry),
                       "games_per_gpu": G, "sims": sims, "energy": E, "net_evals_per_position": (sims // E) * E + 1,
                       "sharding": "games g -> rank g mod N; weights broadcast from rank 0; %s gather of per-step records to rank 0"
                                   % ("RCCL" if args.backend == "nccl" else "gloo"),
                       "backend": args.backend, "weights_broadcast": bcast},
            "engine": {"halves": args.halves, "captured_rounds": captured, "packed_input": bool(getattr(eng, "packed", False)),
                       "positions_in_window": int(moves), "games_restarted_in_window": restarts[0],
                       "host_seconds_outside_stepping_incl_warmup": host_s,
                       "tree_blocks": (eng.halves[0].pool_info() if args.halves == 2 else eng.pool_info()),
                       "graph_replays": (sum(e.n_graph_replays for e in eng.halves) if args.halves == 2 else getattr(eng, "n_graph_replays", 0))},
            "rccl": dict(rccl, tuples_on_rank0=gathered[0],
                         gather="TupleGather: counts all_gather + padded gather to rank 0, 3-stage pipeline (host never waits on a fresh "
                                "collective), %s" % ("beside the step on a side stream" if args.gather_stream == "side" else
                                                     "ordered into the stepping stream between two steps")),
            "roofline_board_advance": {"bound": "hbm", "kernel": "board_advance in situ (make_play + legal set + history move of the step's leaf list; "
                                                                 "k_board_advance_rows, one half-wavefront per leaf, up to 32 768 leaves, k_board_advance above)",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "algorithmic_bytes_per_position": per_leaf,
                         "network_input": "none: the stem reads the packed records (%s)" % ("sgo_stem_packed_dev" if getattr(eng, "packed", False) else "tensor route: k_nn_pack"),
                         "positions_per_launch": per_launch, "avg_launch_ms": avg_ms, "launches": adv_n,
                         "note": "one launch per search round carries games x energy leaves: %d x 1 834 B = %.1f MB, 1.9 us at the HBM peak -- the figure is "
                                 "the floor of one small launch (dependent load -> compute -> store of 4 096 wavefronts), not a bandwidth limit; the same "
                                 "kernel family on a chip-filling batch is roofline_saturated" % (int(per_launch), per_launch * per_leaf / 1e6)},
            "net": {"evals": int(evals), "flops_per_eval": flops,
                    "achieved_tflops": (evals * sym_mult * flops / dt / 1e12) if flops else None,
                    "peak_tflops": 2500.0, "bound": "mfma"},
        }
        if conv_n:
            # the dominant kernel of the step (98 % of the GPU time): the tower convolution, timed launch by launch with HIP
            # events on the launch stream inside the timed region; `achieved` is over ALL its launches (both batch sizes)
            ach = conv_fl / (conv_ms * 1e-3) / 1e12
            from sejonggo_amd import _lib as _L2
            tk = _L2.load().sgo_conv_tower_kernel(-1)      # -1 leaves the selection as it is and returns it
            kname = ("sgo_conv4r::k_conv4r (two 256-thread workgroups per CU, weights L2 -> registers; csrc/sgo_conv4r.hpp)" if getattr(net, "packed_tower", False) else
                     "sgo_conv8w::k_conv8w (one 512-thread workgroup per CU; csrc/sgo_conv8w.hpp)" if tk == 0 else
                     "sgo_conv4w::k_conv4w (two 256-thread workgroups per CU; csrc/sgo_conv4w.hpp)")
            out["roofline"] = {"bound": "mfma", "kernel": kname + ": tower 3x3 convolution 256->256 + bias (+ skip) + ReLU, fp16 in / fp32 accumulate",
                               "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": None,
                               "launches": conv_n, "launches_sampled_1_in": net.conv_event_stride, "avg_launch_ms": conv_ms / conv_n, "flops_per_launch_mean": conv_fl / conv_n,
                               "largest_batch": {"launches": conv_big_n, "avg_launch_ms": conv_big_ms / max(conv_big_n, 1),
                                                 "flops_per_launch": conv_big_fl,
                                                 "achieved": conv_big_fl * conv_big_n / max(conv_big_ms * 1e-3, 1e-12) / 1e12},
                               "share_of_step_time": conv_ms * 1e-3 / dt,
                               "tail_split": {"enabled": bool(getattr(net, "tail_split", False)), "side_stream_flops": conv_side_fl,
                                              "note": "8 192 positions = 9 248 tiles = 36.1 rounds of 256 workgroups; the 29 positions "
                                                      "beyond 36 whole rounds run on a side stream inside the main launches' wall time"}}
            pc = os.path.join(ROOT, "profiles", "r03_pmc_conv.json")
            if os.path.isfile(pc) and (S, G, E) == (19, 1024, 8) and not getattr(net, "packed_tower", False) and tk != 0:
                try:   # HBM bytes per 8192-batch launch: separate --pmc passes; FETCH_SIZE doubled (gfx950 wide-read correction)
                    j = json.load(open(pc))
                    out["roofline"]["traffic"] = (2 * j["FETCH_SIZE"]["mean"] + j["WRITE_SIZE"]["mean"]) * 1024
                    out["roofline"]["traffic_source"] = ("profiles/r03_pmc_conv.json (tools/pmc_conv_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                         "passes of this command, 8192-position launches; algorithmic x + skip + y + weights = 3.63 GB with skip, 2.42 GB without)")
                except Exception:
                    pass
        else:
            out["roofline"] = dict(out["roofline_board_advance"])
        tr = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
        if os.path.isfile(tr) and (S, G, E) == (19, 1024, 8):
            try:
                out["roofline_board_advance"]["traffic"] = json.load(open(tr)).get("traffic_bytes_per_launch_corrected")
                out["roofline_board_advance"]["traffic_source"] = "profiles/%s (tools/pmc_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)" % os.path.basename(tr)
            except Exception:
                pass
        eng.close()
        if (args.avg8_leg and world == 1 and args.symmetry == "random1" and args.net == "resnet" and not args.plain_net
                and (S, sims, G, args.blocks, args.channels) == (19, 400, 1024, 20, 256)):
            try:
                out["config3_avg8"] = avg8_leg(net, S, G, sims, E, local)
            except Exception as ex:
                out["config3_avg8"] = {"error": repr(ex)}
        if args.steady_state and world == 1:
            try:
                out["steady_state"] = steady_state(args, net, S, G, sims, E, local, out["value"])
            except Exception as ex:
                out["steady_state"] = {"error": repr(ex)}
        if args.saturated and world == 1:
            try:
                out["roofline_saturated"] = saturated_advance(S)
            except Exception as ex:
                out["roofline_saturated"] = {"error": repr(ex)}
        if args.cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, S, sims, E, args.blocks, args.channels, args.symmetry)
            except Exception as ex:  # the GPU number stands on its own; say why the comparator is missing
                out["cpu_baseline"] = {"value": None, "error": repr(ex)}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
