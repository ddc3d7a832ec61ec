"""One rank of the multi-process tests (started by sejonggo_amd.distributed.launch_ranks, the same launcher bench.py
--gpus N uses): gathers ragged (s, pi, z) tuples to rank 0 and takes the network weights from rank 0.
usage: dist_worker.py <backend> <out_dir>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend, out_dir = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    from sejonggo_amd.distributed import (init_from_env, tuple_dtype, gather_tuples, shard_games, broadcast_net, net_tensors,
                                          TupleGather, device_identities)
    rank, world, dev = init_from_env(backend)
    dt = tuple_dtype(9)
    n = 3 + 4 * rank          # ragged: rank 0 has 3 tuples, rank 1 has 7
    t = np.zeros(n, dtype=dt)
    t["rank"] = rank
    t["game"] = shard_games(2 * n * world, world, rank)[:n]
    t["move_n"] = np.arange(n)
    t["z"] = 1.0 - 2.0 * rank
    t["pi"] = (np.arange(82, dtype=np.float32) + rank)[None, :]
    t["state"] = (np.arange(dt["state"].shape[0], dtype=np.uint32) * (rank + 1))[None, :]
    out = gather_tuples(t)
    empty = gather_tuples(t[:0] if rank == world - 1 else t[:1])   # one rank contributes nothing
    # the pipelined form the self-play loops use: four batches of different lengths (one of them empty on the last rank),
    # nothing completes before the third submit, batches come back in submit order
    tg = TupleGather(dt, side_stream=(os.environ.get('SGO_TEST_SIDE_STREAM', '1') == '1'))   # the side-stream form (a CUDA stream under nccl)
    done_at, got = [], []
    for k in range(4):
        m = 0 if (k == 2 and rank == world - 1) else min(n, 1 + k + rank)
        b = t[:m].copy()
        b["game_seq"] = k
        r = tg.submit(b)
        done_at.append(len(r))
        got.extend(r)
    got.extend(tg.flush())
    pipe_ok = done_at[:2] == [0, 0] and done_at[2:] == [1, 1] and len(got) == 4 and not tg.inflight
    if rank == 0:
        for k, a in enumerate(got):
            want = sum(0 if (k == 2 and r == world - 1) else min(3 + 4 * r, 1 + k + r) for r in range(world))
            pipe_ok = pipe_ok and a is not None and len(a) == want and bool((a["game_seq"] == k).all()) \
                and list(a["rank"]) == sorted(a["rank"])
    else:
        pipe_ok = pipe_ok and all(a is None for a in got)
    # slots that must GROW: the last rank suddenly sends 60 tuples (33 KB: its own slot is re-allocated at stage 1, every other
    # rank's at stage 2 when it learns the count), then everybody is small again; with the all_gather collective as well
    for coll in ("gather", "all_gather"):
        tg2 = TupleGather(dt, collective=coll, side_stream=False)
        got2 = []
        sizes = [2, 60 if rank == world - 1 else 1, 3, 1, 2]
        for k, m in enumerate(sizes):
            b = np.zeros(m, dtype=dt)
            b["rank"] = rank; b["game_seq"] = 100 + k; b["move_n"] = np.arange(m)
            b["pi"] = np.float32(rank + k)
            got2.extend(tg2.submit(b))
        got2.extend(tg2.flush())
        pipe_ok = pipe_ok and len(got2) == len(sizes)
        if rank == 0:
            for k, a in enumerate(got2):
                want = [(60 if (k == 1 and r == world - 1) else [2, 1, 3, 1, 2][k]) for r in range(world)]
                pipe_ok = pipe_ok and a is not None and len(a) == sum(want) and bool((a["game_seq"] == 100 + k).all())
                o = 0
                for r, w in enumerate(want):
                    part = a[o:o + w]
                    pipe_ok = pipe_ok and bool((part["rank"] == r).all()) and list(part["move_n"]) == list(range(w)) \
                        and bool((part["pi"] == np.float32(r + k)).all())
                    o += w
    ident = device_identities()
    # weights: every rank starts from different values; after the broadcast all hold rank 0's
    from sejonggo_amd.net import PolicyValueNet
    torch.manual_seed(100 + rank)
    net = PolicyValueNet(5, 1, 8, name="r%d" % rank)
    if backend == "nccl":
        net = net.cuda()
    before = float(net.p_fc.weight.double().sum().item())
    info = broadcast_net(net)
    after = float(net.p_fc.weight.double().sum().item())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), before=before, after=after, identical=info["identical"],
             checksum=np.int64(info["checksum"]), nbytes=info["bytes"], n_tensors=len(net_tensors(net)),
             out=(np.frombuffer(out.tobytes(), dtype=np.uint8) if out is not None else np.zeros(0, np.uint8)),
             n_out=(-1 if out is None else len(out)), n_empty=(-1 if empty is None else len(empty)),
             pipe_ok=bool(pipe_ok), ident_world=ident["world"], ident_n=len(ident["devices"]), ident_distinct=ident["distinct"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
