"""world_size-2 gloo test of the only collective on the path: the variable-length gather of (s, pi, z)
tuples to rank 0 (sejonggo_amd/distributed.py), plus the static game sharding."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from sejonggo_amd.distributed import tuple_dtype, gather_tuples, shard_games
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dt = tuple_dtype(9)
    n = 3 + 4 * rank          # ragged: rank 0 has 3 tuples, rank 1 has 7
    t = np.zeros(n, dtype=dt)
    t["rank"] = rank
    t["game"] = shard_games(2 * n, world, rank)[:n]
    t["move_n"] = np.arange(n)
    t["z"] = 1.0 - 2.0 * rank
    t["pi"] = (np.arange(82, dtype=np.float32) + rank)[None, :]
    t["state"] = (np.arange(dt["state"].shape[0], dtype=np.uint32) * (rank + 1))[None, :]
    out = gather_tuples(t)
    empty = gather_tuples(t[:0] if rank == 1 else t[:1])   # one rank contributes nothing
    if rank == 0:
        q.put((out.tobytes(), len(out), len(empty)))
    else:
        assert out is None and empty is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_tuples_world2():
    from sejonggo_amd.distributed import tuple_dtype, shard_games
    assert shard_games(10, 4, 1) == [1, 5, 9]
    assert sorted(sum((shard_games(8192, 8, r) for r in range(8)), [])) == list(range(8192))
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    raw, n, n_empty = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    dt = tuple_dtype(9)
    out = np.frombuffer(raw, dtype=dt)
    assert n == 10 and n_empty == 1
    assert list(out["rank"]) == [0] * 3 + [1] * 7
    assert list(out["game"][:3]) == [0, 2, 4] and list(out["game"][3:6]) == [1, 3, 5]
    assert (out["z"][:3] == 1).all() and (out["z"][3:] == -1).all()
    assert (out["pi"][5] == np.arange(82, dtype=np.float32) + 1).all()
    assert (out["state"][9] == np.arange(dt["state"].shape[0], dtype=np.uint32) * 2).all()
