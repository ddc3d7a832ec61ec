"""The N > 1 path on CPU (gloo): static game sharding, the variable-length gather of (s, pi, z) tuples to rank 0 and
the weight broadcast from rank 0 (sejonggo_amd/distributed.py), started through the launcher bench.py --gpus N uses."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(world, tmp_path, backend="gloo"):
    from sejonggo_amd.distributed import launch_ranks
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    rc = launch_ranks([os.path.join(HERE, "dist_worker.py"), backend, str(tmp_path)], world, env=env, timeout=300)
    assert rc == 0
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]


def _check(res, world):
    from sejonggo_amd.distributed import tuple_dtype
    dt = tuple_dtype(9)
    r0 = res[0]
    out = np.frombuffer(r0["out"].tobytes(), dtype=dt)
    sizes = [3 + 4 * r for r in range(world)]
    assert int(r0["n_out"]) == sum(sizes) == len(out)
    assert int(r0["n_empty"]) == world - 1 if world > 1 else int(r0["n_empty"]) == 0
    assert list(out["rank"]) == sum(([r] * n for r, n in enumerate(sizes)), [])
    o = 0
    for r, n in enumerate(sizes):
        part = out[o:o + n]
        assert list(part["game"]) == [r + world * i for i in range(n)]          # game g -> rank g mod world
        assert (part["z"] == 1.0 - 2.0 * r).all()
        assert (part["pi"][-1] == np.arange(82, dtype=np.float32) + r).all()
        assert (part["state"][0] == np.arange(dt["state"].shape[0], dtype=np.uint32) * (r + 1)).all()
        o += n
    for r in range(1, world):
        assert int(res[r]["n_out"]) == -1 and int(res[r]["n_empty"]) == -1      # only rank 0 receives
    # weights: different before, rank 0's after, checksums equal everywhere
    assert len({float(x["before"]) for x in res}) == world
    assert all(float(x["after"]) == float(res[0]["before"]) for x in res)
    assert all(bool(x["identical"]) for x in res) and len({int(x["checksum"]) for x in res}) == 1
    assert int(r0["n_tensors"]) > 10 and int(r0["nbytes"]) > 0
    # the pipelined side-stream gather (TupleGather: three stages, two submits of latency, submit order kept) and the
    # device identities every rank reports for the bench line
    assert all(bool(x["pipe_ok"]) for x in res)
    assert all(int(x["ident_world"]) == world and int(x["ident_n"]) == world for x in res)


def test_shard_games():
    from sejonggo_amd.distributed import shard_games
    assert shard_games(10, 4, 1) == [1, 5, 9]
    assert sorted(sum((shard_games(8192, 8, r) for r in range(8)), [])) == list(range(8192))


def test_gather_and_broadcast_world2(tmp_path):
    _check(_launch(2, tmp_path), 2)


def test_gather_and_broadcast_world4(tmp_path):
    """Four ranks (half of the node the driver scales to): ragged gathers incl. slot growth on every non-sending rank, the
    pipelined submit order, device identities of four ranks, weights from rank 0."""
    _check(_launch(4, tmp_path), 4)


def test_gather_and_broadcast_world1_runs_the_collectives(tmp_path):
    """A one-rank process group still goes through all_gather / gather / broadcast (no early return)."""
    _check(_launch(1, tmp_path), 1)


def test_launcher_reports_a_failing_rank(tmp_path):
    from sejonggo_amd.distributed import launch_ranks
    script = os.path.join(str(tmp_path), "boom.py")
    open(script, "w").write("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(30)\n")
    assert launch_ranks([script], 2, timeout=60) == 3


@pytest.mark.gpu
def test_rccl_one_rank_on_the_device(tmp_path):
    """backend nccl (= RCCL) with world_size 1 on the real GPU: the same all_gather / gather / broadcast calls bench.py
    issues at N > 1, on CUDA tensors."""
    _check(_launch(1, tmp_path, backend="nccl"), 1)


def test_external_launcher_rendezvous_over_tcp(tmp_path):
    """The driver's way: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P script`.
    No SGO_RDZV_PORT in the environment, so init_from_env must take the launcher's MASTER_ADDR / MASTER_PORT."""
    import subprocess
    from sejonggo_amd.distributed import free_port
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("SGO_RDZV_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(HERE, "dist_worker.py"), "gloo", str(tmp_path)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    _check([np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(2)], 2)
