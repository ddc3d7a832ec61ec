"""Multi-GPU side of the path: games shard statically over ranks (one process per GPU, no exchange during
search); the collectives are (1) one broadcast of the network weights from rank 0 when the replicas are set up and
(2) the gather of finished (s, pi, z) tuples to rank 0 (SURVEY.md §8e -- the reference ships files by scp,
scpy.py:68-76).  Works on RCCL (backend "nccl", CUDA tensors) and on gloo (CPU tensors; the world_size-2 CPU tests).

Variable-length gather (TupleGather): all_gather of one int64 count per rank, then dist.gather of max-padded uint8
blocks (on the 8-GPU xGMI mesh that is 7 concurrent point-to-point transfers into rank 0, a few MB at
most, far below one link's bandwidth), then rank 0 trims the padding -- pipelined over three submits on a side stream
with per-batch staging buffers, so the exchange of step k overlaps the search of steps k+1 and k+2.

`launch_ranks` starts one fresh interpreter per rank (never a fork of a process that may hold a GPU) with the
torch.distributed environment set; bench.py --gpus N and the tests use it."""
import os
import socket
import subprocess
import sys

import numpy as np


def tuple_dtype(size):
    N = size * size
    NW = (N + 31) // 32
    RW = 16 * NW
    return np.dtype([("rank", "<i4"), ("game", "<i4"), ("game_seq", "<i4"), ("move_n", "<i4"), ("action", "<i4"),
                     ("player", "<i4"), ("value", "<f4"), ("z", "<f4"), ("state", "<u4", (RW,)),
                     ("pi", "<f4", (N + 1,))])


def shard_games(n_games_total, world_size, rank):
    """game g -> rank g mod world_size."""
    return [g for g in range(n_games_total) if g % world_size == rank]


# ---------------------------------------------------------------------------------------------- process launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(argv, n_ranks, env=None, timeout=None, master_port=None):
    """Runs `python argv...` once per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, each in
    a fresh interpreter.  The caller's stdout / stderr are inherited (rank 0 prints the result line).  Returns 0, or
    the exit code of the first rank that failed (the others are then terminated), or 124 on timeout."""
    # The launcher itself hosts the rendezvous store on a port the kernel assigns at bind time (SGO_RDZV_PORT): a port that is
    # only *picked* here can be taken by somebody else during the minute a fresh interpreter spends importing torch.  The
    # store is plain TCP on the CPU -- the launcher still never touches a GPU.  MASTER_PORT is exported for code that wants one.
    from datetime import timedelta
    from torch.distributed import TCPStore
    store = TCPStore("127.0.0.1", master_port or 0, n_ranks, is_master=True, timeout=timedelta(seconds=1800), wait_for_workers=False)
    port = store.port
    procs = []
    for r in range(n_ranks):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 SGO_RDZV_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the ranks share the host's cores: without a cap every rank starts one OpenMP thread per core and the small host-side
        # tensor ops of a step crawl (measured: 157 ms instead of 6.5 ms per step with 2 ranks); same cap as torch.distributed.run
        if n_ranks > 1:
            e.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    try:
        return _wait_ranks(procs, n_ranks, timeout)
    finally:
        del store


def _wait_ranks(procs, n_ranks, timeout):
    import time
    t0 = time.time()
    codes = [None] * n_ranks
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = any(c not in (None, 0) for c in codes)
        timed_out = timeout is not None and time.time() - t0 > timeout
        if failed or timed_out:
            first = next((abs(c) for c in codes if c not in (None, 0)), 0)
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            return first if failed else 124
        time.sleep(0.05)
    return max(abs(c) for c in codes)


def init_from_env(backend="nccl"):
    """Initialises torch.distributed from the launcher's environment.  nccl (= RCCL): this rank's device is
    LOCAL_RANK and must exist; gloo: ranks may share devices (LOCAL_RANK mod device count; CPU-only hosts work too).
    Returns (rank, world, local_device or None)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rendezvous without a race on a TCP port where we control it: one rank needs no network at all (in-process HashStore),
    # ranks started by launch_ranks connect to the store their launcher hosts; an external launcher's MASTER_ADDR / MASTER_PORT
    # are used as given.
    kw = {}
    if world == 1:
        kw["store"] = dist.HashStore()
    elif os.environ.get("SGO_RDZV_PORT"):
        from datetime import timedelta
        kw["store"] = dist.TCPStore("127.0.0.1", int(os.environ["SGO_RDZV_PORT"]), world, is_master=False,
                                    timeout=timedelta(seconds=1800))
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    ndev = torch.cuda.device_count()
    if backend == "nccl":
        if local >= ndev:
            raise RuntimeError("rank %d needs device %d but only %d HIP device(s) are visible" % (rank, local, ndev))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), **kw)
        return rank, world, local
    dev = (local % ndev) if ndev > 0 else None
    if dev is not None:
        torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world, **kw)
    return rank, world, dev


def _comm_device():
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


# ---------------------------------------------------------------------------------------------- collectives
def gather_tuples(tuples, device=None, dst=0):
    """Blocking form: this rank's tuples (numpy structured array, tuple_dtype) -> the concatenation over ranks on rank
    `dst` (rank order), None elsewhere.  Without an initialised process group the input is returned; with one -- world
    size 1 included -- the collectives run.  The self-play loops use TupleGather below, which keeps the exchange off the
    stepping stream; this form is one submit + flush of it."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return tuples
    g = TupleGather(tuples.dtype, device=device, dst=dst)
    g.submit(tuples)
    out = g.flush()
    if dist.get_rank() != dst:
        return None
    return out[0] if out else tuples[:0]


class TupleGather(object):
    """Variable-length gather of (s, pi, z) tuples to rank `dst` that keeps the HOST off the critical path:
    every batch goes through three stages, one per `submit`, so the stepping thread never waits for a collective it has just
    issued:

      stage 1 (submit k)    payload -> pinned host block -> device block (async copy); all_gather of the per-rank counts
      stage 2 (submit k+1)  counts read (issued one step ago: complete), blocks padded to the largest, dist.gather to dst
      stage 3 (submit k+2)  dst: device -> pinned host copy, trimmed per rank, handed to the caller in submit order

    On RCCL the gather is 7 concurrent point-to-point transfers into rank 0 over the xGMI mesh (collective="all_gather": one
    ncclAllGather of the padded blocks instead); on gloo (CPU tensors) the same stages run without a stream.  Every tensor a
    collective touches lives in one of three ROTATING SLOTS that are allocated once and only ever grow: a tensor that is freed
    after a collective used it on the communication stream makes the caching allocator poll that stream's events on every
    later allocation (record_stream).  Host-side staging is plain memcpy (see _stage1).  `submit` returns the batches that
    completed (possibly none); `flush` drains."""

    SLOTS = 3

    def __init__(self, dtype, device=None, dst=0, side_stream=True, collective="gather"):
        """side_stream=True (default): copies on a side stream, collectives left to run BESIDE the stepping stream's kernels
        (SURVEY.md §8e's form: step k's gather overlaps the search of steps k+1, k+2).  side_stream=False: the same three stages,
        but every collective is ordered INTO the stepping stream (its next kernels queue behind it).  Measured equal within
        noise at config 2 (9 380 vs 9 423 positions/s, 9 489 without any gather)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.dtype = np.dtype(dtype)
        self.dst = dst
        # "gather": dist.gather to dst; "all_gather": ncclAllGather of the padded blocks (every rank receives everything, only
        # dst keeps it)
        self.collective = collective
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.dev = device if device is not None else _comm_device()
        self.on_gpu = self.dev.type == "cuda"
        self.side = torch.cuda.Stream(device=self.dev) if (self.on_gpu and side_stream) else None
        self.inflight = []          # batches in submit order, each a dict with its stage
        self.n_submitted = 0
        self.bytes_gathered = 0
        self.slots = [None] * self.SLOTS

    # -- persistent buffers ---------------------------------------------------------------------------------------
    def _slot(self, k, nbytes):
        """Slot k with room for nbytes per rank: allocated on first use, re-allocated (larger) only when a batch outgrows it."""
        torch = self.torch
        sl = self.slots[k]
        if sl is None or sl["cap"] < nbytes:
            cap = max(4096, int(nbytes * 1.5))
            holds = self.collective == "all_gather" or self.rank == self.dst
            sl = {"cap": cap,
                  "host_in": torch.empty(cap, dtype=torch.uint8, pin_memory=self.on_gpu),
                  "block": torch.zeros(cap, dtype=torch.uint8, device=self.dev),
                  "cnt_host": torch.zeros(1, dtype=torch.int64, pin_memory=self.on_gpu),
                  "cnt": torch.zeros(1, dtype=torch.int64, device=self.dev),
                  "counts": [torch.zeros(1, dtype=torch.int64, device=self.dev) for _ in range(self.world)],
                  "outs": [torch.zeros(cap, dtype=torch.uint8, device=self.dev) for _ in range(self.world)] if holds else None,
                  "host_out": torch.empty(cap * self.world, dtype=torch.uint8, pin_memory=self.on_gpu) if holds else None}
            self.slots[k] = sl
        return sl

    def _grow(self, b, nbytes):
        """Stage 2 found a peer's batch larger than this slot: move the batch to a bigger slot (its own payload is re-staged
        from the pinned copy)."""
        old = b["slot"]
        k = b["k"]
        self.slots[k] = None
        sl = self._slot(k, nbytes)
        if b["nbytes"]:
            sl["host_in"].numpy()[:b["nbytes"]] = old["host_in"].numpy()[:b["nbytes"]]
            sl["block"][:b["nbytes"]].copy_(sl["host_in"][:b["nbytes"]], non_blocking=True)
        b["slot"] = sl
        b["keep"] = old            # until the batch completes: the old tensors may still be in use by an earlier copy
        return sl

    # -- stages ---------------------------------------------------------------------------------------------------
    def _stream(self):
        import contextlib
        return self.torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()

    def _stage1(self, tuples):
        torch, dist = self.torch, self.dist
        n = len(tuples)
        raw = np.ascontiguousarray(tuples).view(np.uint8).reshape(-1)
        k = self.n_submitted % self.SLOTS
        sl = self._slot(k, raw.size)
        # plain memcpy through the numpy views of the pinned blocks: a torch CPU copy_ of this size goes to the intra-op thread
        # pool, whose workers then spin for a while on every core of the GPU box's CPU share -- ONE such copy per 27-ms step
        # slowed the stepping thread's kernel launches by 30-60 % at config 2 (tools/debug_gather.py: 26.1 -> 42.2 ms per move)
        if raw.size:
            sl["host_in"].numpy()[:raw.size] = raw
        sl["cnt_host"].numpy()[0] = n
        b = {"n": n, "nbytes": raw.size, "stage": 1, "slot": sl, "k": k}
        with self._stream():
            if raw.size:
                sl["block"][:raw.size].copy_(sl["host_in"][:raw.size], non_blocking=True)
            sl["cnt"].copy_(sl["cnt_host"], non_blocking=True)
            b["work"] = dist.all_gather(sl["counts"], sl["cnt"], async_op=True)
            if self.side is None:
                b["work"].wait()          # GPU-side ordering only: the stepping stream runs its next kernels BEHIND the collective
        return b

    def _stage2(self, b):
        dist = self.dist
        with self._stream():
            b["work"].wait()                                             # with a side stream: makes IT wait, never the stepping stream
            sl = b["slot"]
            counts = [int(c.item()) for c in sl["counts"]]               # issued a step ago: no wait worth the name
            b["count_list"] = counts
            width = max(max(counts), 1) * self.dtype.itemsize
            if width > sl["cap"]:
                sl = self._grow(b, width)
            block = sl["block"][:width]
            if self.collective == "all_gather":
                b["outs"] = [o[:width] for o in sl["outs"]]
                b["work"] = dist.all_gather(b["outs"], block, async_op=True)
            elif self.rank == self.dst:
                b["outs"] = [o[:width] for o in sl["outs"]]
                b["work"] = dist.gather(block, b["outs"], dst=self.dst, async_op=True)
            else:
                b["work"] = dist.gather(block, None, dst=self.dst, async_op=True)
            if self.side is None:
                b["work"].wait()
        b["stage"] = 2

    def _stage3(self, b):
        torch = self.torch
        b["stage"] = 3
        with self._stream():
            b["work"].wait()
        if self.rank != self.dst:
            return None
        sl = b["slot"]
        with self._stream():
            off, spans = 0, []
            for o, c in zip(b["outs"], b["count_list"]):
                nb = c * self.dtype.itemsize
                if nb:
                    sl["host_out"][off:off + nb].copy_(o[:nb], non_blocking=True)
                spans.append((off, nb))
                off += nb
            if self.side is not None:
                self.side.synchronize()                                  # this batch's own copies; the compute stream is not involved
            elif self.on_gpu:
                torch.cuda.current_stream().synchronize()
        out = np.frombuffer(sl["host_out"].numpy()[:off].tobytes(), dtype=self.dtype)
        self.bytes_gathered += off
        return out

    # -- driver ---------------------------------------------------------------------------------------------------
    def _advance(self, drain=False):
        done = []
        # oldest first, one stage per call per batch (all of them when draining); collectives are issued in the same order
        # on every rank because every rank submits the same number of batches
        while True:
            moved = False
            for b in list(self.inflight):
                if b["stage"] == 2:
                    done.append(self._stage3(b))
                    self.inflight.remove(b)
                    moved = True
                elif b["stage"] == 1:
                    self._stage2(b)
                    moved = True
            if not drain or not self.inflight or not moved:
                break
        return done

    def submit(self, tuples):
        """Queue this rank's batch (every rank must call submit the same number of times).  Returns the list of batches
        that completed during the call: on `dst` the gathered arrays in submit order, elsewhere Nones."""
        assert tuples.dtype == self.dtype
        done = self._advance()
        assert len(self.inflight) < self.SLOTS
        self.inflight.append(self._stage1(tuples))
        self.n_submitted += 1
        return done

    def flush(self):
        return self._advance(drain=True)


def device_identities():
    """One line per rank: which physical device the rank computes on (UUID where the runtime reports one, PCI address
    otherwise), gathered over the process group -- a SCALE record can then show that N ranks ran on N DISTINCT devices.
    Returns {"world", "backend", "devices": [...], "distinct"}."""
    import torch
    import torch.distributed as dist
    me = "cpu"
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(i)
        uuid = getattr(pr, "uuid", None)
        pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        me = "%s uuid=%s pci=%s" % (getattr(pr, "gcnArchName", pr.name), uuid, pci)
    if not (dist.is_available() and dist.is_initialized()):
        return {"world": 1, "backend": None, "devices": [me], "distinct": 1}
    alls = [None] * dist.get_world_size()
    dist.all_gather_object(alls, me)
    return {"world": dist.get_world_size(), "backend": dist.get_backend(), "devices": alls, "distinct": len(set(alls))}


def net_tensors(net):
    """The weight tensors of a resident net (net.FusedInferenceNet or a torch module), in a fixed order."""
    import torch
    if hasattr(net, "parameters"):
        return [p.data for p in net.parameters()] + [b.data for b in net.buffers()]
    out = []
    for name in ("stem_w", "stem_b"):
        out.append(getattr(net, name))
    for blk in getattr(net, "blocks", []):
        out.extend(blk)
    for name in ("head_w", "head_b", "p_fc_w", "p_fc_b", "v_fc1_w", "v_fc1_b", "v_fc2_w", "v_fc2_b"):
        out.append(getattr(net, name))
    return [t for t in out if torch.is_tensor(t)]


def _checksum(tensors):
    import torch
    total = 0
    for t in tensors:
        b = t.detach().contiguous().reshape(-1).view(torch.uint8).to(torch.int64)
        total = (total * 1000003 + int(b.sum().item()) + 31 * int((b * (torch.arange(b.numel(), device=b.device) % 251 + 1)).sum().item())) % (1 << 61)
    return total


def broadcast_net(net, src=0):
    """the replicas' weights come from rank `src` (one broadcast per tensor, ~47 MB for the 20-block
    net), then every rank's checksum is compared so the replicas are provably identical.  Returns
    {"bytes", "tensors", "checksum", "identical"}; without a process group nothing is sent."""
    import torch
    import torch.distributed as dist
    ts = net_tensors(net)
    nbytes = sum(t.numel() * t.element_size() for t in ts)
    if not (dist.is_available() and dist.is_initialized()):
        return {"bytes": nbytes, "tensors": len(ts), "checksum": _checksum(ts), "identical": True}
    on_cpu = dist.get_backend() != "nccl"
    for t in ts:
        if on_cpu and t.is_cuda:
            c = t.detach().cpu()
            dist.broadcast(c, src=src)
            t.copy_(c)
        else:
            dist.broadcast(t, src=src)
    cs = _checksum(ts)
    mine = torch.tensor([cs], dtype=torch.int64, device=_comm_device())
    alls = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(alls, mine)
    same = all(int(a.item()) == cs for a in alls)
    return {"bytes": nbytes, "tensors": len(ts), "checksum": cs, "identical": same}
