"""Multi-GPU side of the path: games shard statically over ranks (one process per GPU, no exchange during
search); the only collective is the gather of finished (s, pi, z) tuples to rank 0 (SURVEY.md §8e --
the reference ships files by scp, scpy.py:68-76).  Works on RCCL (backend "nccl", CUDA tensors) and on
gloo (CPU tensors, used by the world_size-2 CPU tests).

Variable-length gather: all_gather of one int64 count per rank, then dist.gather of max-padded uint8
blocks (on the 8-GPU xGMI mesh that is 7 concurrent point-to-point transfers into rank 0, a few MB at
most, far below one link's bandwidth), then rank 0 trims the padding."""
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


def gather_tuples(tuples, device=None, dst=0):
    """tuples: numpy structured array (tuple_dtype) of this rank.  Returns the concatenation over ranks on
    rank `dst` (rank order), None elsewhere.  Single-process runs return the input."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tuples
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if dist.get_backend() == "nccl" else torch.device("cpu"))
    itemsize = tuples.dtype.itemsize
    cnt = torch.tensor([len(tuples)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    buf = torch.zeros(mx * itemsize, dtype=torch.uint8, device=dev)
    if len(tuples):
        raw = torch.from_numpy(np.frombuffer(tuples.tobytes(), dtype=np.uint8).copy())
        buf[:raw.numel()] = raw.to(dev)
    if rank == dst:
        outs = [torch.zeros_like(buf) for _ in range(world)]
        dist.gather(buf, outs, dst=dst)
        parts = [np.frombuffer(o.cpu().numpy().tobytes()[:c * itemsize], dtype=tuples.dtype) for o, c in zip(outs, counts)]
        return np.concatenate(parts) if parts else tuples[:0]
    dist.gather(buf, None, dst=dst)
    return None
