"""Multi-GPU sharding helpers.  The env path shards by game (each game owns its state and RNG:
pyhanabi/create.py:36-53), one process per GPU, no data-path collective; torch.distributed (RCCL on
ROCm, gloo on CPU) is used only for rendezvous, barriers and the max-over-ranks timing reduction."""
import os

import torch


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total, rank, world):
    """Contiguous, balanced slice [begin, end) of `total` games for `rank`; slices tile [0, total)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_seed(seed, begin):
    """Game g (global index) is seeded seed + g (create.py:41), so a shard starting at `begin` uses seed+begin."""
    return seed + begin


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (used for the bench's elapsed time)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_params(tensors, src=0):
    """Parameter broadcast rank `src` -> all ranks (the reference's BatchRunner::updateModel across devices,
    rela/batch_runner.h:74-77): one flat bucket per call so RCCL moves a single ~37 MB message over xGMI."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
