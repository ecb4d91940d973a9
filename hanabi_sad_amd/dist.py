"""Multi-GPU sharding helpers.  The env path shards by game (each game owns its state and RNG:
pyhanabi/create.py:36-53), one process per GPU, no data-path collective.  The learner path has three real exchange
steps, all here: the parameter broadcast rank 0 -> actors (`broadcast_params`), the assembly of a learner batch from the
per-GPU replay shards and the scatter of the new priorities back to the owning shards (`ShardedReplay`).
torch.distributed = RCCL on ROCm ("nccl"), gloo in the CPU tests."""
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


def comm_device_for(device):
    """tensors travel on the GPU with RCCL ("nccl"); with the gloo backend (CPU tests, or several ranks sharing one GPU
    in a smoke run) they are staged through host memory"""
    import torch.distributed as dist
    return torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_params(tensors, src=0):
    """Parameter broadcast rank `src` -> all ranks (the reference's BatchRunner::updateModel across devices,
    rela/batch_runner.h:74-77): one flat bucket per call so RCCL moves a single ~37 MB message over xGMI."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors]).to(comm_device_for(tensors[0].device))
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def stratified_positions(canon, total, batch):
    """prioritized_replay.h:300-305 in float32: rand_i = U(0, segment) + i * segment, clamped to sum - 0.1"""
    import numpy as np
    total = np.float32(total)
    seg = np.float32(total / np.float32(batch))
    i = np.arange(batch, dtype=np.float32)
    r = canon.astype(np.float32) * seg + i * seg
    return np.minimum(total - np.float32(0.1), r).astype(np.float32)


def split_positions(pos, shard_sums):
    """Assign each (ascending) global position to the first shard whose inclusive prefix of weight reaches it.
    -> owner [B] int, local target [B] float32 (position minus the weight of the shards before the owner)"""
    import numpy as np
    sums = np.asarray(shard_sums, dtype=np.float64)
    incl = np.cumsum(sums)
    owner = np.searchsorted(incl, pos.astype(np.float64), side="left")
    owner = np.minimum(owner, len(sums) - 1)
    # never hand a position to an empty shard
    nz = np.nonzero(sums > 0)[0]
    if len(nz) == 0:
        raise RuntimeError("ShardedReplay.sample: every shard is empty")
    for j in range(len(owner)):
        if sums[owner[j]] <= 0:
            owner[j] = nz[np.argmin(np.abs(nz - owner[j]))]
    excl = incl - sums
    local = (pos.astype(np.float64) - excl[owner]).astype(np.float32)
    return owner.astype(np.int64), np.maximum(local, np.float32(0))


class ShardedReplay:
    """One prioritized-replay shard per rank behaving, for the learner on `learner_rank`, like ONE
    PrioritizedReplay over the concatenation of the shards (rank order):

      sample(B)           collective.  shard sums/sizes are all-gathered (8 doubles), the learner's shard draws the B
                          canonical uniforms and broadcasts them, every rank cuts the reference's stratified positions
                          into per-shard targets, samples its own quota on its own GPU and ships the rows to the learner
                          (point-to-point; quotas are contiguous runs of the batch because positions ascend).
                          Learner gets (fields, reward, terminal, bootstrap, seq_len), weight; other ranks get None.
      update_priority(p)  collective.  the learner sends every shard the new priorities of the rows it contributed.

    `shard` needs priority_sum() -> (sum, size), draw_canonical(n), sample_at(targets) and update_priority(p) --
    hanabi_sad_amd.replay.DeviceReplay on a GPU, any stand-in with the same methods in the gloo CPU tests."""

    def __init__(self, shard, beta, device, learner_rank=0, host_path=False):
        import torch.distributed as dist
        self.shard, self.beta, self.device, self.learner = shard, float(beta), torch.device(device), learner_rank
        self.host_path = bool(host_path)   # tests: take the collective code path (host positions, sample_at) with one shard
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.comm = comm_device_for(self.device) if self.on else self.device
        self._owner = None

    def _gather_sums(self):
        sm, sz = self.shard.priority_sum()
        if not self.on:
            return [sm], [sz]
        import torch.distributed as dist
        mine = torch.tensor([sm, float(sz)], dtype=torch.float64, device=self.comm)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        out = torch.stack(out).cpu()
        return [float(x) for x in out[:, 0]], [int(x) for x in out[:, 1]]

    def sample(self, batch):
        import numpy as np
        if not self.on and not self.host_path and hasattr(self.shard, "sample"):
            # one shard: the plain device-side sampler draws the same batch (tests/test_sharded_replay_gpu.py) without
            # reading the priority sum back to the host, so the host keeps running ahead of the GPU
            self._owner = np.full(batch, self.rank, dtype=np.int64)
            return self.shard.sample(batch)
        sums, sizes = self._gather_sums()
        canon = torch.zeros(batch, dtype=torch.float32, device=self.comm)
        if self.rank == self.learner:
            canon = torch.from_numpy(self.shard.draw_canonical(batch)).to(self.comm)
        if self.on:
            import torch.distributed as dist
            dist.broadcast(canon, src=self.learner)
        canon = canon.cpu().numpy()
        total = float(np.sum(np.asarray(sums, dtype=np.float64)))
        pos = stratified_positions(canon, total, batch)
        owner, local = split_positions(pos, sums)
        self._owner = owner
        mine = owner == self.rank
        (fields, reward, terminal, bootstrap, seq_len), raw_w = self.shard.sample_at(local[mine])
        names = list(fields.keys())
        parts = [fields[k] for k in names] + [reward, terminal.to(torch.uint8), bootstrap, seq_len, raw_w]
        if not self.on:
            out = parts
        elif self.rank != self.learner:
            import torch.distributed as dist
            if int(mine.sum()) > 0:
                for t in parts:
                    dist.send(t.contiguous().to(self.comm), dst=self.learner)
            return None
        else:
            import torch.distributed as dist
            out = []
            for t in parts:   # assembled tensors: batch is dim 1 for [T, n, ...] tensors and dim 0 for [n] vectors
                shape = list(t.shape)
                shape[1 if t.dim() >= 2 else 0] = batch
                out.append(torch.empty(shape, dtype=t.dtype, device=self.device))
            for k in range(self.world):
                idx = np.nonzero(owner == k)[0]
                if len(idx) == 0:
                    continue
                a, b = int(idx[0]), int(idx[-1]) + 1      # contiguous run
                for t_loc, t_out in zip(parts, out):
                    dim = 1 if t_out.dim() >= 2 else 0
                    if k == self.rank:
                        t_out.narrow(dim, a, b - a).copy_(t_loc)
                    else:
                        shape = list(t_out.shape)
                        shape[dim] = b - a
                        buf = torch.empty(shape, dtype=t_out.dtype, device=self.comm)
                        dist.recv(buf, src=k)
                        t_out.narrow(dim, a, b - a).copy_(buf)
        nf = len(names)
        f = {k: t for k, t in zip(names, out[:nf])}
        reward, terminal, bootstrap, seq_len, raw_w = out[nf:]
        # (N * w / sum)^-beta / max  with N, sum over all shards (prioritized_replay.h:322-333)
        n_total = float(sum(sizes))
        y = torch.pow(n_total * (raw_w / np.float32(total)), -self.beta)
        weight = y / y.max()
        return (f, reward, terminal.bool(), bootstrap, seq_len), weight

    def update_priority(self, priority=None):
        """learner: priority [B] for the batch returned by the last sample(); other ranks: no argument"""
        import numpy as np
        owner = self._owner
        self._owner = None
        n_mine = int((owner == self.rank).sum())
        if not self.on:
            self.shard.update_priority(priority)
            return
        import torch.distributed as dist
        if self.rank == self.learner:
            priority = priority.to(self.device, torch.float32).contiguous()
            for k in range(self.world):
                idx = np.nonzero(owner == k)[0]
                if len(idx) == 0:
                    continue
                part = priority[int(idx[0]):int(idx[-1]) + 1].contiguous()
                if k == self.rank:
                    self.shard.update_priority(part)
                else:
                    dist.send(part.to(self.comm), dst=k)
            if n_mine == 0:
                self.shard.update_priority(torch.empty(0, dtype=torch.float32, device=self.device))
        else:
            part = torch.empty(n_mine, dtype=torch.float32, device=self.comm)
            if n_mine > 0:
                dist.recv(part, src=self.learner)
            self.shard.update_priority(part.to(self.device))
