"""World-size-2/3 gloo tests of the sharded-replay choreography (hanabi_sad_amd/dist.py ShardedReplay): batch assembly
from per-rank shards + priority scatter must behave like ONE PrioritizedReplay over the concatenated shards
(rela/prioritized_replay.h:291-334) fed the same canonical uniforms.  The shard here is a CPU stand-in with the
DeviceReplay shard interface; the GPU shard itself is covered by tests/test_sharded_replay_gpu.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from hanabi_sad_amd.dist import ShardedReplay, rank_world, stratified_positions

T, D, ALPHA, BETA, B = 4, 3, 0.9, 0.6, 16
SIZES = %(sizes)r                     # elements per shard (a shard may be empty)

class FakeShard:
    """CPU stand-in with the DeviceReplay shard interface; element j of shard k carries the tag 1000*k + j"""
    def __init__(self, k, n):
        g = np.random.default_rng(100 + k)
        self.k, self.w = k, (g.random(n).astype(np.float32) * 2 + 0.05) ** np.float32(ALPHA)
        self.rng = np.random.default_rng(7)
        self.last = None
    def priority_sum(self):
        return float(np.sum(self.w.astype(np.float64))), len(self.w)
    def draw_canonical(self, n):
        return self.rng.random(n, dtype=np.float32)
    def sample_at(self, targets):
        acc, ids = np.cumsum(self.w.astype(np.float64)), []
        for t in targets:
            ids.append(int(min(np.searchsorted(acc, np.float64(t), side="left"), len(self.w) - 1)))
        self.last = ids
        n = len(ids)
        tag = torch.tensor([1000 * self.k + i for i in ids], dtype=torch.float32)
        s = tag.view(1, n, 1).expand(T, n, D).contiguous() + torch.arange(T, dtype=torch.float32).view(T, 1, 1)
        a = tag.long().view(1, n, 1).expand(T, n, 1).contiguous()
        fields = {"s": s, "a": a}
        reward = tag.view(1, n).expand(T, n).contiguous() * 0.5
        terminal = (a[:, :, 0] %% 2).bool()
        return (fields, reward, terminal, reward + 1, tag.clone()), torch.from_numpy(self.w[ids].copy()) if n else torch.zeros(0)
    def update_priority(self, p):
        assert len(p) == len(self.last), (len(p), len(self.last))
        for i, v in zip(self.last, p.tolist()):
            self.w[i] = np.float32(v) ** np.float32(ALPHA)
        self.last = None

rank, world = rank_world()
dist.init_process_group("gloo", rank=rank, world_size=world)
shard = FakeShard(rank, SIZES[rank])
sr = ShardedReplay(shard, BETA, "cpu", learner_rank=0)
all_w = [FakeShard(k, SIZES[k]).w for k in range(world)]     # every rank can rebuild every shard's initial weights
for it in range(3):
    res = sr.sample(B)
    if rank == 0:
        (f, reward, terminal, bootstrap, seq_len), weight = res
        # single-buffer emulation of the reference over the concatenation of the shards, same uniforms
        cat = np.concatenate(all_w)
        tags = np.concatenate([1000 * k + np.arange(len(all_w[k])) for k in range(world)])
        canon = np.random.default_rng(7).random((it + 1) * B, dtype=np.float32)[it * B:]
        total = float(np.sum(np.concatenate([w.astype(np.float64) for w in all_w])))
        pos = stratified_positions(canon, total, B)
        acc = np.cumsum(cat.astype(np.float64))
        want = np.minimum(np.searchsorted(acc, pos.astype(np.float64), side="left"), len(cat) - 1)
        got_tags = seq_len.numpy()
        # positions that land within float32 rounding of an element boundary may legitimately resolve to a neighbour
        # (the shards subtract their offset in float32); everything else must be the identical element
        gap = np.minimum(np.abs(acc[want] - pos), np.abs(pos - np.where(want > 0, acc[want - 1], 0)))
        clear = gap > 1e-4 * max(total, 1.0)
        assert clear.sum() >= B - 2, clear
        assert np.array_equal(got_tags[clear], tags[want][clear].astype(np.float32)), (got_tags, tags[want])
        assert torch.equal(f["s"][:, :, 0], seq_len.view(1, B) + torch.arange(T, dtype=torch.float32).view(T, 1))
        assert torch.equal(f["a"][:, :, 0], seq_len.long().view(1, B).expand(T, B))
        assert torch.equal(reward, (seq_len * 0.5).view(1, B).expand(T, B)) and torch.equal(bootstrap, reward + 1)
        assert terminal.dtype == torch.bool and torch.equal(terminal, (f["a"][:, :, 0] %% 2).bool())
        raw = np.array([all_w[int(t) // 1000][int(t) %% 1000] for t in got_tags], dtype=np.float32)
        y = (np.float32(len(cat)) * (raw / np.float32(total))) ** np.float32(-BETA)
        assert np.allclose(weight.numpy(), y / y.max(), rtol=1e-5), (weight, y / y.max())
        newp = torch.tensor(got_tags %% 7 + 0.5 + it, dtype=torch.float32)
        for t, p in zip(got_tags, newp.tolist()):          # sequential like the reference: the last duplicate wins
            all_w[int(t) // 1000][int(t) %% 1000] = np.float32(p) ** np.float32(ALPHA)
        sr.update_priority(newp)
        packed = torch.from_numpy(np.concatenate(all_w))
    else:
        assert res is None
        sr.update_priority()
        packed = torch.empty(sum(SIZES), dtype=torch.float32)
    dist.broadcast(packed, src=0)                           # rank 0's model of every shard after the update
    off = sum(SIZES[:rank])
    all_w = [packed.numpy()[sum(SIZES[:k]):sum(SIZES[:k + 1])].copy() for k in range(world)]
    assert np.array_equal(shard.w, all_w[rank]), (rank, it)  # the scatter reached exactly the owning elements
dist.barrier()
dist.destroy_process_group()
open(os.path.join(%(out)r, "sr%%d.ok" %% rank), "w").write("ok")
'''


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("sizes", [[40, 25], [3, 0, 50]])
def test_sharded_replay_assembles_like_one_buffer_gloo(tmp_path, sizes):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "sizes": sizes, "out": str(tmp_path)})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % len(sizes), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert all((tmp_path / ("sr%d.ok" % r)).exists() for r in range(len(sizes)))


def test_split_positions_respects_shard_boundaries():
    from hanabi_sad_amd.dist import split_positions, stratified_positions
    sums = [10.0, 0.0, 5.0, 25.0]
    pos = np.array([0.0, 9.99, 10.0, 10.01, 14.9, 15.0, 15.2, 39.8], dtype=np.float32)
    owner, local = split_positions(pos, sums)
    assert owner.tolist() == [0, 0, 0, 2, 2, 2, 3, 3]
    assert np.allclose(local, [0, 9.99, 10.0, 0.01, 4.9, 5.0, 0.2, 24.8], atol=1e-5)
    canon = np.full(8, 0.5, dtype=np.float32)
    p = stratified_positions(canon, 40.0, 8)
    assert np.allclose(p, 2.5 + 5.0 * np.arange(8)) and p.dtype == np.float32
    assert stratified_positions(np.full(4, 0.999999, np.float32), 1.0, 4)[-1] <= np.float32(0.9)   # clamp to sum - 0.1
