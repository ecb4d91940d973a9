"""The GPU shard behind hanabi_sad_amd.dist.ShardedReplay: with one shard the sharded path (draw_canonical ->
stratified positions on the host -> hsad_replay_sample_at -> host IS weights) must reproduce
hsad_replay_sample -- ids and rows bit-exactly, weights to rtol 1e-6 -- including eviction and priority updates;
with the buffer cut into two shards on the same GPU the union must equal the single buffer's draw."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]
DEV = "cuda:0"
T, D = 6, 5
FIELDS = [("s", D, torch.float32), ("a", 1, torch.int64)]


def make_rows(rng, n, base):
    s = torch.tensor(rng.standard_normal((n, T, D)).astype(np.float32), device=DEV)
    a = (torch.arange(n, device=DEV).view(n, 1, 1) + base).expand(n, T, 1).contiguous()
    reward = torch.tensor(rng.random((n, T)).astype(np.float32), device=DEV)
    terminal = torch.tensor(rng.integers(0, 2, (n, T)).astype(np.uint8), device=DEV)
    bootstrap = 1 - terminal.float()
    seq_len = torch.tensor(rng.integers(1, T + 1, n).astype(np.float32), device=DEV)
    prio = torch.tensor((rng.random(n) * 3 + 0.01).astype(np.float32), device=DEV)
    return {"s": s, "a": a}, reward, terminal, bootstrap, seq_len, prio


def same_batch(x, y):
    (f1, r1, t1, b1, l1), w1 = x
    (f2, r2, t2, b2, l2), w2 = y
    assert torch.equal(f1["a"], f2["a"]) and torch.equal(f1["s"], f2["s"])
    assert torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(b1, b2) and torch.equal(l1, l2)
    assert torch.allclose(w1, w2, rtol=1e-6, atol=0)


def test_one_shard_equals_plain_sample_including_eviction_and_updates():
    from hanabi_sad_amd.dist import ShardedReplay
    from hanabi_sad_amd.replay import DeviceReplay
    cap, B = 48, 16
    plain = DeviceReplay(cap, 11, 0.9, 0.6, 0, T, FIELDS, DEV)
    shard = DeviceReplay(cap, 11, 0.9, 0.6, 0, T, FIELDS, DEV)
    sr = ShardedReplay(shard, 0.6, DEV, host_path=True)
    rng = np.random.default_rng(0)
    base = 0
    for it in range(6):
        n = 20 if it % 2 == 0 else 7
        rows = make_rows(rng, n, base)
        base += n
        plain.add(*rows)
        shard.add(*rows)
        x, y = plain.sample(B), sr.sample(B)
        same_batch(x, y)
        newp = torch.tensor((rng.random(B) * 2 + 0.05).astype(np.float32), device=DEV)
        plain.update_priority(newp)
        sr.update_priority(newp)
        assert plain.size() == shard.size()
        assert abs(plain.priority_sum()[0] - shard.priority_sum()[0]) < 1e-9
    plain.check_errors()
    shard.check_errors()


def test_two_shards_on_one_gpu_draw_the_same_elements_as_one_buffer():
    """cut 60 sequences into shards of 25 + 35 and serve one stratified draw from both: identical elements"""
    from hanabi_sad_amd.dist import split_positions, stratified_positions
    from hanabi_sad_amd.replay import DeviceReplay
    B = 32
    rng = np.random.default_rng(1)
    rows = make_rows(rng, 60, 0)
    whole = DeviceReplay(128, 5, 0.9, 0.6, 0, T, FIELDS, DEV)
    whole.add(*rows)
    cut = lambda lo, hi: ({k: v[lo:hi].contiguous() for k, v in rows[0].items()},) + tuple(r[lo:hi].contiguous() for r in rows[1:])
    shards = [DeviceReplay(64, 5, 0.9, 0.6, 0, T, FIELDS, DEV) for _ in range(2)]
    shards[0].add(*cut(0, 25))
    shards[1].add(*cut(25, 60))
    canon = shards[0].draw_canonical(B)
    (fw, *_), _ = whole.sample(B)                       # consumes the same first B uniforms (same seed)
    sums = [s.priority_sum()[0] for s in shards]
    assert abs(sum(sums) - whole.priority_sum()[0]) < 1e-6
    pos = stratified_positions(canon, sum(sums), B)
    owner, local = split_positions(pos, sums)
    got = []
    for k, s in enumerate(shards):
        (f, *_), raw = s.sample_at(local[owner == k])
        got.append(f["a"][0, :, 0])
    got = torch.cat(got)
    want = fw["a"][0, :, 0]
    # a position within float32 rounding of an element boundary may resolve to the neighbouring element
    assert int((got != want).sum()) <= 1, (got, want)


def test_two_rank_selfplay_assembles_batches_from_both_shards(tmp_path):
    """End to end on this box's single GPU: two ranks (gloo transport, tensors staged through host memory) each roll
    out their game shard into their own DeviceReplay shard; rank 0 learns from batches assembled from both."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "hanabi_sad_amd.selfplay", "--num_game", "512", "--num_update", "12",
           "--burn_in_frames", "300", "--rnn_hid_dim", "256", "--batchsize", "64", "--dist_backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("Speed: train:") == 2 and "update 0 loss" in out.stdout
