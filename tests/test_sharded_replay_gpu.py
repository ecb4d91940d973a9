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


@pytest.mark.parametrize("extra", [[], pytest.param(["--method", "vdn"], marks=pytest.mark.slow), ["--pred_weight", "0.25"]], ids=["iql", "vdn", "aux"])
def test_three_rank_selfplay_learner_and_free_running_actors(tmp_path, extra):
    """End to end on this box's single GPU: three ranks (gloo transport, tensors staged through host memory) -- rank 0 only learns,
    ranks 1 and 2 roll out their game shards into their own DeviceReplay shards and serve the learner's rounds between their steps
    (dist.ReplayLink); the learner's batches are assembled from both shards, parameters flow back, everybody stops cleanly."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "hanabi_sad_amd.selfplay", "--num_game", "512", "--num_update", "45",
           "--burn_in_frames", "300", "--rnn_hid_dim", "256", "--batchsize", "64", "--dist_backend", "gloo", "--actor_sync_freq", "10"] + extra
    # HSAD_LINK_DEBUG: the learner checks every assembled batch for non-finite values on the host -- the extra host synchronisations of that
    # check are what exposed (10 % of the runs without it, every run with it) importance weights computed OUTSIDE the exchange stream's
    # context in round 4's first pipelined link; a line of it in the output is a failure
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=root, env=dict(os.environ, HSAD_LINK_DEBUG="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "LINK_DEBUG" not in out.stdout, out.stdout[-2000:]
    assert out.stdout.count("Speed: train:") == 1 and "update 40 loss" in out.stdout and ("exchange_ms" in out.stdout or "batch_gather_ms" in out.stdout)


def test_three_rank_selfplay_over_the_kernel_free_ipc_transport():
    """The same three-rank job with the star round's messages on ReplayLink(transport = "ipc") (HSAD_LINK_TRANSPORT): rows, statistics,
    headers and the parameter bucket travel as one-sided device-to-device copies into IPC-mapped landing rings of the receiving process
    (three processes, one GPU: hipIpcOpenMemHandle of another process's allocation on the same device), announced through the rendezvous
    store -- no RCCL / gloo message carries data, no communication kernel waits on a CU (a posted RCCL receive would
    sit next to the learner's whole-chip persistent launches).  The process group (gloo here) only provides the store."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "hanabi_sad_amd.selfplay", "--num_game", "512", "--num_update", "45",
           "--burn_in_frames", "300", "--rnn_hid_dim", "256", "--batchsize", "64", "--dist_backend", "gloo", "--actor_sync_freq", "10"]
    env = dict(os.environ, HSAD_LINK_DEBUG="1", HSAD_LINK_TRANSPORT="ipc", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "LINK_DEBUG" not in out.stdout, out.stdout[-2000:]
    assert out.stdout.count("Speed: train:") == 1 and "update 40 loss" in out.stdout and "transport ipc" in out.stdout, out.stdout[-1500:]


def test_device_side_sharded_draw_equals_the_host_choreography():
    """hsad_replay_stats / _serve / _assemble / _update_owned (what dist.ReplayLink runs, no host round trip) against the host path
    they replace (priority_sum -> stratified_positions / split_positions -> sample_at -> update_priority): same owners, same
    elements, same rows, same weights, same shard state afterwards -- three shards (one empty), a bit-packed field unpacked to bf16"""
    from hanabi_sad_amd.dist import split_positions, stratified_positions
    from hanabi_sad_amd.replay import Bits, DeviceReplay
    B, W = 32, 3
    fields = [("s", D, torch.float32), ("a", 1, torch.int64), ("m", 70, Bits(1))]
    rng = np.random.default_rng(4)

    def rows(n, base):
        f, *rest = make_rows(rng, n, base)
        f["m"] = torch.tensor((rng.random((n, T, 70)) < 0.3).astype(np.float32), device=DEV)
        return (f,) + tuple(rest)
    dev_sh = [DeviceReplay(64, 5, 0.9, 0.6, 0, T, fields, DEV) for _ in range(W)]     # device-side path
    host_sh = [DeviceReplay(64, 5, 0.9, 0.6, 0, T, fields, DEV) for _ in range(W)]    # host choreography
    base = 0
    for k, n in enumerate([25, 0, 35]):
        if n:
            r = rows(n, base)
            dev_sh[k].add(*r)
            host_sh[k].add(*r)
            base += n
    for s in dev_sh + host_sh:
        s.set_field_output("m", "bf16", 128)
    wb = dev_sh[0].wire_bytes()
    cat = lambda parts, get, dim: torch.cat([get(p) for p in parts if get(p).shape[dim] > 0], dim)
    for it in range(3):
        canon = rng.random(B, dtype=np.float32)
        sums = [s.priority_sum()[0] for s in host_sh]
        pos = stratified_positions(canon, float(np.sum(np.asarray(sums, np.float64))), B)
        owner_h, local = split_positions(pos, sums)
        parts = [s.sample_at(local[owner_h == k]) for k, s in enumerate(host_sh)]
        all_stats = torch.stack([s.stats() for s in dev_sh])
        wire_all = torch.zeros(W, B, wb, dtype=torch.uint8, device=DEV)
        owners = [s.serve(torch.tensor(canon, device=DEV), all_stats, k, wire_all[k]) for k, s in enumerate(dev_sh)]
        assert all(torch.equal(o, owners[0]) for o in owners) and np.array_equal(owners[0].cpu().numpy(), owner_h)
        assert 0 < int((owner_h == 0).sum()) < B and not (owner_h == 1).any()           # both non-empty shards serve, the empty one never
        (f, reward, terminal, bootstrap, seq_len), raw_w = dev_sh[0].assemble(wire_all, owners[0])
        assert torch.equal(f["a"], cat(parts, lambda p: p[0][0]["a"], 1)) and torch.equal(f["s"], cat(parts, lambda p: p[0][0]["s"], 1))
        assert torch.equal(f["m"], cat(parts, lambda p: p[0][0]["m"], 1)) and f["m"].dtype == torch.bfloat16
        assert torch.equal(reward, cat(parts, lambda p: p[0][1], 1)) and torch.equal(terminal, cat(parts, lambda p: p[0][2], 1))
        assert torch.equal(bootstrap, cat(parts, lambda p: p[0][3], 1)) and torch.equal(seq_len, cat(parts, lambda p: p[0][4], 0))
        assert torch.equal(raw_w, cat(parts, lambda p: p[1], 0))
        newp = torch.tensor((rng.random(B) * 2 + 0.05).astype(np.float32), device=DEV)
        for k in range(W):
            host_sh[k].update_priority(newp[torch.tensor(owner_h == k, device=DEV)])
            dev_sh[k].answer(newp, k)
            assert host_sh[k].priority_sum() == dev_sh[k].priority_sum()
    for s in dev_sh + host_sh:
        s.check_errors()


def test_serve_from_older_statistics_stretches_the_share_and_scales_the_weights():
    """the star-shaped round of dist.ReplayLink cuts a draw with the (sum, size) every shard reported one round earlier; a shard that
    has pushed since must map its share [0, old sum) onto its present weights and send raw weights scaled by old / present sum, so
    that raw / total is the probability the element was drawn with.  Dyadic priorities, alpha = 1: every number is exact."""
    from hanabi_sad_amd.dist import split_positions, stratified_positions
    from hanabi_sad_amd.replay import DeviceReplay
    B, W = 32, 2
    rng = np.random.default_rng(21)
    shards = [DeviceReplay(96, 5, 1.0, 0.6, 0, T, FIELDS, DEV) for _ in range(W)]
    w = []
    for k, s in enumerate(shards):
        r = list(make_rows(rng, 30, 100 * k))
        prio = (rng.integers(1, 64, 30) / 16.0).astype(np.float32)
        r[5] = torch.tensor(prio, device=DEV)
        s.add(*r)
        w.append(prio)
    old_stats = torch.stack([s.stats() for s in shards])
    more = list(make_rows(rng, 20, 100 + 30))                                # shard 1 pushes 20 sequences after it reported
    prio = (rng.integers(1, 64, 20) / 16.0).astype(np.float32)
    more[5] = torch.tensor(prio, device=DEV)
    shards[1].add(*more)
    w[1] = np.concatenate([w[1], prio])
    wb = shards[0].wire_bytes()
    canon = rng.random(B, dtype=np.float32)
    wire_all = torch.zeros(W, B, wb, dtype=torch.uint8, device=DEV)
    owner = [s.serve(torch.tensor(canon, device=DEV), old_stats, k, wire_all[k]) for k, s in enumerate(shards)][0]
    sums_old = old_stats[:, 0].cpu().numpy()
    pos = stratified_positions(canon, float(sums_old.sum()), B)
    owner_h, _ = split_positions(pos, list(sums_old))
    assert np.array_equal(owner.cpu().numpy(), owner_h) and 0 < int((owner_h == 1).sum()) < B
    (f, *_), raw_w = shards[0].assemble(wire_all, owner)
    tags, raw = f["a"][0, :, 0].cpu().numpy(), raw_w.cpu().numpy()
    prefix = np.concatenate([[0.0], np.cumsum(sums_old)])
    saw_new = False
    for b in range(B):
        k = int(owner_h[b])
        now, told = float(w[k].astype(np.float64).sum()), float(sums_old[k])
        loc = max(float(pos[b]) - prefix[k], 0.0)
        if now != told:
            loc *= now / told
        target = min(max(np.float32(loc), np.float32(0)), np.float32(now) * (np.float32(1) - np.float32(1e-6)))
        i = int(np.searchsorted(np.cumsum(w[k].astype(np.float64)), np.float64(target), side="left"))
        assert tags[b] == (100 * k + i if k == 0 or i < 30 else 100 + i), (b, k, i, tags[b])
        assert raw[b] == np.float32(w[k][i] * np.float32(told / now)), (b, raw[b], w[k][i])
        saw_new |= k == 1 and i >= 30
    assert saw_new                                                           # the sequences pushed after the report can be drawn
    # the probabilities of shard 1's elements under (old share) x (present weights) sum to the old share
    p1 = (sums_old[1] / sums_old.sum()) * (w[1].astype(np.float64) / w[1].astype(np.float64).sum())
    assert abs(p1.sum() - sums_old[1] / sums_old.sum()) < 1e-12
    for s in shards:
        s.check_errors()


def test_late_priorities_reach_exactly_the_owned_elements():
    """two shards, depth 2: serve A, serve B, answer A, answer B -- the running sums follow a host model of the weights"""
    from hanabi_sad_amd.replay import DeviceReplay
    B, W = 16, 2
    rng = np.random.default_rng(9)
    shards = [DeviceReplay(64, 5, 1.0, 0.6, 0, T, FIELDS, DEV) for _ in range(W)]
    model = []
    for k, s in enumerate(shards):
        r = list(make_rows(rng, 30, 100 * k))
        prio = (rng.integers(1, 64, 30) / 16.0).astype(np.float32)
        r[5] = torch.tensor(prio, device=DEV)
        s.add(*r)
        s.set_outstanding(2)
        model.append(prio.astype(np.float64).copy())
    wb = shards[0].wire_bytes()
    pending = []
    for it in range(6):
        if len(pending) == 2:
            owner, ids, newp = pending.pop(0)
            for k, s in enumerate(shards):
                s.answer(torch.tensor(newp, device=DEV), k)
                for i, p in zip(ids[k], newp[owner == k]):
                    model[k][i] = p
            for k, s in enumerate(shards):
                assert s.priority_sum()[0] == float(model[k].sum()), (it, k)
        canon = torch.tensor(rng.random(B, dtype=np.float32), device=DEV)
        all_stats = torch.stack([s.stats() for s in shards])
        wire_all = torch.zeros(W, B, wb, dtype=torch.uint8, device=DEV)
        owner = [s.serve(canon, all_stats, k, wire_all[k]) for k, s in enumerate(shards)][0].cpu().numpy()
        ids = [s.last_ids(B).cpu().numpy()[:int((owner == k).sum())] for k, s in enumerate(shards)]
        (f, *_), raw_w = shards[0].assemble(wire_all, torch.tensor(owner, device=DEV, dtype=torch.int32))
        tags = f["a"][0, :, 0].cpu().numpy()                          # element tag = 100 * shard + index at add time (= ring slot)
        assert np.array_equal(tags, np.concatenate([100 * k + ids[k] for k in range(W)]))
        assert np.array_equal(raw_w.cpu().numpy(), np.concatenate([model[k][ids[k]] for k in range(W)]).astype(np.float32))
        pending.append((owner, ids, (rng.integers(1, 64, B) / 16.0).astype(np.float32)))
    for s in shards:
        s.check_errors()


@pytest.mark.parametrize("mode", ["star", "collective"])
def test_replay_link_over_rccl_single_rank_equals_plain_sampling(mode):
    """dist.ReplayLink with the RCCL backend ("nccl") on this box's one GPU: the learner is the only rank and serves its own shard.
    In a collective round every transport call (broadcast, all_gather_into_tensor, gather, parameter bucket) goes through RCCL on
    the exchange stream; a star round has no peer to talk to here and runs the learner's own serve -> answer -> statistics order.
    The assembled batches must equal what hsad_replay_sample draws from an identical replay with the same uniforms, and the late
    priorities (written back before the draw in a collective round, after it in a star round) must leave both replays in the
    same state."""
    import os
    import socket
    import torch.distributed as dist
    from hanabi_sad_amd.dist import ReplayLink
    from hanabi_sad_amd.replay import DeviceReplay
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        B, cap = 16, 48
        plain = DeviceReplay(cap, 11, 0.9, 0.6, 0, T, FIELDS, DEV)
        shard = DeviceReplay(cap, 11, 0.9, 0.6, 0, T, FIELDS, DEV)
        plain.set_outstanding(3 if mode == "star" else 2)
        rng = np.random.default_rng(0)
        rows = make_rows(rng, 40, 0)
        plain.add(*rows)
        shard.add(*rows)
        link = ReplayLink(shard, B, 0.6, DEV, learner_rank=0, depth=2, param_numel=64, mode=mode)
        prios, got = [], []
        for r in range(6):
            prio = prios[r - 2] if r >= 2 else None
            if r == 3:
                link.stage_params(torch.arange(64, dtype=torch.float32, device=DEV))
            link.begin(prio, params=(r == 3))
            if prio is not None and mode == "collective":
                plain.update_priority(prio)                 # answers the oldest outstanding draw, like the shard does
            x = plain.sample(B)
            if prio is not None and mode == "star":
                plain.update_priority(prio)
            y = link.finish()
            same_batch(x, y)
            prios.append(torch.tensor((rng.random(B) * 2 + 0.05).astype(np.float32), device=DEV))
        torch.cuda.synchronize()
        assert plain.priority_sum() == shard.priority_sum()
        assert torch.equal(link.bucket, torch.arange(64, dtype=torch.float32, device=DEV))
        t = link.timings()
        assert (t["exchange_ms"] >= 0 and t["serve_ms"] > 0) if mode == "star" else (t["batch_gather_ms"] > 0 and t["header_bcast_ms"] > 0), t
        plain.check_errors()
        shard.check_errors()
    finally:
        dist.destroy_process_group()


def test_replay_link_three_rounds_open_over_rccl_single_rank():
    """dist.ReplayLink(ahead=3) on the RCCL backend with the learner as the only rank: the ring of round slots, the down / up streams and
    their events, statistics `ahead` rounds old in the header, priorities `ahead` + 1 rounds after their draw -- everything but a peer.
    Dyadic priorities, alpha = 1: the shard's weight sum must equal a host model that applies every answered batch to the elements the
    assembled batches named; weights finite with maximum 1; the compute stream's wait for a batch is measured (timings)."""
    import socket
    import torch.distributed as dist
    from hanabi_sad_amd.dist import ReplayLink
    from hanabi_sad_amd.replay import DeviceReplay
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        B, AHEAD, U = 16, 3, 12
        rng = np.random.default_rng(2)
        shard = DeviceReplay(64, 5, 1.0, 0.6, 0, T, FIELDS, DEV)
        rows = list(make_rows(rng, 40, 0))
        w = (rng.integers(1, 64, 40) / 16.0).astype(np.float32)
        rows[5] = torch.tensor(w, device=DEV)
        shard.add(*rows)
        link = ReplayLink(shard, B, 0.6, DEV, learner_rank=0, param_numel=64, mode="star", ahead=AHEAD)
        assert link.ahead == AHEAD and len(link.hdrs) == AHEAD + 1
        link.stage_params(torch.arange(64, dtype=torch.float32, device=DEV))
        for i in range(AHEAD):
            link.begin(None, params=(i == 0))
        cur = link.finish()
        tags, prios, sent = [], [], []
        for u in range(U):
            p = prios.pop(0) if prios else None
            if p is not None:
                sent.append(p)
            link.begin(None if p is None else p[1])
            (f, *_), weight = cur
            t = f["a"][0, :, 0].cpu().numpy()
            wt = weight.cpu().numpy()
            assert np.all(np.isfinite(wt)) and np.all(wt > 0) and abs(float(wt.max()) - 1.0) < 1e-6 and t.min() >= 0 and t.max() < 40
            (torch.ones(2048, 2048, device=DEV) @ torch.ones(2048, 2048, device=DEV))          # "the update runs here"
            prios.append((t, torch.tensor((rng.integers(1, 64, B) / 16.0).astype(np.float32), device=DEV)))
            cur = link.finish()
        while link._rounds:
            link.finish()
        torch.cuda.synchronize()
        shard.check_errors()
        model = w.astype(np.float64).copy()
        for t, p in sent:
            for i, v in zip(t, p.cpu().numpy()):
                model[int(i)] = float(v)
        assert shard.priority_sum()[0] == float(model.sum())
        tm = link.timings()
        assert tm["rounds_ahead"] == AHEAD and tm["wait_for_batch_ms"] >= 0 and tm["serve_ms"] > 0 and "header_send_ms" in tm, tm
    finally:
        dist.destroy_process_group()


def test_hsad_comm_c_entry_points_single_rank():
    """hsad_comm_{unique_id, init, bcast_params, gather_batch, scatter_priority, star_round} (csrc/hsad_comm.hip: RCCL bound with dlopen, no
    torch.distributed) on the one GPU of this box: a world of one rank, every call through the real RCCL communicator; the drawn
    batch and the shard state afterwards equal the plain sampler's on an identical replay"""
    import ctypes as C
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.replay import DeviceReplay, _stream
    lib = _lib.load_library()
    uid = (C.c_char * 256)()
    _lib.check(lib.hsad_comm_unique_id(uid, 256))
    comm = C.c_void_p()
    _lib.check(lib.hsad_comm_init(uid, 0, 1, torch.cuda.current_device(), C.byref(comm)))
    try:
        assert lib.hsad_comm_rank(comm) == 0 and lib.hsad_comm_world(comm) == 1
        B, cap = 16, 48
        plain = DeviceReplay(cap, 11, 0.9, 0.6, 0, T, FIELDS, DEV)
        shard = DeviceReplay(cap, 11, 0.9, 0.6, 0, T, FIELDS, DEV)
        rng = np.random.default_rng(0)
        rows = make_rows(rng, 40, 0)
        plain.add(*rows)
        shard.add(*rows)
        st = _stream(torch.device(DEV))
        params = torch.arange(1000, dtype=torch.float32, device=DEV)
        _lib.check(lib.hsad_comm_bcast_params(comm, params.data_ptr(), 1000, 0, st))
        assert torch.equal(params, torch.arange(1000, dtype=torch.float32, device=DEV))
        wb = shard.wire_bytes()
        for it in range(3):
            canon = torch.tensor(shard.draw_canonical(B), device=DEV)
            owner = torch.empty(B, dtype=torch.int32, device=DEV)
            wire, wire_all = torch.zeros(B, wb, dtype=torch.uint8, device=DEV), torch.zeros(1, B, wb, dtype=torch.uint8, device=DEV)
            _lib.check(lib.hsad_comm_gather_batch(comm, shard.h, B, canon.data_ptr(), 0, owner.data_ptr(), wire.data_ptr(),
                                                  wire_all.data_ptr(), st))
            batch, raw_w = shard.assemble(wire_all, owner)
            x = plain.sample(B)
            (fx, rx, tx, bx, lx), wx = x
            assert torch.equal(batch[0]["a"], fx["a"]) and torch.equal(batch[0]["s"], fx["s"]) and torch.equal(batch[1], rx)
            assert torch.equal(batch[4], lx) and (owner == 0).all()
            newp = torch.tensor((rng.random(B) * 2 + 0.05).astype(np.float32), device=DEV)
            _lib.check(lib.hsad_comm_scatter_priority(comm, shard.h, B, newp.data_ptr(), owner.data_ptr(), 0, st))
            plain.update_priority(newp)
            assert plain.priority_sum() == shard.priority_sum()
        # the point-to-point round shape (hsad_comm_star_round): header = uniforms | late priorities | statistics; a rank serves
        # BEFORE it writes the late priorities back, so one more draw is outstanding (a world of one: the root's own part)
        plain.set_outstanding(3)
        shard.set_outstanding(3)
        PRIME, HAS_PRIO, PARAMS = 8, 4, 1
        hdr = torch.zeros(2 * B + 4, dtype=torch.float32, device=DEV)
        owners, prios = [], []
        for it in range(5):
            hdr[:B] = torch.tensor(shard.draw_canonical(B), device=DEV)
            flags = (PRIME if it == 0 else 0) | (PARAMS if it == 2 else 0)
            if it >= 2:
                hdr[B:2 * B] = prios[it - 2]
                flags |= HAS_PRIO
            owner = torch.empty(B, dtype=torch.int32, device=DEV)
            wire, wire_all = torch.zeros(B, wb, dtype=torch.uint8, device=DEV), torch.zeros(1, B, wb, dtype=torch.uint8, device=DEV)
            _lib.check(lib.hsad_comm_star_round(comm, shard.h, B, hdr.data_ptr(), flags, 0, owners[it - 2].data_ptr() if it >= 2 else None,
                                                owner.data_ptr(), wire.data_ptr(), wire_all.data_ptr(), params.data_ptr(), 1000, st))
            stats = hdr[2 * B:].view(torch.float64).cpu().numpy()
            batch, raw_w = shard.assemble(wire_all, owner)
            (fx, rx, tx, bx, lx), wx = plain.sample(B)
            if it >= 2:
                plain.update_priority(prios[it - 2])
            assert torch.equal(batch[0]["a"], fx["a"]) and torch.equal(batch[0]["s"], fx["s"]) and torch.equal(batch[1], rx)
            assert torch.equal(batch[4], lx) and (owner == 0).all() and stats[1] == 40.0
            assert plain.priority_sum() == shard.priority_sum()
            owners.append(owner)
            prios.append(torch.tensor((rng.random(B) * 2 + 0.05).astype(np.float32), device=DEV))
        assert torch.equal(params, torch.arange(1000, dtype=torch.float32, device=DEV))
        plain.check_errors()
        shard.check_errors()
    finally:
        lib.hsad_comm_destroy(comm)


def test_pipelined_star_round_halves_over_two_communicators_single_rank():
    """hsad_comm_star_open / _collect (the star round in two halves, for rounds that overlap: ReplayLink(ahead = 3) for a non-Python host)
    over TWO real RCCL communicators and two streams, three rounds open at once.  A world of one has no peers, so what runs is the
    root's own part: statistics of an EARLIER round in the header, own shard served from them (share stretched onto the present sum),
    late priorities written back four rounds after their draw.  Dyadic priorities, alpha = 1: the shard's weight sum must equal a host
    model that applies every answered batch to the elements the assembled batches named."""
    import ctypes as C
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.replay import DeviceReplay
    lib = _lib.load_library()
    comms = []
    for _ in range(2):
        uid = (C.c_char * 256)()
        _lib.check(lib.hsad_comm_unique_id(uid, 256))
        c = C.c_void_p()
        _lib.check(lib.hsad_comm_init(uid, 0, 1, torch.cuda.current_device(), C.byref(c)))
        comms.append(c)
    down, up = comms
    try:
        B, AHEAD, PRIME, HAS_PRIO = 16, 3, 8, 4
        S = AHEAD + 1
        rng = np.random.default_rng(4)
        shard = DeviceReplay(64, 5, 1.0, 0.6, 0, T, FIELDS, DEV)
        rows = list(make_rows(rng, 40, 0))
        w = (rng.integers(1, 64, 40) / 16.0).astype(np.float32)
        rows[5] = torch.tensor(w, device=DEV)
        shard.add(*rows)
        shard.set_outstanding(AHEAD + 2)
        model = w.astype(np.float64).copy()
        wb = shard.wire_bytes()
        sd, su = torch.cuda.Stream(), torch.cuda.Stream()
        hdrs = [torch.zeros(2 * B + 4, dtype=torch.float32, device=DEV) for _ in range(S)]
        wires = [torch.zeros(1, B, wb, dtype=torch.uint8, device=DEV) for _ in range(S)]
        replies = [torch.zeros(1, 2, dtype=torch.float64, device=DEV) for _ in range(S)]
        prime = torch.zeros(1, 2, dtype=torch.float64, device=DEV)
        owners, prios, tags, done, sums_at_reply = {}, {}, {}, {}, {}
        collected = 0
        torch.cuda.synchronize()

        def open_round(r, answered):
            k = r % S
            hdrs[k][:B] = torch.tensor(shard.draw_canonical(B), device=DEV)
            flags = PRIME if r == 0 else 0
            if answered is not None:
                hdrs[k][B:2 * B] = prios[answered]
                flags |= HAS_PRIO
            sd.wait_stream(torch.cuda.current_stream())
            known = replies[(collected - 1) % S].data_ptr() if collected else prime.data_ptr()
            owners[r] = torch.empty(B, dtype=torch.int32, device=DEV)
            _lib.check(lib.hsad_comm_star_open(down, up, shard.h, B, hdrs[k].data_ptr(), flags, None if r == 0 else known, prime.data_ptr(),
                                               None, 0, sd.cuda_stream, su.cuda_stream))
            _lib.check(lib.hsad_comm_star_collect(down, up, shard.h, B, hdrs[k].data_ptr(), flags,
                                                  owners[answered].data_ptr() if answered is not None else None, owners[r].data_ptr(),
                                                  wires[k].data_ptr(), replies[k].data_ptr(), sd.cuda_stream, su.cuda_stream))
            done[r] = torch.cuda.Event()
            done[r].record(su)

        for r in range(AHEAD):
            open_round(r, None)
        for u in range(8):
            torch.cuda.current_stream().wait_event(done[u])
            (f, *_), raw_w = shard.assemble(wires[u % S], owners[u])
            tags[u] = f["a"][0, :, 0].cpu().numpy()
            assert (owners[u] == 0).all() and tags[u].min() >= 0 and tags[u].max() < 40
            st = hdrs[u % S][2 * B:].view(torch.float64).cpu().numpy()
            assert st[1] == 40.0 and st[0] > 0                                   # the header carried a shard's (sum, size)
            sums_at_reply[u] = float(replies[u % S][0, 0])
            collected = u + 1
            prios[u] = torch.tensor((rng.integers(1, 64, B) / 16.0).astype(np.float32), device=DEV)
            open_round(u + AHEAD, u - 1 if u >= 1 else None)                      # batch u - 1's priorities leave with round u + AHEAD
        torch.cuda.synchronize()
        shard.check_errors()
        # answered: batches 0 .. 6, in that order, each AFTER the draw of the round that carried it
        for b in range(7):
            for i, p in zip(tags[b], prios[b].cpu().numpy()):
                model[int(i)] = float(p)
        assert shard.priority_sum()[0] == float(model.sum())
        # the statistics a reply reports are the shard's sum at that moment: round r answered batch r - AHEAD - 1
        m = w.astype(np.float64).copy()
        for r in range(8):
            b = r - AHEAD - 1
            if b >= 0:
                for i, p in zip(tags[b], prios[b].cpu().numpy()):
                    m[int(i)] = float(p)
            assert sums_at_reply[r] == float(m.sum()), r
    finally:
        for c in comms:
            lib.hsad_comm_destroy(c)
