"""World-size-2/3 gloo tests of the asynchronous learner <-> actor exchange (hanabi_sad_amd/dist.py ReplayLink): rounds opened by the
learner and noticed by polling actors, one packed buffer per rank, priorities piggybacked two rounds late and answered oldest-first,
parameters as one flat bucket -- and the batches must still be what ONE PrioritizedReplay over the concatenated shards would draw
from the same uniforms given the same (late) priority write-backs (rela/prioritized_replay.h:291-345).  The shard is a host stand-in
with the DeviceReplay shard interface; the device kernels behind that interface are covered by tests/test_sharded_replay_gpu.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from hanabi_sad_amd.dist import ReplayLink, rank_world, split_positions, stratified_positions

ALPHA, BETA, B, ROUNDS = 0.9, 0.6, 16, 7
SIZES = %(sizes)r                     # elements per shard; the learner's (rank 0) may be empty: a dedicated learner
MODE, GROW = %(mode)r, %(grow)r       # shape of a round; elements an actor shard pushes between two rounds (star only: older statistics)

def initial(k):
    return (np.random.default_rng(100 + k).random(SIZES[k]).astype(np.float32) * 2 + 0.05) ** np.float32(ALPHA)

class HostShard:
    """the DeviceReplay shard interface on the host: element j of shard k carries the tag 1000 * k + j"""
    def __init__(self, k):
        self.k, self.w, self.rng = k, initial(k), np.random.default_rng(7)
        self.queue, self.depth = [], 1
    def set_outstanding(self, depth): self.depth = depth
    def draw_canonical(self, n): return self.rng.random(n, dtype=np.float32)
    def wire_bytes(self): return 16
    def stats(self): return torch.tensor([float(np.sum(self.w.astype(np.float64))), float(len(self.w))], dtype=torch.float64)
    def serve(self, canon, all_stats, rank, wire_out):
        sums = all_stats[:, 0].numpy()
        pos = stratified_positions(canon.numpy(), float(np.sum(sums)), len(canon))
        owner, local = split_positions(pos, list(sums))
        now, told = float(np.sum(self.w.astype(np.float64))), float(sums[rank])
        scale = np.float32(told / now) if told > 0 and now > 0 else np.float32(1)     # hsad_replay_serve: older statistics
        if told > 0 and now > 0 and told != now:
            local = (local.astype(np.float64) * (now / told)).astype(np.float32)
        acc = np.cumsum(self.w.astype(np.float64))
        ids = [int(min(np.searchsorted(acc, np.float64(t), side="left"), len(self.w) - 1)) for t in local[owner == rank]]
        assert len(self.queue) < self.depth
        self.queue.append((owner, ids))
        slots = wire_out.view(torch.float32).view(-1, 4)
        for j, i in enumerate(ids):
            slots[j, 0], slots[j, 1] = 1000.0 * self.k + i, float(self.w[i] * scale)
        return torch.from_numpy(owner.astype(np.int32))
    def answer(self, prio, rank):
        owner, ids = self.queue.pop(0)
        for i, p in zip(ids, prio.numpy()[owner == rank]):
            self.w[i] = np.float32(p) ** np.float32(ALPHA)
    def assemble(self, wire_all, owner):
        owner = owner.numpy()
        slots = wire_all.view(torch.float32).view(wire_all.shape[0], -1, 4)
        tags, raw = [], []
        for b in range(len(owner)):
            j = b - int(np.argmax(owner == owner[b]))
            tags.append(float(slots[owner[b], j, 0])); raw.append(float(slots[owner[b], j, 1]))
        tags = torch.tensor(tags)
        return ({"tag": tags}, None, None, None, tags.clone()), torch.tensor(raw)

rank, world = rank_world()
dist.init_process_group("gloo", rank=rank, world_size=world)
NP = 1000
link = ReplayLink(HostShard(rank), B, BETA, "cpu", learner_rank=0, depth=2, param_numel=NP, mode=MODE)
assert link.mode == MODE and link.shard.depth == (3 if MODE == "star" else 2)
link.FLAG_SLOTS = 4                                        # fewer flag keys than rounds: the ring of store keys wraps in this test
if rank == 0:
    model = [initial(k) for k in range(world)]            # the single-buffer emulation: every shard's weights, updated when the priorities ARRIVE
    rng = np.random.default_rng(7)
    drawn, batches = [], []
    for r in range(ROUNDS):
        prio = None
        def write_back():
            for t, p in zip(drawn[r - 2][0], drawn[r - 2][1]):        # sequential: the last duplicate wins
                model[int(t) // 1000][int(t) %% 1000] = np.float32(p) ** np.float32(ALPHA)
        if r >= 2:                                         # update r-2 is the newest finished one when round r is opened
            prio = torch.tensor(drawn[r - 2][1])
            if MODE == "collective":                       # ... and is written back before the draw there, after it in a star round
                write_back()
        params = r == 3
        if params:
            link.stage_params(torch.arange(NP // 2, dtype=torch.float32), torch.arange(NP // 2, dtype=torch.float32) + 0.5)
        link.begin(prio, params=params, stop=(r == ROUNDS - 1))
        time.sleep(0.02)                                   # "update r-1 runs here"
        (f, *_ , seq_len), weight = link.finish()
        if GROW:                                           # shards that push between rounds: properties only (the exact rescaling
            got = f["tag"].numpy()                         # is pinned on the device kernels, tests/test_sharded_replay_gpu.py)
            rng.random(B, dtype=np.float32)
            assert all(0 <= int(t) %% 1000 < SIZES[int(t) // 1000] + GROW * (r + 1) and SIZES[int(t) // 1000] > 0 for t in got), got
            w = weight.numpy()
            assert np.all(np.isfinite(w)) and np.all(w > 0) and abs(float(w.max()) - 1.0) < 1e-6
            drawn.append((got, (got %% 7 + 0.5 + r).astype(np.float32)))
            continue
        # what ONE buffer over the concatenation draws from the same uniforms
        cat = np.concatenate(model)
        tags = np.concatenate([1000 * k + np.arange(len(model[k])) for k in range(world)])
        canon = rng.random(B, dtype=np.float32)
        total = float(np.sum(np.concatenate([w.astype(np.float64) for w in model])))
        pos = stratified_positions(canon, total, B)
        acc = np.cumsum(cat.astype(np.float64))
        want = np.minimum(np.searchsorted(acc, pos.astype(np.float64), side="left"), len(cat) - 1)
        got = f["tag"].numpy()
        gap = np.minimum(np.abs(acc[want] - pos), np.abs(pos - np.where(want > 0, acc[want - 1], 0)))
        clear = gap > 1e-4 * max(total, 1.0)               # float32 rounding at an element boundary may pick the neighbour
        assert clear.sum() >= B - 2 and np.array_equal(got[clear], tags[want][clear].astype(np.float32)), (r, got, tags[want])
        raw = np.array([model[int(t) // 1000][int(t) %% 1000] for t in got], dtype=np.float32)
        y = (np.float32(len(cat)) * (raw / np.float32(total))) ** np.float32(-BETA)
        assert np.allclose(weight.numpy(), y / y.max(), rtol=1e-5), (r, weight, y / y.max())
        if r >= 2 and MODE == "star":
            write_back()
        drawn.append((got, (got %% 7 + 0.5 + r).astype(np.float32)))
    t = link.timings()
    keys = {"serve_ms", "exchange_ms", "param_send_ms", "assemble_ms"} if MODE == "star" else \
        {"header_bcast_ms", "stats_allgather_ms", "serve_ms", "batch_gather_ms", "param_bcast_ms", "assemble_ms"}
    assert set(t) >= keys, t
    final = torch.from_numpy(np.concatenate(model))
else:
    n_poll = n_act = 0
    got_params = False
    while True:
        flags = link.poll()
        n_poll += 1
        if flags is None:
            n_act += 1                                      # an actor step would run here; it never blocks on the learner
            time.sleep(0.001)
            continue
        stop = link.serve(flags)
        if GROW and link.shard.w.size:                     # "the actor pushed GROW sequences since"
            link.shard.w = np.concatenate([link.shard.w, (np.arange(GROW, dtype=np.float32) + 1.0 + link.served) / 8])
        if flags & ReplayLink.PARAMS:
            assert torch.equal(link.bucket[:NP // 2], torch.arange(NP // 2, dtype=torch.float32))
            assert torch.equal(link.bucket[NP // 2:], torch.arange(NP // 2, dtype=torch.float32) + 0.5)
            got_params = True
        if stop:
            break
    assert link.served == ROUNDS and got_params and n_act > ROUNDS   # it kept "acting" between the rounds
    final = torch.empty(sum(SIZES), dtype=torch.float32)
dist.broadcast(final, src=0)
lo = sum(SIZES[:rank])
# the late priorities reached exactly the owning elements of every shard (rounds 0 .. ROUNDS-3 were answered)
assert GROW or np.array_equal(link.shard.w, final.numpy()[lo:lo + SIZES[rank]]), rank
assert len(link.shard.queue) == 2                           # the last two draws are still waiting for theirs
dist.barrier()
dist.destroy_process_group()
open(os.path.join(%(out)r, "link%%d.ok" %% rank), "w").write("ok")
'''


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("sizes,mode,grow", [([0, 40], "star", 0), ([12, 0, 50], "star", 0), ([0, 30, 25], "star", 0),
                                             ([0, 30, 25], "star", 3), ([0, 40], "collective", 0), ([12, 0, 50], "collective", 0)])
def test_replay_link_rounds_gloo(tmp_path, sizes, mode, grow):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "sizes": sizes, "out": str(tmp_path), "mode": mode, "grow": grow})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % len(sizes), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert all((tmp_path / ("link%d.ok" % r)).exists() for r in range(len(sizes)))


PIPE_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from hanabi_sad_amd.dist import ReplayLink, rank_world, split_positions, stratified_positions
exec(open(%(host_shard)r).read())                     # ALPHA, BETA, B, SIZES, initial(), HostShard: the stand-in shard of the test above

AHEAD, UPDATES, UPDATE_S, DELAY_S = %(ahead)d, 40, 0.002, 0.003
rank, world = rank_world()
dist.init_process_group("gloo", rank=rank, world_size=world)
NP = 64
link = ReplayLink(HostShard(rank), B, BETA, "cpu", learner_rank=0, param_numel=NP, mode="star", ahead=AHEAD)
assert link.ahead == AHEAD and link.shard.depth == AHEAD + 2       # drawn batches a shard keeps: `ahead` + 1 in flight + the one served before the write-back
if rank == 0:
    model = [initial(k) for k in range(world)]
    sent, prios, answered = [], [], 0
    link.stage_params(torch.arange(NP // 2, dtype=torch.float32), torch.arange(NP // 2, dtype=torch.float32) + 0.5)
    for i in range(AHEAD):                             # selfplay.run_link_learner: `ahead` rounds are opened before the first update
        link.begin(None, params=(i == 0))
    cur = link.finish()
    link.wait_ms, link.wait_n = 0.0, 0                 # (the very first batch has nothing to hide behind)
    for u in range(UPDATES):
        p = prios.pop(0) if prios else None
        if p is not None:
            sent.append(p)
        link.begin(None if p is None else torch.tensor(p[1]), stop=False)
        (f, *_), weight = cur
        got = f["tag"].numpy()
        w = weight.numpy()
        assert np.all(np.isfinite(w)) and np.all(w > 0) and abs(float(w.max()) - 1.0) < 1e-6
        assert all(0 <= int(t) %% 1000 < SIZES[int(t) // 1000] for t in got), got
        time.sleep(UPDATE_S)                           # "update u runs here"
        prios.append((got, (got %% 7 + 0.5 + u).astype(np.float32)))
        cur = link.finish()
    wait = link.timings()["wait_for_batch_ms"]
    while link._rounds:                                # the rounds still open are collected, then everybody stops
        link.finish()
    link.begin(None, stop=True)
    link.finish()
    for tags, pr in sent:                              # what the shards must hold now: every priority that was SENT, in sending order
        for t, p in zip(tags, pr):
            model[int(t) // 1000][int(t) %% 1000] = np.float32(p) ** np.float32(ALPHA)
    final = torch.from_numpy(np.concatenate(model))
    open(os.path.join(%(out)r, "wait_ahead%%d.txt" %% AHEAD), "w").write("%%f" %% wait)
else:
    notice, n_step = [], 0
    while True:
        link.poll()
        now = time.perf_counter()
        while len(notice) < getattr(link, "_known_open", 0):
            notice.append(now)                          # a round is NOTICED now and answered DELAY_S later: pure latency (queued steps in
        if link.served < len(notice) and now >= notice[link.served] + DELAY_S:     # front of it), rounds behind it are served back to back
            flags = link.poll()
            assert flags is not None
            if link.serve(flags):
                break
            continue
        n_step += 1
        time.sleep(0.0002)                              # an actor step would run here
    assert n_step > UPDATES
    final = torch.empty(sum(SIZES), dtype=torch.float32)
dist.broadcast(final, src=0)
lo = sum(SIZES[:rank])
assert np.array_equal(link.shard.w, final.numpy()[lo:lo + SIZES[rank]]), rank     # late priorities landed on exactly the owned elements
dist.barrier()
dist.destroy_process_group()
open(os.path.join(%(out)r, "pipe%%d.ok" %% rank), "w").write("ok")
'''


def test_rounds_opened_three_updates_ahead_hide_the_actors_reply_latency(tmp_path):
    """an actor answers a round some milliseconds after it was opened (a notice + the steps queued in front of it)
    while an update takes less.  Three ranks over gloo, actors that answer every round 3 ms late, 2 ms updates: with rounds opened ONE
    update ahead the learner waits for rows every round; opened THREE ahead (the reference's prefetch depth) the wait disappears --
    and every late priority still reaches exactly the element it was drawn from (priorities now travel four rounds after their draw,
    shards keep five drawn batches)."""
    sizes = [0, 30, 25]
    shard_src = tmp_path / "host_shard.py"
    head = WORKER.split("rank, world = rank_world()")[0].split("ALPHA, BETA, B, ROUNDS")[1]
    shard_src.write_text(("ALPHA, BETA, B, ROUNDS" + head) % {"sizes": sizes, "mode": "star", "grow": 0, "root": ROOT, "out": str(tmp_path)})
    def timing_ok(w):
        # one round ahead: ~ latency - update >= 1 ms of waiting per round; three ahead: the reply has been in for a whole update
        return w[1] > 0.6 and w[3] < 0.5 * w[1] and w[3] < 0.8      # (three processes on shared host cores: margins, not measurements)
    tried = []
    for attempt in range(4):      # the functional checks hold in every attempt; the timing margins in at least one (a shared host has bad minutes)
        waits = {}
        for ahead in (1, 3):
            script = tmp_path / ("pipe%d.py" % ahead)
            script.write_text(PIPE_WORKER % {"root": ROOT, "out": str(tmp_path), "ahead": ahead, "host_shard": str(shard_src)})
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), str(script)]
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
            assert all((tmp_path / ("pipe%d.ok" % r)).exists() for r in range(3))
            for r in range(3):
                (tmp_path / ("pipe%d.ok" % r)).unlink()
            waits[ahead] = float((tmp_path / ("wait_ahead%d.txt" % ahead)).read_text())
        tried.append(waits)
        if timing_ok(waits):
            return
    # four attempts outside the margins: is it the host?  1 ms sleeps that overshoot by more than 1 ms mean that the 2 ms "updates" and 3 ms
    # "latencies" of the workers are not what they say either
    import time
    over = []
    for _ in range(200):
        t0 = time.perf_counter()
        time.sleep(0.001)
        over.append((time.perf_counter() - t0 - 0.001) * 1e3)
    over.sort()
    if over[180] > 1.0:
        pytest.skip("host scheduling noise (1 ms sleeps overshoot by %.1f ms at the 90th percentile): timing margins not checked; waits %s" % (over[180], tried))
    assert False, tried
