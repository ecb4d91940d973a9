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


def _default_store():
    """the rendezvous store of the default process group (a TCPStore under torchrun): the host-side signalling channel of ReplayLink"""
    from torch.distributed import distributed_c10d as c10d
    return c10d._get_default_store()


class _Timer:
    """per-section time of the learner's side of a round: HIP events on the exchange stream (read one round late, so reading never
    waits) or perf_counter on the host (gloo / CPU tests)"""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.ms, self.n, self.pending = {}, 0, []
        self.marks = []

    def start(self):
        """-> the marks of a new timed section (pass it to mark / stop when several sections are open at once: rounds in flight)"""
        self._flush()
        self.marks = []
        self.mark(None)
        return self.marks

    def mark(self, name, marks=None):
        marks = self.marks if marks is None else marks
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
        else:
            import time
            marks.append((name, time.perf_counter()))

    def stop(self, marks=None):
        own = marks is None
        self.pending.append(self.marks if own else marks)      # read later: a round still in flight is kept, not dropped
        if own:
            self.marks = []

    def _flush(self):
        while self.pending:
            marks = self.pending[0]
            if self.cuda and not marks[-1][1].query():
                return                       # still in flight: keep it (and the rounds behind it) for the next call
            for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
                dt = a.elapsed_time(b) if self.cuda else (b - a) * 1e3
                self.ms[name] = self.ms.get(name, 0.0) + dt
            self.n += 1
            self.pending.pop(0)

    def summary(self):
        self._flush()
        return {k: v / max(self.n, 1) for k, v in self.ms.items()}


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _IpcTransport:
    """One-sided point-to-point messages between the ranks of ONE node without a communication kernel: every rank owns landing rings in
    its own device memory (hsad_ipc_alloc), exported by IPC handle through the rendezvous store; a SEND is a plain device-to-device copy
    from the sender's stream into the receiver's ring slot (SDMA over xGMI between GPUs) followed, once the copy has finished, by a key in
    the store (a helper thread waits for the copy's event); a RECEIVE is the host waiting for that key and one local copy out of the slot.
    Why not RCCL here: a posted receive is a kernel that stays resident on CUs until its peer sends, and the learner's fused recurrences
    need every CU (tools/resident_probe.py: a 2.5 ms wait next to an update stretches it 1.5-2 x).  Messages between a pair are matched
    in order, like RCCL's; a ring of `depth` slots per (source, destination, direction) bounds the sender's lead (the star round's own
    causality keeps it far below that: a slot is reused `depth` messages later, after at least one full round trip)."""

    def __init__(self, store, keys, rank, world, learner, device, down_bytes, up_bytes, depth):
        import ctypes as C
        import queue
        import threading
        from . import _lib
        from .composite import _view
        self.lib, self.store, self.keys = _lib.load_library(), store, keys + "ipc/"
        self.rank, self.world, self.learner, self.device, self.depth = rank, world, learner, torch.device(device), int(depth)
        align = lambda n: (int(n) + 255) // 256 * 256
        self.slot = {"down": align(down_bytes), "up": align(up_bytes)}
        # what this rank receives: the learner one "up" ring per actor, an actor one "down" ring from the learner
        self.sources = [p for p in range(world) if p != learner] if rank == learner else [learner]
        self.tag_in = "up" if rank == learner else "down"
        nbytes = len(self.sources) * self.depth * self.slot[self.tag_in]
        hb = self.lib.hsad_ipc_handle_bytes()
        handle, ptr = (C.c_ubyte * hb)(), C.c_void_p()
        # Every rank must end up on the same transport: allocate + export, open the peers' arenas, then agree through the store -- if any
        # rank failed at either step (no dmabuf IPC, no peer access), ALL ranks give up (self.ok = False) and the link uses the backend
        self.base, self.peer_base, self.ok, why = None, {}, True, ""
        rc = self.lib.hsad_ipc_alloc(nbytes, C.byref(ptr), handle, hb)
        if rc:
            why = self.lib.hsad_last_error().decode()
            store.set(self.keys + "handle/%d" % rank, b"")
        else:
            self.base, self.nbytes = ptr.value, nbytes
            self.mine = _view(self.base, nbytes, self.device, self, dtype=torch.uint8)
            store.set(self.keys + "handle/%d" % rank, bytes(handle))
        mine_ok = rc == 0
        for p in ([k for k in range(world) if k != learner] if rank == learner else [learner]):
            h = store.get(self.keys + "handle/%d" % p)
            if len(h) != hb:
                mine_ok = False
                continue
            q = C.c_void_p()
            if self.lib.hsad_ipc_open((C.c_ubyte * hb).from_buffer_copy(h), C.byref(q)):
                mine_ok, why = False, self.lib.hsad_last_error().decode()
            else:
                self.peer_base[p] = q.value
        store.set(self.keys + "ok/%d" % rank, b"1" if mine_ok else ("0 " + why).encode())
        votes = [store.get(self.keys + "ok/%d" % p) for p in range(world)]
        self.ok = all(v == b"1" for v in votes)
        self.sent, self.got = {}, {}              # messages sent to / consumed from a peer so far
        if not self.ok:
            self.why = "; ".join("rank %d: %s" % (p, v.decode()[2:]) for p, v in enumerate(votes) if v != b"1")
            for q in self.peer_base.values():
                self.lib.hsad_ipc_close(C.c_void_p(q))
            if self.base:
                self.lib.hsad_ipc_free(C.c_void_p(self.base))
            self.q = None
            return
        self.q = queue.Queue()
        self.err = None
        self.thread = threading.Thread(target=self._notify, daemon=True)
        self.thread.start()

    def _notify(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            ev, key, _keep = item
            try:
                ev.synchronize()
                self.store.set(key, b"1")
            except Exception as e:       # surfaced by the next send / recv of the owning thread
                self.err = e
                return

    def _ring_index(self, owner, src):
        """position of `src`'s ring inside `owner`'s arena"""
        if owner == self.learner:
            return [p for p in range(self.world) if p != self.learner].index(src)
        return 0

    def send(self, t, peer):
        import ctypes as C
        from . import _lib
        if self.err is not None:
            raise self.err
        tag = "down" if self.rank == self.learner else "up"
        n = self.sent.get(peer, 0)
        self.sent[peer] = n + 1
        nb = t.numel() * t.element_size()
        assert t.is_contiguous() and nb <= self.slot[tag], (nb, self.slot[tag])
        dst = self.peer_base[peer] + (self._ring_index(peer, self.rank) * self.depth + n % self.depth) * self.slot[tag]
        st = torch.cuda.current_stream(self.device)
        _lib.check(self.lib.hsad_ipc_put(C.c_void_p(dst), C.c_void_p(t.data_ptr()), nb, C.c_void_p(st.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(st)
        self.q.put((ev, self.keys + "m/%d/%d/%d" % (self.rank, peer, n), t))
        return None

    def recv(self, t, peer):
        n = self.got.get(peer, 0)
        self.got[peer] = n + 1
        return (t, peer, n)

    def wait(self, work):
        """the host waits for the message, then enqueues the copy out of the landing slot on the current stream"""
        if work is None:
            return
        if self.err is not None:
            raise self.err
        t, peer, n = work
        key = self.keys + "m/%d/%d/%d" % (peer, self.rank, n)
        # polled, not store.wait(): a blocking wait holds the store client's lock, and this process's notifier thread needs the same client
        # to announce the message the peer is waiting for before it sends this one
        import time
        deadline = time.monotonic() + 300.0
        while not self.store.check([key]):
            if self.err is not None:
                raise self.err
            if time.monotonic() > deadline:
                raise RuntimeError("ReplayLink ipc transport: rank %d waited 300 s for message %d of rank %d" % (self.rank, n, peer))
            time.sleep(5e-5)
        self.store.delete_key(key)
        nb = t.numel() * t.element_size()
        off = (self._ring_index(self.rank, peer) * self.depth + n % self.depth) * self.slot[self.tag_in]
        t.view(torch.uint8).reshape(-1).copy_(self.mine[off:off + nb]) if t.dtype != torch.uint8 else t.reshape(-1).copy_(self.mine[off:off + nb])

    def close(self):
        """stop the notifier, then (all ranks, behind a store barrier: nobody unmaps an arena a peer may still be copying into) close the
        peers' mappings and free the own landing arena"""
        import ctypes as C
        if self.q is None:
            return
        self.q.put(None)
        self.thread.join(timeout=10.0)
        self.q = None
        try:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self.store.set(self.keys + "closed/%d" % self.rank, b"1")
            import time
            deadline = time.monotonic() + 10.0
            while not self.store.check([self.keys + "closed/%d" % p for p in range(self.world)]) and time.monotonic() < deadline:
                time.sleep(1e-3)
        except Exception:          # a peer (or the store's owner) is already gone: unmapping is all that is left to do
            pass
        for q in self.peer_base.values():
            self.lib.hsad_ipc_close(C.c_void_p(q))
        self.peer_base = {}
        if self.base:
            self.mine = None
            self.lib.hsad_ipc_free(C.c_void_p(self.base))
            self.base = None


class ReplayLink:
    """The learner <-> actors exchange of a multi-GPU job, asynchronous and packed (the reference's
    PrioritizedReplay::sample / updatePriority + BatchRunner::updateModel across processes: rela/prioritized_replay.h:208-257,
    rela/batch_runner.h:74-77).  One prioritized-replay shard per rank behaves, for the learner, like ONE buffer over the
    concatenation of the shards, but

      * actors never wait for the learner's compute.  A round is opened by the learner BEFORE it issues the kernels of the update
        in flight (`begin`), and runs on a side stream there; actor ranks notice it by polling a counter in the rendezvous store
        between two of their own steps (`poll`) and serve it from their stream (`serve`) -- one receive (the header), two small
        kernels and one send in the default point-to-point shape (see "Two shapes of a round" below).
      * the priorities of batch k travel in the header of a LATER round (k + 2 when rounds are pipelined like that), next to the
        canonical uniforms of the new draw: no separate scatter, no message per shard.  Shards keep their drawn batches in a
        queue and answer the oldest (hsad_replay_set_outstanding = the reference's prefetch depth: its prefetched batches are
        drawn before the priorities of the batches in training are written back, too).
      * nothing on an actor rank reads device memory from the host: shard statistics, stratification, ownership, the draw and the
        packing of the rows are kernels (hsad_replay_stats / _serve / _update_owned); message sizes are fixed ([B] slots of
        hsad_replay_wire_bytes: stored rows with the bit-packed observation, ~13 KB per sequence), so no size ever has to be known.
      * every rank sends ONE buffer per round (one grouped send / recv in RCCL at the learner); the learner unpacks all
        of them with one kernel (hsad_replay_assemble) straight into the batch tensors (bf16 observation operand included).
      * parameters go out as one persistent flat bucket [online | target], staged by the learner on its compute stream
        (`stage_params`) and sent inside a round whose flags say so.
      * an actor keeps its host at most a few steps ahead of its device (hsad_actor_set_run_ahead, selfplay.run_link_actor):
        a round is stream-ordered behind whatever the actor has queued.

    Two shapes of a round (`mode`, default "star"; HSAD_LINK_MODE overrides):

      "star"        every message is point-to-point between the learner and ONE actor, so no actor ever waits for another actor:
                    learner -> actor   header [canonical uniforms | late priorities | (sum, size) of EVERY shard]
                    actor -> learner   its rows of the draw, then its (sum, size) after this round
                    learner -> actor   the parameter bucket, in a PARAMS round
                    The statistics the draw is cut with are the ones the PREVIOUS replies carried (the very first round collects
                    them up front): a shard that has pushed sequences since stretches its share of the positions onto its present
                    weight sum and scales the raw weights it sends, so that raw / (sum of the header's sums) is still the
                    probability the sequence was drawn with (hsad_replay_serve); a shard nobody pushed to is served bit for bit
                    as the collective round serves it.  An actor serves the draw BEFORE it writes the late priorities back (the
                    header's statistics describe the shard before them), so one more drawn batch is outstanding per shard.
                    7 x 1.7 MB arrive at the learner over 7 xGMI links at once; the 39 MB bucket leaves over the same 7 links.
      "collective"  header broadcast, all-gather of the statistics, gather of the rows, parameter broadcast over the whole world
                    on every actor's stream: the statistics are fresh, but an actor that notices the round early has its stream
                    wait inside the first collective for the last one to join (kept as the A/B twin).

    `shard`: hanabi_sad_amd.replay.DeviceReplay, or any stand-in with stats / wire_bytes / serve / answer / assemble /
    draw_canonical / set_outstanding (the gloo CPU tests)."""

    ROUND_KEY = "hsad/link/round"
    FLAG_SLOTS = 64        # flags of round r live in key r % 64: the learner's host is never more than four rounds ahead of its own
    PARAMS, STOP, HAS_PRIO, PRIME = 1, 2, 4, 8   # exchange stream (hdr_ev below) and that stream cannot pass a round an actor has not served

    def __init__(self, shard, batch, beta, device, learner_rank=0, depth=2, param_numel=0, store=None, mode=None, name="", ahead=1, transport=None):
        import os
        import torch.distributed as dist
        self._keys = "hsad/link/%s" % (name + "/" if name else "")     # a second link of the same job (bench.py's A/B) has its own keys
        self.ROUND_KEY = self._keys + "round"
        self.mode = mode or os.environ.get("HSAD_LINK_MODE", "star")
        if self.mode not in ("star", "collective"):
            raise ValueError("ReplayLink mode must be 'star' or 'collective', not %r" % (self.mode,))
        self.shard, self.B, self.beta = shard, int(batch), float(beta)
        self.device = torch.device(device)
        self.learner = int(learner_rank)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.is_learner = self.rank == self.learner
        self.comm = comm_device_for(self.device)
        self.staged = self.comm != self.device           # gloo with GPU shards (smoke runs): tensors travel through host memory
        self.store = store if store is not None else _default_store()
        star = self.mode == "star"
        # `ahead` (star only): how many rounds the learner keeps open at once.  With 1, round u+1 is opened when update u is issued (rounds
        # 1-3); an actor answers a round 1.7-2.6 ms after it was opened while an update takes 1.4 ms, so the learner would wait for rows
        # every round.  With `ahead` = 3 (the reference's prefetch depth, rela/prioritized_replay.h:229-237, selfplay.py prefetch = 3) the
        # batch of update u was requested three updates earlier.  The priorities of a batch then travel `ahead` + 1 rounds after its draw
        # and every shard keeps that many drawn batches (+ the one served before the write-back) outstanding.
        self.ahead = max(1, int(ahead)) if star else 1
        self.lag = self.ahead + 1 if star else int(depth)
        if star and self.ahead == 1:
            self.lag = int(depth)
        shard.set_outstanding(self.lag + 1 if star else self.lag)
        B, wb, d = self.B, shard.wire_bytes(), self.device
        S = self.ahead + 1 if (star and self.is_learner) else 1       # round slots of the learner: a ring, one per round in flight (+ 1)
        if star:     # canonical uniforms | priorities of an earlier batch | every shard's (sum, size) as float64 pairs
            self.hdrs = [torch.zeros(2 * B + 4 * self.world, dtype=torch.float32, device=d) for _ in range(S)]
            self.stats_in = [torch.zeros(self.world, 2, dtype=torch.float64, device=d) for _ in range(S)] if self.is_learner else None
            self.prime_stats = torch.zeros(self.world, 2, dtype=torch.float64, device=d) if self.is_learner else None
        else:
            self.hdrs = [torch.zeros(2 * B, dtype=torch.float32, device=d)]
            self.stats_in = None
            self.all_stats = torch.zeros(self.world, 2, dtype=torch.float64, device=d)
        self.hdr = self.hdrs[0]
        if star:
            self.all_stats = self.hdr[2 * B:].view(torch.float64).view(self.world, 2)
        self.wire = torch.zeros(B, wb, dtype=torch.uint8, device=d)
        self.wire_alls = [torch.zeros(self.world, B, wb, dtype=torch.uint8, device=d) for _ in range(S)] if self.is_learner else None
        self.wire_all = self.wire_alls[0] if self.is_learner else None
        self.bucket = torch.zeros(int(param_numel), dtype=torch.float32, device=d) if param_numel else None
        self.opened = self.served = self.finished = 0
        cuda = d.type == "cuda"
        # Two communicators and two streams at the learner (star): headers and parameters go DOWN, rows and statistics come UP.  Operations
        # on one RCCL communicator run in issue order and send / recv rendezvous; with one communicator the header of round r + 1 would
        # queue behind the receive of round r's replies and the rounds could never overlap.  (new_group is collective: every rank builds
        # its link at the same point of the program.)
        self.g_down = self.g_up = None
        if star and self.world > 1:
            self.g_down = dist.new_group(backend=dist.get_backend())
            self.g_up = dist.new_group(backend=dist.get_backend())
        self.xs = torch.cuda.Stream(d) if (cuda and self.is_learner) else None            # up / the collective round
        self.xd = torch.cuda.Stream(d) if (cuda and self.is_learner and star) else None   # down
        self.hdr_host = [torch.zeros(B, dtype=torch.float32).pin_memory() if cuda else torch.zeros(B) for _ in range(4 + S)]
        self.hdr_ev = [None] * (4 + S)
        self.timer, self.timer_down = _Timer(d), _Timer(d)
        # transport of the star round's point-to-point messages: the process group's backend (RCCL kernels, or gloo through host memory), or
        # "ipc" (HSAD_LINK_TRANSPORT): one-sided device-to-device copies into IPC-mapped landing rings + store flags, no communication kernel
        # resident on any GPU (class _IpcTransport; one node only)
        # Default ("auto"): ipc for an RCCL job whose ranks all sit on this node (what `bench.py --gpus N` / torchrun --nnodes=1 start), the
        # backend otherwise (gloo smoke runs keep their host-staged path unless asked).
        want = (transport if transport is not None else os.environ.get("HSAD_LINK_TRANSPORT", "")) or "auto"
        if want not in ("auto", "backend", "ipc"):
            raise ValueError("ReplayLink transport must be 'auto', 'backend' or 'ipc', not %r" % (want,))
        if want == "auto":
            one_node = int(os.environ.get("LOCAL_WORLD_SIZE", self.world)) == self.world
            want = "ipc" if (dist.get_backend() == "nccl" and one_node) else "backend"
        self.transport, self.ipc = "backend", None
        self.transport_decision = {"asked": (transport if transport is not None else os.environ.get("HSAD_LINK_TRANSPORT", "")) or "auto", "wanted": want}
        if want == "ipc" and star and self.world > 1 and cuda:
            # peer access first (hipDeviceCanAccessPeer between the learner's device and every actor's, both directions, voted through the
            # store so that every rank decides the same way): a pair without it cannot map the other's landing ring -- the backend then
            why = self._peer_access_vote(d)
            if why:
                want = "backend"
                self.transport_decision["peer_access"] = why
                if self.rank == self.learner:
                    print("ReplayLink: no peer access between the learner's and an actor's device (%s) -- using the %s backend" % (why, dist.get_backend()), flush=True)
            else:
                self.transport_decision["peer_access"] = "ok"
        if want == "ipc" and star and self.world > 1 and cuda:
            hdr_bytes = self.hdr.numel() * 4
            ipc = _IpcTransport(self.store, self._keys, self.rank, self.world, self.learner, d, max(hdr_bytes, 4 * int(param_numel)),
                                max(B * wb, 64), 2 * (self.ahead + 2))
            if ipc.ok:
                self.ipc, self.transport, self.staged = ipc, "ipc", False
            else:
                self.transport_decision["ipc_open"] = ipc.why
                if self.rank == self.learner:
                    print("ReplayLink: the ipc transport is not available (%s) -- using the %s backend" % (ipc.why, dist.get_backend()), flush=True)
        self.transport_decision["chosen"] = self.transport
        self.transport_decision["backend"] = dist.get_backend() if self.world > 1 else None
        self._rounds = []           # learner: rounds begun and not yet finished, oldest first
        self._done_ev = {}          # round -> event on the up stream: its replies are in and unpacked
        self._bucket_ev = None      # the last parameter send has left the bucket
        self.wait_ms, self.wait_n = 0.0, 0

    def _peer_access_vote(self, d):
        """-> "" when every (learner, actor) pair of devices can access each other (or sits on one device), else the reason; every rank
        publishes its device index and its own checks in the store and all read all votes: one decision for the job"""
        idx = torch.device(d).index
        idx = torch.cuda.current_device() if idx is None else idx
        st, k = self.store, self._keys + "peer/"
        st.set(k + "dev/%d" % self.rank, str(idx))
        others = [p for p in range(self.world) if p != self.learner] if self.is_learner else [self.learner]
        bad = []
        for p in others:
            q = int(st.get(k + "dev/%d" % p).decode())
            if q == idx:
                continue                      # ranks sharing one device (tests): the same address space
            try:
                ok = q < torch.cuda.device_count() and torch.cuda.can_device_access_peer(idx, q)
            except Exception as e:            # (a process that sees only its own device cannot ask: the ipc open vote decides)
                ok = True
            if not ok:
                bad.append("rank %d device %d -> rank %d device %d" % (self.rank, idx, p, q))
        st.set(k + "vote/%d" % self.rank, "; ".join(bad))
        votes = [st.get(k + "vote/%d" % p).decode() for p in range(self.world)]
        return "; ".join(v for v in votes if v)

    # -- transport (RCCL on the GPU; host staging only for gloo with GPU shards) --
    def _bcast(self, t):
        import torch.distributed as dist
        if not self.staged:
            dist.broadcast(t, src=self.learner)
            return
        h = t.cpu()
        dist.broadcast(h, src=self.learner)
        if not self.is_learner:
            t.copy_(h)

    def _all_gather_stats(self, mine):
        import torch.distributed as dist
        if not self.staged:
            dist.all_gather_into_tensor(self.all_stats.view(-1), mine)
            return
        h = torch.zeros(self.world * 2, dtype=torch.float64)
        dist.all_gather_into_tensor(h, mine.cpu())
        self.all_stats.copy_(h.view(self.world, 2))

    def _gather_wire(self):
        import torch.distributed as dist
        if not self.staged:
            dist.gather(self.wire, list(self.wire_all.unbind(0)) if self.is_learner else None, dst=self.learner)
            return
        h = self.wire.cpu()
        parts = [torch.empty_like(h) for _ in range(self.world)] if self.is_learner else None
        dist.gather(h, parts, dst=self.learner)
        if self.is_learner:
            self.wire_all.copy_(torch.stack(parts))

    def _p2p_begin(self, ops, group=None):
        """ops: [("send" | "recv", tensor, peer)] -> one grouped launch (ncclGroupStart / End in RCCL: the learner's sends to and
        receives from all actors progress at once; two messages between the same pair are matched in the order given)"""
        import torch.distributed as dist
        if not ops:
            return None
        if self.ipc is not None:
            return ("ipc", [self.ipc.send(t.contiguous(), peer) if kind == "send" else self.ipc.recv(t, peer) for kind, t, peer in ops])
        host = {}                                # gloo with GPU shards: ONE host copy of a tensor that goes to several peers
        def _h(t):
            if id(t) not in host:
                host[id(t)] = t.cpu()
            return host[id(t)]
        staged = [(kind, t, peer, (_h(t) if kind == "send" else torch.empty(t.shape, dtype=t.dtype)) if self.staged else t)
                  for kind, t, peer in ops]
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf, peer, group)
                                        for kind, _, peer, buf in staged])
        return works, staged

    def _p2p_end(self, pending):
        if pending is None:
            return
        if pending[0] == "ipc":
            for w in pending[1]:
                self.ipc.wait(w)         # the HOST waits for the store key of the message (sends: nothing to wait for)
            return
        works, staged = pending
        for w in works:
            w.wait()             # RCCL: the current stream waits, the host does not
        if self.staged:
            for kind, t, _, buf in staged:
                if kind == "recv":
                    t.copy_(buf)

    def _p2p(self, ops, group=None):
        self._p2p_end(self._p2p_begin(ops, group))

    def _round_star(self, flags):
        """an ACTOR's side of one point-to-point round (class docstring), stream-ordered on the caller's current stream"""
        B, L, me = self.B, self.learner, self.rank
        assert not self.is_learner
        if flags & self.PRIME:
            self._p2p([("send", self.shard.stats(), L)], self.g_up)
        self._p2p([("recv", self.hdr, L)], self.g_down)
        self.shard.serve(self.hdr[:B], self.all_stats, me, self.wire)
        if flags & self.HAS_PRIO:
            self.shard.answer(self.hdr[B:2 * B], me)
        self._p2p([("send", self.wire, L), ("send", self.shard.stats(), L)], self.g_up)
        if flags & self.PARAMS:
            self._p2p([("recv", self.bucket, L)], self.g_down)
        self.served += 1
        return None

    def _star_open(self, r, flags, canon, prio):
        """the LEARNER's side of round r, first half -- header (and parameters) out on the down stream, the receives of the replies posted
        and the own shard served on the up stream.  Returns the round's record; `_star_collect` finishes it."""
        B, L, me = self.B, self.learner, self.rank
        peers = [k for k in range(self.world) if k != L]
        S = len(self.hdrs)
        k = r % S
        hdr, wire_all, stats_in = self.hdrs[k], self.wire_alls[k], self.stats_in[k]
        cur = torch.cuda.current_stream(self.device) if self.xs is not None else None
        down = torch.cuda.stream(self.xd) if self.xd is not None else _NullCtx()
        up = torch.cuda.stream(self.xs) if self.xs is not None else _NullCtx()
        td, t = self.timer_down, self.timer
        rec = {"r": r, "slot": k, "flags": flags}
        if flags & self.PRIME:          # nobody has told the learner a (sum, size) yet: the actors send theirs first (up)
            if self.xs is not None:
                self.xs.wait_stream(cur)          # (buffers of this link were zero-filled on the caller's stream)
            with up:
                self._p2p([("recv", self.prime_stats[p], p) for p in peers], self.g_up)
                self.prime_stats[L].copy_(self.shard.stats())
            if self.xd is not None:
                self.xd.wait_stream(self.xs)
        if self.finished == 0:          # no round collected yet (the first `ahead` rounds): what the actors said up front
            src_stats = self.prime_stats
        else:                           # the statistics of the newest round this host has collected (`ahead` rounds old)
            last = self.finished - 1
            src_stats = self.stats_in[last % S]
            if self.xd is not None and last in self._done_ev:
                self.xd.wait_event(self._done_ev[last])
        with down:
            if self.xd is not None:
                self.xd.wait_stream(cur)          # everything issued so far: the priorities, the staged bucket
            td.start()
            hdr[:B].copy_(canon, non_blocking=True)
            if self.xd is not None:
                self.hdr_ev[r % len(self.hdr_ev)] = torch.cuda.Event()
                self.hdr_ev[r % len(self.hdr_ev)].record()
            if prio is not None:
                if self.xd is not None:
                    prio.record_stream(self.xd)   # allocated on the compute stream, read here
                hdr[B:2 * B].copy_(prio)
            tail = hdr[2 * B:].view(torch.float64).view(self.world, 2)
            tail.copy_(src_stats)
            rec["used"] = tail.clone()
            built = None
            if self.xd is not None:
                rec["used"].record_stream(self.xs)      # allocated on the down stream, read when the round is collected on the up stream
                built = torch.cuda.Event()
                built.record()
            nccl = self.xd is not None and not self.staged and self.ipc is None
            rec["send"] = self._p2p_begin([("send", hdr, p) for p in peers], self.g_down)
            if nccl or not peers:                 # RCCL: the down stream waits; gloo: the works are waited for when the round is collected
                self._p2p_end(rec.pop("send"))
            td.mark("header_send_ms")
            if flags & self.PARAMS:
                ps = self._p2p_begin([("send", self.bucket, p) for p in peers], self.g_down)
                if nccl or not peers:
                    self._p2p_end(ps)
                    if nccl:
                        self._bucket_ev = torch.cuda.Event()
                        self._bucket_ev.record()
                else:
                    rec["psend"] = ps
                    if self.ipc is not None and self.xd is not None:   # ipc: the puts are async copies on the down stream -- the bucket is busy until the last has left
                        self._bucket_ev = torch.cuda.Event()
                        self._bucket_ev.record()
                td.mark("param_send_ms")
            td.stop()
        with up:
            if self.xs is not None:
                self.xs.wait_event(built)
            rec["marks"] = t.start()
            rec["recv"] = self._p2p_begin([("recv", wire_all[p], p) for p in peers] + [("recv", stats_in[p], p) for p in peers], self.g_up)
            rec["owner"] = self.shard.serve(hdr[:B], tail, me, wire_all[L])
            if flags & self.HAS_PRIO:
                self.shard.answer(hdr[B:2 * B], me)
            stats_in[L].copy_(self.shard.stats())
            t.mark("serve_ms", rec["marks"])
        if self.xs is not None and not self.staged and self.ipc is None:   # RCCL: the stream waits for the replies, the host does not -- the whole round is enqueued now
            self._star_collect(rec)
        return rec

    def _star_collect(self, rec, host_wait=False):
        """second half: replies in, rows unpacked (on the up stream; gloo: called from finish(), the host waits here)"""
        if "result" in rec:
            return
        t, m = self.timer, rec.pop("marks")
        k = rec["slot"]
        with (torch.cuda.stream(self.xs) if self.xs is not None else _NullCtx()):
            import time
            t0 = time.perf_counter()
            self._p2p_end(rec.pop("recv"))
            for key in ("send", "psend"):
                if key in rec:
                    self._p2p_end(rec.pop(key))
            if host_wait:                         # gloo: the host sat here until the last reply was in (RCCL: see finish())
                self.wait_ms += (time.perf_counter() - t0) * 1e3
                self.wait_n += 1
            t.mark("exchange_ms", m)
            batch, raw_w = self.shard.assemble(self.wire_alls[k], rec["owner"])
            used = rec["used"]
            total = used[:, 0].sum().to(torch.float32)
            n_total = used[:, 1].sum().to(torch.float32)
            y = torch.pow(n_total * (raw_w / total), -self.beta)
            weight = y / y.max()            # (inside the stream context: the compute stream only waits for `ev` below)
            t.mark("assemble_ms", m)
            t.stop(m)
            if self.xs is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                self._done_ev[rec["r"]] = ev
                self._done_ev.pop(rec["r"] - 2 * len(self.hdrs), None)
        rec["result"] = (batch, weight)

    # -- one round, every rank (stream-ordered on the caller's current stream) --
    def _round(self, flags):
        if self.mode == "star":
            return self._round_star(flags)          # (actors; the learner's side is _star_open / _star_collect)
        B, t = self.B, self.timer if self.is_learner else None
        if t:
            t.start()
        self._bcast(self.hdr)
        if t:
            t.mark("header_bcast_ms")
        if flags & self.HAS_PRIO:
            self.shard.answer(self.hdr[B:2 * B], self.rank)
        self._all_gather_stats(self.shard.stats())
        if t:
            t.mark("stats_allgather_ms")
        owner = self.shard.serve(self.hdr[:B], self.all_stats, self.rank, self.wire)
        if t:
            t.mark("serve_ms")
        self._gather_wire()
        if t:
            t.mark("batch_gather_ms")
        if flags & self.PARAMS:
            self._bcast(self.bucket)
            if t:
                t.mark("param_bcast_ms")
        out = None
        if self.is_learner:
            batch, raw_w = self.shard.assemble(self.wire_all, owner)
            # (N * w / sum)^-beta / max with N, sum over all shards (prioritized_replay.h:322-333), on the device
            total = self.all_stats[:, 0].sum().to(torch.float32)
            n_total = self.all_stats[:, 1].sum().to(torch.float32)
            y = torch.pow(n_total * (raw_w / total), -self.beta)
            out = (batch, y / y.max())
            t.mark("assemble_ms")
            t.stop()
        self.served += 1
        return out

    # -- learner --
    def stage_params(self, *flats):
        """copy the parameter vectors (online, target) into the persistent bucket on the CURRENT stream; the next begin(params=True)
        broadcasts that snapshot while later updates already change the live parameters"""
        if self._bucket_ev is not None:      # the previous snapshot may still be leaving (rounds are opened several updates ahead)
            torch.cuda.current_stream(self.device).wait_event(self._bucket_ev)
        off = 0
        for f in flats:
            self.bucket[off:off + f.numel()].copy_(f.reshape(-1))
            off += f.numel()

    def begin(self, prio=None, params=False, stop=False):
        """open a round: tell the actors (store), then run the learner's side on the exchange streams.  Call it BEFORE issuing the
        kernels of the update that will run meanwhile; `prio` [B] = aggregated priorities of the OLDEST batch still unanswered.  Up to
        `ahead` rounds may be open at once (star); finish() returns them oldest first."""
        assert self.is_learner and len(self._rounds) < self.ahead, "ReplayLink.begin: %d rounds already open (ahead = %d)" % (len(self._rounds), self.ahead)
        flags = (self.PARAMS if params else 0) | (self.STOP if stop else 0) | (self.HAS_PRIO if prio is not None else 0)
        r = self.opened
        if r == 0 and self.mode == "star":
            flags |= self.PRIME                   # nobody has told the learner a (sum, size) yet: the actors send theirs first
        # the flag ring is only safe while no rank is FLAG_SLOTS rounds behind; the value carries its round so that poll() can tell
        self.store.set(self._keys + "flags/%d" % (r % self.FLAG_SLOTS), "%d %d" % (r, flags))
        self.store.add(self.ROUND_KEY, 1)
        self.opened += 1
        B, k = self.B, r % len(self.hdr_host)
        canon = self.hdr_host[k]
        if self.hdr_ev[k] is not None and not self.hdr_ev[k].query():
            self.hdr_ev[k].synchronize()          # the copy that last read this pinned slot (several rounds ago) -- normally long done
        canon.copy_(torch.from_numpy(self.shard.draw_canonical(B)))
        if self.mode == "star":
            self._rounds.append(self._star_open(r, flags, canon, prio))
            return
        if self.xs is not None:
            self.xs.wait_stream(torch.cuda.current_stream(self.device))   # everything issued so far (the priorities, the staged bucket)
            with torch.cuda.stream(self.xs):
                self.hdr[:B].copy_(canon, non_blocking=True)
                self.hdr_ev[k] = torch.cuda.Event()
                self.hdr_ev[k].record()
                if prio is not None:
                    prio.record_stream(self.xs)           # allocated on the compute stream, read here
                    self.hdr[B:2 * B].copy_(prio)
                self._rounds.append({"r": r, "result": self._round(flags)})
        else:
            self.hdr[:B].copy_(canon)
            if prio is not None:
                self.hdr[B:2 * B].copy_(prio)
            self._rounds.append({"r": r, "result": self._round(flags)})

    def finish(self):
        """-> ((fields, reward, terminal, bootstrap, seq_len), weight) of the OLDEST open round; the current stream waits for the
        exchange stream, the host does not (gloo: the host waits for the replies here).  `wait_for_batch_ms` in timings() is how long
        that wait was: zero when the round was opened early enough."""
        rec = self._rounds.pop(0)
        if self.mode == "star" and "result" not in rec:
            self._star_collect(rec, host_wait=True)
        res = rec["result"]
        self.finished += 1
        if self.xs is not None:
            cur = torch.cuda.current_stream(self.device)
            ev = self._done_ev.get(rec["r"]) if self.mode == "star" else None
            if ev is not None:
                # how long the compute stream will sit in front of this batch: the event pair (arrival on the compute stream, replies
                # unpacked on the up stream) is read one call later, when both have certainly happened
                here = torch.cuda.Event(enable_timing=True)
                here.record(cur)
                self._wait_probe = getattr(self, "_wait_probe", [])
                self._wait_probe.append((here, ev))
                while self._wait_probe and self._wait_probe[0][0].query() and self._wait_probe[0][1].query():
                    h, e = self._wait_probe.pop(0)
                    self.wait_ms += max(0.0, h.elapsed_time(e))
                    self.wait_n += 1
                cur.wait_event(ev)
            else:
                cur.wait_stream(self.xs)
            (f, reward, terminal, bootstrap, seq_len), w = res
            for x in list(f.values()) + [reward, terminal, bootstrap, seq_len, w]:
                if x is not None:
                    x.record_stream(cur)
        return res

    # -- actors --
    poll_every = 1         # ask the rendezvous store only every k-th call (an acting step is ~1 ms; a store round trip goes to rank 0's server)

    def poll(self):
        """has the learner opened a round this rank has not served yet?  -> its flags, or None.  Host-side only (one store query every
        `poll_every` calls; a known backlog is served without asking again)"""
        self._polls = getattr(self, "_polls", 0) + 1
        if getattr(self, "_known_open", 0) <= self.served:
            if self._polls % max(int(self.poll_every), 1):
                return None
            self._known_open = int(self.store.add(self.ROUND_KEY, 0))
            if self._known_open <= self.served:
                return None
        if self._known_open - self.served >= self.FLAG_SLOTS:
            raise RuntimeError("ReplayLink: rank %d is %d rounds behind the learner -- the %d-slot flag ring has wrapped" %
                               (self.rank, self._known_open - self.served, self.FLAG_SLOTS))
        r, flags = self.store.get(self._keys + "flags/%d" % (self.served % self.FLAG_SLOTS)).decode().split()
        if int(r) != self.served:
            raise RuntimeError("ReplayLink: flag slot of round %d holds round %s" % (self.served, r))
        return int(flags)

    def serve(self, flags):
        """serve the round `poll` announced, on the current stream.  After a PARAMS round the new [online | target] parameters are in
        self.bucket (stream-ordered)."""
        assert not self.is_learner
        self._round(flags)
        return bool(flags & self.STOP)

    def timings(self):
        out = self.timer.summary()
        out.update(self.timer_down.summary())
        out.setdefault("param_send_ms", 0.0)
        out["wait_for_batch_ms"] = self.wait_ms / max(self.wait_n, 1)
        out["rounds_ahead"] = self.ahead
        return out

    def close(self):
        if self.ipc is not None:
            self.ipc.close()
