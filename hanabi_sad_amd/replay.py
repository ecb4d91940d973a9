"""Host mirror of the device replay / actor buffers in libhsad (include/hsad.h):

* `aggregate_priority`            <- rela.aggregate_priority            (rela/pybind.cc:92, r2d2_actor.h:10-21)
* `DeviceReplay`                  <- rela.RNNPrioritizedReplay          (rela/pybind.cc:46-58)
* `SequenceWriter`                <- MultiStepBuffer + R2D2Buffer inside R2D2Actor (rela/r2d2_actor.h:23-172)

Transitions are described by an ordered list of (name, width, dtype) fields = the keys of the reference's
obs/action TensorDicts.  All tensors live on the GPU; there is no CPU path."""
import ctypes as C

import torch

from . import _lib

_DT = {torch.float32: 0, torch.int64: 1, torch.uint8: 2, torch.bool: 2}
_TORCH_DT = {0: torch.float32, 1: torch.int64, 2: torch.uint8}


class Bits:
    """field dtype: float32 0.0 / 1.0 values at the API, ONE BIT each in the stored rows (include/hsad.h HSAD_BITS).
    segments: equal parts of the field that each start on a 64-bit word (the players of a VDN row)"""

    def __init__(self, segments=1):
        self.segments = int(segments)

    def __repr__(self):
        return "Bits(%d)" % self.segments


def _api_dtype(dt):
    return torch.float32 if isinstance(dt, Bits) else (torch.uint8 if dt == torch.bool else dt)


def _fields_struct(fields):
    arr = (_lib.Field * len(fields))()
    for i, (_, width, dtype) in enumerate(fields):
        arr[i].width = int(width)
        arr[i].dtype = (3 | (dtype.segments << 8)) if isinstance(dtype, Bits) else _DT[dtype]
    return arr


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def aggregate_priority(priority, seq_len, eta):
    """priority [T,B] f32, seq_len [B] f32 (both on the GPU) -> [B] f32."""
    lib = _lib.load_library()
    if priority.device.type != "cuda":
        raise _lib.HsadError("aggregate_priority needs GPU tensors; there is no CPU path")
    priority = priority.contiguous().float()
    seq_len = seq_len.contiguous().float()
    T, B = priority.shape
    out = torch.empty(B, dtype=torch.float32, device=priority.device)
    _lib.check(lib.hsad_aggregate_priority(priority.data_ptr(), seq_len.data_ptr(), T, B, float(eta), out.data_ptr(),
                                           _stream(priority.device)))
    return out


class DeviceReplay:
    def __init__(self, capacity, seed, alpha, beta, prefetch, seq_len, fields, device="cuda:0"):
        self.lib = _lib.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HsadError("DeviceReplay needs a ROCm device; there is no CPU path")
        self.fields = [(n, int(w), dt) for n, w, dt in fields]
        self.T = int(seq_len)
        self._out = {}
        self._owners = []      # owner[] of the served draws that still wait for their priorities (oldest first)
        self.h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.hsad_replay_create(int(capacity), int(seed), float(alpha), float(beta), int(prefetch),
                                               self.T, len(self.fields), _fields_struct(self.fields), idx,
                                               C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsad_replay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bytes(self):
        return int(self.lib.hsad_replay_bytes(self.h))

    def _check_field_tensors(self, tensors, lead):
        out = []
        for (name, w, dt), t in zip(self.fields, tensors):
            want = _api_dtype(dt)
            t = t.to(want) if t.dtype != want else t
            assert t.is_contiguous() and t.device == self.device, name
            assert tuple(t.shape) in (tuple(lead) + (w,), tuple(lead)) or t.numel() == int(torch.tensor(lead).prod()) * w, \
                (name, tuple(t.shape), lead, w)
            out.append(t)
        return out

    def add(self, fields, reward, terminal, bootstrap, seq_len, priority, n_dev=None):
        """fields: dict name -> [n, T, width]; reward/bootstrap [n,T] f32; terminal [n,T] bool/u8;
        seq_len/priority [n] f32."""
        n = int(priority.shape[0])
        ts = self._check_field_tensors([fields[name] for name, _, _ in self.fields], (n, self.T))
        terminal = terminal.to(torch.uint8).contiguous()
        _lib.check(self.lib.hsad_replay_add(self.h, n, _ptr_array(ts), reward.contiguous().data_ptr(),
                                            terminal.data_ptr(), bootstrap.contiguous().data_ptr(),
                                            seq_len.contiguous().data_ptr(), priority.contiguous().data_ptr(),
                                            None if n_dev is None else n_dev.data_ptr(), _stream(self.device)))

    def set_field_output(self, name, kind, ld=0):
        """what sample() / sample_at() return for bit field `name`: "f32" [T,B,width] float32 (default) | "bf16" [T,B,segments,ld]
        bfloat16, zero-padded to ld per segment (the learner's GEMM operand, no cast pass) | "raw" [T,B,bytes] uint8"""
        k = [n for n, _, _ in self.fields].index(name)
        code = {"f32": 0, "bf16": 1, "raw": 2}[kind]
        _lib.check(self.lib.hsad_replay_set_field_output(self.h, k, code, int(ld)))
        self._out[name] = (kind, int(ld))

    def _alloc_outs(self, n):
        d, T, outs = self.device, self.T, []
        for k, (name, w, dt) in enumerate(self.fields):
            kind, ld = self._out.get(name, ("f32", 0))
            if kind == "bf16":
                outs.append(torch.empty(T, n, dt.segments, ld, dtype=torch.bfloat16, device=d))
            elif kind == "raw":
                outs.append(torch.empty(T, n, self.lib.hsad_replay_field_bytes(self.h, k), dtype=torch.uint8, device=d))
            else:
                outs.append(torch.empty(T, n, w, dtype=_api_dtype(dt), device=d))
        return outs

    def sample(self, batch):
        """-> (dict of fields [T,B,width], reward [T,B], terminal [T,B] bool, bootstrap [T,B], seq_len [B]), weight [B]"""
        d, T = self.device, self.T
        outs = self._alloc_outs(batch)
        reward = torch.empty(T, batch, dtype=torch.float32, device=d)
        terminal = torch.empty(T, batch, dtype=torch.uint8, device=d)
        bootstrap = torch.empty(T, batch, dtype=torch.float32, device=d)
        seq_len = torch.empty(batch, dtype=torch.float32, device=d)
        weight = torch.empty(batch, dtype=torch.float32, device=d)
        _lib.check(self.lib.hsad_replay_sample(self.h, batch, _ptr_array(outs), reward.data_ptr(), terminal.data_ptr(),
                                               bootstrap.data_ptr(), seq_len.data_ptr(), weight.data_ptr(),
                                               _stream(d)))
        fields = {name: t for (name, _, _), t in zip(self.fields, outs)}
        return (fields, reward, terminal.view(torch.bool), bootstrap, seq_len), weight

    # ---- shard interface used by hanabi_sad_amd.dist.ShardedReplay (one DeviceReplay per GPU) ----
    def priority_sum(self):
        """(running weight sum as float, size) of this shard; synchronises the device"""
        sm, sz = C.c_double(0.0), C.c_int32(0)
        _lib.check(self.lib.hsad_replay_priority_sum(self.h, C.byref(sm), C.byref(sz)))
        return sm.value, sz.value

    def draw_canonical(self, n):
        """the next n canonical uniforms of this replay's std::mt19937 stream (what sample(n) would consume), float32"""
        import numpy as np
        out = (C.c_float * n)()
        _lib.check(self.lib.hsad_replay_draw_canonical(self.h, int(n), out))
        return np.frombuffer(out, dtype=np.float32).copy()

    def sample_at(self, targets):
        """targets: host float32 positions in this shard's cumulative-weight space (len may be 0).
        -> (fields dict [T,n,w], reward, terminal(bool), bootstrap [T,n], seq_len [n]), raw_weight [n]"""
        import numpy as np
        targets = np.ascontiguousarray(targets, dtype=np.float32)
        n, d, T = int(targets.shape[0]), self.device, self.T
        outs = self._alloc_outs(n)
        reward = torch.empty(T, n, dtype=torch.float32, device=d)
        terminal = torch.empty(T, n, dtype=torch.uint8, device=d)
        bootstrap = torch.empty(T, n, dtype=torch.float32, device=d)
        seq_len = torch.empty(n, dtype=torch.float32, device=d)
        raw_w = torch.empty(n, dtype=torch.float32, device=d)
        _lib.check(self.lib.hsad_replay_sample_at(self.h, n, targets.ctypes.data_as(C.POINTER(C.c_float)), _ptr_array(outs),
                                                  reward.data_ptr(), terminal.data_ptr(), bootstrap.data_ptr(),
                                                  seq_len.data_ptr(), raw_w.data_ptr(), _stream(d)))
        fields = {name: t for (name, _, _), t in zip(self.fields, outs)}
        return (fields, reward, terminal.view(torch.bool), bootstrap, seq_len), raw_w

    # ---- the same draw without host round trips: shard interface of hanabi_sad_amd.dist.ReplayLink ----
    def stats(self):
        """(running weight sum, size) as a float64 [2] device tensor; stream-ordered, no synchronisation"""
        out = torch.empty(2, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.hsad_replay_stats(self.h, out.data_ptr(), _stream(self.device)))
        return out

    def wire_bytes(self):
        return int(self.lib.hsad_replay_wire_bytes(self.h))

    def serve(self, canon, all_stats, rank, wire_out):
        """one stratified draw over the concatenation of the shards described by all_stats [world, 2]: the positions that fall into
        THIS shard are drawn and written, in batch order, into wire_out[0 ..] ([B, wire_bytes] uint8).  -> owner int32 [B]"""
        B, world = int(canon.numel()), int(all_stats.shape[0])
        assert canon.dtype == torch.float32 and all_stats.dtype == torch.float64 and canon.is_contiguous() and all_stats.is_contiguous()
        assert wire_out.dtype == torch.uint8 and wire_out.is_contiguous() and wire_out.numel() >= B * self.wire_bytes()
        owner = torch.empty(B, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hsad_replay_serve(self.h, B, canon.data_ptr(), all_stats.data_ptr(), world, int(rank), owner.data_ptr(),
                                              wire_out.data_ptr(), _stream(self.device)))
        self._owners.append(owner)
        return owner

    def answer(self, priority, rank):
        """priorities [B] of the OLDEST outstanding served draw: the ones at positions this shard owned are written back"""
        owner = self._owners.pop(0)
        priority = priority.to(self.device, torch.float32).contiguous()
        _lib.check(self.lib.hsad_replay_update_owned(self.h, int(priority.numel()), priority.data_ptr(), owner.data_ptr(), int(rank),
                                                     _stream(self.device)))

    def assemble(self, wire_all, owner):
        """learner: every rank's wire buffer [world, B, wire_bytes] + the draw's owner[] -> the batch as sample() returns it, and the
        raw weights [B] (the caller forms the importance weights over all shards)"""
        world, B = int(wire_all.shape[0]), int(owner.numel())
        d, T = self.device, self.T
        outs = self._alloc_outs(B)
        reward = torch.empty(T, B, dtype=torch.float32, device=d)
        terminal = torch.empty(T, B, dtype=torch.uint8, device=d)
        bootstrap = torch.empty(T, B, dtype=torch.float32, device=d)
        seq_len = torch.empty(B, dtype=torch.float32, device=d)
        raw_w = torch.empty(B, dtype=torch.float32, device=d)
        _lib.check(self.lib.hsad_replay_assemble(self.h, B, world, wire_all.data_ptr(), owner.data_ptr(), _ptr_array(outs),
                                                 reward.data_ptr(), terminal.data_ptr(), bootstrap.data_ptr(), seq_len.data_ptr(),
                                                 raw_w.data_ptr(), _stream(d)))
        fields = {name: t for (name, _, _), t in zip(self.fields, outs)}
        return (fields, reward, terminal.view(torch.bool), bootstrap, seq_len), raw_w

    def update_priority(self, priority):
        priority = priority.to(self.device, torch.float32).contiguous()
        _lib.check(self.lib.hsad_replay_update_priority(self.h, priority.data_ptr(), int(priority.numel()),
                                                        _stream(self.device)))

    def _counts(self):
        s, n = C.c_int32(0), C.c_int32(0)
        _lib.check(self.lib.hsad_replay_size(self.h, C.byref(s), C.byref(n)))
        return s.value, n.value

    def size(self):
        return self._counts()[0]

    def num_add(self):
        return self._counts()[1]

    def get(self, idx):
        d, T = self.device, self.T
        outs = [torch.empty(T, w, dtype=(_api_dtype(dt)), device=d) for _, w, dt in self.fields]
        reward = torch.empty(T, dtype=torch.float32, device=d)
        terminal = torch.empty(T, dtype=torch.uint8, device=d)
        bootstrap = torch.empty(T, dtype=torch.float32, device=d)
        seq_len = torch.empty(1, dtype=torch.float32, device=d)
        _lib.check(self.lib.hsad_replay_get(self.h, int(idx), _ptr_array(outs), reward.data_ptr(), terminal.data_ptr(),
                                            bootstrap.data_ptr(), seq_len.data_ptr(), _stream(d)))
        return {name: t for (name, _, _), t in zip(self.fields, outs)}, reward, terminal.view(torch.bool), bootstrap, seq_len

    def set_outstanding(self, depth):
        """drawn batches that may wait for their priorities at once (the reference's prefetch queue); update_priority answers
        the oldest one"""
        _lib.check(self.lib.hsad_replay_set_outstanding(self.h, int(depth)))

    def last_ids(self, batch):
        out = torch.empty(batch, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hsad_replay_last_ids(self.h, out.data_ptr(), batch, _stream(self.device)))
        return out

    def check_errors(self):
        n = C.c_int32(0)
        _lib.check(self.lib.hsad_replay_error_count(self.h, C.byref(n)))
        if n.value:
            kinds = self.lib.hsad_replay_error_kinds(self.h)
            names = [t for b, t in ((1, "add larger than the ring"), (2, "draw beyond the weight sum"),
                                    (4, "update_priority without a matching draw"),
                                    (8, "sequence writer: push past seq_len or a non-binary value in a bit field")) if kinds & b]
            raise _lib.HsadError("replay logged %d contract violation(s): %s" % (n.value, "; ".join(names) or "?"))


class SequenceWriter:
    def __init__(self, num_envs, multi_step, gamma, seq_len, fields, device="cuda:0"):
        self.lib = _lib.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HsadError("SequenceWriter needs a ROCm device; there is no CPU path")
        self.fields = [(n, int(w), dt) for n, w, dt in fields]
        self.E, self.T = int(num_envs), int(seq_len)
        self._prepacked = set()
        self.h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.hsad_seqwriter_create(self.E, int(multi_step), float(gamma), self.T, len(self.fields),
                                                  _fields_struct(self.fields), idx, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsad_seqwriter_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_prepacked(self, names):
        """bit fields in `names` are handed to push_obs_action as their stored bit words (int64 tensors, e.g. the env's packed
        outputs) instead of float32 0/1 values: a word copy instead of a pack pass"""
        mask = 0
        for k, (n, _, dt) in enumerate(self.fields):
            if n in names:
                if not isinstance(dt, Bits):
                    raise _lib.HsadError("field %s is not a bit field" % n)
                mask |= 1 << k
        _lib.check(self.lib.hsad_seqwriter_set_prepacked(self.h, mask))
        self._prepacked = set(names)

    def push_obs_action(self, fields):
        ts = []
        for name, w, dt in self.fields:
            t = fields[name]
            if name in self._prepacked:
                words = dt.segments * ((w // dt.segments + 63) // 64)
                assert t.dtype == torch.int64 and t.is_contiguous() and t.device == self.device and t.numel() == self.E * words, \
                    (name, t.dtype, tuple(t.shape))
                ts.append(t)
                continue
            want = _api_dtype(dt)
            t = (t.to(want) if t.dtype != want else t).contiguous()
            assert t.device == self.device and t.numel() == self.E * w, (name, tuple(t.shape))
            ts.append(t)
        _lib.check(self.lib.hsad_seqwriter_push_obs_action(self.h, _ptr_array(ts), _stream(self.device)))

    def push_reward_terminal(self, reward, terminal, repeat=1):
        """reward / terminal of this step: [E], or with repeat = k per-game values [E / k] that every k consecutive rows share"""
        reward = reward.to(torch.float32).contiguous()
        terminal = terminal.to(torch.uint8).contiguous()
        if repeat > 1:
            assert reward.numel() * repeat == self.E and terminal.numel() * repeat == self.E
            _lib.check(self.lib.hsad_seqwriter_push_reward_terminal_rep(self.h, reward.data_ptr(), terminal.data_ptr(), int(repeat),
                                                                        _stream(self.device)))
            return
        _lib.check(self.lib.hsad_seqwriter_push_reward_terminal(self.h, reward.data_ptr(), terminal.data_ptr(),
                                                                _stream(self.device)))

    def can_pop(self):
        return bool(self.lib.hsad_seqwriter_can_pop(self.h))

    def pop_transition(self, want_fields=True, want_next=None):
        """-> (obs/action fields of step t, of step t+n, n-step reward, terminal, bootstrap); want_fields / want_next = False
        skip reading the respective rows back from the history ring (None)"""
        d, E = self.device, self.E
        mk = lambda: [torch.empty(E, w, dtype=(_api_dtype(dt)), device=d)
                      for _, w, dt in self.fields]
        cur = mk() if want_fields else None
        nxt = mk() if (want_fields if want_next is None else want_next) else None
        reward = torch.empty(E, dtype=torch.float32, device=d)
        terminal = torch.empty(E, dtype=torch.uint8, device=d)
        bootstrap = torch.empty(E, dtype=torch.float32, device=d)
        _lib.check(self.lib.hsad_seqwriter_pop_transition(
            self.h, _ptr_array(cur) if cur else None, _ptr_array(nxt) if nxt else None, reward.data_ptr(),
            terminal.data_ptr(), bootstrap.data_ptr(), _stream(d)))
        names = [n for n, _, _ in self.fields]
        return (dict(zip(names, cur)) if cur else None, dict(zip(names, nxt)) if nxt else None, reward,
                terminal.view(torch.bool), bootstrap)          # (0 / 1 bytes: reinterpreted, not converted)

    def push_sequence(self, priority):
        priority = priority.to(torch.float32).contiguous()
        _lib.check(self.lib.hsad_seqwriter_push_sequence(self.h, priority.data_ptr(), _stream(self.device)))

    def flush_to_replay(self, replay, eta, out=None):
        """finished sequences -> replay; returns the device counter of how many (written by the kernel; `out` reuses a tensor)"""
        n = out if out is not None else torch.empty(1, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hsad_seqwriter_flush_to_replay(self.h, replay.h, float(eta), n.data_ptr(),
                                                           _stream(self.device)))
        return n
