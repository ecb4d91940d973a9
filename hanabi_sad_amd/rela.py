"""Drop-in names of the reference's `rela` pybind module (rela/pybind.cc:16-93) on the device pipeline.

`RNNPrioritizedReplay(capacity, seed, alpha, beta, prefetch)` keeps the reference signature; the transition
schema and sequence length are taken from the first `bind_schema()` / actor that attaches to it.  The thread
machinery has no counterpart by design: all games advance in lock-step on the GPU (hanabi_sad_amd.actor.DeviceActor).
`BatchRunner` / `R2D2Actor` keep the constructor signatures and carry configuration; `Context.start()` runs the attached
`hanalearn.HanabiThreadLoop`s from ONE background Python thread (ctypes releases the GIL during launches), so a driver
written against the reference -- create_envs / create_threads / ActGroup / context.start() / replay.sample() -- works
unchanged; `Context.step()` is there for drivers that prefer to interleave rollout and learning themselves."""
import os
import sys
import threading
import time as _time

import torch

from .replay import DeviceReplay, aggregate_priority as _aggregate_priority

# BatchRunner's mtxUpdate_ (rela/batch_runner.h:74-77,106-109): a weight update never interleaves with a rollout step that
# the Context thread is enqueueing (both sides enqueue on the same stream; a refresh() between two layers of an act() would
# run one step on mixed weights)
_MODEL_LOCK = threading.RLock()

# The Context threads of a device issue on ONE stream of their own, so that the driver's thread (sampling, the learner's update on its
# current stream) and the rollout overlap on the GPU -- e.g. acting steps next to the learner's BPTT, which occupies half of the XCDs.
# Replay and writer calls from either side are ordered by the library's stream fence; weight updates (BatchRunner.update_model) are
# enqueued ON this stream between two rollout steps; pause() / resume() / terminate() join the two streams.
_ACTOR_STREAMS = {}
_LIVE_CONTEXTS = {}       # device -> number of Context threads currently issuing on its rollout stream
_PACE_NOTICE_SHOWN = False   # Context.start() says once per process that the default pace is on


def _dev_key(device):
    d = torch.device(device)
    return "%s:%d" % (d.type, d.index if d.index is not None else (torch.cuda.current_device() if d.type == "cuda" else 0))


def _loop_device(lp):
    """device key of the GPU a thread loop acts on = its model runner's (pyhanabi/create.py:94-110 round-robins runners, hence
    devices, over threads); None for loops that do not say"""
    try:
        return _dev_key(lp.per_thread[0][0].runner.device)
    except (AttributeError, IndexError, TypeError):
        d = getattr(lp, "device", None)
        return None if d is None else _dev_key(d)


def _actor_stream(device):
    key = _dev_key(device)
    if key not in _ACTOR_STREAMS:
        _ACTOR_STREAMS[key] = torch.cuda.Stream(torch.device(key))
    return _ACTOR_STREAMS[key]


class MultiDeviceError(RuntimeError):
    """One process acting on several GPUs into ONE replay (`--act_device cuda:1,cuda:2`, pyhanabi/create.py:94-101,110) is the
    reference's multi-GPU mechanism because its replay lives in host memory.  Here the replay is device memory next to the games that
    fill it, and the multi-GPU mechanism is one process per GPU with a replay shard each."""

    def __init__(self, have, want):
        super().__init__(
            "this replay lives in the HBM of %s and cannot take sequences from a rollout on %s: a device replay is filled by the games of "
            "its own GPU.  For several acting GPUs launch one process per GPU -- `python -m torch.distributed.run --nproc-per-node N -m "
            "hanabi_sad_amd.selfplay ...` (rank 0 learns, ranks 1..N-1 act into their own replay shards; dist.ReplayLink) -- instead of "
            "--act_device %s,%s in one process" % (have, want, have, want))


class FFTransition:
    """rela.FFTransition fields (rela/pybind.cc:17-23, rela/transition.h:16-35): one feed-forward transition batch.  The reference binds
    the type and nothing produces it from Python (its FFPrioritizedReplay binding is commented out, pybind.cc:34-45); it is here
    because the name is part of the module's surface, with `index` / `pad_like` semantics that follow transition.cc:9-27."""

    def __init__(self, obs=None, action=None, reward=None, terminal=None, bootstrap=None, next_obs=None):
        self.obs, self.action = obs if obs is not None else {}, action if action is not None else {}
        self.reward, self.terminal, self.bootstrap = reward, terminal, bootstrap
        self.next_obs = next_obs if next_obs is not None else {}

    def index(self, i):
        """FFTransition::index (rela/transition.cc:9-27): element i of every batched field"""
        pick = lambda d: {k: v[i] for k, v in d.items()}
        return FFTransition(pick(self.obs), pick(self.action), self.reward[i], self.terminal[i], self.bootstrap[i],
                            pick(self.next_obs))

    def to_dict(self):
        """FFTransition::toDict (rela/transition.cc:29-47): obs and action keys, `next_`-prefixed next_obs, then the scalars"""
        d = dict(self.obs)
        d.update(self.action)
        d.update({"next_" + k: v for k, v in self.next_obs.items()})
        d.update(reward=self.reward, terminal=self.terminal, bootstrap=self.bootstrap)
        return d


class ThreadLoop:
    """rela.ThreadLoop (rela/thread_loop.h:13-60, bound without methods at rela/pybind.cc:60): base of everything a Context runs.
    Subclasses implement `step()` = one iteration of mainLoop's body, and may implement `finished()`; the pause / resume / terminate
    protocol of the reference's class lives in rela.Context here, because one thread drives all loops."""

    def step(self):
        raise NotImplementedError("ThreadLoop.step: one iteration of the loop body (rela/thread_loop.h:20 mainLoop is pure virtual too)")

    def finished(self):
        return False


class RNNTransition:
    """rela.RNNTransition fields (rela/pybind.cc:25-32): obs/action dicts of [T,B,*] tensors, reward, terminal,
    bootstrap [T,B], seq_len [B]."""

    def __init__(self, obs, action, reward, terminal, bootstrap, seq_len):
        self.obs, self.h0, self.action = obs, {}, action
        self.reward, self.terminal, self.bootstrap, self.seq_len = reward, terminal, bootstrap, seq_len


class RNNPrioritizedReplay:
    ACTION_KEYS = ("a", "greedy_a")

    def __init__(self, capacity, seed, alpha, beta, prefetch):
        self.args = (int(capacity), int(seed), float(alpha), float(beta), int(prefetch))
        self.impl = None

    def bind_schema(self, fields, seq_len, device=None):
        device = _dev_key(device if device is not None else "cuda")
        if self.impl is None:
            c, s, a, b, p = self.args
            self.impl = DeviceReplay(c, s, a, b, p, seq_len, fields, device)
            self.device = device
        elif device != self.device:
            raise MultiDeviceError(self.device, device)
        return self.impl

    def size(self):
        return 0 if self.impl is None else self.impl.size()

    def num_add(self):
        return 0 if self.impl is None else self.impl.num_add()

    num_sample = 0         # sample() calls so far: the clock Context's pacing runs the rollout by
    last_sample_time = 0.0  # time.monotonic() of the latest sample(): how a Context notices that a training loop is (still) running

    def sample(self, batchsize, device=None):
        self.num_sample += 1
        self.last_sample_time = _time.monotonic()
        (f, reward, terminal, bootstrap, seq_len), weight = self.impl.sample(batchsize)
        obs = {k: v for k, v in f.items() if k not in self.ACTION_KEYS}
        action = {k: v.squeeze(2) for k, v in f.items() if k in self.ACTION_KEYS}
        if "eps" in obs:
            obs["eps"] = obs["eps"].squeeze(2)
        return RNNTransition(obs, action, reward, terminal, bootstrap, seq_len), weight

    def update_priority(self, priority):
        self.impl.update_priority(priority)

    def get(self, idx):
        f, reward, terminal, bootstrap, seq_len = self.impl.get(idx)
        obs = {k: v.unsqueeze(1) for k, v in f.items() if k not in self.ACTION_KEYS}
        action = {k: v for k, v in f.items() if k in self.ACTION_KEYS}
        return RNNTransition(obs, action, reward, terminal, bootstrap, seq_len[0])


def aggregate_priority(priority, seq_len, eta):
    """rela.aggregate_priority (rela/r2d2_actor.h:10-21); accepts CPU tensors like the reference call site
    (selfplay.py:222-224) and computes on the GPU."""
    dev = priority.device if priority.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    out = _aggregate_priority(priority.to(dev), seq_len.to(dev), eta)
    return out if priority.device.type == "cuda" else out.cpu()


class ContractAgent:
    """The reference's MODEL CONTRACT honoured for an arbitrary model (rela/batch_runner.h:24,76,108; rela/r2d2_actor.h:61-172):
    BatchRunner calls three methods of the agent it is given -- `act(dict) -> dict`, `compute_priority(dict) -> dict`,
    `get_h0(batchsize) -> dict` (on `model._c` when the model is a TorchScript wrapper, else on the model itself) -- with
    tensors shaped [slots, envs, (players), ...] and the hidden state batch-first [slots, envs * players', layers, hidden].
    Any architecture that implements them (a second fc layer, skip connections, other recurrent cores, the OBL models of
    pyhanabi/tools/obl_model.py) acts in the device rollout through this adapter: the observations never leave the GPU, all
    games are ONE slot of `envs` rows, and the n-step priority is the model's own compute_priority on the transition read
    back from the sequence writer, exactly the reference's call.  (The default R2D2Net shape takes the HIP kernels instead.)"""

    cached_q = False          # DeviceActor: priorities come from compute_priority on the popped transition (reference flow)

    def __init__(self, model, device, multi_step, gamma):
        self.model, self.device = model, torch.device(device)
        self.impl = getattr(model, "_c", model)
        self.multi_step, self.gamma = multi_step, gamma
        self.num_player, self.vdn = 1, False
        self.online = self.target = self          # DeviceActor reads agent.online.version for its cached-Q bookkeeping only
        self.version = 0

    def configure(self, num_player, vdn):
        self.num_player, self.vdn = int(num_player), bool(vdn)

    def _lead(self, x, n_rows):
        """[N, ...] rows -> the contract's [1, E, (P), ...]"""
        x = x.to(self.device)
        if self.vdn:
            return x.reshape((1, n_rows // self.num_player, self.num_player) + tuple(x.shape[1:]))
        return x.reshape((1, n_rows) + tuple(x.shape[1:]))

    @staticmethod
    def _hid_in(h):      # [L, N, H] -> [1, N, L, H]
        return h.transpose(0, 1).unsqueeze(0).contiguous()

    def _hid_out(self, h, like):
        return h.to(self.device).reshape(like.shape[1], like.shape[0], like.shape[2]).transpose(0, 1).contiguous()

    def get_h0(self, n):
        with torch.no_grad():
            h = self.impl.get_h0(int(n))
        out = {}
        for k, v in h.items():
            v = v.to(self.device).float()
            if v.dim() == 3 and v.shape[1] != n and v.shape[0] == n:     # batch-first variant
                v = v.transpose(0, 1)
            out[k] = v.contiguous()
        return out

    def act(self, obs, hid, with_q=False):
        n = obs["priv_s"].shape[0]
        d = {k: self._lead(v, n) for k, v in obs.items() if k in ("priv_s", "legal_move", "eps", "own_hand")}
        if "eps" not in d:
            d["eps"] = self._lead(torch.zeros(n, device=self.device), n)
        for k in ("h0", "c0"):
            d[k] = self._hid_in(hid[k])
        with torch.no_grad():
            reply = self.impl.act(d)
        new_hid = {k: self._hid_out(reply[k], hid[k]) for k in ("h0", "c0")}
        out = {"a": reply["a"].to(self.device).reshape(-1).long().contiguous(),
               "greedy_a": reply["greedy_a"].to(self.device).reshape(-1).long().contiguous()}
        return out, new_hid

    def compute_priority_dict(self, cur, nxt, hid, next_hid, reward, terminal, bootstrap):
        """cur / nxt: the popped transition's fields as [E, w] rows (IQL) or [G, P*w] (VDN); -> priority [E]"""
        P = self.num_player if self.vdn else 1
        E = reward.shape[0]

        def shape(name, x):
            if name in ("a", "greedy_a", "eps"):
                return x.reshape((1, E, P) if self.vdn else (1, E))
            return x.reshape((1, E, P, -1) if self.vdn else (1, E, -1))
        d = {}
        for k, v in cur.items():
            d[k] = shape(k, v)
        for k, v in nxt.items():
            if k not in ("a", "greedy_a"):
                d["next_" + k] = shape(k, v)
        d["reward"], d["bootstrap"] = reward.reshape(1, E), bootstrap.reshape(1, E)
        d["terminal"] = terminal.reshape(1, E)
        d["temperature"] = torch.ones_like(d["eps"])      # read (and ignored) by the reference's compute_priority
        for k in ("h0", "c0"):
            d[k] = self._hid_in(hid[k])
            d["next_" + k] = self._hid_in(next_hid[k])
        with torch.no_grad():
            p = self.impl.compute_priority(d)["priority"]
        return p.to(self.device).reshape(-1).float().contiguous()


class BatchRunner:
    """rela.BatchRunner(py_model, device, max_batchsize, methods) (rela/batch_runner.h:17-130): holds the acting copy of the
    agent.  `py_model` with state_dict() carrying `online_net.*` / `target_net.*` of the default R2D2Net shape (the reference
    R2D2Agent), or a plain dict of such tensors, runs on the HIP kernels; any other object that implements the model contract
    (`act` / `compute_priority` / `get_h0`, see ContractAgent) is called as it is.  There is nothing to batch across threads,
    so start()/stop() are no-ops."""

    def __init__(self, py_model, device, max_batchsize=100, methods=None):
        self.device = device
        self.online = self.target = None
        self.model = None                     # set for contract models (no kernels)
        self.update_model(py_model)

    @staticmethod
    def _split(py_model):
        sd = py_model.state_dict() if hasattr(py_model, "state_dict") else dict(py_model)
        on = {k[len("online_net."):]: v for k, v in sd.items() if k.startswith("online_net.")}
        tg = {k[len("target_net."):]: v for k, v in sd.items() if k.startswith("target_net.")}
        if not on:
            raise KeyError("BatchRunner: state_dict has no online_net.* entries")
        return on, (tg or on)

    @staticmethod
    def _kernel_shape(on):
        """an R2D2Net state_dict (1-2 fc layers, 1-3 LSTM layers, dueling + aux heads: pyhanabi/r2d2.py:22-57) and nothing else"""
        from .r2d2 import arch_of, param_order
        nfc, nl = arch_of(on)
        return 1 <= nl <= 3 and set(on) == set(param_order(nfc, nl))

    def update_model(self, py_model):
        """BatchRunner::updateModel (rela/batch_runner.h:74-77)"""
        if getattr(py_model, "device_agent", False):      # an acting agent of this package (e.g. obl.OBLAgent): used as it is
            self.agent_obj = py_model
            return
        with _MODEL_LOCK:
            contract = hasattr(getattr(py_model, "_c", py_model), "act") and not isinstance(py_model, dict)
            try:
                on, tg = self._split(py_model)
                use_kernels = self._kernel_shape(on) and (not contract or not getattr(py_model, "force_contract", False))
            except (KeyError, AttributeError):
                if not contract:
                    raise
                use_kernels = False
            if not use_kernels:
                if not contract:
                    raise KeyError("BatchRunner: the model is neither the default R2D2Net shape nor an object with act / "
                                   "compute_priority / get_h0")
                if self.model is None:
                    self.model = py_model
                elif self.model is not py_model:
                    self.model.load_state_dict(py_model.state_dict())
                return
            if self.online is None:
                from .composite import CNet
                skip = bool(getattr(getattr(py_model, "online_net", None), "skip_connect", False))
                self.online, self.target = CNet(on, self.device, skip_connect=skip), CNet(tg, self.device, skip_connect=skip)    # library-owned nets
                return
            key = _dev_key(self.device)
            st = _ACTOR_STREAMS.get(key)
            if st is not None and not _LIVE_CONTEXTS.get(key, 0):
                # a rollout stream exists but no Context thread is running: whoever acts next does so on its own current stream --
                # drain the rollout stream's backlog into this one and update in line
                torch.cuda.current_stream(st.device).wait_stream(st)
                st = None
            if st is None:
                for net, sd in ((self.online, on), (self.target, tg)):
                    for k, v in sd.items():
                        net.w[k].copy_(v)
                    net.refresh()
                return
            # A rollout stream exists: the acting nets are read there.  Snapshot the new weights on the caller's stream (the learner may go
            # on changing them), then copy + re-derive the kernel operands ON the rollout stream -- the lock puts that between two steps.
            cur = torch.cuda.current_stream(st.device)
            snaps = []
            for net, sd in ((self.online, on), (self.target, tg)):
                snap = {k: v.detach().to(net.w[k].device, torch.float32, copy=True) for k, v in sd.items()}
                for t in snap.values():
                    t.record_stream(st)
                snaps.append((net, snap))
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for net, snap in snaps:
                    for k, v in snap.items():
                        net.w[k].copy_(v)
                    net.refresh()

    def start(self):
        pass

    def stop(self):
        pass

    def make_agent(self, multi_step, gamma, seed=0):
        """the acting agent over this runner's model (R2D2Actor::act / postAct call `act` and `compute_priority` on it)"""
        if getattr(self, "agent_obj", None) is not None:
            return self.agent_obj
        if self.model is not None:
            return ContractAgent(self.model, self.device, multi_step, gamma)
        from .composite import CompositeAgent
        return CompositeAgent(self.online, self.target, multi_step, gamma, seed=seed)


class R2D2Actor:
    """rela.R2D2Actor(runner, multi_step, num_envs, gamma, eta, seq_len, num_player, replay) and the 2-argument eval form
    R2D2Actor(runner, num_player) (rela/pybind.cc:72-84): configuration for hanalearn.HanabiThreadLoop."""

    def __init__(self, runner, *args):
        self.runner, self._num_act = runner, 0
        if len(args) == 1:
            self.num_player, self.replay = int(args[0]), None
            self.multi_step = self.num_envs = self.gamma = self.eta = self.seq_len = None
        elif len(args) == 7:
            self.multi_step, self.num_envs, self.gamma, self.eta, self.seq_len, self.num_player, self.replay = args
        else:
            raise TypeError("R2D2Actor(runner, num_player) or R2D2Actor(runner, multi_step, num_envs, gamma, eta, seq_len, "
                            "num_player, replay)")

    def num_act(self):
        lp = getattr(self, "_loop", None)          # a training loop counts its steps; this actor's share is num_envs per step
        return self._num_act + (lp._train_steps * self._per_step if lp is not None else 0)


class _MarkGroup:
    """the rollout streams' events of one Context iteration: query() = all of them have been reached"""

    def __init__(self, events):
        self.events = events

    def query(self):
        return all(e.query() for e in self.events)


class Context:
    """rela.Context (rela/context.h:18-80): push_env_thread / start / pause / resume / terminate / terminated.
    start() runs every attached loop from one background thread until terminate(); eval loops end by themselves, and
    terminated() turns True once all attached loops have finished (what eval.py:47-51 polls)."""

    def __init__(self):
        self.loops, self._thread, self._paused, self._stop, self._error = [], None, False, False, None
        self._streams = {}         # device key -> the rollout stream of start() (see _ACTOR_STREAMS); step() from the caller's thread uses the caller's
        # pause protocol: every pause() takes a ticket; the loop thread acknowledges the ticket it has SEEN while parked between two
        # steps, so a pause() can only return on an acknowledgement of its own request (an Event could still be set from the
        # previous pause when a resume / pause pair follows within the thread's 1 ms nap)
        self._cv = threading.Condition()
        self._pause_ticket, self._parked_ticket, self._done = 0, 0, False
        self._pace = None          # (replay, steps per sample() call, credit state): an explicit set_pace
        self._run_ahead, self._marks = 0, []
        # the default: pace by whichever replay of the attached training loops is being sampled (see _auto_pace); HSAD_AUTO_PACE=0 or
        # set_pace(False) give the reference's unconditional free-running
        self.auto_pace_steps = self._auto_default = float(os.environ.get("HSAD_AUTO_PACE", "2"))
        self.auto_pace_idle_s = 0.1
        self._auto, self._auto_replays, self._bound = None, None, 0

    def set_pace(self, replay, steps_per_sample=1.0, run_ahead=2):
        """NOT in the reference (rela/context.h free-runs).  On ONE GPU a free-running rollout thread and a training loop share the
        device very unevenly: the rollout's full-chip kernels are always queued, and the learner's persistent launches wait behind each
        of them (16 k sequences/s instead of 65 k).  With a pace the loop thread issues `steps_per_sample` rollout
        steps per `replay.sample()` call of the training loop, once that loop has started sampling (the burn-in before it free-runs),
        and its host never gets more than `run_ahead` steps ahead of the device -- the operating point of selfplay's explicit interleave
        through the reference's two-thread API: rollout on the Context's stream, next to the update on the driver's.

        An unchanged reference driver never calls this and gets the pace anyway: by DEFAULT (set_pace(None), the state a Context is
        created in) the loop thread paces itself by whichever replay of its training loops is being sampled (`auto_pace_steps` per
        sample, default 2 = the measured best operating point) and free-runs whenever nobody has sampled for `auto_pace_idle_s` --
        burn-in, evaluation pauses, a driver that only collects data.  set_pace(replay, k) pins the replay and the ratio (and then
        WAITS for samples instead of falling back to free-running); set_pace(False) is the reference's unconditional free-running."""
        self._auto = None
        if replay is False:
            self._pace, self._run_ahead, self.auto_pace_steps = None, 0, 0.0
            return
        if replay is None and self.auto_pace_steps <= 0:
            self.auto_pace_steps = self._auto_default      # back to the default behaviour, also after a set_pace(False)
        self._pace = None if replay is None else (replay, float(steps_per_sample), [None, 0.0])
        self._run_ahead = int(run_ahead) if replay is not None else 0

    def _auto_pace(self):
        """the default pace: -> (replay, steps per sample, credit state) while a training loop is sampling a replay that the attached loops
        feed, else None (free-running)"""
        if self.auto_pace_steps <= 0:
            return None
        if self._auto_replays is None:
            seen = []
            for lp in self.loops:
                if getattr(lp, "eval_mode", True):
                    continue
                for a in getattr(lp, "actors", []):
                    rp = getattr(a, "replay", None)
                    if rp is not None and hasattr(rp, "num_sample") and not any(rp is x for x in seen):
                        seen.append(rp)
            self._auto_replays = seen
        now = _time.monotonic()
        for rp in self._auto_replays:
            if rp.num_sample > 0 and now - rp.last_sample_time < self.auto_pace_idle_s:
                if self._auto is None or self._auto[0] is not rp:
                    self._auto = (rp, self.auto_pace_steps, [None, 0.0])
                return self._auto
        self._auto = None
        return None

    def _may_step(self):
        """pace gate of the loop thread: True = issue a step now"""
        pace = self._pace if self._pace is not None else self._auto_pace()
        bound = self._run_ahead if self._pace is not None else (2 if pace is not None else 0)
        self._bound = bound
        if bound > 0 and len(self._marks) > bound:
            if not self._marks[0].query():
                return False
            self._marks.pop(0)
        elif bound == 0 and self._marks:
            del self._marks[:]
        if pace is None:
            return True
        replay, per, st = pace
        n = replay.num_sample
        if n == 0:
            return True                          # the training loop has not started: burn-in
        if st[0] is None:
            st[0] = n - 1                        # credits count from the first sample() the thread notices
        if st[1] >= (n - st[0]) * per:
            return False
        st[1] += 1.0
        return True

    def push_env_thread(self, loop):
        """Context::pushThreadLoop (rela/context.h:30-35).  A training loop whose model runner sits on another GPU than the replay it
        feeds is refused HERE (MultiDeviceError), not at its first step on the Context thread."""
        if not hasattr(loop, "step"):
            raise TypeError("push_env_thread: %r is not a rela.ThreadLoop (no step())" % (loop,))
        dev = _loop_device(loop)
        if dev is not None and not getattr(loop, "eval_mode", True):
            for a in getattr(loop, "actors", []):
                rp = getattr(a, "replay", None)
                if rp is None:
                    continue
                have = getattr(rp, "device", None) or getattr(rp, "_claimed", None)
                if have is None:
                    rp._claimed = dev            # the first training loop pushed for a replay decides where the replay will live
                elif have != dev:
                    raise MultiDeviceError(have, dev)
        self.loops.append(loop)
        self._auto_replays = None                # the default pace looks at the replays of ALL attached loops: also of one pushed late
        return len(self.loops)

    def _run(self):
        import time
        try:
            while not self._stop:
                if self._paused:
                    with self._cv:
                        if self._paused and self._parked_ticket != self._pause_ticket:
                            self._parked_ticket = self._pause_ticket
                            self._cv.notify_all()
                    time.sleep(0.001)
                    continue
                if not self._may_step():
                    time.sleep(0)                # yield: the driver's thread is issuing the update this step waits for
                    continue
                busy = False
                for lp in self.loops:
                    if self._paused or self._stop:
                        break
                    if not (hasattr(lp, "finished") and lp.finished()):
                        st = self._streams.get(_loop_device(lp))        # every loop issues on the rollout stream of ITS device
                        with _MODEL_LOCK:
                            if st is not None:
                                with torch.cuda.stream(st):
                                    lp.step()
                            else:
                                lp.step()
                        busy = True
                if self._bound > 0 and self._streams:
                    # ONE mark per iteration, whatever the number of device streams (the run-ahead bound counts steps): the events of
                    # all rollout streams of this iteration, done when all of them are
                    evs = []
                    for st in self._streams.values():
                        e = torch.cuda.Event()
                        e.record(st)
                        evs.append(e)
                    self._marks.append(_MarkGroup(evs))
                if not busy and not self._paused:
                    break
        except Exception as e:   # surfaced by the next Context call from the driver's thread
            self._error = e
        finally:
            with _MODEL_LOCK:
                for key in self._streams:
                    _LIVE_CONTEXTS[key] = max(0, _LIVE_CONTEXTS.get(key, 0) - 1)
            with self._cv:
                self._done = True
                self._cv.notify_all()

    def _check(self):
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def _coalesce(self):
        """merge compatible hanalearn.HanabiThreadLoops (same models and configuration, consecutive seeds) into one batched
        device loop each: the reference pushes one loop per thread -- or per game, in eval.py -- and every one of them would
        otherwise be its own set of kernel launches"""
        last = {}
        for lp in self.loops:
            if not hasattr(lp, "can_absorb") or lp._built or lp.master is not None:
                continue
            key = lp.merge_key()
            head = last.get(key)
            if head is not None and head.can_absorb(lp):
                head.absorb(lp)
            else:
                last[key] = lp

    def _devices(self):
        """the GPUs the attached loops act on, in push order (eval loops and cross-play may sit on several; training loops are held
        to their replay's device by push_env_thread)"""
        out = []
        for lp in self.loops:
            d = _loop_device(lp)
            if d is not None and d not in out:
                out.append(d)
        return out

    def _join_streams(self, to_actor):
        """device-side order between the caller's current stream and the rollout stream of every device (no host synchronisation)"""
        for st in self._streams.values():
            cur = torch.cuda.current_stream(st.device)
            if to_actor:
                st.wait_stream(cur)
            else:
                cur.wait_stream(st)

    def start(self):
        import threading
        self._coalesce()
        global _PACE_NOTICE_SHOWN
        if (self._thread is None and not _PACE_NOTICE_SHOWN and self._pace is None and self.auto_pace_steps > 0 and not os.environ.get("HSAD_QUIET")
                and any(not getattr(lp, "eval_mode", True) for lp in self.loops)):
            # the one behavioural delta an unchanged reference driver gets without asking (rela/context.h:43-50 free-runs): say so, once
            _PACE_NOTICE_SHOWN = True
            print("hanabi_sad_amd.rela.Context: rollout paced by replay.sample() -- %g rollout steps per sample once the training loop samples, "
                  "free-running before that and whenever nobody has sampled for %.1f s.  The reference's Context free-runs unconditionally: "
                  "context.set_pace(False) or HSAD_AUTO_PACE=0 for that (HSAD_QUIET=1 silences this line)."
                  % (self.auto_pace_steps, self.auto_pace_idle_s), file=sys.stderr, flush=True)
        if self._thread is None:
            for dev in self._devices():
                if torch.device(dev).type == "cuda":
                    self._streams[dev] = _actor_stream(dev)
            self._join_streams(True)          # everything the driver set up so far (envs, nets, replay) precedes the first step
            with _MODEL_LOCK:
                for key in self._streams:
                    _LIVE_CONTEXTS[key] = _LIVE_CONTEXTS.get(key, 0) + 1
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def step(self):
        """advance every attached loop by one lock-step iteration from the caller's thread (alternative to start())"""
        self._coalesce()
        for lp in self.loops:
            if not (hasattr(lp, "finished") and lp.finished()):
                with _MODEL_LOCK:
                    lp.step()

    def pause(self):
        """blocks until the loop thread is parked between two steps, like the reference's pause (rela/context.h:52-60 waits
        for every ThreadLoop to reach waitUntilResume)"""
        self._check()
        with self._cv:
            self._pause_ticket += 1
            ticket = self._pause_ticket
            self._paused = True
            if self._thread is not None and self._thread.is_alive():
                self._cv.wait_for(lambda: self._done or self._parked_ticket == ticket)
        self._join_streams(False)             # what the caller reads next (env state, scores, replay) is behind the last rollout step
        self._check()

    def resume(self):
        self._check()
        self._join_streams(True)
        with self._cv:
            self._paused = False

    def terminate(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join()
        self._join_streams(False)
        self._check()

    def terminated(self):
        self._check()
        if self._thread is not None and not self._thread.is_alive():
            return True
        return self._stop or (bool(self.loops) and all(hasattr(lp, "finished") and lp.finished() for lp in self.loops))
