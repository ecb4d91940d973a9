"""Drop-in names of the reference's `rela` pybind module (rela/pybind.cc:16-93) on the device pipeline.

`RNNPrioritizedReplay(capacity, seed, alpha, beta, prefetch)` keeps the reference signature; the transition
schema and sequence length are taken from the first `bind_schema()` / actor that attaches to it.  The thread
machinery has no counterpart by design: all games advance in lock-step on the GPU (hanabi_sad_amd.actor.DeviceActor).
`BatchRunner` / `R2D2Actor` keep the constructor signatures and carry configuration; `Context.start()` runs the attached
`hanalearn.HanabiThreadLoop`s from ONE background Python thread (ctypes releases the GIL during launches), so a driver
written against the reference -- create_envs / create_threads / ActGroup / context.start() / replay.sample() -- works
unchanged; `Context.step()` is there for drivers that prefer to interleave rollout and learning themselves."""
import threading

import torch

from .replay import DeviceReplay, aggregate_priority as _aggregate_priority

# BatchRunner's mtxUpdate_ (rela/batch_runner.h:74-77,106-109): a weight update never interleaves with a rollout step that
# the Context thread is enqueueing (both sides enqueue on the same stream; a refresh() between two layers of an act() would
# run one step on mixed weights)
_MODEL_LOCK = threading.RLock()


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

    def bind_schema(self, fields, seq_len, device="cuda:0"):
        if self.impl is None:
            c, s, a, b, p = self.args
            self.impl = DeviceReplay(c, s, a, b, p, seq_len, fields, device)
        return self.impl

    def size(self):
        return 0 if self.impl is None else self.impl.size()

    def num_add(self):
        return 0 if self.impl is None else self.impl.num_add()

    def sample(self, batchsize, device=None):
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


class BatchRunner:
    """rela.BatchRunner(py_model, device, max_batchsize, methods) (rela/batch_runner.h:17-130): holds the acting copy of the
    agent.  `py_model` is anything with state_dict() carrying `online_net.*` / `target_net.*` (the reference R2D2Agent) or a
    plain dict of such tensors.  There is nothing to batch across threads, so start()/stop() are no-ops."""

    def __init__(self, py_model, device, max_batchsize=100, methods=None):
        self.device = device
        self.online = self.target = None
        self.update_model(py_model)

    @staticmethod
    def _split(py_model):
        sd = py_model.state_dict() if hasattr(py_model, "state_dict") else dict(py_model)
        on = {k[len("online_net."):]: v for k, v in sd.items() if k.startswith("online_net.")}
        tg = {k[len("target_net."):]: v for k, v in sd.items() if k.startswith("target_net.")}
        if not on:
            raise KeyError("BatchRunner: state_dict has no online_net.* entries")
        return on, (tg or on)

    def update_model(self, py_model):
        """BatchRunner::updateModel (rela/batch_runner.h:74-77)"""
        from .r2d2 import R2D2NetKernels
        on, tg = self._split(py_model)
        with _MODEL_LOCK:
            if self.online is None:
                self.online, self.target = R2D2NetKernels(on, self.device), R2D2NetKernels(tg, self.device)
                return
            for net, sd in ((self.online, on), (self.target, tg)):
                for k, v in sd.items():
                    net.w[k].copy_(v)
                net.refresh()

    def start(self):
        pass

    def stop(self):
        pass


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
        return self._num_act


class Context:
    """rela.Context (rela/context.h:18-80): push_env_thread / start / pause / resume / terminate / terminated.
    start() runs every attached loop from one background thread until terminate(); eval loops end by themselves, and
    terminated() turns True once all attached loops have finished (what eval.py:47-51 polls)."""

    def __init__(self):
        self.loops, self._thread, self._paused, self._stop, self._error = [], None, False, False, None
        self._parked = threading.Event()      # set by the loop thread while it sits between two steps with _paused seen

    def push_env_thread(self, loop):
        self.loops.append(loop)
        return len(self.loops)

    def _run(self):
        import time
        try:
            while not self._stop:
                if self._paused:
                    self._parked.set()
                    time.sleep(0.001)
                    continue
                self._parked.clear()
                busy = False
                for lp in self.loops:
                    if self._paused or self._stop:
                        break
                    if not (hasattr(lp, "finished") and lp.finished()):
                        with _MODEL_LOCK:
                            lp.step()
                        busy = True
                if not busy and not self._paused:
                    break
        except Exception as e:   # surfaced by the next Context call from the driver's thread
            self._error = e
        finally:
            self._parked.set()

    def _check(self):
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def start(self):
        import threading
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def step(self):
        """advance every attached loop by one lock-step iteration from the caller's thread (alternative to start())"""
        for lp in self.loops:
            if not (hasattr(lp, "finished") and lp.finished()):
                with _MODEL_LOCK:
                    lp.step()

    def pause(self):
        """blocks until the loop thread is parked between two steps, like the reference's pause (rela/context.h:52-60 waits
        for every ThreadLoop to reach waitUntilResume)"""
        self._check()
        self._paused = True
        if self._thread is not None and self._thread.is_alive():
            self._parked.wait()
        self._check()

    def resume(self):
        self._check()
        self._parked.clear()
        self._paused = False

    def terminate(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join()
        self._check()

    def terminated(self):
        self._check()
        if self._thread is not None and not self._thread.is_alive():
            return True
        return self._stop or (bool(self.loops) and all(hasattr(lp, "finished") and lp.finished() for lp in self.loops))
