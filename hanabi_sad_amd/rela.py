"""Drop-in names of the reference's `rela` pybind module (rela/pybind.cc:16-93) on the device pipeline.

`RNNPrioritizedReplay(capacity, seed, alpha, beta, prefetch)` keeps the reference signature; the transition
schema and sequence length are taken from the first `bind_schema()` / actor that attaches to it.  The thread
machinery (`Context`, `ThreadLoop`, `BatchRunner`, `R2D2Actor`) has no counterpart by design: all games advance
in lock-step on the GPU (hanabi_sad_amd.actor.DeviceActor), so `Context` only keeps the start/pause/terminate
surface for drivers written against the reference."""
import torch

from .replay import DeviceReplay, aggregate_priority as _aggregate_priority


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


class Context:
    """rela.Context surface (rela/context.h:18-80) over lock-step actors: start/pause/resume/terminate flags
    that a driver loop polls; there are no threads to manage."""

    def __init__(self):
        self.actors, self._started, self._paused, self._terminated = [], False, False, False

    def push_env_thread(self, actor):
        self.actors.append(actor)
        return len(self.actors)

    def start(self):
        self._started = True

    def pause(self):
        self._paused = True

    def resume(self):
        self._paused = False

    def terminate(self):
        self._terminated = True

    def terminated(self):
        return self._terminated

    def step(self):
        """advance every attached actor by one lock-step iteration (what the reference's threads do on their own)"""
        if self._started and not self._paused and not self._terminated:
            for a in self.actors:
                a.step()
