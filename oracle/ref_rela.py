"""ORACLE — TEST INFRASTRUCTURE ONLY: ctypes wrapper around oracle/_ref/libref_rela.so, i.e. the REAL
reference rela/ classes (aggregatePriority, MultiStepBuffer, R2D2Buffer, PrioritizedReplay<RNNTransition>)
compiled from /root/reference by oracle/build_ref.sh.  kind = "reference" in the sense of the task statement.
Only tests/ (and smoke/bench baselines) may import this."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libref_rela.so")
_L = None


def available():
    return os.path.exists(PATH)


def lib():
    global _L
    if _L is None:
        import torch  # noqa: F401  (loads libtorch/libc10 so the harness's dependencies resolve)
        _L = C.CDLL(PATH)
        _L.ref_msb_create.restype = C.c_void_p
        _L.ref_r2d2buf_create.restype = C.c_void_p
        _L.ref_replay_create.restype = C.c_void_p
        _L.ref_replay_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
        _L.ref_msb_create.argtypes = [C.c_int, C.c_int, C.c_float]
    return _L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def aggregate_priority(priority, seq_len, eta):
    priority = np.ascontiguousarray(priority, np.float32)
    seq_len = np.ascontiguousarray(seq_len, np.float32)
    T, B = priority.shape
    out = np.zeros(B, np.float32)
    lib().ref_aggregate_priority(_p(priority), _p(seq_len), T, B, C.c_float(eta), _p(out))
    return out


class MultiStepBuffer:
    def __init__(self, multi_step, batchsize, gamma, d):
        self.h = C.c_void_p(lib().ref_msb_create(multi_step, batchsize, gamma))
        self.E, self.d = batchsize, d

    def push_obs_action(self, obs, a):
        obs = np.ascontiguousarray(obs, np.float32)
        a = np.ascontiguousarray(a, np.int64)
        lib().ref_msb_push_obs_action(self.h, _p(obs), self.E, self.d, _p(a))

    def push_reward_terminal(self, r, t):
        r = np.ascontiguousarray(r, np.float32)
        t = np.ascontiguousarray(t, np.uint8)
        lib().ref_msb_push_reward_terminal(self.h, _p(r), _p(t), self.E)

    def can_pop(self):
        return bool(lib().ref_msb_can_pop(self.h))

    def pop(self):
        E, d = self.E, self.d
        out = dict(obs=np.zeros((E, d), np.float32), a=np.zeros(E, np.int64), reward=np.zeros(E, np.float32),
                   terminal=np.zeros(E, np.uint8), bootstrap=np.zeros(E, np.float32),
                   next_obs=np.zeros((E, d), np.float32))
        lib().ref_msb_pop(self.h, _p(out["obs"]), _p(out["a"]), _p(out["reward"]), _p(out["terminal"]),
                          _p(out["bootstrap"]), _p(out["next_obs"]))
        return out


class R2D2Buffer:
    def __init__(self, batchsize, num_player, multi_step, seq_len, d):
        self.h = C.c_void_p(lib().ref_r2d2buf_create(batchsize, num_player, multi_step, seq_len))
        self.E, self.T, self.d = batchsize, seq_len, d

    def push(self, obs, a, reward, terminal, bootstrap, next_obs, priority):
        f = lambda x: np.ascontiguousarray(x, np.float32)
        obs, reward, bootstrap, next_obs, priority = map(f, (obs, reward, bootstrap, next_obs, priority))
        a = np.ascontiguousarray(a, np.int64)
        terminal = np.ascontiguousarray(terminal, np.uint8)
        lib().ref_r2d2buf_push(self.h, _p(obs), self.E, self.d, _p(a), _p(reward), _p(terminal), _p(bootstrap),
                               _p(next_obs), _p(priority))

    def can_pop(self):
        return bool(lib().ref_r2d2buf_can_pop(self.h))

    def pop(self):
        E, T, d = self.E, self.T, self.d
        obs = np.zeros((E, T, d), np.float32)
        a = np.zeros((E, T), np.int64)
        reward = np.zeros((E, T), np.float32)
        terminal = np.zeros((E, T), np.uint8)
        bootstrap = np.zeros((E, T), np.float32)
        seq_len = np.zeros(E, np.float32)
        prio = np.zeros(T * E, np.float32)
        n = lib().ref_r2d2buf_pop(self.h, E, _p(obs), _p(a), _p(reward), _p(terminal), _p(bootstrap), _p(seq_len),
                                  _p(prio))
        assert n > 0
        return dict(n=n, obs=obs[:n], a=a[:n], reward=reward[:n], terminal=terminal[:n], bootstrap=bootstrap[:n],
                    seq_len=seq_len[:n], priority=prio[:T * n].reshape(T, n))


class Replay:
    """rela.RNNPrioritizedReplay (prefetch must be 0 for deterministic single-threaded driving)."""

    def __init__(self, capacity, seed, alpha, beta, T, d, prefetch=0):
        self.h = C.c_void_p(lib().ref_replay_create(capacity, seed, alpha, beta, prefetch))
        self.T, self.d = T, d

    def add(self, obs, a, reward, terminal, bootstrap, seq_len, priority):
        f = lambda x: np.ascontiguousarray(x, np.float32)
        obs, reward, bootstrap, seq_len, priority = map(f, (obs, reward, bootstrap, seq_len, priority))
        a = np.ascontiguousarray(a, np.int64)
        terminal = np.ascontiguousarray(terminal, np.uint8)
        n = obs.shape[0]
        lib().ref_replay_add(self.h, n, self.T, self.d, _p(obs), _p(a), _p(reward), _p(terminal), _p(bootstrap),
                             _p(seq_len), _p(priority))

    def size(self):
        return lib().ref_replay_size(self.h)

    def num_add(self):
        return lib().ref_replay_num_add(self.h)

    def sample(self, B):
        T, d = self.T, self.d
        out = dict(obs=np.zeros((T, B, d), np.float32), a=np.zeros((T, B), np.int64),
                   reward=np.zeros((T, B), np.float32), terminal=np.zeros((T, B), np.uint8),
                   bootstrap=np.zeros((T, B), np.float32), seq_len=np.zeros(B, np.float32),
                   weight=np.zeros(B, np.float32))
        lib().ref_replay_sample(self.h, B, _p(out["obs"]), _p(out["a"]), _p(out["reward"]), _p(out["terminal"]),
                                _p(out["bootstrap"]), _p(out["seq_len"]), _p(out["weight"]))
        return out

    def update_priority(self, p):
        p = np.ascontiguousarray(p, np.float32)
        lib().ref_replay_update_priority(self.h, _p(p), len(p))

    def get(self, idx):
        obs = np.zeros((self.T, self.d), np.float32)
        sl = np.zeros(1, np.float32)
        lib().ref_replay_get(self.h, idx, _p(obs), _p(sl))
        return obs, float(sl[0])
