"""ORACLE — TEST INFRASTRUCTURE ONLY (ctypes wrapper around oracle/liboracle_hanabi.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (hanabi_sad_amd/) never does.

The library is the CPU restatement of the reference HanabiEnv (cpp/hanabi_env.cc:9-205) on top of a
restatement of the absent HLE engine; see the header of hanabi_oracle.cc for what pins it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(fast=False):
    target = "liboracle_hanabi_fast.so" if fast else "liboracle_hanabi.so"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return os.path.join(_HERE, target)


def lib(fast=False):
    global _LIB
    key = "fast" if fast else "std"
    if _LIB is None:
        _LIB = {}
    if key in _LIB:
        return _LIB[key]
    name = "liboracle_hanabi_fast.so" if fast else "liboracle_hanabi.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build(fast)
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    ip64 = C.POINTER(C.c_int64)
    L.orc_env_create.restype = C.c_void_p
    L.orc_env_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int]
    L.orc_env_destroy.argtypes = [C.c_void_p]
    for f in ("feature_size", "num_action", "hand_feature_size", "terminated", "cur_player", "last_score",
              "score", "life", "info", "num_step"):
        fn = getattr(L, "orc_env_" + f)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p]
    L.orc_env_fireworks.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.orc_env_move_is_legal.restype = C.c_int
    L.orc_env_move_is_legal.argtypes = [C.c_void_p, C.c_int]
    L.orc_env_rng_draws.restype = C.c_uint64
    L.orc_env_rng_draws.argtypes = [C.c_void_p]
    L.orc_env_deck_history.restype = C.c_int
    L.orc_env_deck_history.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
    L.orc_env_reset.argtypes = [C.c_void_p, fp, fp, fp, fp]
    L.orc_env_step.restype = C.c_int
    L.orc_env_step.argtypes = [C.c_void_p, ip64, ip64, fp, fp, fp, fp, fp, C.POINTER(C.c_uint8)]
    L.orc_env_state_words.restype = C.c_int
    L.orc_env_state_words.argtypes = [C.c_int, C.c_int]
    L.orc_env_export_state.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.orc_policy_hash.restype = C.c_uint32
    L.orc_policy_hash.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.orc_policy_random.argtypes = [fp, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, ip64, ip64]
    L.orc_vec_rollout.restype = C.c_int64
    L.orc_vec_rollout.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_uint64, ip64, ip64, fp, fp, fp, fp,
                                  fp, C.POINTER(C.c_uint8), ip64, ip64, ip64, ip64]
    _LIB[key] = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class OracleEnv:
    """One game; mirrors hanalearn.HanabiEnv (reference cpp/pybind.cc:16-38)."""

    def __init__(self, players=2, hand_size=5, seed=1, bomb=0, eps_list=(0.0,), max_len=80, sad=False,
                 shuffle_obs=False, shuffle_color=False, knowledge_mode=0, fast=False):
        self.L = lib(fast)
        eps = np.asarray(eps_list, dtype=np.float32)
        self.h = self.L.orc_env_create(players, hand_size, seed, bomb, _fp(eps), len(eps), max_len, int(sad),
                                       int(shuffle_obs), int(shuffle_color), knowledge_mode)
        if not self.h:
            raise ValueError("orc_env_create rejected the configuration")
        self.P, self.H = players, hand_size
        self.F = self.L.orc_env_feature_size(self.h)
        self.A = self.L.orc_env_num_action(self.h)
        self.priv_s = np.zeros((players, self.F), np.float32)
        self.legal = np.zeros((players, self.A), np.float32)
        self.own_hand = np.zeros((players, hand_size * 3), np.float32)
        self.eps = np.zeros((players,), np.float32)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_env_destroy(self.h)
            self.h = None

    def obs(self):
        return {"priv_s": self.priv_s.copy(), "legal_move": self.legal.copy(), "own_hand": self.own_hand.copy(),
                "eps": self.eps.copy()}

    def reset(self):
        self.L.orc_env_reset(self.h, _fp(self.priv_s), _fp(self.legal), _fp(self.own_hand), _fp(self.eps))
        return self.obs()

    def step(self, a, greedy_a=None):
        a = np.ascontiguousarray(a, dtype=np.int64)
        g = a if greedy_a is None else np.ascontiguousarray(greedy_a, dtype=np.int64)
        r = C.c_float(0)
        t = C.c_uint8(0)
        rc = self.L.orc_env_step(self.h, _ip(a), _ip(g), _fp(self.priv_s), _fp(self.legal), _fp(self.own_hand),
                                 _fp(self.eps), C.byref(r), C.byref(t))
        if rc != 0:
            raise RuntimeError("illegal move (rc=%d)" % rc)
        return self.obs(), float(r.value), bool(t.value)

    def terminated(self):
        return bool(self.L.orc_env_terminated(self.h))

    def cur_player(self):
        return self.L.orc_env_cur_player(self.h)

    def get(self, what):
        return getattr(self.L, "orc_env_" + what)(self.h)

    def fireworks(self):
        out = (C.c_int * 5)()
        self.L.orc_env_fireworks(self.h, out)
        return list(out)

    def move_is_legal(self, uid):
        return bool(self.L.orc_env_move_is_legal(self.h, uid))

    def rng_draws(self):
        return int(self.L.orc_env_rng_draws(self.h))

    def deck_history(self):
        buf = (C.c_uint8 * 64)()
        n = self.L.orc_env_deck_history(self.h, buf, 64)
        return list(buf[:n])

    def export_state(self):
        n = self.L.orc_env_state_words(self.P, self.H)
        out = np.zeros((n,), np.int32)
        self.L.orc_env_export_state(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out


def policy_random(legal, policy_seed, game, counter):
    """legal: [P, A] float32 -> (a[P], greedy_a[P]) int64, same spec as the device policy kernel."""
    L = lib()
    legal = np.ascontiguousarray(legal, np.float32)
    P, A = legal.shape
    a = np.zeros((P,), np.int64)
    g = np.zeros((P,), np.int64)
    L.orc_policy_random(_fp(legal), P, A, policy_seed, game, counter, _ip(a), _ip(g))
    return a, g


class OracleVecEnv:
    """E games stepped by the C loop (VectorEnv + thread-loop shape, random-legal policy)."""

    def __init__(self, n_env, seed, game_id0=0, fast=False, **kw):
        self.envs = [OracleEnv(seed=seed + game_id0 + i, fast=fast, **kw) for i in range(n_env)]
        self.L = lib(fast)
        e0 = self.envs[0]
        self.E, self.P, self.F, self.A, self.H = n_env, e0.P, e0.F, e0.A, e0.H
        self.handles = (C.c_void_p * n_env)(*[e.h for e in self.envs])
        self.game_ids = np.arange(game_id0, game_id0 + n_env, dtype=np.int64)
        self.counters = np.zeros((n_env,), np.int64)
        self.priv_s = np.zeros((n_env, self.P, self.F), np.float32)
        self.legal = np.zeros((n_env, self.P, self.A), np.float32)
        self.own_hand = np.zeros((n_env, self.P, self.H * 3), np.float32)
        self.eps = np.zeros((n_env, self.P), np.float32)
        self.reward = np.zeros((n_env,), np.float32)
        self.terminal = np.zeros((n_env,), np.uint8)
        self.a = np.zeros((n_env, self.P), np.int64)
        self.g = np.zeros((n_env, self.P), np.int64)
        self.score_sum = np.zeros((1,), np.int64)
        self.episodes = np.zeros((1,), np.int64)

    def rollout(self, n_iter, policy_seed):
        n = self.L.orc_vec_rollout(self.handles, self.E, n_iter, policy_seed, _ip(self.game_ids),
                                   _ip(self.counters), _fp(self.priv_s), _fp(self.legal), _fp(self.own_hand),
                                   _fp(self.eps), _fp(self.reward),
                                   self.terminal.ctypes.data_as(C.POINTER(C.c_uint8)), _ip(self.a), _ip(self.g),
                                   _ip(self.score_sum), _ip(self.episodes))
        if n < 0:
            raise RuntimeError("oracle rollout hit an illegal move after %d steps" % (-1 - n))
        return n
