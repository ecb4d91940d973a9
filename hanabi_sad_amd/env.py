"""Batched device Hanabi environment: host-side owner of the tensors that libhsad's env kernels
write.  One object = G games = what the reference builds as G `hanalearn.HanabiEnv` objects inside
`HanabiVecEnv`s (pyhanabi/create.py:24-54; rela/env.h:29-108)."""
import ctypes as C

import torch

from . import _lib


class BatchedHanabiEnv:
    def __init__(self, num_games, players=2, hand_size=5, seed=1, bomb=0, eps_list=(0.0,), max_len=80, sad=False,
                 shuffle_obs=False, shuffle_color=False, knowledge_mode=0, device="cuda:0", track_deck_history=True,
                 deal_mode=0, games_per_workgroup=0, threads_per_workgroup=0):
        self.lib = _lib.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HsadError("BatchedHanabiEnv needs a ROCm device (got %s); there is no CPU path" % device)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        eps = (C.c_float * len(eps_list))(*[float(e) for e in eps_list])
        cfg = _lib.EnvConfig(num_games, players, hand_size, int(bomb), int(seed), int(max_len), int(bool(sad)),
                             int(bool(shuffle_obs)), int(bool(shuffle_color)), int(knowledge_mode), len(eps_list),
                             dev_index, int(bool(track_deck_history)), int(deal_mode), int(games_per_workgroup), eps)
        self.h = C.c_void_p()
        _lib.check(self.lib.hsad_env_create(C.byref(cfg), C.byref(self.h)))
        L = self.lib
        self.G, self.P, self.H = num_games, players, hand_size
        self.F = L.hsad_env_feature_size(self.h)
        self.A = L.hsad_env_num_action(self.h)
        self.sad = bool(sad)
        self.knowledge_mode = int(knowledge_mode)
        self.games_per_workgroup = L.hsad_env_games_per_workgroup(self.h)   # kernel shape in use (32 | 64)
        if threads_per_workgroup:
            _lib.check(L.hsad_env_set_threads_per_workgroup(self.h, int(threads_per_workgroup)))
        self.threads_per_workgroup = L.hsad_env_threads_per_workgroup(self.h)   # 128 | 256
        d = self.device
        self.priv_s = torch.zeros(self.G, self.P, self.F, dtype=torch.float32, device=d)
        self.legal_move = torch.zeros(self.G, self.P, self.A, dtype=torch.float32, device=d)
        self.own_hand = torch.zeros(self.G, self.P, 3 * self.H, dtype=torch.float32, device=d)
        self.eps = torch.zeros(self.G, self.P, dtype=torch.float32, device=d)
        self.reward = torch.zeros(self.G, dtype=torch.float32, device=d)
        self.terminal = torch.zeros(self.G, dtype=torch.uint8, device=d)
        self.a = torch.zeros(self.G, self.P, dtype=torch.int64, device=d)
        self.greedy_a = torch.zeros(self.G, self.P, dtype=torch.int64, device=d)
        _lib.check(L.hsad_env_bind_outputs(self.h, self.priv_s.data_ptr(), self.legal_move.data_ptr(),
                                           self.own_hand.data_ptr(), self.eps.data_ptr(), self.reward.data_ptr(),
                                           self.terminal.data_ptr()))

    def enable_packed(self, bf16_row_len=0, keep_float32=True):
        """outputs for device consumers (hsad_env_bind_packed): priv_bits int64 [G,P,ceil(F/64)], legal_bits / own_bits int64 [G,P]
        (bit j = column j of the float32 tensor) and, with bf16_row_len, priv_s_bf16 [G,P,row_len] zero-padded.
        keep_float32=False: the float32 priv_s tensor is no longer written (self.priv_s becomes None)"""
        d, G, P = self.device, self.G, self.P
        self.priv_bits = torch.zeros(G, P, (self.F + 63) // 64, dtype=torch.int64, device=d)
        self.legal_bits = torch.zeros(G, P, dtype=torch.int64, device=d)
        self.own_bits = torch.zeros(G, P, dtype=torch.int64, device=d)
        self.priv_s_bf16 = torch.zeros(G, P, bf16_row_len, dtype=torch.bfloat16, device=d) if bf16_row_len else None
        _lib.check(self.lib.hsad_env_bind_packed(self.h, self.priv_bits.data_ptr(), self.legal_bits.data_ptr(), self.own_bits.data_ptr(),
                                                 self.priv_s_bf16.data_ptr() if bf16_row_len else None, int(bf16_row_len),
                                                 int(bool(keep_float32))))
        if not keep_float32:
            self.priv_s = None

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.hsad_env_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference-shaped accessors (cpp/hanabi_env.h:53-72) --
    def feature_size(self):
        return self.F

    def num_action(self):
        return self.A

    def hand_feature_size(self):
        return self.lib.hsad_env_hand_feature_size(self.h)

    def state_bytes(self):
        return int(self.lib.hsad_env_state_bytes(self.h))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def obs(self):
        """TensorDict view produced by VectorEnv::reset/step (rela/env.h:48-87)."""
        if self.priv_s is None:   # enable_packed(keep_float32=False): the observation only exists as bit words / bf16 rows
            return {"priv_s_bf16": self.priv_s_bf16, "priv_bits": self.priv_bits, "legal_move": self.legal_move, "eps": self.eps,
                    "own_hand": self.own_hand}
        return {"priv_s": self.priv_s, "legal_move": self.legal_move, "eps": self.eps, "own_hand": self.own_hand}

    def reset(self):
        _lib.check(self.lib.hsad_env_reset(self.h, self._stream()))
        return self.obs()

    def step(self, a, greedy_a=None):
        assert a.dtype == torch.int64 and a.is_contiguous() and a.device == self.legal_move.device
        g = greedy_a if greedy_a is not None else (a if self.sad else None)
        _lib.check(self.lib.hsad_env_step(self.h, a.data_ptr(), g.data_ptr() if g is not None else None,
                                          self._stream()))
        return self.obs(), self.reward, self.terminal

    def policy_random(self, policy_seed):
        _lib.check(self.lib.hsad_env_policy_random(self.h, policy_seed, self.a.data_ptr(), self.greedy_a.data_ptr(),
                                                   self._stream()))
        return self.a, self.greedy_a

    def rollout_random(self, n_iter, policy_seed):
        _lib.check(self.lib.hsad_env_rollout_random(self.h, n_iter, policy_seed, self.a.data_ptr(),
                                                    self.greedy_a.data_ptr(), self._stream()))

    def set_partitions(self, n_part):
        """Number of independent game ranges rollout_random overlaps on private HIP streams."""
        _lib.check(self.lib.hsad_env_set_partitions(self.h, int(n_part)))

    def set_rollout_stagger(self, microseconds):
        """phase lock between the partition chains (see include/hsad.h); timing only"""
        _lib.check(self.lib.hsad_env_set_rollout_stagger(self.h, int(microseconds)))

    def set_rollout_chunk(self, iterations_per_launch):
        """persistent rollout: one launch runs this many iterations of every game (0 = one launch per iteration)"""
        _lib.check(self.lib.hsad_env_set_rollout_chunk(self.h, int(iterations_per_launch)))

    def last_rollout_ms(self):
        """average launch duration (ms) on each partition stream of the last partitioned rollout_random"""
        import ctypes as C
        ms, n = (C.c_float * 16)(), C.c_int(0)
        _lib.check(self.lib.hsad_env_last_rollout_ms(self.h, ms, C.byref(n)))
        return [float(ms[k]) for k in range(n.value)]

    def query(self):
        out = torch.zeros(self.G, 16, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hsad_env_query(self.h, out.data_ptr(), self._stream()))
        return out

    def move_is_legal(self, uid):
        uid = uid.to(self.device, torch.int32).contiguous()
        out = torch.zeros(self.G, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.hsad_env_move_is_legal(self.h, uid.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def deck_history(self):
        out = torch.zeros(self.G, 50, dtype=torch.uint8, device=self.device)
        cnt = torch.zeros(self.G, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hsad_env_deck_history(self.h, out.data_ptr(), cnt.data_ptr(), self._stream()))
        return out, cnt

    def export_state(self):
        w = self.lib.hsad_env_state_words(self.h)
        out = torch.zeros(self.G, w, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hsad_env_export_state(self.h, out.data_ptr(), self._stream()))
        return out

    def check_errors(self):
        """Raises if any game hit what the reference treats as assert(false) (hanabi_env.cc:50,63-80)."""
        n, g, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _lib.check(self.lib.hsad_env_error_count(self.h, C.byref(n), C.byref(g), C.byref(c)))
        if n.value:
            what = {1: "illegal move", 2: "illegal greedy move", 3: "step on a finished game"}.get(c.value, "?")
            raise _lib.HsadError("%d game(s) violated the env contract; first: game %d, %s" % (n.value, g.value, what))
