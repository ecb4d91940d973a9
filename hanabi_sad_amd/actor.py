"""Device-resident actor: the reference's R2D2Actor + HanabiThreadLoop (rela/r2d2_actor.h:23-172,
cpp/thread_loop.h:42-88) for ALL games in lock-step on one GPU — no threads, no batcher, no host copies.

IQL layout (reference create.py:115-131: one actor per player): every (game, player) pair is one "env row"
of the agent / sequence writer; reward and terminal are shared by the players of a game.
VDN layout (create.py:98-113: one actor sees [E, P, ·]): the agent still acts on (game, player) rows, but a transition
is one GAME (fields of width P·w, one reward / terminal / priority per game) and priorities sum Q over the players."""
from collections import deque

import torch

from .env import BatchedHanabiEnv
from .r2d2 import R2D2Agent, zero_hidden_rows
from .replay import Bits, DeviceReplay, SequenceWriter


def transition_fields(env, vdn=False):
    """per-step fields of an RNNTransition: obs {priv_s, legal_move, eps, own_hand} + action {a, greedy_a}
    (cpp/hanabi_env.cc:197-204; pyhanabi/r2d2.py:296-303).  IQL: one transition per (game, player); VDN: one per
    game with the players' rows concatenated ([P, w] flattened)."""
    m = env.P if vdn else 1
    # every plane of the observation, the legal-move mask and the own-hand target are 0/1: stored as bits (32x less HBM per
    # stored step and per sampled batch).  The V0-belief encoding (knowledge_mode 1) holds count ratios and stays float32.
    binary = Bits(m)
    obs_dt = binary if getattr(env, "knowledge_mode", 0) == 0 else torch.float32
    return [("priv_s", m * env.F, obs_dt), ("legal_move", m * env.A, binary), ("eps", m, torch.float32),
            ("own_hand", m * 3 * env.H, binary), ("a", m, torch.int64), ("greedy_a", m, torch.int64)]


class DeviceActor:
    def __init__(self, env: BatchedHanabiEnv, agent: R2D2Agent, replay: DeviceReplay, multi_step, gamma, eta, seq_len,
                 vdn=False, packed_obs=None, native=None):
        self.env, self.agent, self.replay = env, agent, replay
        self.G, self.P = env.G, env.P
        self.N = self.G * self.P              # agent rows (hidden state [L, N, H]) in both layouts
        self.vdn = bool(vdn)
        self.E = self.G if self.vdn else self.N   # transitions per step
        self.eta = float(eta)
        self.multi_step = int(multi_step)
        self.writer = SequenceWriter(self.E, multi_step, gamma, seq_len, transition_fields(env, self.vdn), env.device)
        self.cached_q = getattr(agent, "cached_q", True)      # False: a contract model (rela.ContractAgent): reference flow
        # packed observation path: the env kernel writes the observation as the replay's bit words and as the net's bf16 operand
        # from its on-chip bit rows; the float32 observation (the reference's API-boundary format) is not written at all, and
        # neither the bf16 cast nor the row-pack pass runs.  Needs an agent that takes bf16 input (composite.CompositeAgent).
        if packed_obs is None:
            packed_obs = bool(getattr(agent, "accepts_bf16_obs", False)) and self.cached_q and getattr(env, "knowledge_mode", 0) == 0
        self.packed_obs = bool(packed_obs)
        if self.packed_obs:
            env.enable_packed(agent.online.Fp, keep_float32=False)
            self.writer.set_prepacked(("priv_s", "legal_move", "own_hand"))
        if hasattr(agent, "configure"):
            agent.configure(self.P, self.vdn)
        self.hid = agent.get_h0(self.N)
        self.history_hid = deque()
        self.q_hist = deque()                 # (Q_online(s_t, a_t), online weight version) per step still in the n-step window
        self._verify_cached_priority = False   # tests: also run compute_priority on the unpacked transition and compare
        self.n_checked = self.n_checked_stale = 0
        self.num_act = 0          # R2D2Actor::numAct summed over the P per-player actors
        self.n_finished = torch.zeros(1, dtype=torch.int32, device=env.device)
        self._side = torch.cuda.Stream(env.device) if self.cached_q else None      # reset of ended games, see _reset_terminated
        self._side_flush = torch.cuda.Stream(env.device) if self.cached_q else None   # flush of finished sequences, see _flush
        self._reset_pending = False
        # The loop body itself lives in the library (include/hsad.h hsad_actor_*, csrc/hsad_actor.hip): with the library's composite
        # agent and the packed observation path, step() is ONE C call and this class only holds the objects it drives.  The Python
        # body below remains for models behind the reference's contract (rela.ContractAgent), for the Python-orchestrated agent
        # and for the priority cross-check of the tests (verify_cached_priority).
        self.c_actor = None
        if native is None:
            native = self.packed_obs and hasattr(agent, "online") and hasattr(agent.online, "h") and hasattr(agent, "lib") and not getattr(agent.online, "skip", False)
        if native:
            self._make_native(multi_step, gamma, seq_len)

    @property
    def verify_cached_priority(self):
        return self._verify_cached_priority

    @verify_cached_priority.setter
    def verify_cached_priority(self, on):
        """the cross-check runs in the Python loop body, which pushes into this object's own sequence writer -- a native actor has
        handed that role (and closed that writer) to the library's loop: refuse instead of stepping on a freed handle"""
        if on and self.c_actor is not None:
            from . import _lib
            raise _lib.HsadError("verify_cached_priority needs the Python loop body: construct the DeviceActor with native=False "
                                 "(selfplay: --native_actor 0); this actor runs hsad_actor_step and owns no Python-side sequence writer")
        self._verify_cached_priority = bool(on)

    def _make_native(self, multi_step, gamma, seq_len):
        import ctypes as C
        from . import _lib
        env, agent = self.env, self.agent
        self.writer.close()                       # the library's actor owns its own sequence writer
        cfg = _lib.ActorConfig(int(self.vdn), int(multi_step), int(seq_len), int(env.H), int(agent.online.H), float(gamma), float(self.eta),
                               int(agent.seed))
        io = _lib.ActorIO(env.legal_move.data_ptr(), env.own_hand.data_ptr(), env.eps.data_ptr(), env.reward.data_ptr(), env.terminal.data_ptr(),
                          env.priv_bits.data_ptr(), env.legal_bits.data_ptr(), env.own_bits.data_ptr(), env.priv_s_bf16.data_ptr())
        h = C.c_void_p()
        _lib.check(agent.lib.hsad_actor_create(env.h, agent.online.h, agent.target.h, self.replay.h, C.byref(cfg), C.byref(io), C.byref(h)))
        self.c_actor, self._lib = h, agent.lib
        nf = agent.lib.hsad_actor_n_finished_dev(h)
        from .composite import _view
        self.n_finished = _view(nf, 1, env.device, self, dtype=torch.int32)

    def close(self):
        if getattr(self, "c_actor", None):
            self._lib.hsad_actor_destroy(self.c_actor)
            self.c_actor = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def last_reply(self):
        """{a, greedy_a} int64 [N] of the last step (native loop: views of the library's buffers)"""
        import ctypes as C
        from .composite import _view
        g = C.c_void_p()
        a = self._lib.hsad_actor_last_actions(self.c_actor, C.byref(g))
        return {"a": _view(a, self.N, self.env.device, self, dtype=torch.int64), "greedy_a": _view(g.value, self.N, self.env.device, self, dtype=torch.int64)}

    @property
    def last_priority(self):
        """float32 [E] n-step priorities pushed by the last step, None while the n-step window was still filling (native loop)"""
        import ctypes as C
        from .composite import _view
        n = C.c_int32()
        p = self._lib.hsad_actor_last_priority(self.c_actor, C.byref(n))
        return _view(p, n.value, self.env.device, self) if p else None

    @property
    def num_redo(self):
        return int(self._lib.hsad_actor_num_redo(self.c_actor)) if self.c_actor is not None else self.n_checked_stale

    def _rows(self):
        e, N = self.env, self.N
        if self.packed_obs:
            return {"priv_s_bf16": e.priv_s_bf16.view(N, -1), "legal_move": e.legal_move.view(N, e.A), "eps": e.eps.view(N),
                    "own_hand": e.own_hand.view(N, 3 * e.H)}
        return {"priv_s": e.priv_s.view(N, e.F), "legal_move": e.legal_move.view(N, e.A), "eps": e.eps.view(N),
                "own_hand": e.own_hand.view(N, 3 * e.H)}

    def _reset_terminated(self):
        """`if (terminated) reset` at the top of the thread-loop body (cpp/thread_loop.h:46-52).  The reset kernel is latency-bound --
        a handful of lanes each shuffle a deck with mt19937 while the rest of the chip idles (~35 us) -- and nothing between one
        env.step and the next act() touches the env, so the reset of the games that just ended is issued on a side stream right
        after env.step (_prefetch_reset) and overlaps the n-step / sequence / replay bookkeeping of the same iteration; the end of
        step() joins it, so nothing outside step() ever runs next to it."""
        if self._reset_pending:
            self._reset_pending = False       # already issued behind the previous env.step and joined at the end of that step()
        else:
            self.env.reset()

    def _prefetch_reset(self):
        if self._side is None:
            return
        main = torch.cuda.current_stream(self.env.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self.env.reset()
        self._reset_pending = True

    def _join_reset(self):
        if self._reset_pending:
            torch.cuda.current_stream(self.env.device).wait_stream(self._side)

    def set_run_ahead(self, steps):
        """bound how far the host may run ahead of the device (hsad_actor_set_run_ahead; 0 = unbounded): an actor rank of a multi-GPU
        job must be able to serve the learner's round within a few steps, not behind hundreds of queued ones"""
        self._run_ahead, self._step_events = int(steps), []
        if self.c_actor is not None:
            from . import _lib
            _lib.check(self._lib.hsad_actor_set_run_ahead(self.c_actor, int(steps)))

    def _bound_run_ahead(self):
        """the Python body's twin of the library's bound (polling, never a blocking wait)"""
        import time
        e = torch.cuda.Event()
        e.record()
        self._step_events.append(e)
        if len(self._step_events) > self._run_ahead:
            old = self._step_events.pop(0)
            while not old.query():
                time.sleep(0)

    def step(self):
        """one iteration of the thread-loop body: reset-terminated -> act -> step -> postAct"""
        if self.c_actor is not None:
            from . import _lib
            from .composite import _s
            _lib.check(self._lib.hsad_actor_step(self.c_actor, _s(self.env.device)))
            self.num_act = int(self._lib.hsad_actor_num_act(self.c_actor))
            return
        if getattr(self, "_run_ahead", 0) > 0:
            self._bound_run_ahead()
        env, agent, P = self.env, self.agent, self.P
        self._reset_terminated()
        obs = self._rows()
        # historyHidden_.push_back(hidden_): by reference -- R2D2Agent.act returns fresh state tensors and never writes the
        # ones it is given, and zero_hidden_rows below only touches the new ones, which enter the history next step
        self.history_hid.append(dict(self.hid))
        # with_q: Q_online(s_t, a_t) comes out of the pass that picks the action and Q_target(s_t, greedy_t) costs one target
        # pass on the live observation -- exactly the two numbers compute_priority needs from time t (as `obs` n steps from
        # now, as `next_obs` right now).  Two network passes per step instead of the reference's four, no observation ever
        # read back from the n-step ring.
        if not self.cached_q:
            return self._step_contract(obs)
        reply, self.hid = agent.act(obs, self.hid, with_q=True)
        self.q_hist.append((reply["q_online_a"], reply["versions"][0]))
        if self.packed_obs:
            fields = {"priv_s": env.priv_bits, "legal_move": env.legal_bits, "own_hand": env.own_bits, "eps": obs["eps"]}
        else:
            fields = dict(obs)
        fields["a"], fields["greedy_a"] = reply["a"], reply["greedy_a"]
        self.writer.push_obs_action(fields)
        env.step(reply["a"].view(self.G, P), reply["greedy_a"].view(self.G, P))
        self._prefetch_reset()
        self.num_act += self.N               # Tachometer counts P acts per game step in both layouts (utils.py:229-236)
        # postAct: reward / terminal of the game go to each of its players' rows (IQL) or to the game's row (VDN)
        self.writer.push_reward_terminal(env.reward, env.terminal, repeat=1 if self.vdn else P)
        zero_hidden_rows(self.hid, env.terminal, P)                                    # r2d2_actor.h:109-126
        if not self.writer.can_pop():
            self._join_reset()
            return
        hid_s = self.history_hid.popleft()
        qa_s, version_s = self.q_hist.popleft()
        stale = version_s != agent.online.version   # the online weights were synced since step t-n: the reference would
        np_ = P if self.vdn else 1                  # evaluate Q_online(s_{t-n}, a) with the NEW weights, so redo that pass
        N, F, A = self.N, env.F, env.A              # VDN rows [G, P*w] are the same memory as [G*P, w]
        check = self.verify_cached_priority
        cur, nxt, rew, term, boot = self.writer.pop_transition(want_fields=stale or check, want_next=check)
        if stale or check:
            cur_obs = {"priv_s": cur["priv_s"].view(N, F), "legal_move": cur["legal_move"].view(N, A)}
        if stale:
            qa_s = agent.q_of(agent.online, cur_obs, cur["a"].view(-1), hid_s)
        prio = agent.priority_from_q(qa_s, reply["q_target_greedy"], rew, boot, num_player=np_)
        if check:   # tests: the reference's call on the unpacked transition must give the same bits
            nxt_obs = {"priv_s": nxt["priv_s"].view(N, F), "legal_move": nxt["legal_move"].view(N, A)}
            full = agent.compute_priority(cur_obs, cur["a"].view(-1), nxt_obs, hid_s, self.history_hid[-1], rew, boot,
                                          num_player=np_)
            assert torch.equal(prio, full), "cached-Q priorities differ from compute_priority"
            self.n_checked += 1
            self.n_checked_stale += int(stale)
        self.writer.push_sequence(prio)
        self._flush()
        self._join_reset()

    def _flush(self):
        """finished sequences -> replay.  The chain is five small launches, three of them single-workgroup scans (~80 us with the
        chip idle around them); issued on a side stream it overlaps the next step's network passes.  No join here: the library
        orders whoever touches the writer's cursors or the replay next -- on any stream -- behind it (StreamFence in
        csrc/hsad_replay.hip)."""
        if self._side_flush is None:
            self.n_finished = self.writer.flush_to_replay(self.replay, self.eta, out=self.n_finished)
            return
        self._side_flush.wait_stream(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(self._side_flush):
            self.n_finished = self.writer.flush_to_replay(self.replay, self.eta, out=self.n_finished)

    def _step_contract(self, obs):
        """the rest of step() for a model that only offers the reference's contract (act / compute_priority): the n-step
        priority is ITS compute_priority on the popped transition with the hidden states of t-n and t (r2d2_actor.h:128-156)"""
        env, agent, P = self.env, self.agent, self.P
        reply, self.hid = agent.act(obs, self.hid)
        fields = dict(obs)
        fields["a"], fields["greedy_a"] = reply["a"], reply["greedy_a"]
        self.writer.push_obs_action(fields)
        env.step(reply["a"].view(self.G, P), reply["greedy_a"].view(self.G, P))
        self.num_act += self.N
        r = env.reward if self.vdn else env.reward.repeat_interleave(P)
        t = env.terminal if self.vdn else env.terminal.repeat_interleave(P)
        self.writer.push_reward_terminal(r, t)
        zero_hidden_rows(self.hid, env.terminal, P)
        if not self.writer.can_pop():
            return
        hid_s = self.history_hid.popleft()
        cur, nxt, rew, term, boot = self.writer.pop_transition(want_fields=True, want_next=True)
        prio = agent.compute_priority_dict(cur, nxt, hid_s, self.history_hid[-1], rew, term, boot)
        self.writer.push_sequence(prio)
        self.n_finished = self.writer.flush_to_replay(self.replay, self.eta)
