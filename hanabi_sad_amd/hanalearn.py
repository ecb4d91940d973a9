"""Drop-in names of the reference's `hanalearn` pybind module (cpp/pybind.cc:14-56) on the batched device env.

`HanabiEnv(params, eps, max_len, sad, shuffle_obs, shuffle_color, verbose)` keeps the reference constructor; each
object is a 1-game view (G = 1 device env) for code that drives single games (tools, debugging).  Training code
should create one `BatchedHanabiEnv` for all games instead of a Python list of these."""
import torch

from .env import BatchedHanabiEnv
from .rela import ThreadLoop, _dev_key


class HanabiEnv:
    def __init__(self, params, eps_list, max_len, sad, shuffle_obs, shuffle_color, verbose, device=None):
        self.cfg = dict(players=int(params["players"]), hand_size=int(params.get("hand_size", 5)),
                        seed=int(params.get("seed", 1)), bomb=int(params.get("bomb", 0)), eps_list=tuple(eps_list),
                        max_len=int(max_len), sad=bool(sad), shuffle_obs=bool(shuffle_obs), shuffle_color=bool(shuffle_color))
        self.device = device   # None: the current device when driven standalone; the model runner's device inside a thread loop
        self._impl = None      # a 1-game device env, created on first standalone use
        self._vec = None       # (HanabiVecEnv, index) once appended to a vector env: the game then lives in ITS batch
        if verbose:
            print("Hanabi game created, with parameters:", dict(params))

    @property
    def impl(self):
        if self._impl is None:
            dev = self.device if self.device is not None else "cuda:%d" % torch.cuda.current_device()
            self._impl = BatchedHanabiEnv(1, device=dev, **self.cfg)
        return self._impl

    def feature_size(self):
        return self.impl.feature_size()

    def num_action(self):
        return self.impl.num_action()

    def hand_feature_size(self):
        return self.impl.hand_feature_size()

    def _obs(self):
        e = self.impl
        return {"priv_s": e.priv_s[0].clone(), "legal_move": e.legal_move[0].clone(), "eps": e.eps[0].clone(),
                "own_hand": e.own_hand[0].clone()}

    def reset(self):
        self.impl.reset()
        return self._obs()

    def step(self, action):
        """action: {"a": int64 [P], "greedy_a": int64 [P]} -> (obs, reward, terminal) (cpp/hanabi_env.cc:49-113)"""
        dev = self.impl.device
        a = action["a"].to(dev).view(1, -1).contiguous()
        g = action["greedy_a"].to(dev).view(1, -1).contiguous() if "greedy_a" in action else None
        self.impl.step(a, g)
        self.impl.check_errors()
        return self._obs(), float(self.impl.reward[0]), bool(self.impl.terminal[0])

    def _src(self):
        """(device env, row) holding this game: its vector env's batch once that exists, else the private 1-game env"""
        if self._vec is not None and self._vec[0].impl is not None:
            return self._vec[0].impl, self._vec[1]
        return self.impl, 0

    def _q(self, i):
        env, r = self._src()
        return int(env.query()[r, i])

    def terminated(self):
        return bool(self._q(0))

    def get_current_player(self):
        return self._q(1)

    def get_score(self):
        return self._q(2)

    def get_life(self):
        return self._q(3)

    def get_info(self):
        return self._q(4)

    def last_score(self):
        return self._q(5)

    def get_fireworks(self):
        env, r = self._src()
        return [int(x) for x in env.query()[r, 8:13]]

    def move_is_legal(self, uid):
        env, r = self._src()
        return bool(env.move_is_legal(torch.full((env.G,), int(uid)))[r])

    def deck_history(self):
        """dealt cards of the episode as strings like "R1" (colour letter + rank), cpp/hanabi_env.h:112-114"""
        cards, n = self.impl.deck_history()
        return ["RYGWB"[int(c) // 5] + str(int(c) % 5 + 1) for c in cards[0, :int(n[0])]]


class HanabiVecEnv:
    """hanalearn.HanabiVecEnv (cpp/pybind.cc:40-43, rela/env.h:29-108): `append(env)` collects per-game `HanabiEnv`s; the
    games then live in ONE batched device env (created when a thread loop attaches).  The appended games must differ only in
    their seed, and the seeds must be consecutive (what create.py:36-53 builds: seed + game_idx)."""

    def __init__(self):
        self.envs, self.impl = [], None

    def append(self, env):
        if self.impl is not None:
            raise RuntimeError("HanabiVecEnv.append after the device env was built")
        env._vec = (self, len(self.envs))     # (a game absorbed into a merged batch is re-appended there: latest wins)
        self.envs.append(env)

    def size(self):
        return len(self.envs)

    def batched(self, device):
        if self.impl is None:
            if not self.envs:
                raise RuntimeError("empty HanabiVecEnv")
            c0 = self.envs[0].cfg
            for i, e in enumerate(self.envs):
                want = dict(c0, seed=c0["seed"] + i)
                if e.cfg != want:
                    raise ValueError("HanabiVecEnv: game %d differs from game 0 by more than seed = seed0 + index: %s vs %s"
                                     % (i, e.cfg, want))
            self.impl = BatchedHanabiEnv(len(self.envs), device=device, track_deck_history=False, **c0)
        return self.impl


class HanabiThreadLoop(ThreadLoop):
    """hanalearn.HanabiThreadLoop(actor | [actor per player], vec_env, eval) (cpp/thread_loop.h:14-88; bound as a subclass of
    rela.ThreadLoop, cpp/pybind.cc:45-47).  There is no thread:
    `step()` is one iteration of the loop body for all games of the vector env, driven by rela.Context.  Training mode =
    actor.DeviceActor (IQL when given a list of per-player actors, VDN for a single actor with num_player = P); eval mode =
    every player acts greedily with ITS actor's model (cross-play: one runner per seat, eval.py:43-46) until each game has
    finished once.

    The reference creates one loop per thread (create.py:57-76) -- or, for evaluation, one per GAME (eval.py:40-47: 1,000 loops
    of one game).  Loops are built lazily, and rela.Context merges compatible ones (same models, same configuration, consecutive
    seeds) into ONE batched device loop before the first step (`absorb`), so a driver written for the reference still advances
    all its games with one launch per kernel."""

    def __init__(self, actors, vec_env, eval_mode):
        self.per_thread = [list(actors) if isinstance(actors, (list, tuple)) else [actors]]   # actor(s) of every merged thread
        self.is_list = isinstance(actors, (list, tuple))
        self.vec_envs = [vec_env]
        self.eval_mode = bool(eval_mode)
        self.master, self.impl, self.env, self.done = None, None, None, False
        self._train_steps = 0
        self._built = False
        if not vec_env.envs:
            raise RuntimeError("HanabiThreadLoop over an empty HanabiVecEnv")

    # ---- merging (rela.Context) ----
    def merge_key(self):
        a0, c0 = self.per_thread[0], dict(self.vec_envs[0].envs[0].cfg)
        c0.pop("seed")
        cfg = tuple((a.multi_step, a.num_envs if self.eval_mode else None, a.gamma, a.eta, a.seq_len, a.num_player, id(a.replay))
                    for a in a0)
        return (self.eval_mode, self.is_list, tuple(id(a.runner) for a in a0), cfg, tuple(sorted(c0.items())),
                _dev_key(a0[0].runner.device))

    def seed_range(self):
        envs = [e for v in self.vec_envs for e in v.envs]
        return envs[0].cfg["seed"], envs[-1].cfg["seed"]

    def can_absorb(self, other):
        return (not self._built and not other._built and other.master is None and self.merge_key() == other.merge_key()
                and other.seed_range()[0] == self.seed_range()[1] + 1)

    def absorb(self, other):
        self.per_thread += other.per_thread
        self.vec_envs += other.vec_envs
        other.master = self

    @property
    def actors(self):
        return [a for group in self.per_thread for a in group]

    def _build(self):
        from .actor import DeviceActor, transition_fields
        self._built = True
        if len(self.vec_envs) > 1:                 # merged: one vector env over all games, in seed order
            merged = HanabiVecEnv()
            for v in self.vec_envs:
                for e in v.envs:
                    merged.append(e)               # re-points every game at its row of the merged batch
            self.vec_envs = [merged]
        a0 = self.per_thread[0]
        runs = [a.runner for a in a0]
        self.env = self.vec_envs[0].batched(runs[0].device)
        P = self.env.P
        self.vdn = not self.is_list and a0[0].num_player > 1
        if self.eval_mode:
            seats = runs if (self.is_list and len(runs) == P) else [runs[0]] * P
            self.same_model = all(r is seats[0] for r in seats)
            self.agents = []
            for r in (seats[:1] if self.same_model else seats):
                ag = r.make_agent(1, 0.99)
                if hasattr(ag, "configure"):
                    ag.configure(1, False)
                elif getattr(ag, "device_agent", False):
                    pass
                elif ag.target is not ag.online:
                    ag = type(ag)(ag.online, ag.online, 1, 0.99)        # evaluation only ever calls `act`
                self.agents.append(ag)
            rows = self.env.G * P if self.same_model else self.env.G
            self.hids = [ag.get_h0(rows) for ag in self.agents]
            self.env.reset()
        else:
            run = runs[0]
            self.agent = run.make_agent(a0[0].multi_step, a0[0].gamma, seed=self.env.G)
            fields = transition_fields(self.env, self.vdn)
            replay = a0[0].replay.bind_schema(fields, a0[0].seq_len, run.device)
            self.impl = DeviceActor(self.env, self.agent, replay, a0[0].multi_step, a0[0].gamma, a0[0].eta, a0[0].seq_len,
                                    vdn=self.vdn)
            for group, v in zip(self.per_thread, self._thread_sizes()):
                for a in group:
                    a._loop, a._per_step = self, v

    def step(self):
        if self.master is not None:
            return                                  # advanced by the loop that absorbed this one
        if not self._built:
            self._build()
        if not self.eval_mode:
            self.impl.step()
            self._train_steps += 1                  # R2D2Actor::numAct_ += num_envs per act() (r2d2_actor.h:98): R2D2Actor.num_act() multiplies
            return                                  # (no per-actor Python work in the loop: 160 actors cost 25 us per step)
        if self.done:
            return
        env, G, P = self.env, self.env.G, self.env.P
        done = env.query()[:, 0] == 1
        if bool(done.all()):
            self.done = True
            return
        if self.same_model:
            N = G * P
            obs = {"priv_s": env.priv_s.view(N, env.F), "legal_move": env.legal_move.view(N, env.A),
                   "eps": torch.zeros(N, device=env.device)}
            reply, self.hids[0] = self.agents[0].act(obs, self.hids[0])
            a = reply["greedy_a"].view(G, P)
        else:                                       # cross-play: seat p is played by its own model on its own rows
            cols = []
            for p, ag in enumerate(self.agents):
                obs = {"priv_s": env.priv_s[:, p].contiguous(), "legal_move": env.legal_move[:, p].contiguous(),
                       "eps": torch.zeros(G, device=env.device)}
                reply, self.hids[p] = ag.act(obs, self.hids[p])
                cols.append(reply["greedy_a"])
            a = torch.stack(cols, 1)
        a = torch.where(done.unsqueeze(1), torch.full_like(a, env.A - 1), a).contiguous()   # finished games: ignored noop
        env.step(a, a)
        for group, v in zip(self.per_thread, self._thread_sizes()):
            for act in group:
                act._num_act += v
        if bool(done.any()):
            import ctypes as C
            n, g, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            env.lib.hsad_env_error_count(env.h, C.byref(n), C.byref(g), C.byref(c))   # drain the "finished game" notes

    def _thread_sizes(self):
        if not hasattr(self, "_sizes"):
            G, n = self.env.G, len(self.per_thread)
            self._sizes = [G // n] * n if G % n == 0 else [G] + [0] * (n - 1)
        return self._sizes

    def finished(self):
        if self.master is not None:
            return self.master.finished()
        return self.eval_mode and self.done

    def scores(self):
        """lastScore() of every game (eval.py:57-66)"""
        src = self.master or self
        return src.env.query()[:, 5].cpu().tolist()
