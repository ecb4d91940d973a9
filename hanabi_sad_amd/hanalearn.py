"""Drop-in names of the reference's `hanalearn` pybind module (cpp/pybind.cc:14-56) on the batched device env.

`HanabiEnv(params, eps, max_len, sad, shuffle_obs, shuffle_color, verbose)` keeps the reference constructor; each
object is a 1-game view (G = 1 device env) for code that drives single games (tools, debugging).  Training code
should create one `BatchedHanabiEnv` for all games instead of a Python list of these."""
import torch

from .env import BatchedHanabiEnv


class HanabiEnv:
    def __init__(self, params, eps_list, max_len, sad, shuffle_obs, shuffle_color, verbose, device="cuda:0"):
        self.impl = BatchedHanabiEnv(1, players=int(params["players"]), hand_size=int(params.get("hand_size", 5)),
                                     seed=int(params.get("seed", 1)), bomb=int(params.get("bomb", 0)),
                                     eps_list=list(eps_list), max_len=max_len, sad=sad, shuffle_obs=shuffle_obs,
                                     shuffle_color=shuffle_color, device=device)
        if verbose:
            print("Hanabi game created, with parameters:", dict(params))

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

    def _q(self, i):
        return int(self.impl.query()[0, i])

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
        return [int(x) for x in self.impl.query()[0, 8:13]]

    def move_is_legal(self, uid):
        return bool(self.impl.move_is_legal(torch.tensor([uid]))[0])

    def deck_history(self):
        """dealt cards of the episode as strings like "R1" (colour letter + rank), cpp/hanabi_env.h:112-114"""
        cards, n = self.impl.deck_history()
        return ["RYGWB"[int(c) // 5] + str(int(c) % 5 + 1) for c in cards[0, :int(n[0])]]
