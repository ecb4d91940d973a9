"""Batched evaluation and checkpoints.

`evaluate` = pyhanabi/eval.py:19-66: `num_game` fresh games (seed+game_idx, max_len = -1, eps = 0), every player
acts greedily, score = last_score(); the reference spins one thread per game, here all games advance in
lock-step on the GPU and finished games simply stop acting.
`save_weights` / `load_weights` keep the reference's `.pthw` format: torch.save(online_net.state_dict())
with the key names net.0.*, lstm.*_l{0,1}, fc_v.*, fc_a.*, pred.* (common_utils/saver.py:17-61; utils.py:278-299)."""
import numpy as np
import torch

from .env import BatchedHanabiEnv
from .r2d2 import R2D2Agent, R2D2NetKernels


def evaluate(weights, num_game, seed, bomb, sad, *, num_player=2, hand_size=5, device="cuda:0", max_steps=200, precision="bf16",
             shuffle_color=False):
    """-> (mean score, fraction of perfect games, scores list, num perfect) like eval.evaluate.  `weights`: a weight dict, or
    an R2D2Agent / net already on the device (its online net acts for every player)"""
    env = BatchedHanabiEnv(num_game, players=num_player, hand_size=hand_size, seed=seed, bomb=bomb, eps_list=[0.0],
                           max_len=-1, sad=bool(sad), shuffle_color=bool(shuffle_color), device=device, track_deck_history=False)
    if isinstance(weights, R2D2Agent):
        agent = R2D2Agent(weights.online, weights.online, 1, 0.99)
    elif hasattr(weights, "act") and hasattr(weights, "get_h0"):
        agent = weights                      # any acting agent (obl.OBLAgent, rela.ContractAgent): used as it is
    elif hasattr(weights, "trunk"):
        agent = R2D2Agent(weights, weights, 1, 0.99)
    elif precision == "bf16":
        from .composite import CNet, CompositeAgent
        net = CNet(weights, device)
        agent = CompositeAgent(net, net, 1, 0.99)
    else:
        net = R2D2NetKernels.make(weights, device, precision)
        agent = R2D2Agent(net, net, 1, 0.99)
    N = num_game * num_player
    hid = agent.get_h0(N)
    env.reset()
    done = torch.zeros(num_game, dtype=torch.bool, device=device)
    noop = env.A - 1
    for _ in range(max_steps):
        obs = {"priv_s": env.priv_s.view(N, env.F), "legal_move": env.legal_move.view(N, env.A), "eps": env.eps.view(N)}
        reply, hid = agent.act(obs, hid)
        a = reply["a"].view(num_game, num_player)
        # finished games may not be stepped again (HanabiEnv::step asserts !terminated()): park them on a copy that
        # the kernel ignores by keeping their state untouched -> step only the live ones through a masked action
        q = env.query()
        done = q[:, 0] == 1
        if bool(done.all()):
            break
        live = ~done
        if bool(live.all()):
            env.step(a.contiguous(), reply["greedy_a"].view(num_game, num_player).contiguous())
        else:
            # lock-step with stragglers: finished games receive an (ignored) illegal noop and are skipped by the
            # error log below; their last_score is already latched
            aa = torch.where(live.unsqueeze(1), a, torch.full_like(a, noop)).contiguous()
            env.step(aa, aa)
            n, g, c = _drain_errors(env)
    scores = env.query()[:, 5].cpu().numpy().astype(np.int64)
    perfect = int((scores == 25).sum())
    return float(scores.mean()), perfect / num_game, scores.tolist(), perfect


def _drain_errors(env):
    import ctypes as C
    n, g, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    env.lib.hsad_env_error_count(env.h, C.byref(n), C.byref(g), C.byref(c))
    return n.value, g.value, c.value


from .checkpoint import load_weights, save_weights  # noqa: E402,F401  (kept importable from here)
