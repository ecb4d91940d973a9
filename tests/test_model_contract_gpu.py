"""the MODEL CONTRACT of rela::BatchRunner (rela/batch_runner.h:24,76,108; rela/r2d2_actor.h:61-172) -- any
object with `act(dict) -> dict`, `compute_priority(dict) -> dict`, `get_h0(n) -> dict` acts in the rollout, not only the default
R2D2Net shape that the HIP kernels implement.  A torch model with a different architecture (two fc layers + skip connection,
one-layer LSTM) is driven through rela.BatchRunner / R2D2Actor / hanalearn.HanabiThreadLoop; the test checks the tensors it is
called with (shapes and hidden-state bookkeeping of the reference), that its actions and priorities are what reaches the
environment and the replay, and that evaluation runs on it."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
DEV = "cuda:0"


class SkipAgent(nn.Module):
    """NOT the default architecture: fc -> fc (+ skip) -> 1-layer LSTM -> advantage head.  Contract per r2d2_actor.h:
    act: {priv_s [S,E,F], legal_move [S,E,A], eps [S,E], h0/c0 [S,E,L,H]} -> {a, greedy_a [S,E], h0, c0 [S,E,L,H]}
    compute_priority: the transition dict (+ next_*, h0, c0, next_h0, next_c0, reward, terminal, bootstrap) -> {priority [S,E]}"""

    def __init__(self, F, A, H=32):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(F, H), nn.Linear(H, H)
        self.lstm = nn.LSTM(H, H, num_layers=1)
        self.adv = nn.Linear(H, A)
        self.H, self.calls = H, []

    def get_h0(self, n):
        z = torch.zeros(1, n, self.H, device=self.adv.weight.device)
        return {"h0": z, "c0": z.clone()}

    def _core(self, priv, h0, c0):
        x = torch.relu(self.fc1(priv))
        x = x + torch.relu(self.fc2(x))
        o, (h, c) = self.lstm(x.unsqueeze(0), (h0, c0))
        return self.adv(o.squeeze(0)), h, c

    def act(self, d):
        S, E = d["priv_s"].shape[:2]
        self.calls.append(("act", {k: tuple(v.shape) for k, v in d.items()}, d["h0"].clone()))
        h0 = d["h0"].flatten(0, 1).transpose(0, 1).contiguous()
        c0 = d["c0"].flatten(0, 1).transpose(0, 1).contiguous()
        adv, h, c = self._core(d["priv_s"].flatten(0, 1), h0, c0)
        legal = d["legal_move"].flatten(0, 1)
        greedy = ((1 + adv - adv.min()) * legal).argmax(1)
        rnd = legal.multinomial(1).squeeze(1)
        explore = (torch.rand(greedy.shape, device=greedy.device) < d["eps"].flatten(0, 1)).long()
        a = greedy * (1 - explore) + rnd * explore
        shp = (S, E, 1, self.H)
        return {"a": a.view(S, E).cpu(), "greedy_a": greedy.view(S, E).cpu(), "h0": h.transpose(0, 1).reshape(shp).cpu(),
                "c0": c.transpose(0, 1).reshape(shp).cpu()}

    def compute_priority(self, d):
        self.calls.append(("prio", {k: tuple(v.shape) for k, v in d.items()}, d["next_h0"].clone()))
        # a checkable function of the transition: 1 + n-step reward + a / 100
        return {"priority": (1.0 + d["reward"] + d["a"].float() / 100.0).cpu()}


def test_contract_model_drives_rollout_replay_and_eval():
    import hanalearn
    import rela
    from hanabi_sad_amd.selfplay import generate_explore_eps
    G, P, T, n, gamma, eta, alpha = 48, 2, 20, 3, 0.99, 0.9, 0.8
    eps = generate_explore_eps(0.4, 7, 11)
    games = [hanalearn.HanabiEnv({"players": str(P), "hand_size": "5", "seed": str(100 + i), "bomb": "0"}, eps, T, True, False,
                                 False, False) for i in range(G)]
    F, A = games[0].feature_size(), games[0].num_action()
    torch.manual_seed(3)
    model = SkipAgent(F, A).to(DEV)
    runner = rela.BatchRunner(model, DEV, 100, ["act", "compute_priority"])
    assert runner.model is model and runner.online is None       # not the kernels' shape: the contract path
    replay = rela.RNNPrioritizedReplay(4096, 1, alpha, 0.6, 3)
    actors = [rela.R2D2Actor(runner, n, G, gamma, eta, T, 1, replay) for _ in range(P)]
    env = hanalearn.HanabiVecEnv()
    for g in games:
        env.append(g)
    loop = hanalearn.HanabiThreadLoop(actors, env, False)
    ctx = rela.Context()
    ctx.push_env_thread(loop)
    steps = 50
    for _ in range(steps):
        ctx.step()
    loop.env.check_errors()
    N = G * P
    acts = [c for c in model.calls if c[0] == "act"]
    prios = [c for c in model.calls if c[0] == "prio"]
    assert len(acts) == steps and len(prios) == steps - n
    s = acts[0][1]
    assert s["priv_s"] == (1, N, F) and s["legal_move"] == (1, N, A) and s["eps"] == (1, N) and s["h0"] == (1, N, 1, 32)
    sp = prios[0][1]
    for k in ("priv_s", "legal_move", "eps", "own_hand", "a", "greedy_a", "next_priv_s", "next_legal_move", "reward", "terminal",
              "bootstrap", "h0", "c0", "next_h0", "next_c0"):
        assert k in sp, k
    assert sp["next_priv_s"] == (1, N, F) and sp["reward"] == (1, N) and sp["next_h0"] == (1, N, 1, 32)
    # historyHidden_.back(): the next_h0 of the k-th compute_priority is the h0 the act of the same iteration received
    for k, (_, _, nh) in enumerate(prios):
        assert torch.equal(nh, acts[k + n][2])
    # the hidden state of a finished game restarts from zero: every h0 row is zero in the first act, never all-zero later on
    assert float(acts[0][2].abs().sum()) == 0 and float(acts[5][2].abs().sum()) > 0
    assert sum(a.num_act() for a in actors) == steps * N
    # what reached the replay: sequences with the model's actions (legal) and ITS priorities, aggregated the reference's way
    size = replay.size()
    assert size > 0 and size == replay.num_add()
    want_w = 0.0
    for i in range(size):
        tr = replay.get(i)
        L = int(tr.seq_len)
        a, legal = tr.action["a"][:L].view(-1), tr.obs["legal_move"][:L, 0]
        assert (legal.gather(1, a.unsqueeze(1)) == 1).all()
        p = (1.0 + tr.reward[:L] + a.float() / 100.0).cpu().numpy()
        want_w += float(eta * p.max() + (1 - eta) * p.mean()) ** alpha
    total, cnt = replay.impl.priority_sum()
    assert cnt == size and abs(total - want_w) < 1e-4 * want_w
    # evaluation on the same model (eval.py:25-66 shape): greedy, finished games stop
    model.calls.clear()
    egames = [hanalearn.HanabiEnv({"players": str(P), "hand_size": "5", "seed": str(500 + i), "bomb": "0"}, [0.0], -1, True, False,
                                  False, False) for i in range(16)]
    ectx = rela.Context()
    for g in egames:
        v = hanalearn.HanabiVecEnv()
        v.append(g)
        ectx.push_env_thread(hanalearn.HanabiThreadLoop([rela.R2D2Actor(runner, 1) for _ in range(P)], v, True))
    for _ in range(200):
        ectx.step()
        if ectx.terminated():
            break
    assert ectx.terminated() and all(g.terminated() for g in egames)
    assert all(0 <= g.last_score() <= 25 for g in egames) and model.calls[0][1]["priv_s"] == (1, 16 * P, F)
