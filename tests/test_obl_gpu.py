"""the OBL model family (pyhanabi/tools/obl_model.py) on the HIP kernels, against golden vectors generated from
the reference's own classes (tests/golden/make_obl_golden.py), in the fp32-exact mode and in bf16; evaluation and cross-play
with the default R2D2 agent through the reference-shaped loops."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load():
    z = np.load(os.path.join(GOLD, "obl_small.npz"))
    W = {k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("w.")}
    return z, W


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_obl_act_against_reference_golden(precision):
    from hanabi_sad_amd.obl import OBLAgent, OBLNetKernels
    z, W = _load()
    net = OBLNetKernels(W, DEV, precision)
    agent = OBLAgent(net)
    N = z["priv_s"].shape[0]
    t = lambda k: torch.tensor(z[k]).to(DEV)
    hid = {k: t(k)[:, 0].transpose(0, 1).contiguous() for k in ("h0", "c0")}           # [N,1,L,H] -> [L,N,H]
    priv, legal = t("priv_s")[:, 0].contiguous(), t("legal_move")[:, 0].contiguous()
    adv, new_hid = net.advantage(priv, hid)
    tol = 1e-5 if precision == "fp32" else 3e-3
    assert float((adv[:, :net.A].cpu() - torch.tensor(z["adv"])).abs().max()) < tol
    assert float((new_hid["h0"].transpose(0, 1).cpu() - torch.tensor(z["out_h0"])[:, 0]).abs().max()) < tol
    assert float((new_hid["c0"].transpose(0, 1).cpu() - torch.tensor(z["out_c0"])[:, 0]).abs().max()) < 2 * tol
    reply, _ = agent.act({"priv_s": priv, "legal_move": legal, "eps": torch.zeros(N, device=DEV)}, hid)
    assert torch.equal(reply["a"], reply["greedy_a"])                                 # obl_model.py:296-297
    agree = float((reply["a"].cpu() == torch.tensor(z["out_a"]).view(-1)).float().mean())
    assert agree == 1.0 if precision == "fp32" else agree >= 0.9


def test_obl_checkpoint_loader_evaluation_and_cross_play(tmp_path):
    import hanalearn
    import rela
    from hanabi_sad_amd.eval import evaluate
    from hanabi_sad_amd.obl import load_obl_model
    from hanabi_sad_amd.selfplay import init_weights
    z, W = _load()
    legacy = dict(W)
    legacy.update({"core_ffn.1.weight": torch.zeros(2, 2), "core_ffn.1.bias": torch.zeros(2), "pred_t.weight": torch.zeros(1, 64),
                   "pred_t.bias": torch.zeros(1)})                                     # other variants' heads: dropped
    path = str(tmp_path / "obl.pthw")
    torch.save(legacy, path)
    agent = load_obl_model(path, DEV, greedy=True)
    mean, perfect, scores, n_perfect = evaluate(agent, 40, 99, 0, True, device=DEV)   # self-play of the OBL agent, SAD env
    assert len(scores) == 40 and all(0 <= s <= 25 for s in scores) and abs(mean - np.mean(scores)) < 1e-9
    # cross-play (eval.py shape, one runner per seat): seat 0 = OBL, seat 1 = a default R2D2 agent
    W2 = init_weights(838, 64, 21, 5, 1)
    sd = {"online_net." + k: v for k, v in W2.items()}
    runners = [rela.BatchRunner(agent, DEV, 1000, ["act"]), rela.BatchRunner(sd, DEV, 1000, ["act"])]
    games = [hanalearn.HanabiEnv({"players": "2", "hand_size": "5", "seed": str(7 + i), "bomb": "0"}, [0.0], -1, True, False, False,
                                 False) for i in range(24)]
    ctx = rela.Context()
    for g in games:
        v = hanalearn.HanabiVecEnv()
        v.append(g)
        ctx.push_env_thread(hanalearn.HanabiThreadLoop([rela.R2D2Actor(r, 1) for r in runners], v, True))
    for _ in range(200):
        ctx.step()
        if ctx.terminated():
            break
    assert ctx.terminated() and all(g.terminated() and 0 <= g.last_score() <= 25 for g in games)
