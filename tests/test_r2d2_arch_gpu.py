"""Every architecture the reference constructs -- R2D2Net(num_lstm_layer 1-3, num_fc_layer 1-2, skip_connect): utils.load_op_model's
M0-M11, selfplay.py's --num_lstm_layer (pyhanabi/r2d2.py:22-57, utils.py:36-84, selfplay.py:50) -- on the kernels, against golden
vectors from the reference itself (tests/golden/make_r2d2_golden.py): act / compute_priority / loss / every gradient, in fp32
(the exact mode, 1e-4) and bf16 (the library's composite entry points; tolerances of tests/test_r2d2_precision_gpu.py), plus the
fused whole-sequence forward at H = 256 / 512 against the fp32 torch restatement those vectors pin."""
import os

import numpy as np
import pytest
import torch

from tests import r2d2_torch_ref as ref
from tests.test_r2d2_precision_gpu import TOL, maxerr, relerr

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["r2d2_fc2_skip_small", "r2d2_skip_small", "r2d2_lstm1_small", "r2d2_lstm3_fc2_small"]


def hid(z, k):
    h = torch.tensor(z[k])
    G, Pp, L, H = h.shape
    return h.reshape(G * Pp, L, H).transpose(0, 1).contiguous().to(DEV)


def make_agent(z, precision):
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    skip = bool(z["arch"][2])
    ms, gm = int(z["meta"][8]), float(z["gamma"][0])
    if precision == "fp32":
        from hanabi_sad_amd.r2d2 import R2D2Agent
        from hanabi_sad_amd.r2d2_f32 import R2D2NetF32
        return R2D2Agent(R2D2NetF32(Won, DEV, skip_connect=skip), R2D2NetF32(Wtg, DEV, skip_connect=skip), ms, gm)
    from hanabi_sad_amd.composite import CNet, CompositeAgent
    return CompositeAgent(CNet(Won, DEV, skip_connect=skip), CNet(Wtg, DEV, skip_connect=skip), ms, gm)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", CASES)
def test_act_and_compute_priority_against_golden(name, precision):
    tol = TOL[precision]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    agent = make_agent(z, precision)
    flat = lambda k: torch.tensor(z[k]).flatten(0, 1).to(DEV)
    obs = {"priv_s": flat("act.priv_s"), "legal_move": flat("act.legal_move"), "eps": torch.zeros(flat("act.priv_s").shape[0], device=DEV)}
    h = {"h0": hid(z, "act.h0"), "c0": hid(z, "act.c0")}
    reply, nh = agent.act(obs, h)
    torch.cuda.synchronize()
    want_g = torch.tensor(z["act.out_greedy_a"]).reshape(-1)
    if precision == "fp32":
        assert torch.equal(reply["greedy_a"].cpu(), want_g)
    else:
        assert (reply["greedy_a"].cpu() == want_g).float().mean() >= 0.85      # near-ties of random-init advantages may flip
    G, L = want_g.shape[0], int(z["arch"][0])
    assert maxerr(nh["h0"].transpose(0, 1), z["act.out_h0"].reshape(G, L, -1)) <= tol["hidden"]
    assert maxerr(nh["c0"].transpose(0, 1), z["act.out_c0"].reshape(G, L, -1)) <= 2 * tol["hidden"]
    nobs = {"priv_s": flat("prio.next_priv_s"), "legal_move": flat("prio.next_legal_move")}
    p = agent.compute_priority(obs, flat("prio.a"), nobs, h, {"h0": hid(z, "prio.next_h0"), "c0": hid(z, "prio.next_c0")},
                               flat("prio.reward"), flat("prio.bootstrap"))
    d = (p.cpu() - torch.tensor(z["prio.out"]).reshape(-1)).abs()
    if precision == "fp32":
        assert float(d.max()) <= tol["priority"]
    else:       # a flipped next-greedy action at a near-tie moves one priority; the rest is within the bf16 tolerance
        assert float(d.median()) <= tol["priority"] and (d > 4 * tol["priority"]).float().mean() <= 0.15


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", CASES)
def test_learner_loss_and_gradients_against_golden(name, precision):
    tol = TOL[precision]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    ms, gm = int(z["meta"][8]), float(z["gamma"][0])
    if precision == "fp32":
        from hanabi_sad_amd.r2d2 import R2D2Learner
        lr = R2D2Learner(Won, Wtg, ms, gm, device=DEV, precision="fp32", skip_connect=bool(z["arch"][2]))
    else:
        from hanabi_sad_amd.composite import CompositeLearner
        lr = CompositeLearner(Won, Wtg, ms, gm, device=DEV)
    t = lambda k: torch.tensor(z[k]).to(DEV)
    batch = {k: t("loss." + k) for k in ("priv_s", "legal_move", "a", "reward", "bootstrap", "seq_len", "own_hand")}
    for tag, pw in (("rl", 0.0), ("aux", 0.25)):
        loss, prio = lr.loss(batch, t("loss.weight"), pw)
        torch.cuda.synchronize()
        assert maxerr(loss, z["loss.%s.loss" % tag]) <= tol["loss"] and maxerr(prio, z["loss.%s.priority" % tag]) <= tol["priority"], (tag,)
        rel = {}
        for k, g in lr.grad.items():
            want = torch.tensor(z["loss.%s.grad.%s" % (tag, k)])
            if want.abs().max() == 0:
                assert g.abs().max() < 1e-6, k
                continue
            rel[k] = relerr(g, want)
        # bf16: 2 x the measured error, which for two fc layers is dominated by ReLU-mask flips of near-zero first-layer units
        # (net.0.* 2.3-3.4 %, net.2.* 1.1 %; tools/emulate_bf16_grad.py reproduces these figures in plain torch from bf16 operand
        # rounding alone); the one-fc-layer cases stay at the default architecture's level
        gtol = tol["grad_rel"] * (1.5 if precision == "fp32" or int(z["arch"][1]) == 1 else 8.0)
        assert max(rel.values()) <= gtol, (tag, rel)
        if precision == "bf16" and int(z["arch"][1]) == 2:
            assert max(v for k, v in rel.items() if not k.startswith("net.")) <= 1.5 * tol["grad_rel"], (tag, rel)
    if precision == "bf16":
        lr.check_sync()


@pytest.mark.parametrize("H,T,B,nl,nfc", [(256, 24, 64, 1, 1), (256, 16, 32, 3, 2), (512, 20, 64, 3, 1), (512, 20, 128, 1, 2)])
def test_fused_forward_schedules_for_other_depths(H, T, B, nl, nfc):
    """H in {256, 512}, rows % 32 == 0: the learner's LSTM runs as fused persistent launches for any depth (3 layers at H = 512: a
    fused pair + a single); loss, priorities and gradients against fp32 autograd of the torch restatement, and against the
    projection-GEMM + per-layer recurrence schedule"""
    from hanabi_sad_amd.composite import CompositeLearner
    from hanabi_sad_amd.selfplay import init_weights
    from tests.test_r2d2_kernels_gpu import _rand_batch
    F, A = 783, 21
    W, Wt = init_weights(F, H, A, 5, 21, nl, nfc), init_weights(F, H, A, 5, 22, nl, nfc)
    batch, weight = _rand_batch(T, B, F, A)
    L = CompositeLearner(W, Wt, 3, 0.999, device=DEV)
    res = {}
    for fused in (True, False):
        L.set_fused(fused)
        loss, prio = L.loss(batch, weight, 0.25)
        torch.cuda.synchronize()
        res[fused] = (loss.clone(), prio.clone(), {k: v.clone() for k, v in L.grad.items()})
    L.check_sync()
    Wd = {k: v.to(DEV).requires_grad_(True) for k, v in W.items()}
    rloss, rprio = ref.loss(Wd, {k: v.to(DEV) for k, v in Wt.items()}, batch, 3, 0.999, 0.25)
    (rloss * weight).mean().backward()
    for fused in (True, False):
        loss, prio, grad = res[fused]
        d = ((prio - rprio).abs() / (1 + rprio.abs())).flatten()
        assert float(torch.quantile(d, 0.99)) < 1e-2, fused
        assert float(torch.quantile((loss - rloss).abs() / (1 + rloss.abs()), 0.9)) < 4e-2, fused
        bad = {k: relerr(grad[k], Wd[k].grad) for k in Wd if Wd[k].grad is not None and relerr(grad[k], Wd[k].grad) > (5e-2 if nfc == 2 else 2e-2)}
        assert not bad, (fused, bad)
