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
from tests.near_tie import assert_sequence_outliers_are_near_ties, top2_gap
from tests.test_r2d2_precision_gpu import TOL, maxerr, record, relerr

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
# relative Frobenius error of any parameter's gradient at the larger shapes below = 2 x the worst measured on MI355X (0.0047, pred.weight
# at H = 512, 3 layers: tools/arch_grad_errors.py, profiles/r04_r2d2_measured_errors.json "arch_fused.*"); net.* gradients are compared under
# the bf16 activation pattern, so no architecture needs more
ARCH_GRAD_REL = 1.0e-2
CASES = ["r2d2_fc2_skip_small", "r2d2_skip_small", "r2d2_lstm1_small", "r2d2_lstm3_fc2_small"]


def cb_weight(z):
    return torch.tensor(z["loss.weight"])


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
    # the fp32 restatement of the reference (pinned to these very golden vectors by tests/test_r2d2_golden_cpu.py) supplies what the
    # golden file does not store: the advantage margins behind each argmax
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    skip, ms, gm = bool(z["arch"][2]), int(z["meta"][8]), float(z["gamma"][0])
    adv, _, _ = ref.net_act(Won, obs["priv_s"].cpu(), h["h0"].cpu(), h["c0"].cpu(), skip)
    gap = top2_gap(adv, obs["legal_move"].cpu())
    flipped = torch.nonzero(reply["greedy_a"].cpu() != want_g).flatten().tolist()
    if precision == "fp32":
        assert not flipped
    for r in flipped:          # a differing greedy action is only acceptable where the reference's own margin is below the Q tolerance
        assert float(gap[r]) < 2 * tol["q"], ("greedy action differs without a near-tie of the reference", r, float(gap[r]))
    G, L = want_g.shape[0], int(z["arch"][0])
    assert maxerr(nh["h0"].transpose(0, 1), z["act.out_h0"].reshape(G, L, -1)) <= tol["hidden"]
    assert maxerr(nh["c0"].transpose(0, 1), z["act.out_c0"].reshape(G, L, -1)) <= 2 * tol["hidden"]
    nobs = {"priv_s": flat("prio.next_priv_s"), "legal_move": flat("prio.next_legal_move")}
    nh_in = {"h0": hid(z, "prio.next_h0"), "c0": hid(z, "prio.next_c0")}
    p = agent.compute_priority(obs, flat("prio.a"), nobs, h, nh_in, flat("prio.reward"), flat("prio.bootstrap"))
    d = (p.cpu() - torch.tensor(z["prio.out"]).reshape(-1)).abs()
    # an outlier must be a near-tie of the NEXT state's greedy action in the reference, and must then equal -- within the same
    # tolerance -- the priority the reference's formula gives with the runner-up action
    nadv, _, _ = ref.net_act(Won, nobs["priv_s"].cpu(), nh_in["h0"].cpu(), nh_in["c0"].cpu(), skip)
    nlegal = nobs["legal_move"].cpu()
    ngap = top2_gap(nadv, nlegal)
    runner_up = ((1 + nadv - nadv.min()) * nlegal).topk(2, dim=1).indices[:, 1]
    qa, _, _, _ = ref.net_forward(Won, obs["priv_s"].cpu().unsqueeze(0), obs["legal_move"].cpu().unsqueeze(0), flat("prio.a").cpu().unsqueeze(0),
                                  h["h0"].cpu(), h["c0"].cpu())
    tqa2, _, _, _ = ref.net_forward(Wtg, nobs["priv_s"].cpu().unsqueeze(0), nlegal.unsqueeze(0), runner_up.unsqueeze(0), nh_in["h0"].cpu(),
                                    nh_in["c0"].cpu())
    alt = (flat("prio.reward").cpu() + flat("prio.bootstrap").cpu() * gm ** ms * tqa2.squeeze(0) - qa.squeeze(0)).abs()
    outliers = torch.nonzero(d > tol["priority"]).flatten().tolist()
    if precision == "fp32":
        assert not outliers, float(d.max())
    for r in outliers:
        assert float(ngap[r]) < 2 * tol["q"], ("priority outlier without a near-tie of the reference", r, float(d[r]), float(ngap[r]))
        assert abs(float(p[r]) - float(alt[r])) <= tol["priority"], ("outlier is not the runner-up's priority either", r, float(p[r]), float(alt[r]))
    record(precision, "arch_act_priority." + name, greedy_flips=len(flipped), priority_outliers=len(outliers),
           priority_max_excl_ties=float(d[d <= tol["priority"]].max()) if bool((d <= tol["priority"]).any()) else 0.0)


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
    two_fc = True          # (the activation-pattern property below is applied to every architecture; it MATTERS with two fc layers)
    if two_fc and precision == "bf16":
        # Two fc layers in bf16: the ReLU decision of a first-layer unit whose pre-activation is within the bf16 rounding of zero comes
        # out the other way, and each such unit moves the net.* gradients by a whole term (2.3-3.4 % of their norm on these cases;
        # tools/emulate_bf16_grad.py).  That is held to a PROPERTY instead of a wide tolerance: (1) every unit whose decision differs
        # between the two arithmetics provably has an fp32 pre-activation within the rounding bound of zero (bf16_relu_masks), and
        # (2) against the fp32 network evaluated UNDER the bf16 activation pattern the gradients meet the ordinary tolerance.
        Wd = {k: v.clone().requires_grad_(True) for k, v in Won.items()}
        cb = {k: v.cpu() for k, v in batch.items()}
        masks, flips = ref.bf16_relu_masks(Wd, cb["priv_s"])
        assert all(worst <= 1.0 for _, worst in flips), ("a ReLU decision flips at a unit that is NOT within the bf16 rounding of zero", flips)
    for tag, pw in (("rl", 0.0), ("aux", 0.25)):
        loss, prio = lr.loss(batch, t("loss.weight"), pw)
        torch.cuda.synchronize()
        assert maxerr(loss, z["loss.%s.loss" % tag]) <= tol["loss"] and maxerr(prio, z["loss.%s.priority" % tag]) <= tol["priority"], (tag,)
        masked = None
        if two_fc and precision == "bf16":
            for v in Wd.values():
                v.grad = None
            ml, _ = ref.loss(Wd, Wtg, cb, ms, gm, pw, online_masks=masks)
            (ml * cb_weight(z)).mean().backward()
            masked = {k: v.grad for k, v in Wd.items()}
        rel = {}
        for k, g in lr.grad.items():
            want = torch.tensor(z["loss.%s.grad.%s" % (tag, k)])
            if want.abs().max() == 0:
                assert g.abs().max() < 1e-6, k
                continue
            if masked is not None and k.startswith("net."):
                want = masked[k]                 # the fp32 network under the bf16 activation pattern (see above)
            rel[k] = relerr(g, want)
        record(precision, "arch_learner.%s.%s" % (name, tag), grad_rel_max=max(rel.values()),
               relu_flips=float(sum(n for n, _ in flips)) if masked is not None else 0.0)
        assert max(rel.values()) <= 1.5 * tol["grad_rel"], (tag, rel)
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
    tol = TOL["bf16"]
    Wd = {k: v.to(DEV).requires_grad_(True) for k, v in W.items()}
    Wtd = {k: v.to(DEV) for k, v in Wt.items()}
    rloss, rprio = ref.loss(Wd, Wtd, batch, 3, 0.999, 0.25)
    (rloss * weight).mean().backward()
    want = {k: v.grad.clone() for k, v in Wd.items() if v.grad is not None}
    flips = []
    # net.* gradients against the fp32 network under the bf16 activation pattern, every flipped unit proven to sit within the bf16 rounding
    # of zero (see the golden test; one fc layer has the effect too: 150-2,000 flipped units move net.0.weight by 0.4-1.3 % at these sizes)
    masks, flips = ref.bf16_relu_masks(Wd, batch["priv_s"])
    assert all(worst <= 1.0 for _, worst in flips), flips
    for v in Wd.values():
        v.grad = None
    ml, _ = ref.loss(Wd, Wtd, batch, 3, 0.999, 0.25, online_masks=masks)
    (ml * weight).mean().backward()
    want.update({k: v.grad.clone() for k, v in Wd.items() if k.startswith("net.")})
    with torch.no_grad():
        h0 = torch.zeros(nl, B, H, device=DEV)
        _, _, rq, _ = ref.net_forward({k: v.detach() for k, v in Wd.items()}, batch["priv_s"], batch["legal_move"], batch["a"], h0, h0.clone())
        gap = top2_gap(rq, batch["legal_move"])
    for fused in (True, False):
        loss, prio, grad = res[fused]
        # maxima, with every outlier a provable near-tie of the fp32 reference (no quantiles)
        n_tie, p_max, l_max = assert_sequence_outliers_are_near_ties(prio, rprio.detach(), loss, rloss.detach(), gap, 3, tol["full_q"],
                                                                    tol["full_loss"], tag=("fused", fused))
        rel = {k: relerr(grad[k], want[k]) for k in want if float(want[k].norm()) > 0}
        record("bf16", "arch_fused.H%d_T%d_B%d_L%d_fc%d.%s" % (H, T, B, nl, nfc, "fused" if fused else "chunked"), near_tie_priorities=n_tie,
               priority_max_excl_ties=p_max, loss_max_excl_ties=l_max, grad_rel_max=max(rel.values()),
               relu_flips=float(sum(n for n, _ in flips)))
        assert n_tie <= 8, n_tie
        assert max(rel.values()) <= ARCH_GRAD_REL, (fused, rel)
