"""CPU: the fp32 torch restatement (tests/r2d2_torch_ref.py) against golden vectors produced by the REFERENCE
agent (tests/golden/make_r2d2_golden.py imports /root/reference/pyhanabi/r2d2.py with PYTORCH_JIT=0)."""
import os

import numpy as np
import pytest
import torch

from tests import r2d2_torch_ref as ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = dict(rtol=2e-5, atol=2e-6)


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def hid_to_LNH(h):
    """reference actor layout [G, P', L, H] (batch first) -> [L, G*P', H]"""
    G, Pp, L, H = h.shape
    return torch.tensor(h).reshape(G * Pp, L, H).transpose(0, 1).contiguous()


def test_act_matches_reference_iql():
    z = load("r2d2_iql_sad_small")
    W = ref.weights_from_npz(z, "online_net.")
    priv, legal = torch.tensor(z["act.priv_s"]).flatten(0, 1), torch.tensor(z["act.legal_move"]).flatten(0, 1)
    g, h, c = ref.greedy_act(W, priv, legal, hid_to_LNH(z["act.h0"]), hid_to_LNH(z["act.c0"]))
    assert np.array_equal(g.numpy(), z["act.out_greedy_a"].reshape(-1))
    assert np.array_equal(g.numpy(), z["act.out_a"].reshape(-1))          # eps = 0
    G = priv.shape[0]
    assert np.allclose(h.transpose(0, 1).numpy(), z["act.out_h0"].reshape(G, 2, -1), **TOL)
    assert np.allclose(c.transpose(0, 1).numpy(), z["act.out_c0"].reshape(G, 2, -1), **TOL)


def test_compute_priority_matches_reference_iql():
    z = load("r2d2_iql_sad_small")
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    f = lambda k: torch.tensor(z[k]).flatten(0, 1)
    meta = z["meta"]
    p = ref.compute_priority(Won, Wtg, f("act.priv_s"), f("act.legal_move"), f("prio.a"), f("prio.next_priv_s"),
                             f("prio.next_legal_move"), hid_to_LNH(z["act.h0"]), hid_to_LNH(z["act.c0"]),
                             hid_to_LNH(z["prio.next_h0"]), hid_to_LNH(z["prio.next_c0"]), f("prio.reward"),
                             f("prio.bootstrap"), int(meta[8]), float(z["gamma"][0]))
    assert np.allclose(p.numpy(), z["prio.out"].reshape(-1), **TOL)


@pytest.mark.parametrize("tag,pw", [("rl", 0.0), ("aux", 0.25)])
def test_loss_priority_and_gradients_match_reference_iql(tag, pw):
    z = load("r2d2_iql_sad_small")
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    for v in Won.values():
        v.requires_grad_(True)
    batch = {k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("loss.") and k.count(".") == 1}
    meta = z["meta"]
    loss, prio = ref.loss(Won, Wtg, batch, int(meta[8]), float(z["gamma"][0]), pw)
    assert np.allclose(loss.detach().numpy(), z["loss.%s.loss" % tag], **TOL)
    assert np.allclose(prio.detach().numpy(), z["loss.%s.priority" % tag], **TOL)
    total = (loss * batch["weight"]).mean()
    assert np.allclose(total.item(), z["loss.%s.total" % tag][0], **TOL)
    total.backward()
    for k, v in Won.items():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        want = z["loss.%s.grad.%s" % (tag, k)]
        assert np.allclose(g.numpy(), want, rtol=1e-4, atol=1e-6), k


def test_vdn_act_priority_loss_and_gradients_match_reference():
    """VDN layouts ([.., P, ..], Q summed over the players of a game; r2d2.py:254-258,316-345,386-412)"""
    z = load("r2d2_vdn_small")
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    meta = z["meta"]
    assert int(meta[0]) == 1
    P = z["act.priv_s"].shape[2]
    f2 = lambda k: torch.tensor(z[k]).flatten(0, 2)
    g, h, c = ref.greedy_act(Won, f2("act.priv_s"), f2("act.legal_move"), hid_to_LNH(z["act.h0"]), hid_to_LNH(z["act.c0"]))
    assert np.array_equal(g.numpy(), z["act.out_greedy_a"].reshape(-1))
    p = ref.compute_priority(Won, Wtg, f2("act.priv_s"), f2("act.legal_move"), f2("prio.a"), f2("prio.next_priv_s"),
                             f2("prio.next_legal_move"), hid_to_LNH(z["act.h0"]), hid_to_LNH(z["act.c0"]),
                             hid_to_LNH(z["prio.next_h0"]), hid_to_LNH(z["prio.next_c0"]),
                             torch.tensor(z["prio.reward"]).flatten(), torch.tensor(z["prio.bootstrap"]).flatten(),
                             int(meta[8]), float(z["gamma"][0]), num_player=P)
    assert np.allclose(p.numpy(), z["prio.out"].reshape(-1), **TOL)
    for v in Won.values():
        v.requires_grad_(True)
    batch = {k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("loss.") and k.count(".") == 1}
    assert batch["priv_s"].dim() == 4
    loss, prio = ref.loss(Won, Wtg, batch, int(meta[8]), float(z["gamma"][0]), 0.0)
    assert np.allclose(loss.detach().numpy(), z["loss.rl.loss"], **TOL)
    assert np.allclose(prio.detach().numpy(), z["loss.rl.priority"], **TOL)
    (loss * batch["weight"]).mean().backward()
    for k, v in Won.items():
        g_ = v.grad if v.grad is not None else torch.zeros_like(v)
        assert np.allclose(g_.numpy(), z["loss.rl.grad.%s" % k], rtol=1e-4, atol=1e-6), k


ARCH_CASES = ["r2d2_fc2_skip_small", "r2d2_skip_small", "r2d2_lstm1_small", "r2d2_lstm3_fc2_small"]


@pytest.mark.parametrize("name", ARCH_CASES)
def test_other_architectures_match_reference(name):
    """R2D2Net(num_lstm_layer 1-3, num_fc_layer 1-2, skip_connect) as utils.load_op_model / --num_lstm_layer construct them
    (r2d2.py:22-57, utils.py:46-57, selfplay.py:50): act applies the skip connection, forward does not"""
    z = load(name)
    nl, nfc, skip = (int(v) for v in z["arch"])
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    assert ref.num_layers(Won) == nl and ("net.2.weight" in Won) == (nfc == 2)
    meta = z["meta"]
    f = lambda k: torch.tensor(z[k]).flatten(0, 1)
    g, h, c = ref.greedy_act(Won, f("act.priv_s"), f("act.legal_move"), hid_to_LNH(z["act.h0"]), hid_to_LNH(z["act.c0"]), bool(skip))
    assert np.array_equal(g.numpy(), z["act.out_greedy_a"].reshape(-1))
    G = g.shape[0]
    assert np.allclose(h.transpose(0, 1).numpy(), z["act.out_h0"].reshape(G, nl, -1), **TOL)
    p = ref.compute_priority(Won, Wtg, f("act.priv_s"), f("act.legal_move"), f("prio.a"), f("prio.next_priv_s"),
                             f("prio.next_legal_move"), hid_to_LNH(z["act.h0"]), hid_to_LNH(z["act.c0"]),
                             hid_to_LNH(z["prio.next_h0"]), hid_to_LNH(z["prio.next_c0"]), f("prio.reward"),
                             f("prio.bootstrap"), int(meta[8]), float(z["gamma"][0]), skip_connect=bool(skip))
    assert np.allclose(p.numpy(), z["prio.out"].reshape(-1), **TOL)
    for v in Won.values():
        v.requires_grad_(True)
    batch = {k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("loss.") and k.count(".") == 1}
    loss, prio = ref.loss(Won, Wtg, batch, int(meta[8]), float(z["gamma"][0]), 0.25)
    assert np.allclose(loss.detach().numpy(), z["loss.aux.loss"], **TOL)
    assert np.allclose(prio.detach().numpy(), z["loss.aux.priority"], **TOL)
    (loss * batch["weight"]).mean().backward()
    for k, v in Won.items():
        g_ = v.grad if v.grad is not None else torch.zeros_like(v)
        assert np.allclose(g_.numpy(), z["loss.aux.grad.%s" % k], rtol=1e-4, atol=1e-6), k
