"""Generates tests/golden/r2d2_*.npz by importing the REFERENCE agent (pyhanabi/r2d2.py) in the authoring
container (PYTORCH_JIT=0: the TorchScript path does not compile on torch 2.10).  The fixture
holds data only: the randomly initialised weights, the inputs and the reference's outputs.  Run:

    PYTORCH_JIT=0 python tests/golden/make_r2d2_golden.py

Small hidden size keeps the fixtures small; full-size (H=512) parity is checked against the plain fp32 torch
restatement in tests/r2d2_torch_ref.py, which itself is pinned by these vectors."""
import os
import sys

os.environ.setdefault("PYTORCH_JIT", "0")
sys.path.insert(0, "/root/reference/pyhanabi")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import r2d2  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))


class Stat(dict):
    class _S:
        def feed(self, v):
            self.v = v

    def __missing__(self, k):
        self[k] = Stat._S()
        return self[k]


def synth_batch(rng, T, B, F, A, H5, num_player=None):
    shp = (T, B) if num_player is None else (T, B, num_player)
    priv_s = (rng.random(shp + (F,)) < 0.15).astype(np.float32)
    legal = (rng.random(shp + (A,)) < 0.4).astype(np.float32)
    legal[..., A - 1] = 0
    empty = legal.sum(-1) == 0
    legal[..., 0][empty] = 1
    # uniform-legal actions
    a = np.zeros(shp, np.int64)
    it = np.nditer(a, flags=["multi_index"])
    for _ in it:
        idx = np.flatnonzero(legal[it.multi_index])
        a[it.multi_index] = rng.choice(idx)
    own = np.zeros(shp + (H5, 3), np.float32)
    cls = rng.integers(0, 4, shp + (H5,))
    for k in range(3):
        own[..., k] = (cls == k)
    own = own.reshape(shp + (H5 * 3,))
    seq_len = rng.integers(max(2, T // 2), T + 1, B).astype(np.float32)
    reward = ((rng.random((T, B)) < 0.2) * rng.integers(1, 3, (T, B))).astype(np.float32)
    t_idx = np.arange(T)[:, None]
    terminal = (t_idx >= seq_len[None, :] - 1)
    bootstrap = (t_idx + 3 < seq_len[None, :]).astype(np.float32)
    mask = (t_idx < seq_len[None, :])
    priv_s *= mask.reshape(mask.shape + (1,) * (priv_s.ndim - 2))
    legal *= mask.reshape(mask.shape + (1,) * (legal.ndim - 2))
    own *= mask.reshape(mask.shape + (1,) * (own.ndim - 2))
    reward *= mask
    a *= mask.reshape(mask.shape + (1,) * (a.ndim - 2))
    return dict(priv_s=priv_s, legal_move=legal, a=a, own_hand=own, seq_len=seq_len, reward=reward,
                terminal=terminal, bootstrap=bootstrap * mask)


def state_arrays(agent):
    return {"w." + k: v.detach().cpu().numpy() for k, v in agent.state_dict().items()}


def run_case(name, vdn, F, A, HID, T, B, G, hand=5, multi_step=3, gamma=0.999, pred_weight=0.25, seed=0, num_lstm_layer=2, num_fc_layer=1,
             skip_connect=False):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    NLL = num_lstm_layer
    agent = r2d2.R2D2Agent(vdn, multi_step, gamma, 0.9, "cpu", F, HID, A, NLL, hand, False, num_fc_layer=num_fc_layer,
                           skip_connect=skip_connect)
    # decorrelate target from online so the double-DQN path is exercised
    with torch.no_grad():
        for p in agent.target_net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    out = dict(state_arrays(agent))
    out["meta"] = np.array([int(vdn), F, A, HID, T, B, G, hand, multi_step], np.int64)
    out["gamma"] = np.array([gamma], np.float64)
    out["arch"] = np.array([num_lstm_layer, num_fc_layer, int(skip_connect)], np.int64)
    P = 2

    # ---- act (eps = 0 -> deterministic greedy branch; r2d2.py:247-303) ----
    if vdn:
        priv = (rng.random((G, 1, P, F)) < 0.15).astype(np.float32)
        legal = (rng.random((G, 1, P, A)) < 0.4).astype(np.float32)
        legal[..., 0] = 1
        eps = np.zeros((G, 1, P), np.float32)
        nh = G * P
    else:
        priv = (rng.random((G, 1, F)) < 0.15).astype(np.float32)
        legal = (rng.random((G, 1, A)) < 0.4).astype(np.float32)
        legal[..., 0] = 1
        eps = np.zeros((G, 1), np.float32)
        nh = G
    h0 = (rng.standard_normal((G, nh // G, NLL, HID)) * 0.3).astype(np.float32)
    c0 = (rng.standard_normal((G, nh // G, NLL, HID)) * 0.3).astype(np.float32)
    obs = {"priv_s": torch.tensor(priv), "legal_move": torch.tensor(legal), "eps": torch.tensor(eps),
           "h0": torch.tensor(h0), "c0": torch.tensor(c0)}
    with torch.no_grad():
        rep = agent.act(obs)
    out.update({"act.priv_s": priv, "act.legal_move": legal, "act.h0": h0, "act.c0": c0,
                "act.out_a": rep["a"].numpy(), "act.out_greedy_a": rep["greedy_a"].numpy(),
                "act.out_h0": rep["h0"].numpy(), "act.out_c0": rep["c0"].numpy()})

    # ---- compute_priority (r2d2.py:305-361; `temperature` is a dead lookup, F6a) ----
    nxt_priv = (rng.random(priv.shape) < 0.15).astype(np.float32)
    nxt_legal = (rng.random(legal.shape) < 0.4).astype(np.float32)
    nxt_legal[..., 0] = 1
    a = np.zeros(legal.shape[:-1], np.int64)
    it = np.nditer(a, flags=["multi_index"])
    for _ in it:
        a[it.multi_index] = rng.choice(np.flatnonzero(legal[it.multi_index]))
    nh0 = (rng.standard_normal(h0.shape) * 0.3).astype(np.float32)
    nc0 = (rng.standard_normal(c0.shape) * 0.3).astype(np.float32)
    rew = rng.random((G, 1)).astype(np.float32)
    boot = (rng.random((G, 1)) < 0.8).astype(np.float32)
    inp = {"priv_s": torch.tensor(priv), "legal_move": torch.tensor(legal), "a": torch.tensor(a),
           "next_priv_s": torch.tensor(nxt_priv), "next_legal_move": torch.tensor(nxt_legal),
           "temperature": torch.zeros_like(torch.tensor(eps)), "h0": torch.tensor(h0), "c0": torch.tensor(c0),
           "next_h0": torch.tensor(nh0), "next_c0": torch.tensor(nc0), "reward": torch.tensor(rew),
           "bootstrap": torch.tensor(boot)}
    with torch.no_grad():
        pr = agent.compute_priority(inp)["priority"]
    out.update({"prio.a": a, "prio.next_priv_s": nxt_priv, "prio.next_legal_move": nxt_legal, "prio.next_h0": nh0,
                "prio.next_c0": nc0, "prio.reward": rew, "prio.bootstrap": boot, "prio.out": pr.numpy()})

    # ---- loss / td_error / gradients (r2d2.py:383-499; selfplay.py:219-231) ----
    b = synth_batch(rng, T, B, F, A, hand, P if vdn else None)

    class Batch:
        pass
    batch = Batch()
    batch.obs = {"priv_s": torch.tensor(b["priv_s"]), "legal_move": torch.tensor(b["legal_move"]),
                 "own_hand": torch.tensor(b["own_hand"])}
    batch.h0 = {}
    batch.action = {"a": torch.tensor(b["a"])}
    batch.reward = torch.tensor(b["reward"])
    batch.terminal = torch.tensor(b["terminal"])
    batch.bootstrap = torch.tensor(b["bootstrap"])
    batch.seq_len = torch.tensor(b["seq_len"])
    weight = torch.tensor(rng.random(B).astype(np.float32) * 0.5 + 0.5)
    for k, v in b.items():
        out["loss." + k] = v
    out["loss.weight"] = weight.numpy()
    for tag, pw in (("rl", 0.0), ("aux", pred_weight)):
        if vdn and pw > 0:
            continue  # VDN + aux crashes in the reference
        agent.zero_grad()
        obs_copy = {k: v.clone() for k, v in batch.obs.items()}
        batch2 = Batch()
        batch2.__dict__.update(batch.__dict__)
        batch2.obs = obs_copy
        batch2.action = {"a": batch.action["a"].clone()}
        stat = Stat()
        loss, priority = agent.loss(batch2, pw, stat)
        total = (loss * weight).mean()
        total.backward()
        out["loss.%s.loss" % tag] = loss.detach().numpy()
        out["loss.%s.priority" % tag] = priority.detach().numpy()
        out["loss.%s.total" % tag] = np.array([total.item()], np.float32)
        for k, p in agent.online_net.named_parameters():
            out["loss.%s.grad.%s" % (tag, k)] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        if pw > 0:
            out["loss.aux.avg_xent"] = np.array([stat["aux1"].v], np.float32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "saved:", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    only = sys.argv[1:]
    cases = {
        "r2d2_iql_sad_small": lambda n: run_case(n, False, 838, 21, 64, 12, 6, 10, seed=1),
        "r2d2_vdn_small": lambda n: run_case(n, True, 783, 21, 64, 9, 4, 5, seed=2),
        # round 3: the architectures utils.load_op_model / --num_lstm_layer construct (r2d2.py:22-57, utils.py:46-57, selfplay.py:50)
        "r2d2_fc2_skip_small": lambda n: run_case(n, False, 783, 21, 64, 10, 6, 8, seed=3, num_fc_layer=2, skip_connect=True),
        "r2d2_skip_small": lambda n: run_case(n, False, 783, 21, 64, 8, 4, 8, seed=4, skip_connect=True),
        "r2d2_lstm1_small": lambda n: run_case(n, False, 838, 21, 64, 10, 6, 8, seed=5, num_lstm_layer=1),
        "r2d2_lstm3_fc2_small": lambda n: run_case(n, False, 838, 21, 64, 8, 4, 6, seed=6, num_lstm_layer=3, num_fc_layer=2),
    }
    for name, fn in cases.items():
        if not only or name in only:
            fn(name)
