"""Generates tests/golden/op_zoo/ -- an Other-Play model zoo in the layout utils.load_op_model reads (models/op/<method>/M{idx}.pthw,
pyhanabi/utils.py:36-84) -- from the REFERENCE: for one index of each architecture class (M0 default, M3 skip connection, M6 two
fc layers, M9 both; utils.py:46-57) a randomly initialised pyhanabi/r2d2.py R2D2Agent's online_net.state_dict(), plus what that
reference agent's greedy_act answers on a fixed input (op_zoo_expected.npz).  Data only (tensors).  Run in the authoring container:

    PYTORCH_JIT=0 python tests/golden/make_op_zoo_fixture.py"""
import os
import sys

os.environ.setdefault("PYTORCH_JIT", "0")
sys.path.insert(0, "/root/reference/pyhanabi")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import r2d2  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "op_zoo", "models", "op", "sad")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(11)
N, F, H, A = 12, 783, 64, 21
priv = (rng.random((N, F)) < 0.15).astype(np.float32)
legal = (rng.random((N, A)) < 0.4).astype(np.float32)
legal[:, 0] = 1
h0 = (rng.standard_normal((2, N, H)) * 0.3).astype(np.float32)
c0 = (rng.standard_normal((2, N, H)) * 0.3).astype(np.float32)
out = dict(priv_s=priv, legal_move=legal, h0=h0, c0=c0)
for idx in (0, 3, 6, 9):
    num_fc, skip = (1 if idx < 6 else 2), (3 <= idx < 6 or idx >= 9)
    torch.manual_seed(100 + idx)
    # hid_dim 64 instead of the zoo's 512 keeps the fixture small; load_op_model reads the dimensions off the tensors
    agent = r2d2.R2D2Agent(False, 3, 0.999, 0.9, "cpu", F, H, A, 2, 5, False, num_fc_layer=num_fc, skip_connect=skip)
    torch.save(agent.online_net.state_dict(), os.path.join(OUT, "M%d.pthw" % idx))
    with torch.no_grad():
        g, hid = agent.greedy_act(torch.tensor(priv), torch.tensor(legal), {"h0": torch.tensor(h0), "c0": torch.tensor(c0)})
        adv, _ = agent.online_net.act(torch.tensor(priv), {"h0": torch.tensor(h0), "c0": torch.tensor(c0)})
    out["M%d.greedy_a" % idx] = g.numpy()
    out["M%d.adv" % idx] = adv.numpy()
    out["M%d.out_h0" % idx] = hid["h0"].numpy()
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "op_zoo_expected.npz"), **out)
print("wrote", sorted(os.listdir(OUT)))
