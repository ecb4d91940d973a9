"""Generates tests/golden/ref_small.pthw (+ _legacy.pthw, _expected.npz) from the REFERENCE: a randomly initialised
pyhanabi/r2d2.py R2D2Agent's online_net.state_dict() written with torch.save -- exactly what common_utils/saver.py:17-61 stores
and utils.load_weight (utils.py:278-299) reads -- so that the loaders of hanabi_sad_amd/checkpoint.py are tested against a file
the reference itself produced.  Data only (tensors).  Run in the authoring container:

    PYTORCH_JIT=0 python tests/golden/make_pthw_fixture.py

  ref_small.pthw           state_dict of R2D2Net(in 838, hid 64, out 21, 2 LSTM layers, hand 5)
  ref_small_legacy.pthw    the same without pred.* (files from before the auxiliary head) plus an unknown key
  ref_small_expected.npz   the reference agent's greedy_act on a fixed input with those weights (what a loaded agent must do)"""
import os
import sys
from collections import OrderedDict

os.environ.setdefault("PYTORCH_JIT", "0")
sys.path.insert(0, "/root/reference/pyhanabi")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import r2d2  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(20260928)
agent = r2d2.R2D2Agent(False, 3, 0.999, 0.9, "cpu", 838, 64, 21, 2, 5, False)
sd = agent.online_net.state_dict()
torch.save(sd, os.path.join(OUT, "ref_small.pthw"))
legacy = OrderedDict((k, v) for k, v in sd.items() if not k.startswith("pred."))
legacy["obsolete.weight"] = torch.zeros(3)
torch.save(legacy, os.path.join(OUT, "ref_small_legacy.pthw"))
rng = np.random.default_rng(5)
N = 12
priv = (rng.random((N, 838)) < 0.15).astype(np.float32)
legal = (rng.random((N, 21)) < 0.4).astype(np.float32)
legal[:, 0] = 1
h0 = (rng.standard_normal((2, N, 64)) * 0.3).astype(np.float32)
c0 = (rng.standard_normal((2, N, 64)) * 0.3).astype(np.float32)
with torch.no_grad():
    g, hid = agent.greedy_act(torch.tensor(priv), torch.tensor(legal), {"h0": torch.tensor(h0), "c0": torch.tensor(c0)})
np.savez(os.path.join(OUT, "ref_small_expected.npz"), priv_s=priv, legal_move=legal, h0=h0, c0=c0, greedy_a=g.numpy(),
         out_h0=hid["h0"].numpy(), out_c0=hid["c0"].numpy())
print("wrote", sorted(sd.keys()))
