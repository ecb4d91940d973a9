"""Generates tests/golden/obl_small.npz from the REFERENCE's OBL model family (pyhanabi/tools/obl_model.py:18-303: PublicLSTMNet
with the private / public input split, its R2D2Agent.act) in the authoring container.  The reference file instantiates a
512-unit model on "cuda:0" at import time (obl_model.py:305-315), which cannot run here (no GPU), so the module text is
executed up to that statement and a small model is built on the CPU from its classes.  Data only: random-init weights (with the
reference's state_dict key names), inputs, and the reference's outputs.

    PYTORCH_JIT=0 python tests/golden/make_obl_golden.py"""
import os
import sys

os.environ.setdefault("PYTORCH_JIT", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

SRC = "/root/reference/pyhanabi/tools/obl_model.py"
OUT = os.path.dirname(os.path.abspath(__file__))
text = open(SRC).read()
ns = {"__name__": "obl_model_ref"}
exec(compile(text[:text.index("obl_model = R2D2Agent(")], SRC, "exec"), ns)      # classes only, no module-level model
torch.manual_seed(77)
H, N = 64, 24
agent = ns["R2D2Agent"](False, 1, 0.999, 0.9, "cpu", (783, 658, 533), H, 21, 2)
sd = agent.online_net.state_dict()
rng = np.random.default_rng(9)
priv = (rng.random((N, 1, 838)) < 0.15).astype(np.float32)
priv[:, :, :125] = 0
legal = (rng.random((N, 1, 21)) < 0.4).astype(np.float32)
legal[:, :, 0] = 1
h0 = (rng.standard_normal((N, 1, 2, H)) * 0.3).astype(np.float32)
c0 = (rng.standard_normal((N, 1, 2, H)) * 0.3).astype(np.float32)
with torch.no_grad():
    reply = agent.act({"priv_s": torch.tensor(priv), "legal_move": torch.tensor(legal), "eps": torch.zeros(N, 1),
                       "h0": torch.tensor(h0), "c0": torch.tensor(c0)})
    # the advantages behind the action, for a numeric comparison (online_net.act on the sliced inputs, obl_model.py:262-268)
    p = torch.tensor(priv).squeeze(1)[:, :783][:, 125:]
    adv, _ = agent.online_net.act(p, p[:, 125:], {"h0": torch.tensor(h0), "c0": torch.tensor(c0)})
out = {"w." + k: v.numpy() for k, v in sd.items()}
out.update(priv_s=priv, legal_move=legal, h0=h0, c0=c0, out_a=reply["a"].numpy(), out_greedy_a=reply["greedy_a"].numpy(),
           out_h0=reply["h0"].numpy(), out_c0=reply["c0"].numpy(), adv=adv.numpy())
np.savez_compressed(os.path.join(OUT, "obl_small.npz"), **out)
print(sorted(sd.keys()), {k: v.shape for k, v in out.items() if not k.startswith("w.")})
