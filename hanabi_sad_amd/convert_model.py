"""Export a trained R2D2 net as a stand-alone TorchScript module for a search runtime (what pyhanabi/tools/convert_model.py:21-84
produces: `<model>.sparta`, a module whose forward takes {"s" [B, in_dim], "h0", "c0" [B, L, H]} and returns the advantage head "a"
[B, A] and the new "h0" / "c0" in the same batch-first layout).  Torch only -- the file is consumed outside this stack; no kernel of
the library is involved.

    python -m hanabi_sad_amd.convert_model --model model.pthw [--device cpu]"""
import argparse
from typing import Dict

import torch
from torch import nn


class SearchNet(torch.jit.ScriptModule):
    """parameter names follow R2D2Net's state_dict (net.0, lstm, fc_v, fc_a: pyhanabi/r2d2.py:22-57) so that a trained file loads by
    name; the value head is carried but not evaluated (the consumer ranks actions by advantage)"""

    def __init__(self, in_dim, hid_dim, out_dim, num_lstm_layer, num_fc_layer=1):
        super().__init__()
        self.in_dim = in_dim
        layers = [nn.Linear(in_dim, hid_dim), nn.ReLU()]
        for _ in range(1, num_fc_layer):         # R2D2Net's ff_layers (pyhanabi/r2d2.py:36-40): net.2, net.4, ...
            layers += [nn.Linear(hid_dim, hid_dim), nn.ReLU()]
        self.net = nn.Sequential(*layers)
        self.lstm = nn.LSTM(hid_dim, hid_dim, num_layers=num_lstm_layer)
        self.fc_v = nn.Linear(hid_dim, 1)
        self.fc_a = nn.Linear(hid_dim, out_dim)

    @torch.jit.script_method
    def forward(self, obs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        state = (obs["h0"].transpose(0, 1).contiguous(), obs["c0"].transpose(0, 1).contiguous())     # [B, L, H] -> [L, B, H]
        s = obs["s"]
        assert s.size(1) == self.in_dim
        o, (h, c) = self.lstm(self.net(s.unsqueeze(0)), state)
        return {"a": self.fc_a(o).squeeze(0), "h0": h.transpose(0, 1).contiguous(), "c0": c.transpose(0, 1).contiguous()}


def convert(model_path, device="cpu", save_path=None):
    """-> (module, path it was saved to).  Shapes come from the file: in_dim / hid from net.0.weight, A from fc_a.weight, the number of
    fc layers from net.2.* and of LSTM layers from the lstm.weight_ih_l* keys (load_weights also accepts a whole agent's `online_net.*`
    file); the auxiliary head `pred` is the ONLY part of the file the exported module may leave out."""
    from .checkpoint import load_weights
    from .r2d2 import arch_of
    sd = load_weights(model_path, device)
    hid, in_dim = sd["net.0.weight"].shape
    out_dim = sd["fc_a.weight"].shape[0]
    num_fc, layers = arch_of(sd)
    m = SearchNet(int(in_dim), int(hid), int(out_dim), layers, num_fc).to(device)
    own = m.state_dict()
    missing = [k for k in own if k not in sd]
    dropped = [k for k in sd if k not in own and not k.startswith("pred.")]
    if missing or dropped:
        raise KeyError("convert_model: %s lacks %s / holds %s that the exported net would silently drop" % (model_path, missing, dropped))
    m.load_state_dict({k: sd[k].to(device) for k in own})
    save_path = save_path or model_path.rsplit(".", 1)[0] + ".sparta"
    torch.jit.save(m, save_path)
    return m, save_path


def main(argv=None):
    p = argparse.ArgumentParser(description="export an R2D2 net as TorchScript (pyhanabi/tools/convert_model.py)")
    p.add_argument("--model", type=str, required=True)
    p.add_argument("--device", type=str, default="cpu")
    args = p.parse_args(argv)
    _, path = convert(args.model, args.device)
    print("saving model to:", path)


if __name__ == "__main__":
    main()
