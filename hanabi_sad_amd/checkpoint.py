"""Checkpoints in the reference's `.pthw` format and its model loaders.

A `.pthw` file is `torch.save(R2D2Net.state_dict())` with the keys net.0.*, lstm.{weight,bias}_{ih,hh}_l{0,1}, fc_v.*, fc_a.*,
pred.* (common_utils/saver.py:17-61 writes them, pyhanabi/utils.py:278-299 reads them back).  The kernels keep their master
weights under exactly those names, so published weights map 1:1.

  save_weights / load_weights      plain round trip of a weight dict
  load_weight(target, file)        utils.load_weight: fill a net's weight dict from a file, keeping the net's own value (with a
                                   warning) for keys the file lacks and dropping keys the net does not have (legacy files)
  load_sad_model(files, device)    utils.load_sad_model: one acting agent per file, dimensions read off the tensors
  load_op_model(method, i, j, ..)  utils.load_op_model: the Other-Play zoo models/op/<method>/M{idx}.pthw; idx selects the
                                   architecture (1-2 fc layers, skip connection); all twelve run on the kernels"""
import os

import torch

from . import _lib
from .r2d2 import PARAM_ORDER, R2D2Agent, R2D2NetKernels, arch_of, param_order


def save_weights(weights, path):
    """online_net.state_dict() in the reference's key names -> `.pthw`"""
    torch.save({k: weights[k].detach().cpu().clone() for k in param_order(*arch_of(weights))}, path)


def load_weights(path, device="cpu"):
    """-> {R2D2Net.state_dict() name: fp32 tensor}.  Accepts a bare net (what `save_weights` and the reference's savers write) or a
    whole agent's state_dict (`online_net.*` / `target_net.*`: the online net is taken)."""
    sd = torch.load(path, map_location=device)
    if any(k.startswith("online_net.") for k in sd):
        sd = {k[len("online_net."):]: v for k, v in sd.items() if k.startswith("online_net.")}
    names = param_order(*arch_of(sd))
    missing = [k for k in names if k not in sd]
    if missing:
        raise KeyError("checkpoint lacks %s (expected R2D2Net.state_dict() keys)" % missing)
    return {k: sd[k].float() for k in names}


def load_weight(target, weight_file, device="cpu", verbose=True):
    """utils.load_weight (pyhanabi/utils.py:278-299) on a weight dict: `target` (name -> tensor, e.g. R2D2NetKernels.w or
    init_weights(...)) is updated IN PLACE from the file.  Keys missing in the file keep the target's value ("warning: k not
    loaded" -- e.g. pred.* in checkpoints written before the auxiliary head existed); keys the target does not have are
    ignored ("removing: k not used").  Shapes must agree.  Returns (loaded, kept, dropped) key lists."""
    sd = torch.load(weight_file, map_location=device)
    loaded, kept, dropped = [], [], []
    for k, v in target.items():
        if k not in sd:
            if verbose:
                print("warning: %s not loaded" % k)
            kept.append(k)
            continue
        if tuple(sd[k].shape) != tuple(v.shape):
            raise ValueError("%s: checkpoint shape %s does not match the net's %s" % (k, tuple(sd[k].shape), tuple(v.shape)))
        v.copy_(sd[k].to(v.device, v.dtype))
        loaded.append(k)
    for k in sd:
        if k not in target:
            if verbose:
                print("removing: %s not used" % k)
            dropped.append(k)
    return loaded, kept, dropped


def _dims(sd):
    H = sd["fc_a.weight"].shape[1]
    return sd["net.0.weight"].shape[1], H, sd["fc_a.weight"].shape[0]


def _blank(in_dim, hid_dim, out_dim, hand_size=5, num_fc_layer=1, num_lstm_layer=2):
    from .selfplay import init_weights
    return init_weights(in_dim, hid_dim, out_dim, hand_size, 0, num_lstm_layer=num_lstm_layer, num_fc_layer=num_fc_layer)


def agent_from_file(weight_file, device="cuda:0", multi_step=3, gamma=0.999, precision="bf16", skip_connect=False):
    """an acting agent (online = target = the file's weights) on the HIP kernels.  The architecture (1-2 fc layers, 1-3 LSTM layers)
    is read off the file's keys; skip_connect is a constructor flag in the reference and therefore here.  The reference default in
    bf16 runs the Python-orchestrated kernels or the composite entry points alike; every other architecture runs through the
    library's composite entry points (bf16) or the fp32 exact mode."""
    from .composite import CNet, CompositeAgent
    from .r2d2 import arch_of
    sd = torch.load(weight_file, map_location="cpu")
    nfc, nl = arch_of(sd)
    if not 1 <= nl <= 3 or any(k.startswith("net.") and k.split(".")[1] not in ("0", "2") for k in sd):
        raise _lib.HsadError("%s: R2D2Net with %d LSTM layers / input MLP keys %s is outside what the reference's loaders construct"
                             % (weight_file, nl, [k for k in sd if k.startswith("net.")]))
    in_dim, hid, out_dim = _dims(sd)
    hand = sd["pred.weight"].shape[0] // 3 if "pred.weight" in sd else 5
    W = _blank(in_dim, hid, out_dim, hand, nfc, nl)
    load_weight(W, weight_file, verbose=False)
    if precision == "fp32":
        from .r2d2_f32 import R2D2NetF32
        net = R2D2NetF32(W, device, skip_connect=skip_connect)
        return R2D2Agent(net, net, multi_step, gamma)
    if (nfc, nl) == (1, 2) and not skip_connect:
        net = R2D2NetKernels.make(W, device, precision)
        return R2D2Agent(net, net, multi_step, gamma)
    net = CNet(W, device, skip_connect=skip_connect)
    return CompositeAgent(net, net, multi_step, gamma)


def load_sad_model(weight_files, device="cuda:0"):
    """utils.load_sad_model (pyhanabi/utils.py:19-33): R2D2Agent(vdn=False, multi_step 3, gamma 0.999, eta 0.9, 2 LSTM layers,
    hand 5) per file with input / output dims taken from net.0.weight / fc_a.weight"""
    return [agent_from_file(f, device, 3, 0.999) for f in weight_files]


def op_model_arch(idx):
    """architecture of models/op/<method>/M{idx}.pthw (utils.py:46-57): (num_fc_layer, skip_connect)"""
    return (1 if idx < 6 else 2), (3 <= idx < 6 or idx >= 9)


def load_op_model(method, idx1, idx2, device="cuda:0", root=None, precision="bf16"):
    """utils.load_op_model (pyhanabi/utils.py:36-84): the two-player Other-Play zoo, all twelve architectures -- M0-2 default,
    M3-5 skip connection, M6-8 two fc layers, M9-11 both -- on the kernels."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    folder = os.path.join(root, "models", "op", method)
    agents = []
    for idx in (idx1, idx2):
        if idx is None:
            continue
        path = os.path.join(folder, "M%d.pthw" % idx)
        if not os.path.exists(path):
            raise FileNotFoundError("Cannot find weight at: %s" % path)
        num_fc, skip = op_model_arch(idx)
        sd = torch.load(path, map_location="cpu")
        if (2 if "net.2.weight" in sd else 1) != num_fc:
            raise _lib.HsadError("M%d should have %d fc layer(s) (utils.py:46-57); the file has keys %s" % (idx, num_fc, [k for k in sd if k.startswith("net.")]))
        agents.append(agent_from_file(path, device, 3, 0.999, precision, skip_connect=skip))
    return agents
