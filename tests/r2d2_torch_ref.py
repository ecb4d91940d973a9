"""TEST INFRASTRUCTURE: plain PyTorch fp32 restatement of the R2D2 network / agent math
(reference pyhanabi/r2d2.py:13-157 R2D2Net, :159-499 R2D2Agent), written functionally over a weight dict
with the reference's state_dict key names.  Pinned against golden vectors generated from the reference
itself (tests/golden/r2d2_*.npz, tests/test_r2d2_golden_cpu.py) and used as the fp32 reference for the
full-size HIP kernel tests.  Never imported by the product package."""
import torch
import torch.nn.functional as F


def weights_from_npz(z, prefix):
    """prefix: 'online_net.' or 'target_net.' -> dict of tensors keyed like R2D2Net.state_dict()."""
    out = {}
    for k in z.files:
        if k.startswith("w." + prefix):
            out[k[len("w." + prefix):]] = torch.tensor(z[k])
    return out


def num_layers(W):
    return len([k for k in W if k.startswith("lstm.weight_ih_l")])


def lstm(W, x, h0, c0):
    """x [T,N,H]; h0/c0 [L,N,H] -> o [T,N,H], h [L,N,H], c [L,N,H]; gate order i,f,g,o like torch.nn.LSTM."""
    hs, cs = [], []
    inp = x
    for l in range(num_layers(W)):
        wih, whh = W["lstm.weight_ih_l%d" % l], W["lstm.weight_hh_l%d" % l]
        b = W["lstm.bias_ih_l%d" % l] + W["lstm.bias_hh_l%d" % l]
        h, c = h0[l], c0[l]
        outs = []
        for t in range(inp.shape[0]):
            g = inp[t] @ wih.t() + h @ whh.t() + b
            i, f, gg, o = g.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, 0)
        hs.append(h)
        cs.append(c)
    return inp, torch.stack(hs, 0), torch.stack(cs, 0)


def mlp(W, priv_s, masks=None):
    """masks: None = ReLU; else one 0/1 tensor per fc layer that REPLACES the layer's own ReLU decision (x = a * mask) -- the fp32
    network under the activation pattern of another arithmetic (bf16_relu_masks), see the two-fc-layer gradient tests"""
    a = priv_s @ W["net.0.weight"].t() + W["net.0.bias"]                  # r2d2.py:42-46
    x = F.relu(a) if masks is None else a * masks[0]
    if "net.2.weight" in W:                                               # num_fc_layer = 2
        a = x @ W["net.2.weight"].t() + W["net.2.bias"]
        x = F.relu(a) if masks is None else a * masks[1]
    return x


def bf16_relu_masks(W, priv_s):
    """The activation pattern of the input MLP when its GEMM operands and stored activations are bf16 (fp32 accumulation, fp32 bias) --
    the arithmetic of the production kernels -- next to the proof obligation that goes with it.  Returns (masks, flips): `masks` for
    mlp(..., masks=), `flips` = per layer (n_flipped, worst |fp32 pre-activation| / bound over the flipped units), where `bound` is the
    largest distance the bf16 arithmetic can move that unit's pre-activation: u * sum_k |x_k w_jk| for the operand roundings
    (u = 2^-9 per rounded operand; 0/1 inputs are exact) + |W| |x_fp32 - x_bf16| for what the layer below handed up + fp32 summation
    noise.  worst <= 1 means: every unit whose ReLU decision differs between the two arithmetics has an fp32 pre-activation within the
    bf16 rounding of zero, i.e. the reference's own decision there is not robust to the rounding of its inputs."""
    u, noise = 2.0 ** -9, 4e-6
    r = lambda t: t.bfloat16().float()
    with torch.no_grad():
        masks, flips = [], []
        x32, x16 = priv_s, r(priv_s)
        exact_in = bool((x32 == x16).all())
        for name in ("net.0", "net.2"):
            if name + ".weight" not in W:
                break
            w, b = W[name + ".weight"].detach(), W[name + ".bias"].detach()
            a32 = x32 @ w.t() + b
            a16 = x16 @ r(w).t() + b
            bound = ((1.0 if exact_in else 2.0) * u + noise) * (x16.abs() @ w.abs().t()) + (x32 - x16).abs() @ w.abs().t() + noise * b.abs()
            flip = (a32 > 0) != (a16 > 0)
            ratio = (a32.abs() / bound.clamp(min=1e-30))[flip]
            flips.append((int(flip.sum()), float(ratio.max()) if ratio.numel() else 0.0))
            masks.append((a16 > 0).float())
            x32, x16 = F.relu(a32), r(F.relu(a16))
            exact_in = False
    return masks, flips


def trunk(W, priv_s, h0, c0, masks=None):
    return lstm(W, mlp(W, priv_s, masks), h0, c0)


def net_act(W, priv_s, h0, c0, skip_connect=False):
    """R2D2Net.act (r2d2.py:65-78): priv_s [N,F] -> advantage [N,A], new hidden.  skip_connect applies HERE only (forward ignores it)."""
    x = mlp(W, priv_s.unsqueeze(0))
    o, h, c = lstm(W, x, h0, c0)
    if skip_connect:
        o = o + x
    return (o @ W["fc_a.weight"].t() + W["fc_a.bias"]).squeeze(0), h, c


def net_forward(W, priv_s, legal_move, action, h0, c0, masks=None):
    """R2D2Net.forward (r2d2.py:80-122) on [T,N,*]: qa, greedy action, q, lstm output."""
    o, _, _ = trunk(W, priv_s, h0, c0, masks)
    a = o @ W["fc_a.weight"].t() + W["fc_a.bias"]
    v = o @ W["fc_v.weight"].t() + W["fc_v.bias"]
    legal_a = a * legal_move
    q = v + legal_a - legal_a.mean(2, keepdim=True)                          # _duel: mean over ALL actions
    qa = q.gather(2, action.unsqueeze(2)).squeeze(2)
    legal_q = (1 + q - q.min()) * legal_move                                  # global min (r2d2.py:113)
    return qa, legal_q.argmax(2), q, o


def greedy_act(W, priv_s, legal_move, h0, c0, skip_connect=False):
    adv, h, c = net_act(W, priv_s, h0, c0, skip_connect)
    legal_adv = (1 + adv - adv.min()) * legal_move                            # r2d2.py:242
    return legal_adv.argmax(1), h, c


def zeros_hid(W, n, like):
    H = W["fc_v.weight"].shape[1]
    z = torch.zeros(num_layers(W), n, H, dtype=like.dtype, device=like.device)
    return z, z.clone()


def td_error(Won, Wtg, priv_s, legal_move, action, reward, bootstrap, seq_len, multi_step, gamma, online_masks=None):
    """R2D2Agent.td_error (r2d2.py:383-428): IQL layouts [T,B,*]; VDN layouts [T,B,P,*] are flattened to B*P rows
    (flat_4d, r2d2.py:363-381) and the Q-values summed over the players of a game."""
    T, B = priv_s.shape[:2]
    P = 0
    if priv_s.dim() == 4:
        P = priv_s.shape[2]
        priv_s, legal_move, action = priv_s.flatten(1, 2), legal_move.flatten(1, 2), action.flatten(1, 2)
    h0, c0 = zeros_hid(Won, priv_s.shape[1], priv_s)
    online_qa, greedy_a, _, lstm_o = net_forward(Won, priv_s, legal_move, action, h0, c0, online_masks)
    with torch.no_grad():
        target_qa, _, _, _ = net_forward(Wtg, priv_s, legal_move, greedy_a, h0, c0)
    if P:
        online_qa, target_qa = online_qa.view(T, B, P).sum(-1), target_qa.view(T, B, P).sum(-1)
    with torch.no_grad():
        target_qa = torch.cat([target_qa[multi_step:], target_qa[:multi_step]], 0)
        target_qa[-multi_step:] = 0
        target = reward + bootstrap * (gamma ** multi_step) * target_qa
    mask = (torch.arange(T, device=seq_len.device).unsqueeze(1) < seq_len.unsqueeze(0)).float()
    return (target - online_qa) * mask, lstm_o


def aux_xent(W, lstm_o, own_hand, seq_len):
    """aux_task_iql + cross_entropy (r2d2.py:133-156, 430-440): own_hand [T,B,hand*3]."""
    T, B, _ = own_hand.shape
    tgt = own_hand.view(T, B, -1, 3)
    slot = tgt.sum(3)
    logit = (lstm_o @ W["pred.weight"].t() + W["pred.bias"]).view(tgt.shape)
    logq = F.log_softmax(logit, -1)
    xent = -((tgt * logq).sum(-1) * slot).sum(-1) / slot.sum(-1).clamp(min=1e-6)
    return xent.sum(0), (xent.sum(0) / seq_len).mean()


def loss(Won, Wtg, batch, multi_step, gamma, pred_weight, online_masks=None):
    """R2D2Agent.loss (r2d2.py:461-499) -> per-sequence loss [B], priority [T,B]."""
    err, lstm_o = td_error(Won, Wtg, batch["priv_s"], batch["legal_move"], batch["a"], batch["reward"],
                           batch["bootstrap"], batch["seq_len"], multi_step, gamma, online_masks)
    rl = F.smooth_l1_loss(err, torch.zeros_like(err), reduction="none").sum(0)
    out = rl
    if pred_weight > 0:
        x, _ = aux_xent(Won, lstm_o, batch["own_hand"], batch["seq_len"])
        out = rl + pred_weight * x
    return out, err.abs()


def compute_priority(Won, Wtg, priv_s, legal_move, a, next_priv_s, next_legal_move, h0, c0, next_h0, next_c0, reward,
                     bootstrap, multi_step, gamma, num_player=1, skip_connect=False):
    """R2D2Agent.compute_priority, IQL, flat [N,*] inputs with hidden [L,N,H] (r2d2.py:305-361)."""
    qa, _, _, _ = net_forward(Won, priv_s.unsqueeze(0), legal_move.unsqueeze(0), a.unsqueeze(0), h0, c0)
    next_a, _, _ = greedy_act(Won, next_priv_s, next_legal_move, next_h0, next_c0, skip_connect)
    tqa, _, _, _ = net_forward(Wtg, next_priv_s.unsqueeze(0), next_legal_move.unsqueeze(0), next_a.unsqueeze(0), next_h0,
                               next_c0)
    qa, tqa = qa.squeeze(0), tqa.squeeze(0)
    if num_player > 1:                                                        # VDN: sum over the players of a game
        qa, tqa = qa.view(-1, num_player).sum(1), tqa.view(-1, num_player).sum(1)
    target = reward + bootstrap * (gamma ** multi_step) * tqa
    return (target - qa).abs()
