"""Developer tool (CPU): what gradient error do bf16 MATMUL OPERANDS alone cause?  Plain torch autograd over the golden cases with every
matmul operand (weights, activations, upstream gradients) and every stored activation rounded to bf16, fp32 accumulation -- the
rounding points of the kernels.  Its output matched the kernels' measured errors to two digits (e.g. net.0.weight 2.3 % / 3.4 % for the
two-fc-layer cases: ReLU-mask flips of near-zero first-layer units), which is how the tolerances of tests/test_r2d2_arch_gpu.py were
told apart from bugs.  usage: python tools/emulate_bf16_grad.py r2d2_fc2_skip_small ..."""
import numpy as np, torch, sys, os
sys.path.insert(0,'/root/repo')
from tests import r2d2_torch_ref as ref
import torch.nn.functional as F
class BL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        xb, wb = x.bfloat16().float(), w.bfloat16().float()
        ctx.save_for_backward(xb, wb)
        return xb @ wb.t()
    @staticmethod
    def backward(ctx, g):
        xb, wb = ctx.saved_tensors
        gb = g.bfloat16().float()
        return gb @ wb, gb.reshape(-1, gb.shape[-1]).t() @ xb.reshape(-1, xb.shape[-1])
def lin(x, w, b): return BL.apply(x, w) + b
def rb(x): return (x.bfloat16().float() - x).detach() + x   # straight-through bf16 rounding of a stored activation
def fwd(W, priv, legal, a, nl):
    x = rb(F.relu(lin(priv, W["net.0.weight"], W["net.0.bias"])))
    if "net.2.weight" in W: x = rb(F.relu(lin(x, W["net.2.weight"], W["net.2.bias"])))
    T,B,_ = x.shape; H = x.shape[-1]
    inp = x
    for l in range(nl):
        h = torch.zeros(B,H); c = torch.zeros(B,H); outs=[]
        for t in range(T):
            g = lin(inp[t], W["lstm.weight_ih_l%d"%l], 0) + lin(h, W["lstm.weight_hh_l%d"%l], 0) + W["lstm.bias_ih_l%d"%l] + W["lstm.bias_hh_l%d"%l]
            i,f,gg,o = g.chunk(4,1)
            c = torch.sigmoid(f)*c + torch.sigmoid(i)*torch.tanh(gg)
            h = rb(torch.sigmoid(o)*torch.tanh(c)); outs.append(h)
        inp = torch.stack(outs,0)
    o = inp
    av = lin(o, W["fc_a.weight"], W["fc_a.bias"]); v = lin(o, W["fc_v.weight"], W["fc_v.bias"])
    la = av*legal; q = v + la - la.mean(2, keepdim=True)
    qa = q.gather(2, a.unsqueeze(2)).squeeze(2)
    return qa, ((1+q-q.min())*legal).argmax(2), o
for name in sys.argv[1:]:
    z = np.load('/root/repo/tests/golden/%s.npz'%name)
    nl = int(z["arch"][0]) if "arch" in z.files else 2
    Won, Wtg = ref.weights_from_npz(z,"online_net."), ref.weights_from_npz(z,"target_net.")
    for v in Won.values(): v.requires_grad_(True)
    b = {k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("loss.") and k.count(".")==1}
    qa, greedy, o = fwd(Won, b["priv_s"], b["legal_move"], b["a"], nl)
    with torch.no_grad(): tqa,_,_ = fwd(Wtg, b["priv_s"], b["legal_move"], greedy, nl)
    T = qa.shape[0]; ms=3
    tq = torch.cat([tqa[ms:], tqa[:ms]],0); tq[-ms:]=0
    target = b["reward"] + b["bootstrap"]*(0.999**ms)*tq
    mask = (torch.arange(T).unsqueeze(1) < b["seq_len"].unsqueeze(0)).float()
    err = (target - qa)*mask
    loss = F.smooth_l1_loss(err, torch.zeros_like(err), reduction="none").sum(0)
    (loss*b["weight"]).mean().backward()
    rel = {k: float((v.grad - torch.tensor(z["loss.rl.grad."+k])).norm()/torch.tensor(z["loss.rl.grad."+k]).norm().clamp(min=1e-12)) for k,v in Won.items() if v.grad is not None}
    print(name, {k: round(x,4) for k,x in rel.items() if x>3e-3})
