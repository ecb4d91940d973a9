import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd.composite import CompositeLearner
from tests.test_r2d2_kernels_gpu import _rand_batch, _rand_net
from tests.test_composite_abi_gpu import relerr
DEV = "cuda:0"
F, A, H, T, B = 838, 21, 512, 80, 128
W, Wt = _rand_net(F, H, A, seed=13), _rand_net(F, H, A, seed=14)
batch, weight = _rand_batch(T, B, F, A)
batch["priv_s"] = (batch["priv_s"] > 0.8).float()
b16 = dict(batch); del b16["priv_s"]
b16["priv_s_bf16"] = torch.zeros(T, B, 1, 896, dtype=torch.bfloat16, device=DEV)
b16["priv_s_bf16"][:, :, 0, :F] = batch["priv_s"]
flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else None
res = []
for name, bt in (("f32 a", batch), ("bf16 a", b16), ("f32 b", batch), ("bf16 b", b16)):
    L = CompositeLearner(W, Wt, 3, 0.999, lr=1e-3, device=DEV)
    if flags is not None:
        L.set_fused(flags)
    l, p = L.loss(bt, weight, 0.25)
    torch.cuda.synchronize()
    res.append((name, l.clone(), p.clone(), {k: v.clone() for k, v in L.grad.items()}))
    L.check_sync()
for i in range(1, 4):
    n0, l0, p0, g0 = res[0]; n, l, p, g = res[i]
    bad = {k: "%.2e" % relerr(g[k], g0[k]) for k in g if not torch.equal(g[k], g0[k])}
    print(n0, "vs", n, "loss eq", torch.equal(l, l0), "prio eq", torch.equal(p, p0), "differing grads:", bad)
