import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd.composite import CompositeLearner
from tests.test_r2d2_kernels_gpu import _rand_batch, _rand_net
from tests.test_composite_abi_gpu import relerr
DEV = "cuda:0"
F, A, H, T, B = 838, 21, 512, 80, 128
W, Wt = _rand_net(F, H, A, seed=13), _rand_net(F, H, A, seed=14)
batch, weight = _rand_batch(T, B, F, A)
ref = None
for who in range(3):
    L = CompositeLearner(W, Wt, 3, 0.999, lr=1e-3, device=DEV)
    for it in range(3):
        l, p = L.loss(batch, weight, 0.25)
        torch.cuda.synchronize()
        g = {k: v.clone() for k, v in L.grad.items()}
        if ref is None:
            L2 = None
            ref = g
        bad = {k: "%.1e" % relerr(g[k], ref[k]) for k in g if not torch.equal(g[k], ref[k])}
        print("learner %d eval %d: differing from (learner 0, eval 0):" % (who, it), bad if bad else "none", flush=True)
    L.check_sync()
L.set_fused(0x39 | (1 << 8) | (1 << 25))
l, p = L.loss(batch, weight, 0.25); torch.cuda.synchronize()
bad = {k: "%.1e" % relerr(L.grad[k], ref[k]) for k in ref if not torch.equal(L.grad[k], ref[k])}
print("32 x 32 blocks vs (learner 0, eval 0):", bad)
