"""Developer tool: are the learner's LSTM weight gradients bit-stable from update to update, and across schedule switches?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.composite import CompositeLearner
from tests.test_r2d2_kernels_gpu import _rand_batch, _rand_net
DEV="cuda:0"
F,H,T,B,A=838,512,80,128,21
W, Wt = _rand_net(F, H, A, seed=13), _rand_net(F, H, A, seed=14)
batch, weight = _rand_batch(T, B, F, A)
for seq in ([1, 0x201, 1, 1, 0x201, 1], [1, 1, 0x201, 0x201, 1]):
    L = CompositeLearner(W, Wt, 3, 0.999, device=DEV)
    first = {}
    for flags in seq:
        L.set_fused(flags)
        loss, prio = L.loss(batch, weight, 0.25)
        torch.cuda.synchronize()
        g = {k: v.clone() for k, v in L.grad.items()}
        if flags in first:
            bad = {k: float((g[k] - first[flags][k]).abs().max() / first[flags][k].abs().max()) for k in g if k.startswith("lstm.weight") and not torch.equal(g[k], first[flags][k])}
            print([hex(x) for x in seq], "revisit", hex(flags), "differs:", bad)
        else:
            first[flags] = g
    L.check_sync()
    L.close()
