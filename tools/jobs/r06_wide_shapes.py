"""wide vs 32 x 32 blocking of the four-stage BPTT at other batch sizes: bit equality of the LSTM weight gradients, run-to-run determinism"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd.composite import CompositeLearner
from tests.test_r2d2_kernels_gpu import _rand_batch, _rand_net
F, A, H = 838, 21, 512
CFG = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(80, 64), (80, 32), (24, 96), (80, 128), (7, 64), (1, 32)]
for T, B in CFG:
    W, Wt = _rand_net(F, H, A, seed=13), _rand_net(F, H, A, seed=14)
    batch, weight = _rand_batch(T, B, F, A)
    L = CompositeLearner(W, Wt, 3, 0.999, device="cuda:0")
    res = {}
    for rep in range(3):
        for name, flags in (("wide", 0x39 | (1 << 8)), ("32x32", 0x39 | (1 << 8) | (1 << 25))):
            L.set_fused(flags)
            loss, prio = L.loss(batch, weight, 0.25)
            torch.cuda.synchronize()
            g = {k: v.clone() for k, v in L.grad.items()}
            res.setdefault(name, []).append(g)
    L.check_sync()
    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    keys = list(res["wide"][0].keys())
    print("T %d B %d" % (T, B), flush=True)
    for k in keys:
        same_ab = torch.equal(res["wide"][0][k], res["32x32"][0][k])
        det_w = all(torch.equal(res["wide"][0][k], res["wide"][i][k]) for i in (1, 2))
        det_o = all(torch.equal(res["32x32"][0][k], res["32x32"][i][k]) for i in (1, 2))
        if not (same_ab and det_w and det_o):
            print("   %-24s wide==32x32 %s (rel %.2e)  wide deterministic %s (rel %.2e)  32x32 deterministic %s (rel %.2e)" % (
                k, same_ab, rel(res["wide"][0][k], res["32x32"][0][k]), det_w, max(rel(res["wide"][0][k], res["wide"][i][k]) for i in (1, 2)),
                det_o, max(rel(res["32x32"][0][k], res["32x32"][i][k]) for i in (1, 2))))
    del L
