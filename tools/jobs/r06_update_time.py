"""ms per learner update (configs[2] shape) of the composite learner, median of 10 blocks of 50 updates: for A/B runs under developer switches"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd import _lib
if os.environ.get("UPD_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UPD_LIB"])      # A/B against another build of the library
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
from tests.test_r2d2_kernels_gpu import _rand_batch
F, H, A, T, B = 838, 512, 21, 80, int(os.environ.get("UPD_B", "128"))
W = init_weights(F, H, A, 5, 1)
L = CompositeLearner(W, W, 3, 0.999, device="cuda:0")
if len(sys.argv) > 1:
    L.set_fused(int(sys.argv[1], 0))
batch, weight = _rand_batch(T, B, F, A)
for _ in range(20):
    L.loss(batch, weight, 0.0); L.optimizer_step()
torch.cuda.synchronize()
ts = []
for blk in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.loss(batch, weight, 0.0); L.optimizer_step()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 50)
L.check_sync()
ts.sort()
print("ms per update: median %.4f min %.4f max %.4f   (%s)" % (ts[5], ts[0], ts[-1], " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("HSAD_") or k == "UPD_LIB")))
