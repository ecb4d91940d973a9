"""rocprofv3 --kernel-trace --stats -- python tools/learner_update_breakdown.py N [set_fused flags] : kernels of N composite learner updates"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_r2d2_kernels_gpu import _rand_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F, H, A, T, B = 838, 512, 21, 80, 128
W = init_weights(F, H, A, 5, 1)
L = CompositeLearner(W, W, 3, 0.999, device="cuda:0")
if len(sys.argv) > 2:
    L.set_fused(int(sys.argv[2], 0))
batch, weight = _rand_batch(T, B, F, A)
b16 = dict(batch); del b16["priv_s"]
b16["priv_s_bf16"] = torch.zeros(T, B, 1, 896, dtype=torch.bfloat16, device="cuda:0")
b16["priv_s_bf16"][:, :, 0, :F] = batch["priv_s"]
for _ in range(3):
    L.loss(b16, weight, 0.0); L.optimizer_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    L.loss(b16, weight, 0.0); L.optimizer_step()
torch.cuda.synchronize()
print("ms per update", (time.perf_counter() - t0) / N * 1e3)
L.check_sync()
