"""learner update vs the split-K factor of the weight-gradient GEMMs"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
from tests.test_r2d2_kernels_gpu import _rand_batch
F, H, A, T, B = 838, 512, 21, 80, 128
W = init_weights(F, H, A, 5, 0)
batch, weight = _rand_batch(T, B, F, A)
lr = CompositeLearner(W, W, 3, 0.999, device="cuda:0")
lr.loss(batch, weight, 0.0)
for split in (8, 4, 5, 8):
    _lib.check(lr.lib.hsad_r2d2_learner_set_schedule(lr.h, 4, split))
    for _ in range(5):
        lr.loss(batch, weight, 0.0); lr.optimizer_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        lr.loss(batch, weight, 0.0); lr.optimizer_step()
    torch.cuda.synchronize()
    print("wgrad_split=%d  %.3f ms/update" % (split, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
lr.check_sync()
