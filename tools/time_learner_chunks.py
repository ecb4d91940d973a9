"""learner update time vs the number of time chunks of the layer pipeline (hsad_r2d2_learner_set_schedule)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
from tests.test_r2d2_kernels_gpu import _rand_batch
F, H, A, T, B = 838, 512, 21, 80, 128
W = init_weights(F, H, A, 5, 1)
batch, weight = _rand_batch(T, B, F, A)
for chunks in (2, 4, 5, 8, 10):
    for split in (8,):
        L = CompositeLearner(W, W, 3, 0.999, device="cuda:0")
        L.chunks, L.wgrad_split = chunks, split
        for _ in range(4):
            L.loss(batch, weight, 0.0); L.optimizer_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            L.loss(batch, weight, 0.0); L.optimizer_step()
        torch.cuda.synchronize()
        print("chunks %2d split %2d: %.3f ms/update" % (chunks, split, (time.perf_counter() - t0) / 40 * 1e3), flush=True)
        L.check_sync()
        L.close()
