"""raw per-step stamps of the forward launch's online layers for a few steps, all 16 members of row block 0: developer aid"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
from tests.test_r2d2_kernels_gpu import _rand_batch
F, H, A, T, B = 838, 512, 21, 80, 128
lib = _lib.load_library()
W = init_weights(F, H, A, 5, 1)
L = CompositeLearner(W, W, 3, 0.999, device="cuda:0")
batch, weight = _rand_batch(T, B, F, A)
for _ in range(3):
    L.loss(batch, weight, 0.0); L.optimizer_step()
torch.cuda.synchronize()
KREC, KNB, KT, KK = 6, 16, 96, 12
NW = 2 * KREC * KNB * KT * KK
buf = (C.c_uint64 * NW)()
_lib.check(lib.hsad_lstm_debug_enable(2)); _lib.check(lib.hsad_lstm_debug_trace(buf, NW))
L.loss(batch, weight, 0.0); L.optimizer_step()
_lib.check(lib.hsad_lstm_debug_trace(buf, NW))
f = (np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(2, KREC, KNB, KT, KK) * 0.01)[0]
_lib.check(lib.hsad_lstm_debug_enable(0))
base = f[0, :, 1, 0].min()
print("layer 0 / layer 1 signal of step t (us after the first step begins), member 0; layer 1's lag behind layer 0")
for t in (1, 5, 10, 20, 40, 60, 78):
    print("  t=%2d  L0 %7.2f  L1 %7.2f  lag %.2f   target L0 %7.2f L1 %7.2f" % (t, f[0, 0, t, 6] - base, f[1, 0, t, 6] - base, f[1, 0, t, 6] - f[0, 0, t, 6], f[2, 0, t, 6] - base, f[3, 0, t, 6] - base))
for rec in (1, 0):
    print("online layer %d: per member (rows), steps 40..43: stamps 0 step begins, 1 counter seen / h DMA issued, 2 DMA issued + X half, 3 h landed, 4 h MFMA done, 5 cell staged, 6 signalled, 7 background done; us after the previous step's LAST signal" % rec)
    for t in (40, 41, 42):
        last = f[rec, :, t - 1, 6].max()
        print("  step %d (previous signals spread %.2f)" % (t, last - f[rec, :, t - 1, 6].min()))
        for nb in range(16):
            r = f[rec, nb, t]
            print("    nb%2d prev signal %6.2f | " % (nb, f[rec, nb, t - 1, 6] - last) + " ".join("%d:%6.2f" % (k, r[k] - last) for k in range(8)) + "   prev bg done %6.2f" % (f[rec, nb, t - 1, 7] - last))
