"""Developer tool: phase timers of the fused forward / backward recurrence kernels inside real learner updates"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
from tests.test_r2d2_kernels_gpu import _rand_batch
F, H, A, T, B = 838, 512, 21, 80, 128
lib = _lib.load_library()
W = init_weights(F, H, A, 5, 1)
L = CompositeLearner(W, W, 3, 0.999, device="cuda:0")
if len(sys.argv) > 1:
    L.set_fused(int(sys.argv[1], 0))
batch, weight = _rand_batch(T, B, F, A)
for _ in range(3):
    L.loss(batch, weight, 0.0); L.optimizer_step()
torch.cuda.synchronize()
_lib.check(lib.hsad_lstm_debug_enable(1))
buf = (C.c_uint64 * 32)()
_lib.check(lib.hsad_lstm_debug_timing32(buf, 1))
N = 10
for _ in range(N):
    L.loss(batch, weight, 0.0); L.optimizer_step()
torch.cuda.synchronize()
_lib.check(lib.hsad_lstm_debug_timing32(buf, 1))
_lib.check(lib.hsad_lstm_debug_enable(0))
us = lambda i, div: buf[i] / 100.0 / (N * div * T)
print("forward  layer 0 (2 nets): " + "  ".join("%s %.2f" % (n, us(i, 2)) for i, n in enumerate(["wait h", "h dma+x2", "landed", "h mfma", "late x", "cell", "publish"])))
print("forward  stacked (2 nets): " + "  ".join("%s %.2f" % (n, us(8 + i, 2)) for i, n in enumerate(["wait h", "h dma+x2", "landed", "h mfma", "late x", "cell", "publish"])))
names = ["wait above", "x load+mfma", "wait own", "h load+mfma", "reduce+cell", "publish"]
print("backward top layer       : " + "  ".join("%s %.2f" % (n, us(16 + i, 1)) for i, n in enumerate(names)) + "  | sum %.2f" % sum(us(16 + i, 1) for i in range(6)))
print("backward lower layer     : " + "  ".join("%s %.2f" % (n, us(24 + i, 1)) for i, n in enumerate(names)) + "  | sum %.2f" % sum(us(24 + i, 1) for i in range(6)))
L.check_sync()
