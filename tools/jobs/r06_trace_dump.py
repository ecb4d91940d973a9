"""dump raw per-step stamps (us, relative) of the BPTT launch's records for a few steps: developer aid"""
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
tr = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(2, KREC, KNB, KT, KK) * 0.01
_lib.check(lib.hsad_lstm_debug_enable(0))
b = tr[1]
base = b[0, 0, T - 1, 0]
names = {0: "top", 1: "proj", 2: "lower", 3: "sink"}
for t in range(T - 1, T - 4, -1):
    for j in (0, 1, 2, 3):
        row = b[j, 0, t]
        print("t=%2d %-5s nb0 " % (t, names[j]) + " ".join("%d:%7.2f" % (k, row[k] - base) if row[k] else "%d:   -   " % k for k in (0, 8, 2, 9, 10, 3, 4, 7, 5, 6, 11)))
    sig = b[2, :8, t, 5] - base
    print("      lower signals of all members:", " ".join("%.2f" % x for x in sig))
print("prologue of the BPTT launch (us relative to the earliest workgroup entry): entry, handshake done, first loop top, first signal; last signal of the launch")
e0 = min(b[j, n, T - 1, 1] for j in (0, 1, 2, 3) for n in range(8) if b[j, n, T - 1, 1] > 0)
for j in (0, 1, 2, 3):
    for n in (0, 3, 7):
        r = b[j, n, T - 1]
        print("  %-5s nb%d entry %7.2f handshake %7.2f loop top %7.2f first signal %7.2f | step 0 signalled %8.2f" % (names[j], n, r[1] - e0, r[11] - e0, r[0] - e0, r[5] - e0, b[j, n, 0, 5] - e0))
for t in ():
    for j in (0, 2):
        row = b[j, 0, t]
        print("t=%2d %-5s nb0 " % (t, names[j]) + " ".join("%d:%7.2f" % (k, row[k] - base) if row[k] else "%d:   -   " % k for k in (0, 8, 2, 9, 10, 3, 4, 7, 5, 6, 11)))
    print("      lower signals:", " ".join("%.2f" % (x - base) for x in b[2, :8, t, 5]), " top signals:", " ".join("%.2f" % (x - base) for x in b[0, :8, t, 5]))
