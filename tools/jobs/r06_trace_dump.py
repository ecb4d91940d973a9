"""dump raw per-step stamps (us, relative) of the BPTT launch's records for a few steps: developer aid"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
from tests.test_r2d2_kernels_gpu import _rand_batch
F, H, A, T, B = 838, 512, 21, 80, int(os.environ.get("TRACE_B", "128"))
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

print("lifetime of every workgroup of the BPTT launch (us after the earliest entry): last step done / exit, rows = row block (XCD), columns = unit block 0..7")
for j in (0, 1, 2, 3):
    print(" stage %s" % names[j])
    for rb in range(B // 16):
        r = b[j, :8, T + rb]
        print("   rb%d  handshake %s | last step %s | exit %s" % (rb, " ".join("%5.1f" % (x - e0) for x in r[:, 1]), " ".join("%6.1f" % (x - e0) for x in r[:, 2]), " ".join("%6.1f" % (x - e0) for x in r[:, 3])))
ent = np.array([b[j, n, T + rb, 0] for j in range(4) for n in range(8) for rb in range(B // 16)])
ex = np.array([b[j, n, T + rb, 3] for j in range(4) for n in range(8) for rb in range(B // 16)])
print("entries span %.1f us, exits %.1f .. %.1f us after the earliest entry" % (ent.max() - ent.min(), ex.min() - ent.min(), ex.max() - ent.min()))
print("per row block: physical XCD, shader clock over the launch (MHz), top / lower last step (us)")
for rb in range(B // 16):
    r = b[0, 0, T + rb]
    mhz = (r[6] - r[5]) * 100.0 / max(r[3] - r[1], 1e-9)      # counters were scaled by 0.01 above; r[3] - r[1] is in us
    print("   rb%d  xcc %d  %7.0f MHz   top %6.1f  lower %6.1f  sink %6.1f" % (rb, int(round(r[4] * 100)), mhz, r[2] - e0, b[2, 0, T + rb, 2] - e0, b[3, 0, T + rb, 2] - e0))
for rep in range(4):
    L.loss(batch, weight, 0.0); L.optimizer_step()
    _lib.check(lib.hsad_lstm_debug_enable(2))
    L.loss(batch, weight, 0.0); L.optimizer_step()
    _lib.check(lib.hsad_lstm_debug_trace(buf, NW))
    _lib.check(lib.hsad_lstm_debug_enable(0))
    b = (np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(2, KREC, KNB, KT, KK) * 0.01)[1]
    e0 = min(b[j, n, T + rb, 0] for j in range(4) for n in range(8) for rb in range(B // 16))
    print("traced launch %d: per row block xcc / MHz / top, lower, sink last step" % (rep + 2))
    for rb in range(B // 16):
        r = b[0, 0, T + rb]
        mhz = (r[6] - r[5]) * 100.0 / max(r[3] - r[1], 1e-9)
        print("   rb%d  xcc %d  %7.0f MHz   top %6.1f  lower %6.1f  sink %6.1f" % (rb, int(round(r[4] * 100)), mhz, r[2] - e0, b[2, 0, T + rb, 2] - e0, b[3, 0, T + rb, 2] - e0))
    print("   progress of the top / lower layer's unit block 0 per row block: us (after the earliest entry) at which steps 70, 60, .. 0 were signalled, then us per step in each interval")
    for j in (0, 2):
        for rb in range(B // 16):
            cp = b[j, 0, T + 8 + rb, :8][::-1] - e0       # steps 70 .. 0
            print("   %-5s rb%d  " % (names[j], rb) + " ".join("%6.1f" % x for x in cp) + "   | " + " ".join("%.2f" % ((cp[i + 1] - cp[i]) / 10) for i in range(7)) + "  first 9 steps %.2f" % ((cp[0] - b[j, 0, T + rb, 1] + e0) / 9))
    # the traced row block's phases, mean over steps 1..T-2 and over its 8 unit blocks: where does a slow launch lose its time?
    for j in (0, 2):
        st = b[j, :8, 1:T - 1]                      # [nb][t][k]
        nxt = b[j, :8, 0:T - 2]                     # step t - 1 (runs after step t)
        ph = {"loop top -> polls satisfied": st[:, :, 8] - st[:, :, 0], "-> seen by all": st[:, :, 2] - st[:, :, 8], "-> first quarter": st[:, :, 9] - st[:, :, 2],
              "-> whole tile": st[:, :, 10] - st[:, :, 9], "-> products": st[:, :, 3] - st[:, :, 10], "-> cell math staged": st[:, :, 4] - st[:, :, 3],
              "-> stores issued": st[:, :, 7] - st[:, :, 4], "-> drained, signalled": st[:, :, 5] - st[:, :, 7], "-> copy done": st[:, :, 6] - st[:, :, 5],
              "step": nxt[:, :, 0] - st[:, :, 0]}
        print("   rb0 %-5s " % names[j] + "  ".join("%s %.2f" % (k, np.nanmean(v[:, :]) if k != "loop top -> polls satisfied" else np.nanmean(v[0])) for k, v in ph.items()))
        d = (nxt[0, :, 0] - st[0, :, 0])
        print("   rb0 %-5s step time by step (us, t = T-2 .. 1): " % names[j] + " ".join("%.1f" % x for x in d[::-1]))
    if rep == 3:
        for j in (0, 2):
            print("   rb0 %s, first 14 steps of the launch, per step: members' (polls satisfied -> signalled) max/min | member 0: poll wait, first quarter, whole tile, products, cell, stores, drain" % names[j])
            for t in range(T - 2, T - 16, -1):
                st = b[j, :8, t]
                work = st[:, 5] - st[:, 8]
                m = st[0]
                print("     t=%2d work max %.2f min %.2f (slowest member %d) | %.2f %.2f %.2f %.2f %.2f %.2f %.2f | signals %s" % (
                    t, work.max(), work.min(), int(work.argmax()), m[8] - m[0], m[9] - m[2], m[10] - m[9], m[3] - m[10], m[4] - m[3], m[7] - m[4], m[5] - m[7],
                    " ".join("%.1f" % (x - e0) for x in st[:, 5])))
