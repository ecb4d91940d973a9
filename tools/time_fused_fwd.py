"""Developer tool: time the fused persistent forward (2 nets x 2 layers, T=80, B=128, H=512) and print its phase timers."""
import os, sys
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.r2d2 import check_sync, gate_block_perm, lstm_forward_fused

DEV = "cuda:0"
T, Bn, H = int(sys.argv[1]) if len(sys.argv) > 1 else 80, 128, 512
nnet, nl = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 2)
lib = _lib.load_library()
perm = gate_block_perm(H, DEV)
g = torch.Generator().manual_seed(0)
xs = [torch.randn(T, Bn, H, generator=g).to(DEV).to(torch.bfloat16) for _ in range(nnet)]
nets = [[((torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV)[perm].to(torch.bfloat16).contiguous(),
          (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(DEV)[perm].to(torch.bfloat16).contiguous(),
          torch.zeros(4 * H, device=DEV)) for _ in range(nl)] for _ in range(nnet)]
for keep, dbg in ((True, 0), (True, 0), (True, 1), (False, 0)):
    _lib.check(lib.hsad_lstm_debug_enable(dbg))
    plan = {}
    for _ in range(3):
        lstm_forward_fused(xs, nets, keep=keep, plan=plan, unpack=False)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    _lib.check(lib.hsad_lstm_debug_timing(buf, 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        lstm_forward_fused(xs, nets, keep=keep, plan=plan, unpack=False)
    e1.record()
    torch.cuda.synchronize()
    check_sync()
    _lib.check(lib.hsad_lstm_debug_timing(buf, 1))
    ms = e0.elapsed_time(e1) / n
    print("dbg=%d " % dbg + "keep=%d nnet=%d nl=%d T=%d: %.1f us per launch (incl. allocation/memset), %.2f us per step" % (keep, nnet, nl, T, ms * 1e3, ms * 1e3 / (T + nl - 1)))
    if not dbg:
        continue
    names = ["wait h", "issue+x-mfma", "h landed", "h-mfma", "late x", "cell", "publish"]
    for base, tag in ((0, "layer 0"), (8, "stacked")):
        v = [buf[base + i] / 100.0 / (n * nnet * T) for i in range(7)]
        print("  %s us/step: " % tag + "  ".join("%s %.2f" % (nm, x) for nm, x in zip(names, v)) + "  | sum %.2f" % sum(v))
