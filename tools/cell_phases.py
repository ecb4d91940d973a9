"""phase timers of the 256 x 256 fused cell kernels (developer instantiation, hsad_lstm_debug_enable): the phase-interleaved core (gemm8_kernel)
reports shader clocks per item (k loop | epilogue | store drain); the one-barrier kernel (HSAD_CELL_PP=0) wall-clock time workgroup 0
spends per launch in  wait = s_waitcnt + barrier at the top of a k step | mfma = fragment reads + MFMAs of a k step |
epilogue = cell update + state loads / stores | rest.    python tools/cell_phases.py [rows] [state_outputs 0|1]"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.r2d2 import _s
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
state = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H = 512; d = torch.device("cuda:0")
lib = _lib.load_library()
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(N, H, generator=g).to(d).to(torch.bfloat16); h16 = torch.randn(N, H, generator=g).to(d).to(torch.bfloat16)
W = (torch.randn(4 * H, 2 * H, generator=g) / 32).to(d).to(torch.bfloat16); b = torch.randn(4 * H, generator=g).to(d)
c0 = torch.randn(N, H, generator=g).to(d); c1 = torch.empty_like(c0); h1 = torch.empty_like(c0)
o16 = torch.empty(N, H, dtype=torch.bfloat16, device=d)
def cell():
    _lib.check(lib.hsad_lstm_cell_fused(N, H, H, x.data_ptr(), H, h16.data_ptr(), W.data_ptr(), b.data_ptr(), c0.data_ptr(),
                                        c1.data_ptr() if state else None, h1.data_ptr() if state else None, o16.data_ptr(), _s(d)))
for dbg in (0, 1):
    _lib.check(lib.hsad_lstm_debug_enable(dbg))
    for _ in range(100): cell()
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    _lib.check(lib.hsad_lstm_debug_timing(buf, 1))
    K = 40
    t0 = time.perf_counter()
    for _ in range(K): cell()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    _lib.check(lib.hsad_lstm_debug_timing(buf, 1))
    print("debug %d: %.1f us per launch (%.0f TF)" % (dbg, dt * 1e6, 2 * N * 2048 * 1024 / dt / 1e12))
    if dbg and os.environ.get("HSAD_CELL_PP", "1") != "0":
        # gemm8_kernel's developer timers: shader clocks of workgroup 0 in its k loops | epilogues up to the last store issued | waiting for
        # those stores (a vmcnt(0) the product does not have: it is what the next item's first counted waits would see) | items
        for g, o in (("wave 0 (early row)", 0), ("wave 4 (late row)", 8)):
            cyc = [buf[o + i] / K for i in range(4)]
            n = max(cyc[3], 1)
            print("  %s per launch: %d items; per item: k loop %.0f cycles | epilogue %.0f | store drain %.0f" % (g, n, cyc[0] / n, cyc[1] / n, cyc[2] / n))
    elif dbg:
        us = [buf[i] / K / 100.0 for i in range(4)]
        print("  workgroup 0 per launch: wait %.1f us | mfma %.1f us | epilogue %.1f us | rest %.1f us | sum %.1f us" % (us[2], us[3], us[1], us[0], sum(us)))
_lib.check(lib.hsad_lstm_debug_enable(0))
