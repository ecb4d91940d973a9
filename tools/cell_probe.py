"""Developer tool: a few launches of the fused inference LSTM cell (actors) and of the learner's projection GEMM, for
rocprofv3 --pmc / --kernel-trace runs.  python tools/cell_probe.py [rows] [reps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.r2d2 import gemm_nt, _s
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, d = 512, torch.device("cuda:0")
lib = _lib.load_library()
x = torch.randn(N, H, device=d).to(torch.bfloat16); h16 = torch.randn(N, H, device=d).to(torch.bfloat16)
W = (torch.randn(4 * H, 2 * H, device=d) / 32).to(torch.bfloat16); b = torch.randn(4 * H, device=d)
c0 = torch.randn(N, H, device=d); c1 = torch.empty_like(c0); h1 = torch.empty_like(c0)
o16 = torch.empty(N, H, dtype=torch.bfloat16, device=d)
def cell():
    _lib.check(lib.hsad_lstm_cell_fused(N, H, H, x.data_ptr(), H, h16.data_ptr(), W.data_ptr(), b.data_ptr(), c0.data_ptr(),
                                        c1.data_ptr(), h1.data_ptr(), o16.data_ptr(), _s(d)))
A = torch.randn(10240, 512, device=d).to(torch.bfloat16); B = torch.randn(2048, 512, device=d).to(torch.bfloat16)
C = torch.empty(10240, 2048, device=d)
for f, name, fl in ((cell, "cell %d x 2048 x 1024" % N, 2 * N * 2048 * 1024),
                    (lambda: gemm_nt(A, B, 10240, 2048, 512, out32=C), "gemm 10240x2048x512", 2 * 10240 * 2048 * 512)):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(R): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
    print("%s: %.1f us  %.0f TF" % (name, dt * 1e6, fl / dt / 1e12))
