"""the 256 x 256 fused cell kernels side by side, one subprocess per variant (HSAD_CELL_PP is read once per process), 300 warm-up
launches and 5 x 100 timed ones each, plus a checksum of the outputs:  pp0 the one-barrier k loop, pp1 the phase-interleaved k loop
(default), and its ablations (results are garbage, the time is what counts): pp11 no operand DMA, pp12 no MFMA, pp14 no stagger of
the two wave rows, pp19 neither DMA nor fragment reads.    python tools/cell_hints.py [rows] [variants ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from hanabi_sad_amd import _lib
from hanabi_sad_amd.r2d2 import _s
N = int(sys.argv[1]); H = 512; d = torch.device("cuda:0")
lib = _lib.load_library()
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(N, H, generator=g).to(d).to(torch.bfloat16); h16 = torch.randn(N, H, generator=g).to(d).to(torch.bfloat16)
W = (torch.randn(4 * H, 2 * H, generator=g) / 32).to(d).to(torch.bfloat16); b = torch.randn(4 * H, generator=g).to(d)
c0 = torch.randn(N, H, generator=g).to(d); c1 = torch.empty_like(c0); h1 = torch.empty_like(c0)
o16 = torch.empty(N, H, dtype=torch.bfloat16, device=d)
def cell():
    _lib.check(lib.hsad_lstm_cell_fused(N, H, H, x.data_ptr(), H, h16.data_ptr(), W.data_ptr(), b.data_ptr(), c0.data_ptr(),
                                        c1.data_ptr(), h1.data_ptr(), o16.data_ptr(), _s(d)))
for _ in range(300): cell()
ts = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): cell()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100 * 1e6)
print("pp %%2s: %%s us   checksum %%.6f" %% (os.environ.get("HSAD_CELL_PP", "1"), " ".join("%%.1f" %% t for t in ts), float(c1.double().sum() + h1.double().sum())))
''' % ROOT
rows = sys.argv[1] if len(sys.argv) > 1 else "32768"
for hint in (sys.argv[2:] or ["pp0", "pp1", "pp11", "pp12", "pp14", "pp19", "pp0", "pp1"]):
    env = dict(os.environ, HSAD_CELL_PP=hint[2:])
    out = subprocess.run([sys.executable, "-c", CHILD, rows], env=env, capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-600:])
