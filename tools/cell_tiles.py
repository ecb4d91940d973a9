"""the fused cell kernel's tilings side by side (HSAD_CELL_TILE is read once per process: one subprocess per variant): time and a
checksum of the outputs.  python tools/cell_tiles.py [rows]"""
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
def cell_tg():      # the target pass of an acting step: only the bf16 layer output leaves the kernel
    _lib.check(lib.hsad_lstm_cell_fused(N, H, H, x.data_ptr(), H, h16.data_ptr(), W.data_ptr(), b.data_ptr(), c0.data_ptr(),
                                        None, None, o16.data_ptr(), _s(d)))
for _ in range(5): cell_tg()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): cell_tg()
torch.cuda.synchronize(); dt_tg = (time.perf_counter() - t0) / 40
print("  (bf16 output only: %%.1f us  %%.0f TF)" %% (dt_tg * 1e6, 2 * N * 2048 * 1024 / dt_tg / 1e12))
for _ in range(5): cell()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): cell()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
print("tile %%s: %%.1f us  %%.0f TF   checksum c %%.6f h %%.6f h16 %%.6f" %% (os.environ.get("HSAD_CELL_TILE", "auto"), dt * 1e6, 2 * N * 2048 * 1024 / dt / 1e12,
      float(c1.double().sum()), float(h1.double().sum()), float(o16.double().sum())))
''' % ROOT
rows = sys.argv[1] if len(sys.argv) > 1 else "32768"
for tile in ("256", "128"):
    env = dict(os.environ, HSAD_CELL_TILE=tile)
    out = subprocess.run([sys.executable, "-c", CHILD, rows], env=env, capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-600:])
