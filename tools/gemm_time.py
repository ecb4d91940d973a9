"""Developer tool: time one bf16 GEMM shape (M N K) through hsad_gemm_nt_bf16."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.r2d2 import gemm_nt
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (10240, 2048, 512)))
A = torch.randn(M, K, device="cuda:0").to(torch.bfloat16); B = torch.randn(N, K, device="cuda:0").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda:0")
for _ in range(5): gemm_nt(A, B, M, N, K, out32=C)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): gemm_nt(A, B, M, N, K, out32=C)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("%dx%dx%d: %.1f us  %.0f TF" % (M, N, K, dt * 1e6, 2 * M * N * K / dt / 1e12))
