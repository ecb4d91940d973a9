"""Developer tool (run under rocprofv3 --pmc ...): the LSTM input-projection GEMM 10240x2048x512, 20 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.r2d2 import gemm_nt
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (10240, 2048, 512)))
A = torch.randn(M, K, device="cuda:0").to(torch.bfloat16); B = torch.randn(N, K, device="cuda:0").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda:0")
for _ in range(20):
    gemm_nt(A, B, M, N, K, out32=C)
torch.cuda.synchronize()
