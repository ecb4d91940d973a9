"""Developer tool: throughput of the R2D2 network kernels at BASELINE configs[2] shapes (run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.r2d2 import gemm_nt, lstm_layer_forward, gate_block_perm
DEV = "cuda:0"

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

for (M, N, K) in [(10240, 2048, 512), (10240, 512, 896), (10240, 512, 2048), (2048, 512, 10240), (131072, 2048, 512), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16); B = torch.randn(N, K, device=DEV).to(torch.bfloat16)
    C = torch.empty(M, N, device=DEV)
    dt = timeit(lambda: gemm_nt(A, B, M, N, K, out32=C))
    dt_t = timeit(lambda: torch.matmul(A, B.t()))
    print("gemm %6dx%5dx%5d: %7.1f us  %6.1f TF   (torch/hipBLASLt bf16: %7.1f us %6.1f TF)" % (M, N, K, dt * 1e6, 2 * M * N * K / dt / 1e12, dt_t * 1e6, 2 * M * N * K / dt_t / 1e12))
T, Bn, H = 80, 128, 512
perm = gate_block_perm(H, DEV)
Whh = (torch.randn(4 * H, H, device=DEV) / H ** 0.5)[perm].to(torch.bfloat16).contiguous()
gates = torch.randn(T, Bn, 4 * H, device=DEV)
dt = timeit(lambda: lstm_layer_forward(gates.clone(), Whh, None, None), n=10)
dtc = timeit(lambda: gates.clone(), n=10)
print("lstm layer fwd T=80 B=128 H=512: %.1f us total, %.2f us/step (clone %.1f us)" % (dt * 1e6, (dt - dtc) * 1e6 / T, dtc * 1e6))
lstm = torch.nn.LSTM(H, H, 1).to(DEV)
x = torch.randn(T, Bn, H, device=DEV)
with torch.no_grad():
    dt = timeit(lambda: lstm(x), n=10)
print("torch nn.LSTM (MIOpen) 1 layer fwd fp32: %.1f us" % (dt * 1e6))
Bn = 131072
gates = torch.randn(1, Bn, 4 * H, device=DEV)
h0 = torch.randn(Bn, H, device=DEV); c0 = torch.randn(Bn, H, device=DEV)
dt = timeit(lambda: lstm_layer_forward(gates, Whh, h0, c0), n=10)
print("lstm step Bn=131072: %.1f us  (%.1f TF)" % (dt * 1e6, 2 * Bn * 4 * H * H / dt / 1e12))
