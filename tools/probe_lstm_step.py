import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.r2d2 import lstm_layer_forward, gate_block_perm
DEV = "cuda:0"
def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (T, Bn, H) in [(80, 128, 512), (80, 32, 512), (80, 128, 64), (80, 128, 256), (80, 512, 512), (400, 128, 512)]:
    perm = gate_block_perm(H, DEV)
    Whh = (torch.randn(4 * H, H, device=DEV) / H ** 0.5)[perm].to(torch.bfloat16).contiguous()
    gates = torch.randn(T, Bn, 4 * H, device=DEV)
    dt = timeit(lambda: lstm_layer_forward(gates, Whh, None, None))
    print("T=%d Bn=%d H=%d: %.2f us/step" % (T, Bn, H, dt * 1e6 / T))
