"""One leg per invocation, run under rocprofv3 by tools/mfma_util.sh:
  calib    a plain bf16 GEMM of known MFMA count (8192^3 fp32-out on gemm8_kernel<G8_F32>, the 256 x 256 core: 2 * 8192^3 FLOP = 33,554,432 wave-level
           v_mfma_f32_32x32x16_bf16) -- pins what SQ_VALU_MFMA_BUSY_CYCLES counts per instruction on this chip in the same tool chain
  learner  composite learner updates at configs[2] (lstm_fused_fwd_kernel / lstm_fused_bwd_kernel inside real updates)
  actor    steady-state acting steps at 16,384 games (gemm8_kernel<G8_CELL> inside real steps)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
leg = sys.argv[1]
dev = "cuda:0"
if leg == "calib":
    from hanabi_sad_amd.r2d2 import gemm_nt
    M = N = K = 8192
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev)
    for _ in range(6):
        gemm_nt(A, B, M, N, K, out32=C)
    torch.cuda.synchronize()
elif leg == "learner":
    import bench
    bench.learner_bench(dev, updates=6, warmup=2, gemm_probe=False)
elif leg == "actor":
    import bench
    bench.actor_bench(dev, games=16384, steps=20, warmup=100)
print("mfma probe leg %s done" % leg)
