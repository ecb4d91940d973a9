"""the acting step's input-layer GEMM pair (32768 x 512 x 896, one observation, two nets) on the 128 x 128 kernel and on the
phase-interleaved 256 x 256 kernel (hsad_gemm_set_pp)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.r2d2 import _s
lib = _lib.load_library()
DEV = "cuda:0"
for M, N, K in ((32768, 512, 896), (32768, 512, 512), (65536, 512, 896)):
    A = (torch.rand(M, K, device=DEV) < 0.15).to(torch.bfloat16)
    B = [(torch.randn(N, K, device=DEV) / K ** 0.5).to(torch.bfloat16) for _ in range(2)]
    bias = [torch.randn(N, device=DEV) for _ in range(2)]
    o = [torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    st, p = _s(torch.device(DEV)), (lambda t: t.data_ptr())
    for on in (0, 1, 0, 1):
        lib.hsad_gemm_set_pp(on)
        run = lambda: _lib.check(lib.hsad_gemm_nt_bf16_pair(p(A), p(A), K, p(B[0]), p(B[1]), K, M, N, K, p(bias[0]), p(bias[1]), None, None, 0,
                                                           p(o[0]), p(o[1]), N, 1, st))
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print("%d x %d x %d pair  %s  %.1f us  %.0f TFLOP/s" % (M, N, K, "phase-interleaved 256x256" if on else "128x128                  ", us, 4.0 * M * N * K / us / 1e6), flush=True)
lib.hsad_gemm_set_pp(1)
