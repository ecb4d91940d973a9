"""Measure (GPU): relative Frobenius error of every parameter's bf16 gradient at the shapes of tests/test_r2d2_arch_gpu.py's
test_fused_forward_schedules_for_other_depths, against the plain fp32 reference and against the fp32 network under the bf16
activation pattern (tests/r2d2_torch_ref.bf16_relu_masks) -- what ARCH_GRAD_REL in that test is 2 x of."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
from tests import r2d2_torch_ref as ref
from tests.test_r2d2_kernels_gpu import _rand_batch
from tests.test_r2d2_precision_gpu import relerr
DEV = "cuda:0"
for H, T, B, nl, nfc in [(256, 24, 64, 1, 1), (256, 16, 32, 3, 2), (512, 20, 64, 3, 1), (512, 20, 128, 1, 2), (512, 80, 128, 2, 1)]:
    F, A = 783, 21
    W, Wt = init_weights(F, H, A, 5, 21, nl, nfc), init_weights(F, H, A, 5, 22, nl, nfc)
    batch, weight = _rand_batch(T, B, F, A)
    L = CompositeLearner(W, Wt, 3, 0.999, device=DEV)
    loss, prio = L.loss(batch, weight, 0.25)
    torch.cuda.synchronize()
    grad = {k: v.clone() for k, v in L.grad.items()}
    Wd = {k: v.to(DEV).requires_grad_(True) for k, v in W.items()}
    Wtd = {k: v.to(DEV) for k, v in Wt.items()}
    rl, _ = ref.loss(Wd, Wtd, batch, 3, 0.999, 0.25)
    (rl * weight).mean().backward()
    plain = {k: v.grad.clone() for k, v in Wd.items() if v.grad is not None}
    masks, flips = ref.bf16_relu_masks(Wd, batch["priv_s"])
    for v in Wd.values():
        v.grad = None
    ml, _ = ref.loss(Wd, Wtd, batch, 3, 0.999, 0.25, online_masks=masks)
    (ml * weight).mean().backward()
    masked = {k: v.grad.clone() for k, v in Wd.items() if v.grad is not None}
    print("H=%d T=%d B=%d L=%d fc=%d flips %s" % (H, T, B, nl, nfc, flips))
    for k in plain:
        if float(plain[k].norm()) > 0:
            print("   %-22s plain %.4f  masked %.4f" % (k, relerr(grad[k], plain[k]), relerr(grad[k], masked[k])))
    L.close()
