"""learner update through the composite entry points: fused forward recurrences vs the chunk-pipelined schedule"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
dev = "cuda:0"
F, H, A, T, B = 838, 512, 21, 80, 128
W = init_weights(F, H, A, 5, 0)
seq_len = torch.randint(40, 81, (B,)).float().to(dev)
mask = (torch.arange(T, device=dev).unsqueeze(1) < seq_len.unsqueeze(0)).float()
legal = (torch.rand(T, B, A, device=dev) < 0.4).float(); legal[..., 0] = 1
a = torch.multinomial(legal.view(-1, A), 1).view(T, B)
batch = {"priv_s": (torch.rand(T, B, F, device=dev) < 0.15).float() * mask.unsqueeze(2), "legal_move": legal * mask.unsqueeze(2),
         "a": a * mask.long(), "reward": (torch.rand(T, B, device=dev) < 0.05).float() * mask, "bootstrap": mask.clone(),
         "seq_len": seq_len, "own_hand": torch.zeros(T, B, 15, device=dev)}
weight = torch.ones(B, device=dev)
lr = CompositeLearner(W, W, 3, 0.999, device=dev)
for fused in [int(x, 0) for x in sys.argv[1:]] or (1 | (1 << 8), 9 | (1 << 8), 9 | (2 << 8), 9 | (4 << 8), 1 | (2 << 8), 1 | (4 << 8), 3 | (1 << 8), 0, 5 | (1 << 8), 1 | (1 << 8)):
    lr.set_fused(fused)
    def upd():
        lr.loss(batch, weight, 0.0); lr.optimizer_step()
    for _ in range(5): upd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): upd()
    t_issue = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lr.loss(batch, weight, 0.0, compute_grad=False)
    e1.record(); torch.cuda.synchronize()
    print("fused=0x%x  %.3f ms/update  (host issue %.3f ms)  %.1f k sequences/s   forward only %.3f ms" % (fused, dt * 1e3, t_issue * 1e3, B / dt / 1e3, e0.elapsed_time(e1) / 50), flush=True)
lr.check_sync()
