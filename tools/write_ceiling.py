"""Developer tool: what a pure streaming WRITE reaches on this GPU (the env kernel is 98 % writes): torch fill / copy of
env-observation-sized and larger buffers, timed over back-to-back launches between one pair of events."""
import sys, torch
d = "cuda:0"
def timed(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (448, 1024, 4096):
    x = torch.empty(mb << 20, dtype=torch.uint8, device=d); y = torch.empty_like(x)
    t = timed(lambda: x.zero_()); print("fill  %5d MB: %7.1f us  %.2f TB/s written" % (mb, t * 1e6, (mb << 20) / t / 1e12))
    xf = x.view(torch.float32)
    t = timed(lambda: xf.fill_(1.0)); print("fill1 %5d MB: %7.1f us  %.2f TB/s written" % (mb, t * 1e6, (mb << 20) / t / 1e12))
    t = timed(lambda: y.copy_(x)); print("copy  %5d MB: %7.1f us  %.2f TB/s read+written" % (mb, t * 1e6, 2 * (mb << 20) / t / 1e12))
