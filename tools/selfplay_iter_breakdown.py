"""interleaved actor step + learner update on one GPU: wall per iteration vs the two halves alone"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.selfplay import Trainer, parse_args
args = parse_args(["--num_game", "16384", "--replay_buffer_size", "65536", "--sad", "1"])
tr = Trainer(args, "cuda:0")
for _ in range(120):
    tr.actor.step()
for _ in range(5):
    tr.learner_update()
torch.cuda.synchronize()
def timeit(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_issue / n * 1e3
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
print("actor step      %.3f ms (host issue %.3f)" % timeit(tr.actor.step, N))
print("learner update  %.3f ms (host issue %.3f)" % timeit(tr.learner_update, N))
def both():
    tr.actor.step(); tr.learner_update()
print("interleaved     %.3f ms (host issue %.3f)" % timeit(both, N))
