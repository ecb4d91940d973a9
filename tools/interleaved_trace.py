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
t0 = time.perf_counter()
N = 60
for _ in range(N):
    tr.actor.step(); tr.learner_update()
torch.cuda.synchronize()
print("interleaved ms", (time.perf_counter() - t0) / N * 1e3)
