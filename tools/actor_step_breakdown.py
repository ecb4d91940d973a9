"""rocprofv3 --kernel-trace --stats -- python tools/actor_step_breakdown.py : kernels of N steady-state actor steps only
(per-step time of each kernel = TotalDurationNs / N)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.selfplay import Trainer, parse_args
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
args = parse_args(["--num_game", "16384", "--replay_buffer_size", "65536", "--sad", "1"])
tr = Trainer(args, "cuda:0")
for _ in range(N):
    tr.actor.step()
torch.cuda.synchronize()
print("steps", N)
