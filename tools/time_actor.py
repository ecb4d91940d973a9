"""Developer tool: time of one DeviceActor.step() (reset -> act -> env step -> n-step pop -> compute_priority -> sequence
push -> replay flush) at a given number of games, 2-player SAD IQL, H=512."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.selfplay import Trainer, parse_args
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
args = parse_args(["--num_game", str(G), "--replay_buffer_size", "65536", "--method", os.environ.get("METHOD", "iql")])
tr = Trainer(args, "cuda:0")
for _ in range(120): tr.actor.step()      # past the first episode ends: steady-state rate of finished sequences
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 160
for _ in range(n): tr.actor.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("G=%d: actor step %.2f ms -> %.2f M acts/s (%.2f M game-steps/s)" % (G, dt * 1e3, G * 2 / dt / 1e6, G / dt / 1e6))
