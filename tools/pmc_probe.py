"""Run under `rocprofv3 --pmc WRITE_SIZE|FETCH_SIZE --kernel-trace`: a few rollout iterations at the bench
configuration plus a calibration kernel of known traffic (fill of exactly the step kernel's algorithmic
output bytes), so the counter can be corrected as MI355X_MICROARCH.md §HBM prescribes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv
G = 65536
EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]
env = BatchedHanabiEnv(G, seed=1, eps_list=EPS, device="cuda:0", track_deck_history=False)
CHUNK = 50
env.set_rollout_chunk(CHUNK)   # as bench.py launches it: one env_rollout_kernel dispatch = CHUNK iterations of all G games
env.rollout_random(4 * CHUNK, 5)
env.set_rollout_chunk(0)
env.set_partitions(3); env.set_rollout_stagger(30)   # the launch-per-iteration path: every env_kernel<3,...> dispatch covers G/3 games
env.rollout_random(30, 5)
env.set_partitions(1)
torch.cuda.synchronize()
for _ in range(10):
    env.reset(); a, g = env.policy_random(5); env.step(a, g)
torch.cuda.synchronize()
# calibration: write exactly G*2*783*4 bytes (= priv_s) with a plain streaming fill, and read+write copy
x = torch.empty(G * 2 * 783, dtype=torch.float32, device="cuda:0")
y = torch.empty_like(x)
for _ in range(5):
    x.fill_(1.0)
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
env.check_errors()
print("probe done; priv_s bytes", x.numel() * 4)
