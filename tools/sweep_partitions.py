"""Developer tool: rollout throughput vs number of stream partitions (run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv
G = 65536
for K in (1, 2, 3, 4, 6, 8, 12, 16):
    env = BatchedHanabiEnv(G, seed=1, eps_list=[0.1], device="cuda:0", track_deck_history=False)
    env.set_partitions(K)
    env.rollout_random(30, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.rollout_random(200, 5); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env.check_errors()
    print("K=%2d  %.1f us/iter  %.1f M steps/s" % (K, dt / 200 * 1e6, G * 200 / dt / 1e6))
    env.close()
