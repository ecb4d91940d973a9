"""Developer tool: rollout throughput vs number of stream partitions / stagger (run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv
G = 65536
import itertools
cases = [(1, 0)] + [(3, x) for x in (22, 24, 26, 28, 30, 32, 34, 36, 40)]
for K, stag in cases:
    env = BatchedHanabiEnv(G, seed=1, eps_list=[0.1], device="cuda:0", track_deck_history=False)
    env.set_partitions(K)
    env.set_rollout_stagger(stag)
    env.rollout_random(30, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.rollout_random(300, 5); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env.check_errors()
    print("K=%2d stagger=%3d us: %.1f us/iter  %.1f M steps/s" % (K, stag, dt / 300 * 1e6, G * 300 / dt / 1e6))
    env.close()
