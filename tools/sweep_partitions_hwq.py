"""Developer tool: rollout throughput vs number of phase-locked partitions and lock time (python tools/sweep_partitions_hwq.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv
G = 65536
cases = [(3, 30), (3, 26), (4, 26), (4, 22), (4, 18), (4, 14), (5, 18), (5, 14), (5, 10)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for K, stag in cases:
    env = BatchedHanabiEnv(G, seed=1, eps_list=[0.1], device="cuda:0", track_deck_history=False)
    env.set_partitions(K); env.set_rollout_stagger(stag)
    env.rollout_random(30, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.rollout_random(300, 5); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env.check_errors()
    print("K=%2d lock=%3d us: %.1f us/iter  %.1f M steps/s" % (K, stag, dt / 300 * 1e6, G * 300 / dt / 1e6), flush=True)
    env.close()
