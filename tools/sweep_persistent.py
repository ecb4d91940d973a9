"""Developer tool: persistent rollout (hsad_env_set_rollout_chunk) vs phase-locked partitions: us per iteration.
python tools/sweep_persistent.py [G]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv
G = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cases = [("part", 3, 30), ("pers", 10, 0), ("pers", 20, 0), ("pers", 25, 0), ("pers", 50, 0), ("pers", 50, 3), ("pers", 100, 0), ("pers", 200, 0), ("pers", 400, 0)]
for kind, a, b in cases:
    env = BatchedHanabiEnv(G, seed=1, eps_list=[0.1], device="cuda:0", track_deck_history=False)
    if kind == "part":
        env.set_partitions(a); env.set_rollout_stagger(b)
    else:
        env.set_rollout_chunk(a); env.set_rollout_stagger(b)
    env.rollout_random(100, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.rollout_random(400, 5); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env.check_errors()
    print("%s %3d stagger/lock %2d us: %.1f us/iter  %.1f M steps/s" % (kind, a, b, dt / 400 * 1e6, G * 400 / dt / 1e6), flush=True)
    env.close()
