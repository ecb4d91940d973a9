"""env step / reset / rollout time at a given size for both workgroup sizes"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv
EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]
for (P, H, G, sad) in ((2, 5, 16384, True), (2, 5, 6400, True), (2, 5, 65536, False), (5, 4, 16384, True), (3, 5, 16384, True)):
    for gpw in (32, 64):
        for th in (128, 256):
            env = BatchedHanabiEnv(G, players=P, hand_size=H, seed=1, eps_list=EPS, sad=sad, shuffle_color=P > 2, device="cuda:0",
                                   track_deck_history=False, games_per_workgroup=gpw, threads_per_workgroup=th)
            for _ in range(30):
                env.reset(); a, g = env.policy_random(5); env.step(a, g)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ts, tr = 0.0, 0.0
            for _ in range(50):
                ev[0].record(); env.reset(); ev[1].record(); a, g = env.policy_random(5); ev2 = torch.cuda.Event(enable_timing=True); ev2.record(); env.step(a, g); ev[2].record()
                torch.cuda.synchronize()
                tr += ev[0].elapsed_time(ev[1]); ts += ev2.elapsed_time(ev[2])
            env.set_rollout_chunk(50)
            env.rollout_random(50, 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); env.rollout_random(100, 3); torch.cuda.synchronize()
            ro = (time.perf_counter() - t0) / 100 * 1e3
            print("P%d G%6d gpw%d threads%d: reset %.1f us  step %.1f us  rollout %.1f us/iter" % (P, G, gpw, th, tr / 50 * 1e3, ts / 50 * 1e3, ro * 1e3), flush=True)
            del env
