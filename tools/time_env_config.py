"""Developer tool: rollout throughput of an arbitrary env configuration (players, hand, games, sad, shuffle_color)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv
P, H, G, sad, sc = (int(x) for x in sys.argv[1:6])
K, lock = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (1, 0)
CHUNK = int(sys.argv[8]) if len(sys.argv) > 8 else 0     # > 0: persistent rollout, iterations per launch
EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]
env = BatchedHanabiEnv(G, players=P, hand_size=H, seed=1, eps_list=EPS, max_len=80, sad=bool(sad), shuffle_color=bool(sc),
                       device="cuda:0", track_deck_history=False)
env.set_partitions(K); env.set_rollout_stagger(lock); env.set_rollout_chunk(CHUNK)
env.rollout_random(50, 5); torch.cuda.synchronize()
t0 = time.perf_counter(); env.rollout_random(200, 5); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
env.check_errors()
byt = P * (env.F + env.A + 3 * H + 1) * 4 + 5 + P * 8 * (1 + sad) + 256
print("K=%d lock=%d chunk=%d " % (K, lock, CHUNK), end=""); print("P=%d hand=%d G=%d sad=%d shuffle_color=%d: %.1f us/iter, %.1f M env-steps/s, %.2f TB/s algorithmic (%.0f B/step)" %
      (P, H, G, sad, sc, dt * 1e6, G / dt / 1e6, byt * G / dt / 1e12, byt))
