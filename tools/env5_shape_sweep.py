"""configs[4] per GPU (16,384 five-player games): the persistent rollout kernel by workgroup shape (games per workgroup x threads)"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from hanabi_sad_amd import BatchedHanabiEnv
import bench
eps = bench.EPS
for gpw in (32, 64):
    for thr in (128, 256):
        try:
            env = BatchedHanabiEnv(16384, players=5, hand_size=4, seed=1, eps_list=eps, max_len=80, sad=True, shuffle_color=True, device="cuda:0",
                                   track_deck_history=False, games_per_workgroup=gpw, threads_per_workgroup=thr)
        except Exception as e:
            print(gpw, thr, "ERR", str(e)[:80]); continue
        env.set_rollout_chunk(50); env.set_rollout_stagger(0)
        env.rollout_random(100, 7)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.rollout_random(200, 7); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 200)
        env.check_errors()
        b = bench.algorithmic_bytes_per_step(env.P, env.F, env.A, env.H, True)
        ms = sorted(ts)[2]
        print("gpw %d threads %d: %.4f ms/iter  %.1f M steps/s  %.0f GB/s (%.2f of 8 TB/s)" % (env.games_per_workgroup, env.threads_per_workgroup, ms, 16384 / ms / 1e3, b * 16384 / (ms * 1e-3) / 1e9, b * 16384 / (ms * 1e-3) / 8e12))
        del env
