"""Is the env kernel's 3.3 ms / 4.0 ms per 50-iteration launch a property of the PROCESS (memory placement of the observation buffers)?
Within one process: create the env, time it, destroy it, give the memory back (empty_cache), allocate a spacer of varying size, repeat."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd import BatchedHanabiEnv
dev = "cuda:0"
eps = [0.1 * 0.5 ** i for i in range(8)]
def run(tag, spacer_mb=0):
    sp = torch.empty(spacer_mb * 1024 * 1024, dtype=torch.uint8, device=dev) if spacer_mb else None
    env = BatchedHanabiEnv(65536, players=2, hand_size=5, seed=1, eps_list=eps, max_len=80, sad=False, device=dev, track_deck_history=False)
    env.set_rollout_chunk(50)
    env.rollout_random(50, 12345)
    torch.cuda.synchronize()
    ms = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout_random(200, 12345); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 4)
    ms.sort()
    print("%-28s priv_s @ 0x%x (mod 2 MB: %7d KB)  ms per 50-iteration launch: median %.3f  min %.3f max %.3f" % (
        tag, env.priv_s.data_ptr(), (env.priv_s.data_ptr() % (2 << 20)) // 1024, ms[len(ms) // 2], ms[0], ms[-1]), flush=True)
    env.close(); del env, sp
    torch.cuda.empty_cache()
for i, mb in enumerate([0, 0, 1, 3, 64, 0, 700, 0, 5, 0]):
    run("pass %d spacer %4d MB" % (i, mb), mb)
