"""Developer tool: per-phase s_memtime breakdown of the reset / step kernels (run on the GPU box)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv, _lib

G = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = BatchedHanabiEnv(G, seed=1, eps_list=[0.1], device="cuda:0", track_deck_history=False)
env.rollout_random(30, 5)
nb = (G + 63) // 64
buf = torch.zeros(nb, 8, dtype=torch.int64, device="cuda:0")
_lib.check(env.lib.hsad_env_debug_timing(env.h, buf.data_ptr()))
for name in ("reset", "step", "fused"):
    if name == "reset":
        buf.zero_(); env.reset(); torch.cuda.synchronize()
    elif name == "fused":
        buf.zero_(); env.rollout_random(1, 7); torch.cuda.synchronize()
    else:
        a, g = env.policy_random(5); buf.zero_(); env.step(a, g); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.float64)
    b = b[b[:, 5] > 0] / 100.0  # wall_clock64 ticks are 10 ns -> microseconds
    t0 = b[:, 0].min()
    d = np.diff(b[:, :6], axis=1)
    print(name, "waves", len(b), "kernel span %.1f us" % (b[:, 5].max() - t0))
    print("  phase mean us: load %.1f logic %.1f build %.1f writeback %.1f stream %.1f" % tuple(d.mean(0)))
    print("  phase max  us: load %.1f logic %.1f build %.1f writeback %.1f stream %.1f" % tuple(d.max(0)))
    print("  start spread %.1f us  end mean %.1f us" % ((b[:, 0] - t0).max(), (b[:, 5] - t0).mean()))
    if name in ("reset", "fused"):
        ok = (b[:, 6] > 0) & (b[:, 7] > 0)
        print("  reset logic split us: window prefetch %.1f, initial deal %.1f, eps/perm/LA/publish %.1f" % (
            (b[ok, 6] - b[ok, 1]).mean(), (b[ok, 7] - b[ok, 6]).mean(), (b[ok, 2] - b[ok, 7]).mean()))
