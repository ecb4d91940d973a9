"""does a start-up stagger between the workgroups that share a CU move the persistent env kernel out of its slow state?
usage: HSAD_ENV_STAGGER_MODE=m python tools/jobs/r06_env_stagger.py <stagger us> ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd import BatchedHanabiEnv
eps = [0.1 * 0.5 ** i for i in range(8)]
env = BatchedHanabiEnv(65536, players=2, hand_size=5, seed=1, eps_list=eps, max_len=80, sad=False, device="cuda:0", track_deck_history=False)
env.set_rollout_chunk(50)
for us in [int(x) for x in sys.argv[1:]] or [0]:
    env.set_rollout_stagger(us)
    env.rollout_random(100, 12345)
    torch.cuda.synchronize()
    ms = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout_random(200, 12345); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 4)
    ms.sort()
    print("mode %s stagger %3d us: ms per 50-iteration launch median %.3f (min %.3f max %.3f)" % (os.environ.get("HSAD_ENV_STAGGER_MODE", "0"), us, ms[3], ms[0], ms[-1]), flush=True)
env.check_errors()
