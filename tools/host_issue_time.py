"""Developer tool: host-side issue time vs device time of one learner update (is the update launch-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NO_TORCH"] = "1"
src = open(os.path.join(os.path.dirname(__file__), "time_learner.py")).read().split("for _ in range(5): upd()")[0]
exec(src)
for _ in range(3): upd()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): upd()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("issue %.2f ms/update, total %.2f ms/update" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
