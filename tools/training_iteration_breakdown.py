"""rocprofv3 --kernel-trace --stats -- python tools/training_iteration_breakdown.py [N] [games] : the kernels of N whole learner iterations on
rollout data (selfplay.Trainer.learner_update: prioritized sample -> loss fwd + BPTT -> clip + Adam -> priority write-back), what the
`learner_iteration_ms_on_rollout_data` field of bench.py's actor leg times.  Also prints the host's issue time per iteration (the loop
without a device wait) next to the device-paced time.  tools/update_timeline.py on the kernel trace gives one iteration as a timeline."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.selfplay import Trainer, parse_args
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
games = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
act_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # rollout steps issued in front of every update (selfplay's --act_steps_per_update), on the rollout stream
tr = Trainer(parse_args(["--num_game", str(games), "--replay_buffer_size", "65536", "--sad", "1"]), "cuda:0")
tr.act_step(120)
tr.join_rollout()
assert tr.replay.size() >= tr.args.batchsize
def iteration():
    if act_steps:
        tr.act_step(act_steps)
    tr.learner_update()
for _ in range(5):
    iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    iteration()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("iteration: %.3f ms device-paced, host issue %.3f ms (B = %d: %.1f k sequences/s)" % (t_all / N * 1e3, t_issue / N * 1e3, tr.args.batchsize,
                                                                                      tr.args.batchsize / (t_all / N) / 1e3))
# the host alone: with the queue drained before every call nothing the host does waits for the device
hs = []
for _ in range(10):
    torch.cuda.synchronize()
    t = time.perf_counter()
    iteration()
    hs.append(time.perf_counter() - t)
hs.sort()
print("host issue time of one iteration on an empty queue: median %.3f ms" % (hs[len(hs) // 2] * 1e3))
tr.join_rollout()
torch.cuda.synchronize()
tr.learner.check_sync()
tr.replay.check_errors()
