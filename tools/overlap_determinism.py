"""selfplay.Trainer with the rollout on its own stream (--overlap_rollout 1) against the strictly alternating single-stream loop: every
cross-stream dependency is enforced in host issue order, so the two must see the same data and produce the same losses (up to the
float-atomic bias sums of the BPTT launch)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.selfplay import Trainer, parse_args
G = sys.argv[1] if len(sys.argv) > 1 else "1024"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
BURN, BUF, BATCH = (sys.argv[3:6] + ["1000", "8192", "64"][len(sys.argv[3:6]):])
out = {}
for o in (0, 1, 0, 1):
    tr = Trainer(parse_args(["--num_game", G, "--sad", "1", "--seed", "5", "--burn_in_frames", BURN, "--replay_buffer_size", BUF,
                             "--batchsize", BATCH, "--overlap_rollout", str(o), "--actor_sync_freq", "5"]), "cuda:0")
    tr.act_step(int(BURN))          # a FIXED number of burn-in steps: the polling loop of the driver depends on timing
    losses = []
    for u in range(N):
        tr.act_step(1)
        loss, g = tr.learner_update()
        losses.append(loss)
    tr.join_rollout()
    torch.cuda.synchronize()
    tr.env.check_errors(); tr.replay.check_errors()
    ls = torch.stack([l.detach().float() for l in losses]).cpu()
    out.setdefault(o, []).append((ls, tr.replay.num_add(), tr.replay.size(), tr.replay.priority_sum()[0], tr.actor.num_act))
    print("overlap=%d  num_add %d size %d priority_sum %.6f  loss[0,1,N/2,N-1] = %s" % (o, tr.replay.num_add(), tr.replay.size(), tr.replay.priority_sum()[0],
          [round(float(ls[i]), 6) for i in (0, 1, N // 2, N - 1)]), flush=True)
    del tr
    torch.cuda.empty_cache()
for K in (10, 20, 40, 80, N):
    d_same0 = (out[0][0][0][:K] - out[0][1][0][:K]).abs().max()
    d_same1 = (out[1][0][0][:K] - out[1][1][0][:K]).abs().max()
    d_cross = (out[0][0][0][:K] - out[1][0][0][:K]).abs().max()
    print("first %3d updates, max |loss difference|: alternating run vs run %.3e, overlapped run vs run %.3e, alternating vs overlapped %.3e" % (K, d_same0, d_same1, d_cross))
