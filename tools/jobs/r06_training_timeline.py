"""60 iterations of one-GPU self-play training (bench.py's training_bench loop) for a rocprofv3 --kernel-trace; tools/update_timeline.py
prints the last adam-to-adam span = one steady-state iteration (rollout step + update + draw)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hanabi_sad_amd.selfplay import Trainer, parse_args
args = parse_args(["--num_game", "6400", "--replay_buffer_size", "131072", "--sad", "1"])
tr = Trainer(args, "cuda:0")
tr.act_step(130); tr.join_rollout()
for _ in range(60):
    tr.act_step(1); tr.learner_update()
tr.join_rollout(); torch.cuda.synchronize()
tr.learner.check_sync()
