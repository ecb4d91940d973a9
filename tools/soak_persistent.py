"""Developer tool: 2,000 iterations of all 65,536 games, persistent rollout (137 iterations per launch) against the
launch-per-iteration path with 3 phase-locked partitions: state dump, observations, rewards, actions must be identical."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import BatchedHanabiEnv
EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]
def mk(chunk, parts=1, lock=0, **kw):
    e = BatchedHanabiEnv(65536, seed=99, eps_list=EPS, device="cuda:0", track_deck_history=False, **kw)
    e.set_partitions(parts); e.set_rollout_stagger(lock); e.set_rollout_chunk(chunk); return e
for kw in ({}, {"sad": True, "shuffle_color": True}):
    a, b = mk(0, 3, 30, **kw), mk(137, **kw)
    for blk in range(4):
        a.rollout_random(500, 3); b.rollout_random(500, 3); torch.cuda.synchronize()
        a.check_errors(); b.check_errors()
        assert torch.equal(a.export_state(), b.export_state()) and torch.equal(a.priv_s, b.priv_s) and torch.equal(a.reward, b.reward)
        assert torch.equal(a.terminal, b.terminal) and torch.equal(a.legal_move, b.legal_move) and torch.equal(a.a, b.a)
    print("soak OK", kw, "2000 iterations, persistent (137 per launch) == 3 phase-locked partitions")
    a.close(); b.close()
