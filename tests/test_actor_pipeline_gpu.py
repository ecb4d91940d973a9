"""GPU: the device actor pipeline end to end (env -> act -> n-step/sequence writer -> priority -> replay ->
learner update), small sizes.  Component parity is covered elsewhere; here the data flow and its invariants."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def test_actor_fills_replay_with_well_formed_sequences_and_learner_trains():
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    args = parse_args(["--num_game", "96", "--rnn_hid_dim", "64", "--batchsize", "16", "--replay_buffer_size", "2048",
                       "--burn_in_frames", "64", "--max_len", "40", "--act_base_eps", "0.4", "--sad", "1",
                       "--pred_weight", "0.25", "--num_update_between_sync", "5", "--actor_sync_freq", "2"])
    tr = Trainer(args, "cuda:0")
    for _ in range(60):
        tr.actor.step()
    tr.env.check_errors()
    tr.replay.check_errors()
    n = tr.replay.size()
    assert n >= 64 and tr.replay.num_add() == n
    assert tr.actor.num_act == 60 * 96 * 2
    T, F, A = args.max_len, tr.env.F, tr.env.A
    for idx in range(0, n, max(1, n // 25)):
        f, reward, terminal, bootstrap, seq_len = tr.replay.get(idx)
        L = int(seq_len.item())
        term = terminal.cpu().numpy()
        assert 1 <= L <= T and term[L - 1] and not term[:L - 1].any() and term[L:].all()           # padding: terminal = 1
        assert f["priv_s"][L:].abs().sum() == 0 and reward[L:].abs().sum() == 0 and bootstrap[L:].sum() == 0
        legal = f["legal_move"][:L]
        a = f["a"][:L, 0]
        assert (legal.gather(1, a.unsqueeze(1)) == 1).all()                                        # actions were legal
        assert (legal.gather(1, f["greedy_a"][:L]) == 1).all()
        assert (f["priv_s"][:L, :125] == 0).all()                                                   # own-hand block hidden
        assert bootstrap[:L].cpu().numpy()[max(0, L - args.multi_step):].sum() == 0                 # no bootstrap past the end
        assert ((f["own_hand"][:L].view(L, 5, 3).sum(2) <= 1).all())
    w0 = tr.learner.flat.clone()
    losses = []
    for _ in range(6):
        tr.actor.step()
        loss, g_norm = tr.learner_update()
        losses.append(float(loss))
        assert np.isfinite(losses[-1]) and np.isfinite(float(g_norm))
    tr.replay.check_errors()
    assert not torch.equal(w0, tr.learner.flat)
    assert torch.equal(tr.act_online.w["fc_a.weight"], tr.learner.online.w["fc_a.weight"]) or tr.num_update % 2 != 1
