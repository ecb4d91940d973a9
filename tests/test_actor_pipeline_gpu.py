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
    # the same padding through the sampling path (the flush stores only the len real steps; readers materialise the rest)
    (f, reward, terminal, bootstrap, seq_len), weight = tr.replay.sample(16)
    for b in range(16):
        L = int(seq_len[b].item())
        assert 1 <= L <= T and f["priv_s"][L:, b].abs().sum() == 0 and f["legal_move"][L:, b].abs().sum() == 0
        assert f["a"][L:, b].abs().sum() == 0 and reward[L:, b].abs().sum() == 0 and bootstrap[L:, b].sum() == 0
        assert terminal[L:, b].all() and bool(terminal[L - 1, b]) and not terminal[:L - 1, b].any()
        assert f["legal_move"][:L, b].sum(1).min() >= 1
    tr.replay.update_priority(torch.ones(16, device="cuda:0"))
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


@pytest.mark.parametrize("games", [1024, 4096])
def test_cached_q_priorities_on_the_fused_cell_kernels(games):
    """the same bit-for-bit check at batch sizes that take the fused GEMM + cell kernels (2,048 rows: 128 x 128 tiles;
    8,192 rows: 256 x 256 tiles), where the hidden state also carries its bf16 copy from step to step"""
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    args = parse_args(["--num_game", str(games), "--batchsize", "16", "--replay_buffer_size", "4096", "--burn_in_frames", "64",
                       "--act_base_eps", "0.4", "--sad", "1", "--native_actor", "0"])
    tr = Trainer(args, "cuda:0")
    tr.actor.verify_cached_priority = True
    for it in range(24):
        if it in (9, 10):
            for net in (tr.act_online, tr.act_target):
                net.w["fc_a.weight"].mul_(1.02)
                net.w["lstm.weight_hh_l0"].mul_(0.99)
                net.refresh()
        tr.actor.step()
    assert "h0_16" in tr.actor.hid and tr.actor.n_checked >= 18 and tr.actor.n_checked_stale >= 3
    tr.env.check_errors()
    tr.replay.check_errors()


@pytest.mark.parametrize("method", ["iql", "vdn"])
def test_actor_priorities_from_cached_q_equal_compute_priority_bit_for_bit(method):
    """DeviceActor forms the n-step priorities from Q_online(s_t, a_t) / Q_target(s_t, greedy_t) computed when step t was
    acted on; the reference's call (compute_priority on the transition read back from the n-step ring, four network
    passes, r2d2.py:305-361 / r2d2_actor.h:128-150) must give the same bits every step -- also across actor weight syncs,
    where the cached online value is stale and the actor redoes that one pass with the new weights."""
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    args = parse_args(["--num_game", "64", "--rnn_hid_dim", "64", "--batchsize", "16", "--replay_buffer_size", "1024",
                       "--burn_in_frames", "64", "--max_len", "40", "--act_base_eps", "0.4", "--sad", "1",
                       "--method", method, "--native_actor", "0"])
    tr = Trainer(args, "cuda:0")
    tr.actor.verify_cached_priority = True
    for it in range(60):
        if it in (20, 21, 35):          # a weight sync (with really different weights) inside the n-step window
            for net in (tr.act_online, tr.act_target):
                net.w["fc_a.weight"].mul_(1.02)
                net.w["lstm.weight_hh_l0"].mul_(0.99)
                net.refresh()
        tr.actor.step()
    assert tr.actor.n_checked >= 50 and 6 <= tr.actor.n_checked_stale <= 9
    # the greedy action act() hands out is the argmax compute_priority would recompute on (next_obs, next_hid)
    agent = tr.actor.agent
    obs = tr.actor._rows()
    assert tr.actor.packed_obs and "priv_s" not in obs       # the loop above ran on the env's packed observation outputs
    obs["priv_s"] = obs["priv_s_bf16"][:, :tr.env.F].float().contiguous()
    hid = {k: v.clone() for k, v in tr.actor.hid.items()}
    reply, _ = agent.act(dict(obs, eps=torch.zeros_like(obs["eps"])), hid)
    z = torch.zeros(tr.actor.E, device="cuda:0")
    p1 = agent.compute_priority(obs, reply["a"], obs, hid, hid, z, z + 1, num_player=tr.actor.P if tr.actor.vdn else 1)
    p2 = agent.compute_priority(obs, reply["a"], obs, hid, hid, z, z + 1, num_player=tr.actor.P if tr.actor.vdn else 1,
                                next_greedy_a=reply["greedy_a"])
    assert torch.equal(p1, p2)


def test_reference_named_mirrors():
    """hanalearn.HanabiEnv / rela.RNNPrioritizedReplay / rela.aggregate_priority keep the reference call surface."""
    from hanabi_sad_amd import hanalearn, rela
    from oracle.oracle import OracleEnv, policy_random
    params = {"players": "2", "hand_size": "5", "seed": "77", "bomb": "0"}
    env = hanalearn.HanabiEnv(params, [0.1, 0.2], 80, True, False, False, False)
    ref = OracleEnv(players=2, hand_size=5, seed=77, eps_list=[0.1, 0.2], max_len=80, sad=True)
    assert (env.feature_size(), env.num_action(), env.hand_feature_size()) == (838, 21, 125)
    assert env.terminated()
    o, ro = env.reset(), ref.reset()
    n = 0
    while not ref.terminated():
        assert np.array_equal(o["priv_s"].cpu().numpy(), ro["priv_s"]) and env.get_current_player() == ref.cur_player()
        a, g = policy_random(ro["legal_move"], 3, 0, n)
        assert all(env.move_is_legal(int(u)) == ref.move_is_legal(int(u)) for u in range(20))
        o, r, t = env.step({"a": torch.tensor(a), "greedy_a": torch.tensor(g)})
        ro, rr, rt = ref.step(a, g)
        assert (r, t) == (rr, rt)
        n += 1
    assert env.terminated() and env.get_score() == ref.get("score") and env.last_score() == ref.get("last_score")
    assert env.get_fireworks() == ref.fireworks() and env.get_life() == ref.get("life") and env.get_info() == ref.get("info")
    assert len(env.deck_history()) == len(ref.deck_history())
    p = torch.tensor([[1., 2], [3, 4], [5, 6]])
    assert np.allclose(rela.aggregate_priority(p, torch.tensor([2., 3]), 0.9).numpy(), [2.9, 5.8], rtol=1e-6)
    rep = rela.RNNPrioritizedReplay(64, 1, 0.9, 0.6, 3)
    assert rep.size() == 0 and rep.num_add() == 0


def test_batched_greedy_evaluation_and_checkpoint_roundtrip(tmp_path):
    """eval.evaluate semantics (fresh games, eps = 0, max_len = -1, greedy) + `.pthw` key compatibility."""
    from hanabi_sad_amd.eval import evaluate, load_weights, save_weights
    from hanabi_sad_amd.selfplay import init_weights
    W = init_weights(838, 64, 21, 5, seed=3)
    path = str(tmp_path / "model0.pthw")
    save_weights(W, path)
    sd = torch.load(path)
    assert list(sd.keys())[:2] == ["net.0.weight", "net.0.bias"] and sd["lstm.weight_hh_l1"].shape == (256, 64)
    W2 = load_weights(path)
    assert all(torch.equal(W[k], W2[k]) for k in W)
    mean, perfect, scores, n_perfect = evaluate(W2, 200, seed=9917, bomb=0, sad=True, device="cuda:0")
    assert len(scores) == 200 and all(0 <= s <= 25 for s in scores) and 0 <= mean <= 25
    mean2, _, scores2, _ = evaluate(W2, 200, seed=9917, bomb=0, sad=True, device="cuda:0")
    assert scores == scores2                      # deterministic: greedy policy, fixed seeds


@pytest.mark.parametrize("method,games,hid", [("iql", 64, 64), ("vdn", 64, 64), ("iql", 1024, 512)])
def test_library_actor_step_equals_the_python_body_bit_for_bit(method, games, hid):
    """hsad_actor_step (include/hsad.h; csrc/hsad_actor.hip) against actor.DeviceActor's Python body on the same seeds: the actions of every
    step, the priorities of every step -- across actor weight syncs inside the n-step window, where the library redoes the online pass
    on the unpacked transition -- the carried state, and every sequence that reached the replay must be identical bits.  The Python body
    is in turn pinned to the reference's compute_priority (tests above) and to the oracle env + reference buffers (test_actor_oracle_e2e)."""
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    common = ["--num_game", str(games), "--rnn_hid_dim", str(hid), "--batchsize", "16", "--replay_buffer_size", "8192", "--burn_in_frames", "64",
              "--max_len", "30", "--act_base_eps", "0.4", "--sad", "1", "--method", method, "--seed", "11"]
    trs = [Trainer(parse_args(common + ["--native_actor", str(k)]), "cuda:0") for k in (1, 0)]
    nat, py = trs[0].actor, trs[1].actor
    assert nat.c_actor is not None and py.c_actor is None
    from hanabi_sad_amd import HsadError
    with pytest.raises(HsadError):           # the Python cross-check cannot be switched on over the library's loop
        nat.verify_cached_priority = True
    assert nat.verify_cached_priority is False
    if games == 1024:                        # the multi-GPU actor loop's bound on the host's run-ahead changes no result
        nat.set_run_ahead(2)
        py.set_run_ahead(2)
    rec_p = []
    orig_push = py.writer.push_sequence

    def push(prio):
        rec_p.append(prio.clone())
        orig_push(prio)
    py.writer.push_sequence = push
    steps = 50
    for it in range(steps):
        if it in (20, 21, 35):
            for tr in trs:
                for net in (tr.act_online, tr.act_target):
                    net.w["fc_a.weight"].mul_(1.02)
                    net.w["lstm.weight_hh_l0"].mul_(0.99)
                    net.refresh()
        rec_p.clear()
        nat.step()
        py.step()
        pr = nat.last_priority
        assert (pr is None) == (len(rec_p) == 0), it
        if pr is not None:
            assert torch.equal(pr, rec_p[0]), it
    torch.cuda.synchronize()
    assert 6 <= nat.num_redo <= 9 and nat.num_act == py.num_act == steps * nat.N
    import ctypes as C
    h, c = C.c_void_p(), C.c_void_p()
    from hanabi_sad_amd import _lib
    from hanabi_sad_amd.composite import _view
    _lib.check(nat._lib.hsad_actor_state(nat.c_actor, C.byref(h), C.byref(c)))
    n = py.hid["h0"].numel()
    assert torch.equal(_view(h.value, n, "cuda:0", nat), py.hid["h0"].reshape(-1)) and torch.equal(_view(c.value, n, "cuda:0", nat), py.hid["c0"].reshape(-1))
    for tr in trs:
        tr.env.check_errors()
        tr.replay.check_errors()
    ra, rb = trs[0].replay, trs[1].replay
    assert ra.size() == rb.size() > 0 and ra.num_add() == rb.num_add()
    assert ra.priority_sum() == rb.priority_sum()
    for i in range(0, ra.size(), max(1, ra.size() // 97)):
        fa, rwa, ta, ba, sa = ra.get(i)
        fb, rwb, tb, bb, sb = rb.get(i)
        assert all(torch.equal(fa[k], fb[k]) for k in fa) and torch.equal(rwa, rwb) and torch.equal(ta, tb) and torch.equal(ba, bb) and torch.equal(sa, sb), i


def test_run_ahead_bound_keeps_the_host_within_a_few_steps_of_the_device():
    """hsad_actor_set_run_ahead(k): when hsad_actor_step(t) returns, step t - k has left the device (an actor rank serves a learner's
    round behind at most k queued steps instead of behind everything its host managed to enqueue)"""
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    tr = Trainer(parse_args(["--num_game", "4096", "--rnn_hid_dim", "512", "--batchsize", "16", "--replay_buffer_size", "16384",
                             "--burn_in_frames", "64", "--sad", "1", "--seed", "3"]), "cuda:0")
    act = tr.actor
    assert act.c_actor is not None
    act.set_run_ahead(2)
    marks = []
    for it in range(40):
        act.step()
        e = torch.cuda.Event()
        e.record()
        marks.append(e)
        if it >= 3:
            assert marks[it - 3].query(), it
    act.set_run_ahead(0)
    for it in range(5):
        act.step()
    torch.cuda.synchronize()
    tr.env.check_errors()
    tr.replay.check_errors()

