"""GPU: the reference-shaped training driver end to end: epochs with Tachometer / Stopwatch /
MultiCounter output, per-epoch evaluation, top-k saving into --save_dir with the reference's key names, --load_model; and an
agent loaded from a `.pthw` written by the reference acts like the reference agent (tests/golden/ref_small*)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_agent_loaded_from_a_reference_checkpoint_acts_like_the_reference(precision):
    from hanabi_sad_amd.checkpoint import agent_from_file, load_sad_model
    z = np.load(os.path.join(GOLD, "ref_small_expected.npz"))
    agent = agent_from_file(os.path.join(GOLD, "ref_small.pthw"), DEV, precision=precision)
    t = lambda k: torch.tensor(z[k]).to(DEV)
    obs = {"priv_s": t("priv_s"), "legal_move": t("legal_move"), "eps": torch.zeros(z["priv_s"].shape[0], device=DEV)}
    reply, hid = agent.act(obs, {"h0": t("h0"), "c0": t("c0")})
    tol = 1e-5 if precision == "fp32" else 2e-3
    assert float((hid["h0"].cpu() - torch.tensor(z["out_h0"])).abs().max()) < tol
    assert float((hid["c0"].cpu() - torch.tensor(z["out_c0"])).abs().max()) < 2 * tol
    agree = float((reply["greedy_a"].cpu() == torch.tensor(z["greedy_a"])).float().mean())
    assert agree == 1.0 if precision == "fp32" else agree >= 0.9
    assert len(load_sad_model([os.path.join(GOLD, "ref_small.pthw")] * 2, DEV)) == 2


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_load_op_model_runs_every_architecture_class_of_the_zoo(precision):
    """utils.load_op_model (pyhanabi/utils.py:36-84): M0-2 default, M3-5 skip connection, M6-8 two fc layers, M9-11 both.  The files under
    tests/golden/op_zoo were written by the reference (make_op_zoo_fixture.py), one per class; the loaded agents must answer like the
    reference's agents: same greedy actions, same new hidden state"""
    from hanabi_sad_amd.checkpoint import load_op_model
    z = np.load(os.path.join(GOLD, "op_zoo_expected.npz"))
    root = os.path.join(GOLD, "op_zoo")
    t = lambda k: torch.tensor(z[k]).to(DEV)
    for i, j in ((0, 3), (6, 9)):
        agents = load_op_model("sad", i, j, DEV, root=root, precision=precision)
        assert len(agents) == 2
        for idx, agent in zip((i, j), agents):
            obs = {"priv_s": t("priv_s"), "legal_move": t("legal_move"), "eps": torch.zeros(z["priv_s"].shape[0], device=DEV)}
            reply, hid = agent.act(obs, {"h0": t("h0"), "c0": t("c0")})
            tol = 1e-5 if precision == "fp32" else 2e-3
            assert float((hid["h0"].cpu() - torch.tensor(z["M%d.out_h0" % idx])).abs().max()) < tol, idx
            want = torch.tensor(z["M%d.greedy_a" % idx])
            got = reply["greedy_a"].cpu()
            if precision == "fp32":
                assert torch.equal(got, want), idx
            else:       # a bf16 flip is only acceptable between advantages the reference itself has within 5e-3 of each other
                adv = torch.tensor(z["M%d.adv" % idx])
                for r in torch.nonzero(got != want).flatten().tolist():
                    assert abs(float(adv[r, got[r]] - adv[r, want[r]])) < 5e-3, (idx, r)
    with pytest.raises(FileNotFoundError):
        load_op_model("sad", 1, None, DEV, root=root)


def test_epoch_loop_logs_evaluates_and_saves(tmp_path):
    from hanabi_sad_amd import selfplay
    from hanabi_sad_amd.checkpoint import load_weights
    save_dir = str(tmp_path / "exp1")
    argv = ["--save_dir", save_dir, "--num_game", "64", "--rnn_hid_dim", "64", "--batchsize", "16", "--replay_buffer_size", "2048",
            "--burn_in_frames", "64", "--max_len", "40", "--num_epoch", "2", "--epoch_len", "5", "--num_eval_game", "48",
            "--stopwatch", "1", "--pred_weight", "0.25", "--load_model", os.path.join(GOLD, "ref_small.pthw"), "--seed", "7"]
    old = sys.stdout
    try:
        selfplay.main(argv)
    finally:
        sys.stdout = old
    log = open(os.path.join(save_dir, "train.log")).read()
    assert log.lstrip().startswith("{")                          # the pprint'ed args: the saved configuration
    for needle in ("*****loading pretrained model*****", "beginning of epoch:  1", "EPOCH: 1", "Speed: train:", "@@@Time",
                   "sync and updating", "sample data", "forward & backward", "update model", "updating priority",
                   "1:loss", "1:grad_norm", "epoch 1, eval score:", "model saved: True"):
        assert needle in log, needle
    w = load_weights(os.path.join(save_dir, "model0.pthw"))       # reference key names, loadable back
    ref = torch.load(os.path.join(GOLD, "ref_small.pthw"))
    assert set(w) == set(ref) and not torch.equal(w["fc_a.weight"], ref["fc_a.weight"])   # trained away from the loaded weights


@pytest.mark.parametrize("nl", [1, 3])
def test_num_lstm_layer_flag_trains_and_saves_that_architecture(tmp_path, nl):
    """--num_lstm_layer (pyhanabi/selfplay.py:50): the epoch loop with a 1- / 3-layer LSTM end to end -- acting, replay, learner, evaluation,
    checkpoint with the reference's key names for that depth, loadable back as an agent"""
    from hanabi_sad_amd import selfplay
    from hanabi_sad_amd.checkpoint import agent_from_file, load_weights
    save_dir = str(tmp_path / ("exp_l%d" % nl))
    argv = ["--save_dir", save_dir, "--num_game", "64", "--rnn_hid_dim", "64", "--batchsize", "16", "--replay_buffer_size", "2048",
            "--burn_in_frames", "64", "--max_len", "40", "--num_epoch", "1", "--epoch_len", "5", "--num_eval_game", "32", "--seed", "3",
            "--num_lstm_layer", str(nl)]
    old = sys.stdout
    try:
        selfplay.main(argv)
    finally:
        sys.stdout = old
    log = open(os.path.join(save_dir, "train.log")).read()
    assert "epoch 0, eval score:" in log and "model saved: True" in log
    w = load_weights(os.path.join(save_dir, "model0.pthw"))
    assert sorted(k for k in w if k.startswith("lstm.weight_ih")) == ["lstm.weight_ih_l%d" % l for l in range(nl)]
    agent = agent_from_file(os.path.join(save_dir, "model0.pthw"), DEV)
    assert agent.get_h0(4)["h0"].shape[0] == nl


def test_the_early_draw_sees_the_same_data_as_the_draw_behind_the_optimizer_step():
    """selfplay --early_draw 1 (default, round 6): priority write-back and the next draw are issued between the forward half and the BPTT of
    an update, on a stream of their own that waits for the forward half only -- instead of behind the optimizer step on the caller's stream.
    The order of the replay operations (flush -> write-back -> draw -> next flush) is the same, so the loop must see the same batches: the
    per-update losses and the replay's final state agree with --early_draw 0 and with the strictly alternating loop."""
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    out = []
    for early, overlap in ((0, 1), (1, 1), (1, 0)):
        tr = Trainer(parse_args(["--num_game", "1024", "--sad", "1", "--seed", "5", "--burn_in_frames", "1000", "--replay_buffer_size", "8192",
                                 "--batchsize", "64", "--overlap_rollout", str(overlap), "--actor_sync_freq", "5", "--early_draw", str(early)]), "cuda:0")
        assert (tr.draw_stream is not None) == bool(early)
        tr.act_step(120)
        losses = []
        for u in range(40):
            tr.act_step(1)
            loss, g_norm = tr.learner_update()
            losses.append(loss.detach().float())
        tr.join_rollout()
        torch.cuda.synchronize()
        tr.env.check_errors()
        tr.replay.check_errors()
        tr.learner.check_sync()
        out.append((torch.stack(losses).cpu(), tr.replay.num_add(), tr.replay.size(), tr.replay.priority_sum()[0], tr.actor.num_act))
        del tr
        torch.cuda.empty_cache()
    (l0, *s0) = out[0]
    for (l1, *s1) in out[1:]:
        assert s0[0] == s1[0] and s0[1] == s1[1] and s0[3] == s1[3]
        assert torch.allclose(l0, l1, rtol=1e-4, atol=1e-5), (l0 - l1).abs().max()
        assert abs(s0[2] - s1[2]) <= 1e-3 * abs(s0[2])


def test_overlapped_rollout_sees_the_same_data_as_the_alternating_loop():
    """selfplay --overlap_rollout 1 (the default): rollout steps on a stream of their own next to the update on the caller's.  Every
    cross-stream dependency (sequence flush -> sample, sample / update_priority -> next flush, optimizer step -> actor weight sync ->
    next update) is enforced in host issue order, so the overlapped loop must see exactly the data of the strictly alternating one: the
    per-update losses agree (to the float-atomic bias sums of the BPTT launch) and the replay ends in the same state."""
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    out = []
    for overlap in (0, 1):
        tr = Trainer(parse_args(["--num_game", "1024", "--sad", "1", "--seed", "5", "--burn_in_frames", "1000", "--replay_buffer_size", "8192",
                                 "--batchsize", "64", "--overlap_rollout", str(overlap), "--actor_sync_freq", "5"]), "cuda:0")
        assert (tr.act_stream is not None) == bool(overlap)
        tr.act_step(120)                               # a fixed burn-in: the driver's polling loop depends on timing
        losses = []
        for u in range(40):
            tr.act_step(1)
            loss, g_norm = tr.learner_update()
            losses.append(loss.detach().float())
        tr.join_rollout()
        torch.cuda.synchronize()
        tr.env.check_errors()
        tr.replay.check_errors()
        tr.learner.check_sync()
        out.append((torch.stack(losses).cpu(), tr.replay.num_add(), tr.replay.size(), tr.replay.priority_sum()[0], tr.actor.num_act))
        del tr
        torch.cuda.empty_cache()
    (l0, *s0), (l1, *s1) = out
    assert s0[0] == s1[0] and s0[1] == s1[1] and s0[3] == s1[3]
    assert torch.allclose(l0, l1, rtol=1e-4, atol=1e-5), (l0 - l1).abs().max()
    assert abs(s0[2] - s1[2]) <= 1e-3 * abs(s0[2])
