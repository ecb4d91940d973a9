"""CPU tests of the run-level utilities (hanabi_sad_amd/common.py, checkpoint.py): behaviour of the
reference's TopkSaver / Stopwatch / MultiCounter / Tachometer / Logger and of its `.pthw` loaders, the latter against files
written by the reference itself (tests/golden/ref_small*.pthw, tests/golden/make_pthw_fixture.py)."""
import os
import sys

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_FLAGS = ["save_dir", "method", "shuffle_obs", "shuffle_color", "pred_weight", "num_eps", "load_model", "seed", "gamma", "eta",
             "train_bomb", "eval_bomb", "sad", "num_player", "hand_size", "lr", "eps", "grad_clip", "num_lstm_layer",
             "rnn_hid_dim", "train_device", "batchsize", "num_epoch", "epoch_len", "num_update_between_sync", "multi_step",
             "burn_in_frames", "replay_buffer_size", "priority_exponent", "priority_weight", "max_len", "prefetch", "num_thread",
             "num_game_per_thread", "act_base_eps", "act_eps_alpha", "act_device", "actor_sync_freq"]   # selfplay.py:26-86


def test_every_reference_flag_is_accepted():
    from hanabi_sad_amd.selfplay import parse_args
    a = parse_args([])
    for f in REF_FLAGS:
        assert hasattr(a, f), f
    a = parse_args(["--num_thread", "80", "--num_game_per_thread", "80", "--method", "vdn", "--act_device", "cuda:1,cuda:2"])
    assert a.num_game == 6400
    with pytest.raises(SystemExit):
        parse_args(["--shuffle_obs", "1"])


def test_topk_saver_follows_the_reference_slot_for_slot(tmp_path):
    """tests/golden/topk_saver_trace.json: the reference's TopkSaver driven with fixed score sequences (make_topk_trace.py): per call the
    returned flag and which model{i}.pthw holds which score afterwards -- quirks included (the -inf seed entry, slot 0 rewritten while
    the list fills, strict comparison)"""
    import json
    from hanabi_sad_amd.common import TopkSaver
    trace = json.load(open(os.path.join(GOLD, "topk_saver_trace.json")))
    for name, case in trace.items():
        d = tmp_path / name
        s = TopkSaver(str(d), case["topk"])
        for i, step in enumerate(case["steps"]):
            flag = s.save(None, {"w": torch.tensor([step["score"]])}, step["score"])
            assert flag is step["saved"], (name, i)
            files = {f: float(torch.load(str(d / f))["w"]) for f in sorted(os.listdir(d)) if f.startswith("model")}
            assert files == pytest.approx(step["files"]), (name, i)
    s = TopkSaver(str(tmp_path / "forced"), 2)
    assert s.save(None, {"w": torch.tensor([0.1])}, 0.1, save_latest=True, force_save_name="model_epoch50") is True
    assert float(torch.load(str(tmp_path / "forced" / "model_epoch50.pthw"))["w"]) == pytest.approx(0.1)
    assert float(torch.load(str(tmp_path / "forced" / "latest.pthw"))["w"]) == pytest.approx(0.1)


def test_stopwatch_multicounter_tachometer_output(capsys):
    from hanabi_sad_amd.common import MultiCounter, Stopwatch, Tachometer, ValueStats, num2str, sec2str
    sw = Stopwatch()
    for _ in range(3):
        sw.time("sample data")
        sw.time("forward & backward")
    sw.summary()
    out = capsys.readouterr().out
    assert "@@@Time" in out and "sample data" in out and "forward & backward" in out and "@@@total time per iter" in out
    st = MultiCounter(None)
    for v in (3.0, 1.0, 2.0):
        st["loss"].feed(v)
    st.inc("saved")
    st.summary(7)
    out = capsys.readouterr().out
    assert "7:loss" in out and "avg:   2.0000" in out and "min:   1.0000[   1]" in out and "max:   3.0000[   0]" in out
    assert "saved: 1/1" in out
    v = ValueStats("x")
    assert v.summary() == "x[0]"
    assert sec2str(3725) == "1H 02M 05S" and num2str(999) == "999" and num2str(1500) == "1.500K" and num2str(2.5e6) == "2.500M"

    class A:
        n = 0

        def num_act(self):
            return self.n

    class R:
        def num_add(self):
            return 40

        def size(self):
            return 30
    a, t = A(), Tachometer()
    t.start()
    a.n = 1000
    import time
    time.sleep(0.05)
    train, act, add = t.lap([a], R(), 128, 2)
    out = capsys.readouterr().out
    assert "Speed: train:" in out and "buffer_size: 30" in out and "Total Sample: train: 128, act: 1.000K" in out
    assert act / train == pytest.approx(1000 / 128) and add / train == pytest.approx(40 / 128)


def test_logger_tees_stdout(tmp_path, capsys):
    from hanabi_sad_amd.common import Logger
    path = str(tmp_path / "exp" / "train.log")
    old = sys.stdout
    try:
        sys.stdout = Logger(path)
        print("hello log")
    finally:
        sys.stdout = old
    assert "hello log" in open(path).read()


def test_load_weight_semantics_on_files_written_by_the_reference(capsys):
    from hanabi_sad_amd.checkpoint import load_weight, load_weights, op_model_arch
    from hanabi_sad_amd.selfplay import init_weights
    ref = torch.load(os.path.join(GOLD, "ref_small.pthw"), map_location="cpu")
    W = init_weights(838, 64, 21, 5, 3)
    assert set(W) == set(ref)                                   # same key names as R2D2Net.state_dict()
    loaded, kept, dropped = load_weight(W, os.path.join(GOLD, "ref_small.pthw"))
    assert not kept and not dropped and all(torch.equal(W[k], ref[k]) for k in ref)
    assert all(torch.equal(v, ref[k]) for k, v in load_weights(os.path.join(GOLD, "ref_small.pthw")).items())
    # legacy file: no pred.* (keep the net's own), an unknown key (ignored)
    W2 = init_weights(838, 64, 21, 5, 4)
    own_pred = W2["pred.weight"].clone()
    loaded, kept, dropped = load_weight(W2, os.path.join(GOLD, "ref_small_legacy.pthw"))
    out = capsys.readouterr().out
    assert sorted(kept) == ["pred.bias", "pred.weight"] and dropped == ["obsolete.weight"]
    assert "warning: pred.weight not loaded" in out and "removing: obsolete.weight not used" in out
    assert torch.equal(W2["pred.weight"], own_pred) and torch.equal(W2["fc_a.weight"], ref["fc_a.weight"])
    with pytest.raises(ValueError):
        load_weight(init_weights(838, 32, 21, 5, 0), os.path.join(GOLD, "ref_small.pthw"))
    assert [op_model_arch(i) for i in (0, 2, 3, 5, 6, 8, 9, 11)] == [(1, False), (1, False), (1, True), (1, True), (2, False),
                                                                     (2, False), (2, True), (2, True)]


def test_convert_model_exports_a_loadable_torchscript_net(tmp_path):
    """hanabi_sad_amd.convert_model (pyhanabi/tools/convert_model.py): the exported module reloads with torch.jit.load and its forward
    equals a plain torch evaluation of the same weights (ReLU(Linear) -> 2-layer LSTM step -> fc_a), state in batch-first layout"""
    import torch
    from hanabi_sad_amd.checkpoint import save_weights
    from hanabi_sad_amd.convert_model import convert
    g = torch.Generator().manual_seed(0)
    F, H, A, L, B = 40, 16, 7, 2, 5
    sd = {"net.0.weight": torch.randn(H, F, generator=g) * 0.2, "net.0.bias": torch.randn(H, generator=g) * 0.1,
          "fc_v.weight": torch.randn(1, H, generator=g), "fc_v.bias": torch.randn(1, generator=g),
          "fc_a.weight": torch.randn(A, H, generator=g), "fc_a.bias": torch.randn(A, generator=g),
          "pred.weight": torch.randn(15, H, generator=g), "pred.bias": torch.randn(15, generator=g)}
    for l in range(L):
        sd["lstm.weight_ih_l%d" % l] = torch.randn(4 * H, H, generator=g) * 0.2
        sd["lstm.weight_hh_l%d" % l] = torch.randn(4 * H, H, generator=g) * 0.2
        sd["lstm.bias_ih_l%d" % l] = torch.randn(4 * H, generator=g) * 0.1
        sd["lstm.bias_hh_l%d" % l] = torch.randn(4 * H, generator=g) * 0.1
    path = str(tmp_path / "model0.pthw")
    save_weights(sd, path)
    _, out = convert(path)
    assert out.endswith("model0.sparta")
    m = torch.jit.load(out)
    s, h0, c0 = torch.randn(B, F, generator=g), torch.randn(B, L, H, generator=g) * 0.3, torch.randn(B, L, H, generator=g) * 0.3
    got = m({"s": s, "h0": h0, "c0": c0})
    x = torch.relu(s @ sd["net.0.weight"].T + sd["net.0.bias"])
    hs, cs = [], []
    for l in range(L):
        gates = x @ sd["lstm.weight_ih_l%d" % l].T + sd["lstm.bias_ih_l%d" % l] + h0[:, l] @ sd["lstm.weight_hh_l%d" % l].T + sd["lstm.bias_hh_l%d" % l]
        i, f, gg, o = gates.chunk(4, 1)
        c = torch.sigmoid(f) * c0[:, l] + torch.sigmoid(i) * torch.tanh(gg)
        x = torch.sigmoid(o) * torch.tanh(c)
        hs.append(x)
        cs.append(c)
    assert torch.allclose(got["a"], x @ sd["fc_a.weight"].T + sd["fc_a.bias"], atol=1e-5)
    assert torch.allclose(got["h0"], torch.stack(hs, 1), atol=1e-5) and torch.allclose(got["c0"], torch.stack(cs, 1), atol=1e-5)


def test_convert_model_keeps_a_second_fc_layer_and_accepts_an_agent_file(tmp_path):
    """a num_fc_layer = 2 checkpoint (net.2.*) must not lose net.2.* on export, and a whole agent's state_dict
    (online_net.* / target_net.*) is accepted by load_weights, convert_model and action_matrix alike"""
    import torch
    from hanabi_sad_amd.checkpoint import load_weights
    from hanabi_sad_amd.convert_model import convert
    g = torch.Generator().manual_seed(1)
    F, H, A, B = 24, 8, 5, 3
    sd = {"net.0.weight": torch.randn(H, F, generator=g) * 0.3, "net.0.bias": torch.randn(H, generator=g) * 0.1,
          "net.2.weight": torch.randn(H, H, generator=g) * 0.5, "net.2.bias": torch.randn(H, generator=g) * 0.1,
          "fc_v.weight": torch.randn(1, H, generator=g), "fc_v.bias": torch.randn(1, generator=g),
          "fc_a.weight": torch.randn(A, H, generator=g), "fc_a.bias": torch.randn(A, generator=g),
          "pred.weight": torch.randn(15, H, generator=g), "pred.bias": torch.randn(15, generator=g),
          "lstm.weight_ih_l0": torch.randn(4 * H, H, generator=g) * 0.3, "lstm.weight_hh_l0": torch.randn(4 * H, H, generator=g) * 0.3,
          "lstm.bias_ih_l0": torch.randn(4 * H, generator=g) * 0.1, "lstm.bias_hh_l0": torch.randn(4 * H, generator=g) * 0.1}
    agent = {"online_net." + k: v for k, v in sd.items()}
    agent.update({"target_net." + k: v + 1 for k, v in sd.items()})
    path = str(tmp_path / "agent.pthw")
    torch.save(agent, path)
    w = load_weights(path)
    assert set(w) == set(sd) and all(torch.equal(w[k], sd[k]) for k in sd)          # the online net, prefix stripped
    m, out = convert(path)
    assert "net.2.weight" in m.state_dict() and torch.equal(m.state_dict()["net.2.weight"], sd["net.2.weight"])
    s, h0, c0 = torch.randn(B, F, generator=g), torch.zeros(B, 1, H), torch.zeros(B, 1, H)
    got = torch.jit.load(out)({"s": s, "h0": h0, "c0": c0})
    x = torch.relu(torch.relu(s @ sd["net.0.weight"].T + sd["net.0.bias"]) @ sd["net.2.weight"].T + sd["net.2.bias"])
    gates = x @ sd["lstm.weight_ih_l0"].T + sd["lstm.bias_ih_l0"] + sd["lstm.bias_hh_l0"]
    i, f, gg, o = gates.chunk(4, 1)
    h = torch.sigmoid(o) * torch.tanh(torch.sigmoid(i) * torch.tanh(gg))
    assert torch.allclose(got["a"], h @ sd["fc_a.weight"].T + sd["fc_a.bias"], atol=1e-5)
