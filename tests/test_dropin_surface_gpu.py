"""A driver written against the reference's `hanalearn` / `rela` names (the calls of pyhanabi/create.py:24-76,98-145,
selfplay.py:152-245 and eval.py:25-66, in the same order) running on the device pipeline through the mirror modules."""
import time

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
DEV = "cuda:0"


class TinyAgent:
    """stands in for the reference's torch R2D2Agent: all the mirrors need is state_dict() with online_net.* / target_net.*"""

    def __init__(self, in_dim, hid, out_dim, hand, seed):
        from hanabi_sad_amd.selfplay import init_weights
        W = init_weights(in_dim, hid, out_dim, hand, seed)
        self.sd = {"online_net." + k: v for k, v in W.items()}
        self.sd.update({"target_net." + k: v.clone() for k, v in W.items()})

    def state_dict(self):
        return self.sd


def create_envs(hanalearn, num_env, seed, num_player, hand_size, bomb, eps, max_len, sad):
    games = []
    for i in range(num_env):
        params = {"players": str(num_player), "hand_size": str(hand_size), "seed": str(seed + i), "bomb": str(bomb)}
        games.append(hanalearn.HanabiEnv(params, eps, max_len, sad, False, False, False))
    return games


@pytest.mark.parametrize("method", ["iql", "vdn"])
def test_reference_shaped_training_driver(method):
    import hanalearn
    import rela                     # the top-level module names the reference's drivers import (create.py:16-21)
    assert rela.__file__.endswith(".so") and hanalearn.__file__.endswith(".so")
    from hanabi_sad_amd.r2d2 import R2D2Learner
    from hanabi_sad_amd.selfplay import generate_explore_eps
    num_thread, per_thread, P, hand, n, gamma, eta, T, B = 2, 96, 2, 5, 3, 0.999, 0.9, 80, 32
    eps = generate_explore_eps(0.1, 7, 80)
    games = create_envs(hanalearn, num_thread * per_thread, 7, P, hand, 0, eps, T, True)
    F, A = games[0].feature_size(), games[0].num_action()
    assert (F, A) == (838, 21)
    agent = TinyAgent(F, 256, A, hand, 3)
    replay = rela.RNNPrioritizedReplay(4096, 1, 0.9, 0.6, 3)
    runner = rela.BatchRunner(agent, DEV, 100, ["act", "compute_priority"])
    runner.start()
    context, threads, actors = rela.Context(), [], []
    for t in range(num_thread):
        if method == "vdn":
            acts = rela.R2D2Actor(runner, n, per_thread, gamma, eta, T, P, replay)
            actors.append(acts)
        else:
            acts = [rela.R2D2Actor(runner, n, per_thread, gamma, eta, T, 1, replay) for _ in range(P)]
            actors.extend(acts)
        env = hanalearn.HanabiVecEnv()
        for g in range(per_thread):
            env.append(games[t * per_thread + g])
        th = hanalearn.HanabiThreadLoop(acts, env, False)
        threads.append(th)
        context.push_env_thread(th)
    context.start()
    t0 = time.time()
    while replay.size() < 4 * B:                       # burn in (selfplay.py:182-186)
        assert time.time() - t0 < 120
        time.sleep(0.05)
    on = {k[len("online_net."):]: v for k, v in agent.state_dict().items() if k.startswith("online_net.")}
    learner = R2D2Learner(on, on, n, gamma, device=DEV)
    for it in range(4):                                 # train-loop body (selfplay.py:208-244)
        batch, weight = replay.sample(B, DEV)
        obs, act = batch.obs, batch.action
        if method == "vdn":
            v4 = lambda x: x.view(x.shape[0], x.shape[1], P, -1)
            b = {"priv_s": v4(obs["priv_s"]), "legal_move": v4(obs["legal_move"]), "a": act["a"], "own_hand": v4(obs["own_hand"])}
        else:
            b = {"priv_s": obs["priv_s"], "legal_move": obs["legal_move"], "a": act["a"].squeeze(-1) if act["a"].dim() == 3 else act["a"],
                 "own_hand": obs["own_hand"]}
        b.update(reward=batch.reward, bootstrap=batch.bootstrap, seq_len=batch.seq_len)
        loss, priority = learner.loss(b, weight, 0.0)
        learner.optimizer_step()
        replay.update_priority(rela.aggregate_priority(priority, batch.seq_len, eta))
        assert torch.isfinite(loss).all()
    context.pause()
    n_act = sum(a.num_act() for a in actors)
    assert n_act > 0 and replay.num_add() >= replay.size() > 0
    context.terminate()
    assert context.terminated()
    # the per-thread loops were merged into ONE batched device loop (consecutive seeds, same models): one launch per kernel
    assert threads[1].master is threads[0] and threads[0].env.G == num_thread * per_thread
    threads[0].env.check_errors()
    per_actor = [a.num_act() for a in actors]
    assert len(set(per_actor)) == 1 and per_actor[0] % per_thread == 0          # R2D2Actor::numAct_ += num_envs per act()


def test_context_pace_runs_the_rollout_by_the_training_loops_samples():
    """rela.Context.set_pace (an addition to the reference's free-running Context for one-GPU jobs): the loop thread free-runs while
    nobody samples (burn-in), then issues `steps_per_sample` rollout steps per replay.sample() call of the training loop and stops
    when that loop stops; set_pace(None) returns to free-running"""
    import hanalearn
    import rela
    from hanabi_sad_amd.selfplay import generate_explore_eps
    per_thread, P, hand, n, gamma, eta, T, B = 64, 2, 5, 3, 0.999, 0.9, 80, 16
    eps = generate_explore_eps(0.1, 7, 80)
    games = create_envs(hanalearn, per_thread, 11, P, hand, 0, eps, T, True)
    agent = TinyAgent(games[0].feature_size(), 128, games[0].num_action(), hand, 3)
    replay = rela.RNNPrioritizedReplay(2048, 1, 0.9, 0.6, 3)
    runner = rela.BatchRunner(agent, DEV, 100, ["act", "compute_priority"])
    acts = [rela.R2D2Actor(runner, n, per_thread, gamma, eta, T, 1, replay) for _ in range(P)]
    env = hanalearn.HanabiVecEnv()
    for g in games:
        env.append(g)
    context = rela.Context()
    context.push_env_thread(hanalearn.HanabiThreadLoop(acts, env, False))
    context.set_pace(replay, 2.0)
    context.start()
    steps = lambda: acts[0].num_act() // per_thread
    t0 = time.time()
    while replay.size() < 4 * B:                       # nobody samples yet: free-running burn-in
        assert time.time() - t0 < 120
        time.sleep(0.02)

    def settled():
        a = steps()
        time.sleep(0.3)
        return a if steps() == a else None
    replay.sample(B, DEV)                               # the first sample() starts the clock: two steps of credit
    replay.update_priority(torch.ones(B, device=DEV))
    t0 = time.time()
    while (base := settled()) is None:                  # the thread spends its credit, then waits
        assert time.time() - t0 < 60
    for k in range(1, 6):
        replay.sample(B, DEV)
        replay.update_priority(torch.ones(B, device=DEV))
        t0 = time.time()
        while steps() < base + 2 * k:
            assert time.time() - t0 < 60
            time.sleep(0.005)
        time.sleep(0.2)
        assert steps() == base + 2 * k, (k, steps(), base)     # exactly two more steps per sample, never a third
    context.set_pace(None)
    t0 = time.time()
    while steps() < base + 40:                          # free-running again
        assert time.time() - t0 < 60
        time.sleep(0.01)
    context.pause()
    context.terminate()


def test_an_unchanged_driver_is_paced_by_default():
    """the reference's driver (selfplay.py:180-244) calls context.start() and then loops over replay.sample() --
    nothing else.  Without any set_pace call the Context must run the rollout at its default ratio (2 steps per sample) while that
    loop samples -- the measured operating point of tools/time_dropin.py -- and free-run before (burn-in) and after it."""
    import hanalearn
    import rela
    from hanabi_sad_amd.selfplay import generate_explore_eps
    per_thread, P, hand, n, gamma, eta, T, B = 64, 2, 5, 3, 0.999, 0.9, 80, 16
    eps = generate_explore_eps(0.1, 7, 80)
    games = create_envs(hanalearn, per_thread, 21, P, hand, 0, eps, T, True)
    agent = TinyAgent(games[0].feature_size(), 128, games[0].num_action(), hand, 3)
    replay = rela.RNNPrioritizedReplay(2048, 1, 0.9, 0.6, 3)
    runner = rela.BatchRunner(agent, DEV, 100, ["act", "compute_priority"])
    acts = [rela.R2D2Actor(runner, n, per_thread, gamma, eta, T, 1, replay) for _ in range(P)]
    env = hanalearn.HanabiVecEnv()
    for g in games:
        env.append(g)
    context = rela.Context()
    context.push_env_thread(hanalearn.HanabiThreadLoop(acts, env, False))
    assert context.auto_pace_steps == 2.0
    context.start()                                     # ... and no set_pace
    steps = lambda: acts[0].num_act() // per_thread
    t0 = time.time()
    while replay.size() < 4 * B:                        # burn-in free-runs
        assert time.time() - t0 < 120
        time.sleep(0.02)
    a = steps()
    time.sleep(0.3)
    assert steps() > a + 10                             # still free-running: nobody samples
    context.auto_pace_idle_s = 1.0                      # (a wide window so that the assertions below cannot race the fall-back)
    N = 150
    replay.sample(B, DEV)                               # the thread notices the training loop within one of its steps ...
    replay.update_priority(torch.ones(B, device=DEV))
    time.sleep(0.25)
    s0 = steps()
    time.sleep(0.1)
    assert steps() == s0                                # ... spends that sample's credit and waits for the next one
    for _ in range(N):                                  # from then on: two steps per sample, no more, no fewer
        replay.sample(B, DEV)
        replay.update_priority(torch.ones(B, device=DEV))
    t0 = time.time()
    while steps() < s0 + 2 * N:
        assert time.time() - t0 < 0.9, (steps() - s0, N)
        time.sleep(0.002)
    time.sleep(0.05)
    s1 = steps()
    assert s1 - s0 == 2 * N, (s1 - s0, N)
    time.sleep(1.5)                                     # the training loop has stopped: free-running again
    assert steps() > s1 + 10
    context.set_pace(False)                             # the reference's unconditional free-running stays available
    b = steps()
    for _ in range(20):
        replay.sample(B, DEV)
        replay.update_priority(torch.ones(B, device=DEV))
        time.sleep(0.01)
    assert steps() - b > 2 * 20 + 10
    context.pause()
    context.terminate()


def test_action_matrix_tool_counts_like_the_reference_loop():
    """hanabi_sad_amd.action_matrix (pyhanabi/tools/action_matrix.py:31-107): whole greedy self-play games collected in VDN layout
    through rela / hanalearn, and the conditional action matrix computed over all sequences at once -- against the reference's
    per-step loop (its analyze(), restated here) on the same dataset"""
    from hanabi_sad_amd import action_matrix as am
    from hanabi_sad_amd.selfplay import init_weights
    W = init_weights(838, 128, 21, 5, 4)
    dataset, context = am.create_dataset(W, True, DEV, dataset_size=48, num_game=12, max_len=80, seed=3)
    assert dataset.size() == 48
    normed, counts = am.analyze(dataset)
    want = np.zeros((20, 20))
    n_steps = 0
    for i in range(dataset.size()):
        ep = dataset.get(i)
        action = ep.action["a"]
        assert action.shape[1] == 2
        for t in range(int(ep.seq_len.item()) - 1):
            a0, a1 = (int(action[t][0]), int(action[t + 1][1])) if t % 2 == 0 else (int(action[t][1]), int(action[t + 1][0]))
            assert a0 < 20 and a1 < 20                  # the mover never plays the noop
            want[a0][a1] += 1
            n_steps += 1
    assert n_steps > 48 * 5 and np.array_equal(counts, want)
    rows = want.sum(1) > 0
    assert np.allclose(normed[rows], want[rows] / want[rows].sum(1, keepdims=True))
    context.terminate()


def test_reference_shaped_eval_driver():
    """eval.py:25-66: one game per env, greedy actors, poll context.terminated(), read game.last_score()"""
    from hanabi_sad_amd import hanalearn, rela
    num_game, P = 40, 2
    games = create_envs(hanalearn, num_game, 11, P, 5, 0, [0.0], -1, True)
    agent = TinyAgent(games[0].feature_size(), 256, games[0].num_action(), 5, 5)
    runner = rela.BatchRunner(agent, DEV, 1000, ["act"])
    runner.start()
    context = rela.Context()
    env = hanalearn.HanabiVecEnv()
    for g in games:
        env.append(g)
    context.push_env_thread(hanalearn.HanabiThreadLoop([rela.R2D2Actor(runner, 1) for _ in range(P)], env, True))
    context.start()
    t0 = time.time()
    while not context.terminated():
        assert time.time() - t0 < 120
        time.sleep(0.05)
    context.terminate()
    runner.stop()
    scores = [g.last_score() for g in games]
    assert len(scores) == num_game and all(0 <= s <= 25 for s in scores)
    assert all(g.terminated() for g in games)


@pytest.mark.parametrize("cross_play", [False, True])
def test_eval_driver_with_one_loop_per_game_like_eval_py(cross_play):
    """pyhanabi/eval.py:25-66 verbatim in shape: ONE vector env + thread loop per game, one R2D2Actor per seat (its own runner:
    cross-play when the runners differ), context.start(), poll terminated(), game.last_score().  The context merges the
    per-game loops into one batched loop; self-play scores must equal hanabi_sad_amd.eval.evaluate on the same seeds."""
    import hanalearn
    import rela
    from hanabi_sad_amd.eval import evaluate
    num_game, P, seed = 48, 2, 2024
    games = create_envs(hanalearn, num_game, seed, P, 5, 0, [0.0], -1, True)
    agents = [TinyAgent(games[0].feature_size(), 64, games[0].num_action(), 5, 5 + (i if cross_play else 0)) for i in range(P)]
    runners = [rela.BatchRunner(ag, DEV, 1000, ["act"]) for ag in agents]
    context = rela.Context()
    loops = []
    for g in games:
        env = hanalearn.HanabiVecEnv()
        env.append(g)
        loops.append(hanalearn.HanabiThreadLoop([rela.R2D2Actor(runners[i], 1) for i in range(P)], env, True))
        context.push_env_thread(loops[-1])
    for r in runners:
        r.start()
    context.start()
    t0 = time.time()
    while not context.terminated():
        assert time.time() - t0 < 120
        time.sleep(0.05)
    context.terminate()
    for r in runners:
        r.stop()
    assert all(lp.master is loops[0] for lp in loops[1:]) and loops[0].env.G == num_game
    scores = [g.last_score() for g in games]
    assert all(g.terminated() for g in games) and all(0 <= s <= 25 for s in scores)
    if not cross_play:
        W = {k[len("online_net."):]: v for k, v in agents[0].state_dict().items() if k.startswith("online_net.")}
        _, _, want, _ = evaluate(W, num_game, seed, 0, True, device=DEV)
        assert scores == want
    else:
        assert not loops[0].same_model and len(loops[0].agents) == P


def _golden_batch(z):
    import rela
    t = lambda k: torch.tensor(z[k]).to(DEV)
    obs = {"priv_s": t("loss.priv_s"), "legal_move": t("loss.legal_move"), "own_hand": t("loss.own_hand")}
    T, B = obs["priv_s"].shape[:2]
    return rela.RNNTransition(obs, {"a": t("loss.a")}, t("loss.reward"), torch.zeros(T, B, device=DEV), t("loss.bootstrap"),
                              t("loss.seq_len")), t("loss.weight")


class _Meter:
    def __init__(self):
        self.v = []

    def feed(self, x):
        self.v.append(float(x))


@pytest.mark.parametrize("tag,pw", [("rl", 0.0), ("aux", 0.25)])
def test_the_reference_agent_api_trains_on_the_kernels(tag, pw):
    """VERDICT r4 missing 2: the learner half of the reference's driver.  `import r2d2` gives an nn.Module R2D2Agent with the reference's
    constructor; its parameters alias the library's flat weights under the reference's state_dict names; the train-loop body below is the
    reference's (pyhanabi/selfplay.py:128-149, 218-241) call for call: agent.loss(batch, pred_weight, stat) -> (loss * weight).mean() ->
    backward() -> clip_grad_norm_ -> torch.optim.Adam.step() -> zero_grad().  Loss, priorities and every gradient are compared with the
    golden vectors written from the reference agent (tests/golden/r2d2_iql_sad_small.npz) at the bf16 kernels' tolerances, and the Adam
    step with torch's own step on a copy of the parameters."""
    import os
    import r2d2
    from tests import r2d2_torch_ref as ref
    from tests.test_composite_abi_gpu import GOLD, maxerr, relerr
    z = np.load(os.path.join(GOLD, "r2d2_iql_sad_small.npz"))
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    in_dim, hid, out_dim = Won["net.0.weight"].shape[1], Won["fc_v.weight"].shape[1], Won["fc_a.weight"].shape[0]
    hand, n_layer = Won["pred.weight"].shape[0] // 3, len([k for k in Won if k.startswith("lstm.weight_ih_l")])
    agent = r2d2.R2D2Agent(False, int(z["meta"][8]), float(z["gamma"][0]), 0.9, DEV, in_dim, hid, out_dim, n_layer, hand, False)
    assert isinstance(agent, torch.nn.Module) and isinstance(agent.online_net, torch.nn.Module)
    sd = {"online_net." + k: v for k, v in Won.items()}
    sd.update({"target_net." + k: v for k, v in Wtg.items()})
    assert set(agent.state_dict().keys()) == set(sd.keys())
    agent.load_state_dict(sd)
    agent = agent.to(DEV)                                                  # the reference's driver does this; a no-op here
    for k, p in agent.online_net.named_parameters():                      # the Parameters ARE the library's weights
        assert p.data_ptr() == agent.online_net.cnet.w[k].data_ptr() and p.requires_grad
    assert not any(p.requires_grad for p in agent.target_net.parameters())
    optim = torch.optim.Adam(agent.online_net.parameters(), lr=1e-3, eps=1.5e-5)
    batch, weight = _golden_batch(z)
    stat = {"rl_loss": _Meter(), "aux1": _Meter()}
    loss, priority = agent.loss(batch, pw, stat)
    assert loss.requires_grad and not priority.requires_grad
    assert maxerr(loss, z["loss.%s.loss" % tag]) <= 1.2e-3 and maxerr(priority, z["loss.%s.priority" % tag]) <= 1.5e-3
    before = {k: p.detach().clone() for k, p in agent.online_net.named_parameters()}
    loss = (loss * weight).mean()
    loss.backward()
    for k, p in agent.online_net.named_parameters():
        want = torch.tensor(z["loss.%s.grad.%s" % (tag, k)])
        if want.abs().max() == 0:
            assert p.grad.abs().max() < 1e-6, k
        else:
            assert relerr(p.grad, want) <= 8.5e-3, (k, relerr(p.grad, want))
    g_norm = torch.nn.utils.clip_grad_norm_(agent.online_net.parameters(), 5.0)
    assert torch.isfinite(g_norm)
    grads = {k: p.grad.detach().clone() for k, p in agent.online_net.named_parameters()}
    optim.step()
    optim.zero_grad()
    # torch's Adam really stepped the library's weights (first step: p -= lr * g / (|g| + eps))
    for k, p in agent.online_net.named_parameters():
        want = before[k] - 1e-3 * grads[k] / (grads[k].abs() + 1.5e-5)
        assert torch.allclose(p.detach(), want, atol=2e-6), k
        assert p.grad is None
    assert len(stat["rl_loss"].v) == 1 and (pw == 0 or len(stat["aux1"].v) == 1)
    # ... and the next forward pass sees the new weights (the bf16 operands are re-derived lazily): a second loss differs, a third repeats it
    l2, _ = agent.loss(batch, pw, None)
    l3, _ = agent.loss(batch, pw, None)
    assert not torch.equal(l2.detach(), torch.tensor(z["loss.%s.loss" % tag]).to(DEV)) and torch.equal(l2.detach(), l3.detach())
    with pytest.raises(Exception, match="called again before this loss was backpropagated"):
        l2.mean().backward()             # the learner holds the activations of l3's forward pass: a stale backward is refused, not silently wrong
    l3.mean().backward()
    agent.sync_target_with_online()
    for (k, p), (_, q) in zip(agent.online_net.named_parameters(), agent.target_net.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), k
    with pytest.raises(Exception):
        agent.to("cpu")


def test_the_reference_agent_api_acts_and_clones_and_steps_with_the_fused_optimizer():
    """act / compute_priority on the reference's tensor contract ([1, E, ...] batched by rela.BatchRunner in the reference), clone(), and the
    library's fused clip + Adam + zero_grad as a torch.optim.Optimizer (HsadAdam) next to torch.optim.Adam on a clone: same parameters after
    three steps to fp32 rounding."""
    import os
    import r2d2
    from tests import r2d2_torch_ref as ref
    from tests.test_composite_abi_gpu import GOLD, maxerr
    z = np.load(os.path.join(GOLD, "r2d2_iql_sad_small.npz"))
    Won, Wtg = ref.weights_from_npz(z, "online_net."), ref.weights_from_npz(z, "target_net.")
    in_dim, hid, out_dim = Won["net.0.weight"].shape[1], Won["fc_v.weight"].shape[1], Won["fc_a.weight"].shape[0]
    hand, n_layer = Won["pred.weight"].shape[0] // 3, len([k for k in Won if k.startswith("lstm.weight_ih_l")])
    torch.manual_seed(5)
    a1 = r2d2.R2D2Agent(False, int(z["meta"][8]), float(z["gamma"][0]), 0.9, DEV, in_dim, hid, out_dim, n_layer, hand, False)
    sd = {"online_net." + k: v for k, v in Won.items()}
    sd.update({"target_net." + k: v for k, v in Wtg.items()})
    a1.load_state_dict(sd)
    a2 = a1.clone(DEV)
    for (k, p), (_, q) in zip(a1.state_dict().items(), a2.state_dict().items()):
        assert torch.equal(p, q) and p.data_ptr() != q.data_ptr(), k
    # act: the golden vectors (written from the reference agent) are already in the reference's [obsize, ibsize, ...] contract
    t = lambda k: torch.tensor(z[k])
    E = t("act.priv_s").shape[0]
    reply = a1.act({"priv_s": t("act.priv_s"), "legal_move": t("act.legal_move"), "eps": torch.zeros(E, 1), "h0": t("act.h0"), "c0": t("act.c0")})
    assert reply["a"].shape == (E, 1) and reply["a"].device.type == "cpu" and reply["h0"].shape == (E, 1, n_layer, hid)
    assert torch.equal(reply["a"], reply["greedy_a"])                                        # eps = 0
    agree = float((reply["greedy_a"].view(-1) == t("act.out_greedy_a").view(-1)).float().mean())
    assert agree >= 0.9, agree                                                              # (near-ties flip under bf16: test_r2d2_precision has the proofs)
    assert maxerr(reply["h0"], z["act.out_h0"]) <= 2e-2 and maxerr(reply["c0"], z["act.out_c0"]) <= 2e-2
    assert set(a1.get_h0(3).keys()) == {"h0", "c0"} and a1.get_h0(3)["h0"].shape == (n_layer, 3, hid)
    pin = {k[len("prio."):]: t(k) for k in z.files if k.startswith("prio.") and k != "prio.out"}
    pin.update({"priv_s": t("act.priv_s"), "legal_move": t("act.legal_move"), "h0": t("act.h0"), "c0": t("act.c0")})
    pr = a1.compute_priority(pin)["priority"]
    assert pr.shape == t("prio.out").shape and maxerr(pr, z["prio.out"]) <= 1.5e-3
    # three training steps: torch.optim.Adam + clip_grad_norm_ on a1, HsadAdam (clip inside) on a2
    batch, weight = _golden_batch(z)
    o1 = torch.optim.Adam(a1.online_net.parameters(), lr=1e-3, eps=1.5e-5)
    o2 = r2d2.HsadAdam(a2.online_net.parameters(), a2, lr=1e-3, eps=1.5e-5, max_grad_norm=5.0)
    for it in range(3):
        n1 = None
        for ag, opt in ((a1, o1), (a2, o2)):
            loss, prio = ag.loss(batch, 0.25, None)
            (loss * weight).mean().backward()
            if opt is o1:
                n1 = torch.nn.utils.clip_grad_norm_(ag.online_net.parameters(), 5.0)
            opt.step()
            opt.zero_grad()
        assert abs(float(o2.grad_norm) - float(n1)) <= 1e-3 * float(n1), (it, float(o2.grad_norm), float(n1))
    for (k, p), (_, q) in zip(a1.online_net.named_parameters(), a2.online_net.named_parameters()):
        assert torch.allclose(p.detach(), q.detach(), atol=5e-5), (k, float((p - q).abs().max()))
