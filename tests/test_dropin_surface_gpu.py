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
