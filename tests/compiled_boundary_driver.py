"""Driver for tests/test_compiled_boundary_gpu.py, run as a SUBPROCESS with the directory of the compiled modules first on sys.path: the calls
of pyhanabi/create.py:24-76,98-145, selfplay.py:152-245 and eval.py:25-66 in the reference's order, through the COMPILED `rela` / `hanalearn`
modules (bindings/*.cc built by __graft_entry__.build() into build/): real extension modules with a real .so __file__.
usage: python compiled_boundary_driver.py <build dir> <repo root> train-iql | train-vdn | eval"""
import sys
import time

build_dir, root, what = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
sys.path.insert(0, build_dir)          # `import rela` / `import hanalearn` resolve to the extension modules, not to the repository's mirror packages

import torch
import rela
import hanalearn

assert rela.__file__.endswith(".so") and hanalearn.__file__.endswith(".so") and build_dir in rela.__file__, (rela.__file__, hanalearn.__file__)
assert type(rela.Context).__module__ != "builtins" and "pybind11" in repr(type(rela.Context)), repr(type(rela.Context))
DEV = "cuda:0"


class TinyAgent:
    """stands in for the reference's torch R2D2Agent: what BatchRunner needs is state_dict() with online_net.* / target_net.*"""

    def __init__(self, in_dim, hid, out_dim, hand, seed):
        from hanabi_sad_amd.selfplay import init_weights
        W = init_weights(in_dim, hid, out_dim, hand, seed)
        self.sd = {"online_net." + k: v for k, v in W.items()}
        self.sd.update({"target_net." + k: v.clone() for k, v in W.items()})

    def state_dict(self):
        return self.sd


def create_envs(num_env, seed, num_player, hand_size, bomb, eps, max_len, sad):
    games = []
    for i in range(num_env):
        params = {"players": str(num_player), "hand_size": str(hand_size), "seed": str(seed + i), "bomb": str(bomb)}
        games.append(hanalearn.HanabiEnv(params, eps, max_len, sad, False, False, False))
    return games


def train(method):
    from hanabi_sad_amd.r2d2 import R2D2Learner
    from hanabi_sad_amd.selfplay import generate_explore_eps
    num_thread, per_thread, P, hand, n, gamma, eta, T, B = 2, 96, 2, 5, 3, 0.999, 0.9, 80, 32
    eps = generate_explore_eps(0.1, 7, 80)
    games = create_envs(num_thread * per_thread, 7, P, hand, 0, eps, T, True)
    F, A = games[0].feature_size(), games[0].num_action()
    assert (F, A) == (838, 21) and games[0].hand_feature_size() == 125      # HandSize x BitsPerCard (cpp/hanabi_env.h:66-72)
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
    del env, th, acts                                   # keep_alive: the Context holds its loops, a vector env its games
    context.start()
    t0 = time.time()
    while replay.size() < 4 * B:                       # burn in (selfplay.py:182-186); the Context's thread free-runs like the reference's
        assert time.time() - t0 < 120
        time.sleep(0.05)
    on = {k[len("online_net."):]: v for k, v in agent.state_dict().items() if k.startswith("online_net.")}
    learner = R2D2Learner(on, on, n, gamma, device=DEV)
    for it in range(4):                                 # train-loop body (selfplay.py:208-244)
        batch, weight = replay.sample(B, DEV)
        obs, act = batch.obs, batch.action
        assert obs["priv_s"].shape == (T, B, (P if method == "vdn" else 1) * F) and batch.terminal.dtype == torch.bool
        if method == "vdn":
            v4 = lambda x: x.view(x.shape[0], x.shape[1], P, -1)
            b = {"priv_s": v4(obs["priv_s"]), "legal_move": v4(obs["legal_move"]), "a": act["a"], "own_hand": v4(obs["own_hand"])}
        else:
            b = {"priv_s": obs["priv_s"], "legal_move": obs["legal_move"], "a": act["a"], "own_hand": obs["own_hand"]}
        b.update(reward=batch.reward, bootstrap=batch.bootstrap, seq_len=batch.seq_len)
        loss, priority = learner.loss(b, weight, 0.0)
        learner.optimizer_step()
        replay.update_priority(rela.aggregate_priority(priority, batch.seq_len, eta))
        assert torch.isfinite(loss).all()
        if it == 1:                                     # BatchRunner::updateModel (selfplay.py:239-241) while the loop runs
            runner.update_model(agent)
    context.pause()
    n_act = sum(a.num_act() for a in actors)
    n_again = sum(a.num_act() for a in actors)
    assert n_act > 0 and n_act == n_again                # parked between two steps
    assert replay.num_add() >= replay.size() > 0
    context.resume()
    time.sleep(0.05)
    context.pause()
    assert sum(a.num_act() for a in actors) > n_act
    context.terminate()
    assert context.terminated()
    # the per-thread loops were merged into ONE batched device loop (consecutive seeds, same models): one launch per kernel
    assert threads[1].merged() and not threads[0].merged() and threads[0].num_games() == num_thread * per_thread
    threads[0].check_errors()
    per_actor = [a.num_act() for a in actors]
    assert len(set(per_actor)) == 1 and per_actor[0] % per_thread == 0          # R2D2Actor::numAct_ += num_envs per act()
    assert all(0 <= g.get_score() <= 25 and 0 <= g.get_life() <= 3 for g in games[:8])
    print("compiled boundary: %s training driver OK, %d acts, replay %d / %d added" % (method, n_act, replay.size(), replay.num_add()))


def evaluate():
    from hanabi_sad_amd.eval import evaluate as py_evaluate
    num_game, P, seed = 48, 2, 2024
    games = create_envs(num_game, seed, P, 5, 0, [0.0], -1, True)
    agent = TinyAgent(games[0].feature_size(), 64, games[0].num_action(), 5, 5)
    runner = rela.BatchRunner(agent, DEV, 1000, ["act"])
    runner.start()
    context = rela.Context()
    loops = []
    for g in games:                                      # eval.py:40-47: ONE vector env + thread loop per game, one actor per seat
        env = hanalearn.HanabiVecEnv()
        env.append(g)
        loops.append(hanalearn.HanabiThreadLoop([rela.R2D2Actor(runner, 1) for _ in range(P)], env, True))
        context.push_env_thread(loops[-1])
    context.start()
    t0 = time.time()
    while not context.terminated():
        assert time.time() - t0 < 120
        time.sleep(0.05)
    context.terminate()
    runner.stop()
    assert all(lp.merged() for lp in loops[1:]) and loops[0].num_games() == num_game
    scores = [g.last_score() for g in games]
    assert all(g.terminated() for g in games) and all(0 <= s <= 25 for s in scores)
    W = {k[len("online_net."):]: v for k, v in agent.state_dict().items() if k.startswith("online_net.")}
    _, _, want, _ = py_evaluate(W, num_game, seed, 0, True, device=DEV)      # the library's own batched evaluation on the same seeds
    assert scores == want, (scores, want)
    print("compiled boundary: eval driver OK, mean score %.2f" % (sum(scores) / len(scores)))


if what == "train-iql":
    train("iql")
elif what == "train-vdn":
    train("vdn")
else:
    evaluate()
