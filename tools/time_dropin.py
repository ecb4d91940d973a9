"""The drop-in path, measured: a driver written against the reference's own API (pyhanabi/create.py:14-131 create_envs / create_threads,
selfplay.py:143-206) -- hanalearn.HanabiEnv per game, HanabiVecEnv + R2D2Actor + HanabiThreadLoop per "thread", rela.Context.start(),
the trainer polling replay.size() / sample() / update_priority() and calling runner.update_model() -- next to the native driver
(selfplay.Trainer) on the same configuration (2-player SAD IQL, H = 512, 2 LSTM layers, 16,384 games in the reference's launch shape
num_thread x num_game_per_thread = 64 x 256).

    python tools/time_dropin.py [games_per_thread] [threads] [seconds]

Prints: acting rate of the Context thread (acts / s = games x players x steps / s, utils.Tachometer's unit, pyhanabi/utils.py:229-236)
with the trainer idle, and with the real learner loop on the driver's thread (its own stream: the rollout issues on the Context's); then
the native loop's rate.  Compare the Speed line with the one `python -m hanabi_sad_amd.selfplay` prints for the same game count (rollout
and update interleaved on one stream)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hanabi_sad_amd                                  # registers the reference-named mirrors `hanalearn`, `rela`
from hanabi_sad_amd import hanalearn, rela
from hanabi_sad_amd.selfplay import Trainer, generate_explore_eps, init_weights, parse_args

GPT = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 64
SEC = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
DEV, P, T, NSTEP, GAMMA, ETA, SEED = "cuda:0", 2, 80, 3, 0.999, 0.9, 1
eps = generate_explore_eps(0.1, 7, 80)


class Agent:
    """what the reference hands to BatchRunner: an object whose state_dict() holds online_net.* / target_net.* of the R2D2Net shape"""

    def __init__(self, w):
        self.w = w

    def state_dict(self):
        d = {"online_net." + k: v for k, v in self.w.items()}
        d.update({"target_net." + k: v for k, v in self.w.items()})
        return d


probe = hanalearn.HanabiEnv({"players": str(P), "hand_size": "5", "seed": "1", "bomb": "0"}, eps, T, True, False, False, False)
F, A = probe.feature_size(), probe.num_action()
W = init_weights(F, 512, A, 5, SEED)
runner = rela.BatchRunner(Agent(W), DEV, 100, ["act", "compute_priority"])
assert runner.online is not None                       # the kernels, not the contract path
replay = rela.RNNPrioritizedReplay(65536, SEED, 0.9, 0.6, 3)
t0 = time.perf_counter()
ctx, loops, all_actors = rela.Context(), [], []
for th in range(NT):                                   # create.py:57-131: one vector env and one actor per player per thread
    venv = hanalearn.HanabiVecEnv()
    for g in range(GPT):
        venv.append(hanalearn.HanabiEnv({"players": str(P), "hand_size": "5", "seed": str(SEED + th * GPT + g), "bomb": "0"}, eps, T, True,
                                        False, False, False))
    actors = [rela.R2D2Actor(runner, NSTEP, GPT, GAMMA, ETA, T, 1, replay) for _ in range(P)]
    all_actors += actors
    loop = hanalearn.HanabiThreadLoop(actors, venv, False)
    loops.append(loop)
    ctx.push_env_thread(loop)
t_build = time.perf_counter() - t0
G = GPT * NT


def num_act():
    return sum(a.num_act() for a in all_actors)


ctx.start()
while replay.size() < 2048:                            # burn-in, selfplay.py:196-199
    time.sleep(0.05)
n0, t0 = num_act(), time.perf_counter()
time.sleep(SEC)
n1, t1 = num_act(), time.perf_counter()
idle_rate = (n1 - n0) / (t1 - t0)
# the trainer's side of selfplay.py:208-244 next to the running Context, on the driver's thread and stream: sample -> loss -> backward ->
# clip + Adam -> update_priority; actor model sync every 10 updates, target sync every 2,500 (the reference's defaults)
# -- with the reference's OWN agent calls (pyhanabi/selfplay.py:128-149, 218-241): `import r2d2` is the torch face over the kernels
import r2d2
agent = r2d2.R2D2Agent(False, NSTEP, GAMMA, ETA, DEV, F, 512, A, 2, 5, False)
sd0 = {"online_net." + k: v for k, v in W.items()}
sd0.update({"target_net." + k: v for k, v in W.items()})
agent.load_state_dict(sd0)
optim = torch.optim.Adam(agent.online_net.parameters(), lr=6.25e-5, eps=1.5e-5)


def one_update(u):
    if u % 2500 == 0:
        agent.sync_target_with_online()
    if u % 10 == 0:
        runner.update_model(agent)
    batch, weight = replay.sample(128, DEV)
    loss, priority = agent.loss(batch, 0.0, None)                 # priority [T, B] per step (r2d2.py:488-499)
    loss = (loss * weight).mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(agent.online_net.parameters(), 5.0)
    optim.step()
    optim.zero_grad()
    replay.update_priority(rela.aggregate_priority(priority, batch.seq_len, ETA))     # selfplay.py:236-240


results = []
# the FIRST line is the unchanged reference driver: no pacing call of any kind (the Context paces itself by the replay's sample() calls)
for label, stream, pace in (("UNCHANGED driver (no set_pace call: auto pace)  ", torch.cuda.Stream(DEV), "default"),
                            ("set_pace(False): the reference's free-running   ", torch.cuda.Stream(DEV), False),
                            ("free-running rollout, high-priority stream      ", torch.cuda.Stream(DEV, priority=-1), False),
                            ("Context.set_pace(replay, 1 step per sample)     ", torch.cuda.Stream(DEV), 1.0),
                            ("Context.set_pace(replay, 2 steps per sample)    ", torch.cuda.Stream(DEV), 2.0)):
    if pace == "default":
        pass                                            # what a driver written for the reference does: nothing
    elif pace is False:
        auto = ctx.auto_pace_steps
        ctx.set_pace(False)
    else:
        ctx.auto_pace_steps = auto
        ctx.set_pace(replay, pace)
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for u in range(20):
            one_update(u + 1)
        torch.cuda.synchronize()
        n0, t0, it = num_act(), time.perf_counter(), 0
        while time.perf_counter() - t0 < SEC:
            for _ in range(10):
                it += 1
                one_update(it)
        torch.cuda.synchronize()
        n1, t1 = num_act(), time.perf_counter()
    torch.cuda.current_stream().wait_stream(stream)
    results.append((label, it * 128 / (t1 - t0), (n1 - n0) / (t1 - t0), replay.size()))
agent._learner.check_sync()
ctx.pause()
ctx.terminate()
print("drop-in API (hanalearn / rela mirrors): %d threads x %d games = %d games built in %.2f s, merged into %d batched loop(s)"
      % (NT, GPT, G, t_build, sum(1 for l in loops if l.master is None)))
print("  Context thread alone        : %.2f M acts/s  (%.3f ms per step of all games)" % (idle_rate / 1e6, G * P / idle_rate * 1e3))
for label, train_rate, busy_rate, size in results:
    print("  + learner on the driver thread, %s: Speed: train: %.1f, act: %.1f, buffer_size: %d   (sequences/s, acts/s)" % (label, train_rate, busy_rate, size))
del ctx, loops, runner, replay, agent, optim
torch.cuda.empty_cache()
args = parse_args(["--num_game", str(G), "--replay_buffer_size", "65536", "--sad", "1"])
tr = Trainer(args, DEV)
for _ in range(120):
    tr.actor.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 300
for _ in range(K):
    tr.actor.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("native driver (selfplay.Trainer, hsad_actor_step): %.2f M acts/s  (%.3f ms per step)" % (G * P / dt / 1e6, dt * 1e3))
