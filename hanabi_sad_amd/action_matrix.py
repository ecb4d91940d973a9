"""The conditional action matrix of a trained agent, P(partner's next action | my action) -- the analysis of
pyhanabi/tools/action_matrix.py:31-107 on this stack: a dataset of whole self-play games in VDN layout (both players' actions side
by side, create_dataset there) collected through the reference-named `rela` / `hanalearn` surface on the batched device loop, and the
count matrix of analyze() computed over all sequences at once instead of an .item() per step.

    python -m hanabi_sad_amd.action_matrix --weight model.pthw --save_fig matrix.png [--save_npy matrix.npy]

Rows / columns follow the move-uid order D1-5 P1-5 C1-5 R1-5 (tools/action_matrix.py:110-131)."""
import argparse
import time

import numpy as np
import torch

IDX2ACTION = ["D1", "D2", "D3", "D4", "D5", "P1", "P2", "P3", "P4", "P5", "C1", "C2", "C3", "C4", "C5", "R1", "R2", "R3", "R4", "R5"]


def create_dataset(weights, sad, device="cuda:0", dataset_size=1000, num_game=100, max_len=80, seed=1):
    """weights: an R2D2Net state_dict (`online_net.`-less keys).  -> (rela.RNNPrioritizedReplay holding `dataset_size` greedy self-play
    games, the paused rela.Context): uniform sampling (priority exponent 0), eps = 0, one VDN transition per game so that the two
    players' trajectories stay together (tools/action_matrix.py:31-88)"""
    from . import hanalearn, rela

    class _Agent:
        def state_dict(self):
            d = {"online_net." + k: v for k, v in weights.items()}
            d.update({"target_net." + k: v for k, v in weights.items()})
            return d
    runner = rela.BatchRunner(_Agent(), device, 100, ["act", "compute_priority"])
    replay = rela.RNNPrioritizedReplay(dataset_size, 1, 0.0, 1.0, 0)
    games = [hanalearn.HanabiEnv({"players": "2", "hand_size": "5", "seed": str(seed + g), "bomb": "0"}, [0.0], max_len, bool(sad), False,
                                 False, False) for g in range(num_game)]
    context = rela.Context()
    for g in games:                                     # one game per "thread", like the reference: Context merges them into one loop
        env = hanalearn.HanabiVecEnv()
        env.append(g)
        actor = rela.R2D2Actor(runner, 1, 1, 0.99, 0.9, max_len, 2, replay)
        context.push_env_thread(hanalearn.HanabiThreadLoop(actor, env, False))
    runner.start()
    context.start()
    while replay.size() < dataset_size:
        time.sleep(0.05)
    context.pause()
    for _ in range(2):                                  # the buffer holds more than its capacity until a draw evicts the oldest
        _, w = replay.sample(min(10, dataset_size), device)       # (tools/action_matrix.py:78-82, prioritized_replay.h:291-299)
        replay.update_priority(w.detach())
    return replay, context


def analyze_sequences(action, seq_len, num_action=20):
    """action int64 [T, B, 2] (player 0 moves on even steps, the idle player's entry is the noop), seq_len [B] -> (row-normalised
    matrix, counts) [num_action, num_action]: counts[a, b] = number of times a move a was followed by the partner's move b
    (tools/action_matrix.py:91-107)"""
    T, B, _ = action.shape
    t = torch.arange(T - 1, device=action.device).view(-1, 1)
    mover = (t % 2).expand(T - 1, B)                                          # who moves at step t
    a0 = action[:-1].gather(2, mover.unsqueeze(2)).squeeze(2)
    a1 = action[1:].gather(2, (1 - mover).unsqueeze(2)).squeeze(2)
    valid = t < (seq_len.view(1, -1).long() - 1)
    valid &= (a0 < num_action) & (a1 < num_action)
    flat = (a0 * num_action + a1)[valid]
    counts = torch.bincount(flat, minlength=num_action * num_action).view(num_action, num_action).double()
    return (counts / counts.sum(1, keepdim=True)).cpu().numpy(), counts.cpu().numpy()


def analyze(dataset, batch=256):
    """every stored game of an rela.RNNPrioritizedReplay (what create_dataset returns)"""
    n = dataset.size()
    acts, lens = [], []
    for i in range(n):
        ep = dataset.get(i)
        acts.append(ep.action["a"].view(ep.action["a"].shape[0], 1, -1))
        lens.append(ep.seq_len.view(1))
    return analyze_sequences(torch.cat(acts, 1), torch.cat(lens))


def plot(mat, title, savefig):
    import matplotlib
    matplotlib.use("agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(figsize=(8, 8))
    ax.matshow(mat)
    ax.set_title(title)
    ax.set_xticks(range(20))
    ax.set_xticklabels(IDX2ACTION)
    ax.set_yticks(range(20))
    ax.set_yticklabels(IDX2ACTION)
    plt.tight_layout()
    plt.savefig(savefig)


def main(argv=None):
    from .checkpoint import load_weights
    p = argparse.ArgumentParser(description="conditional action matrix of a 2-player agent (pyhanabi/tools/action_matrix.py)")
    p.add_argument("--weight", required=True, type=str, help=".pthw file of an R2D2 agent (online_net.* keys or a bare net)")
    p.add_argument("--sad", type=int, default=-1, help="-1: from the file name like the reference ('sad' or 'aux' in it)")
    p.add_argument("--save_fig", type=str, default="")
    p.add_argument("--save_npy", type=str, default="")
    p.add_argument("--device", type=str, default="cuda:0")
    args = p.parse_args(argv)
    w = load_weights(args.weight, args.device)       # strips an agent file's online_net. prefix itself
    fname = args.weight.split("/")[-1]
    sad = ("sad" in fname or "aux" in fname) if args.sad < 0 else bool(args.sad)
    dataset, context = create_dataset(w, sad, args.device)
    normed, counts = analyze(dataset)
    context.terminate()
    np.set_printoptions(precision=2, suppress=True, linewidth=250)
    print(normed)
    if args.save_npy:
        np.save(args.save_npy, normed)
    if args.save_fig:
        plot(normed, "action_matrix", args.save_fig)


if __name__ == "__main__":
    main()
