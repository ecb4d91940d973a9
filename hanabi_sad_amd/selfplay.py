"""selfplay-style driver on the device pipeline (reference pyhanabi/selfplay.py:89-281, IQL path):
build env / agent / replay / actor, burn in, then alternate rollout and learner updates.

    python -m hanabi_sad_amd.selfplay --num_game 4096 --num_update 200

Single GPU: actors and learner share the device and alternate in lock-step.  Multi-GPU
(torchrun, one rank per GPU): every rank rolls out its shard of games into its own replay shard; rank 0 is
the learner: its batches are assembled from ALL shards (stratified over their concatenation), the new priorities go
back to the owning shards, and the online/target parameters are broadcast over RCCL every --actor_sync_freq updates
 — see hanabi_sad_amd/dist.py."""
import argparse
import contextlib
import os
import time

import numpy as np
import torch

from .actor import DeviceActor, transition_fields
from .env import BatchedHanabiEnv
from .r2d2 import PARAM_ORDER, R2D2Agent, R2D2Learner, R2D2NetKernels, arch_of, check_sync, param_order
from .replay import DeviceReplay, aggregate_priority


def generate_explore_eps(base_eps, alpha, num_env):
    """pyhanabi/utils.py:367-379"""
    if num_env == 1:
        return [0.0 if base_eps < 1e-6 else base_eps]
    out = []
    for i in range(num_env):
        e = base_eps ** (1 + i / (num_env - 1) * alpha)
        out.append(0.0 if e < 1e-6 else e)
    return out


def init_weights(in_dim, hid_dim, out_dim, hand_size, seed, num_lstm_layer=2, num_fc_layer=1):
    """random-init R2D2Net parameters with nn.Linear / nn.LSTM's default U(-1/sqrt(fan), 1/sqrt(fan)) init,
    keyed like the reference state_dict (so `.pthw` checkpoints are interchangeable)"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = lambda shape, fan: (torch.rand(*shape, generator=g) * 2 - 1) / fan ** 0.5
    H = hid_dim
    W = {"net.0.weight": u((H, in_dim), in_dim), "net.0.bias": u((H,), in_dim),
         "fc_v.weight": u((1, H), H), "fc_v.bias": u((1,), H), "fc_a.weight": u((out_dim, H), H),
         "fc_a.bias": u((out_dim,), H), "pred.weight": u((hand_size * 3, H), H), "pred.bias": u((hand_size * 3,), H)}
    if num_fc_layer == 2:
        W["net.2.weight"], W["net.2.bias"] = u((H, H), H), u((H,), H)
    for l in range(num_lstm_layer):
        for k in ("weight_ih", "weight_hh"):
            W["lstm.%s_l%d" % (k, l)] = u((4 * H, H), H)
        for k in ("bias_ih", "bias_hh"):
            W["lstm.%s_l%d" % (k, l)] = u((4 * H,), H)
    return W


class Trainer:
    def __init__(self, args, device="cuda:0", rank=0, world=1):
        self.args, self.device, self.rank, self.world = args, torch.device(device), rank, world
        eps = generate_explore_eps(args.act_base_eps, args.act_eps_alpha, args.num_eps)
        # several ranks: rank 0 is a dedicated learner (the reference's training thread owns its GPU, too); its env only provides the
        # dimensions, and its replay shard stays empty
        self.acting = world == 1 or rank != 0
        self.env = BatchedHanabiEnv(args.num_game if self.acting else 1, players=args.num_player, hand_size=args.hand_size,
                                    seed=args.seed + rank * args.num_game, bomb=args.train_bomb, eps_list=eps,
                                    max_len=args.max_len, sad=bool(args.sad), shuffle_color=bool(args.shuffle_color),
                                    device=device, track_deck_history=False)
        W = init_weights(self.env.F, args.rnn_hid_dim, self.env.A, args.hand_size, args.seed, num_lstm_layer=args.num_lstm_layer)
        self.param_names = list(param_order(1, args.num_lstm_layer))
        # only the learner rank holds optimizer state; actor ranks receive parameters by broadcast
        # bf16 (production): agent and learner are the library's composite entry points (include/hsad.h hsad_r2d2_*: the whole
        # kernel schedule behind one C call each); fp32 (exact mode): the same schedule orchestrated from r2d2.py / r2d2_f32.py
        composite = args.precision == "bf16" and not getattr(args, "python_schedule", 0)
        if composite:
            from .composite import CNet, CompositeAgent, CompositeLearner
            self.learner = CompositeLearner(W, W, args.multi_step, args.gamma, lr=args.lr, eps=args.eps, grad_clip=args.grad_clip,
                                            device=device) if rank == 0 else None
            self.act_online, self.act_target = CNet(W, device), CNet(W, device)
            self.agent = CompositeAgent(self.act_online, self.act_target, args.multi_step, args.gamma, seed=args.seed + 17 * rank)
        else:
            self.learner = R2D2Learner(W, W, args.multi_step, args.gamma, lr=args.lr, eps=args.eps, grad_clip=args.grad_clip,
                                       device=device, precision=args.precision) if rank == 0 else None
            # the actors run their own copies of the agent, refreshed every actor_sync_freq updates
            # (ActGroup.update_model / BatchRunner::updateModel, create.py:143-145)
            self.act_online = R2D2NetKernels.make(W, device, args.precision)
            self.act_target = R2D2NetKernels.make(W, device, args.precision)
            self.agent = R2D2Agent(self.act_online, self.act_target, args.multi_step, args.gamma, seed=args.seed + 17 * rank)
        self.vdn = args.method == "vdn"
        fields = transition_fields(self.env, self.vdn)
        # the reference's capacity is split evenly over the per-GPU shards
        n_shards = max(world - 1, 1)
        self.replay = DeviceReplay(max(args.batchsize, args.replay_buffer_size // n_shards if self.acting else 0), args.seed + rank,
                                   args.priority_exponent, args.priority_weight, args.prefetch, args.max_len, fields, device)
        # the learner's first GEMM reads the observation as zero-padded bf16 rows: have the sampler expand the stored bits into
        # exactly that (no float32 batch, no cast pass)
        self.bf16_batch = composite and self.env.knowledge_mode == 0
        if self.bf16_batch:
            self.replay.set_field_output("priv_s", "bf16", self.act_online.Fp)
        from .dist import ShardedReplay
        self.sharded = ShardedReplay(self.replay, args.priority_weight, device, learner_rank=0) if world == 1 else None
        self.actor = DeviceActor(self.env, self.agent, self.replay, args.multi_step, args.gamma, args.eta, args.max_len,
                                 vdn=self.vdn, native=None if getattr(args, "native_actor", 1) else False) if self.acting else None
        self.num_update = 0
        self._drawn = None           # the batch of the next update, drawn ahead (--draw_ahead)
        self._drawn_ev = None        # ... and the event behind that draw when it was issued on the draw stream
        self.draw_stream = None      # (set below)
        # --overlap_rollout: the rollout issues on a stream of its own, next to the update on the caller's stream (one host thread feeds
        # both; replay and writer calls from the two sides are ordered by the library's stream fence).  The reference's actor threads
        # run concurrently with its training thread the same way (selfplay.py:208-244, rela/context.h:43-50).
        self.act_stream = None
        if self.acting and self.learner is not None and getattr(args, "overlap_rollout", 0) and torch.device(device).type == "cuda":
            self.act_stream = torch.cuda.Stream(torch.device(device))
            self.act_stream.wait_stream(torch.cuda.current_stream(torch.device(device)))     # everything built so far precedes the first step
        # --draw_ahead with the composite learner: priority write-back and the next draw are issued BETWEEN the two halves of the update, on a
        # stream of their own that waits for the forward half only.  Behind the optimizer step on the caller's stream (round 5) the chain of small
        # launches -- priorities, draw (one workgroup, 65 us), row unpacking -- sat between two updates: 0.15 ms in front of every forward launch.
        # Replay calls from the three streams are ordered by the library's stream fence in the order they are issued, which is unchanged:
        # rollout step's flush -> priority write-back -> draw -> next rollout step's flush.
        if (self.learner is not None and composite and getattr(args, "draw_ahead", 0) and getattr(args, "early_draw", 1)
                and self.sharded is not None and torch.device(device).type == "cuda"):
            self.draw_stream = torch.cuda.Stream(torch.device(device))
            self.draw_stream.wait_stream(torch.cuda.current_stream(torch.device(device)))

    def act_step(self, n=1):
        """n rollout steps, on the rollout stream when there is one"""
        if self.act_stream is None:
            for _ in range(n):
                self.actor.step()
            return
        with torch.cuda.stream(self.act_stream):
            for _ in range(n):
                self.actor.step()

    def join_rollout(self):
        """the caller's stream waits for the rollout issued so far (before evaluation, checkpoints, error checks)"""
        if self.act_stream is not None:
            torch.cuda.current_stream(self.act_stream.device).wait_stream(self.act_stream)

    def update_actor_model(self):
        if self.act_stream is not None:      # behind the optimizer step that wrote the weights, between two rollout steps
            self.act_stream.wait_stream(torch.cuda.current_stream(self.act_stream.device))
        with (torch.cuda.stream(self.act_stream) if self.act_stream is not None else contextlib.nullcontext()):
            if self.learner is not None:
                for k in self.param_names:
                    self.act_online.w[k].copy_(self.learner.online.w[k])
                    self.act_target.w[k].copy_(self.learner.target.w[k])
            self.act_online.refresh()
            self.act_target.refresh()
        self.join_rollout()                  # the next update overwrites what the copy reads

    def learner_update(self, stopwatch=None):
        """one learner iteration (selfplay.py:208-244) of a single-GPU job (several GPUs: run_link_learner / run_link_actor below,
        where the actor ranks never wait for this).  stopwatch: a common.Stopwatch that receives the reference's five sections, each closed by a
        device synchronisation like there (selfplay.py:215-241) -- off by default: the fences cost throughput."""
        a = self.args

        def mark(key, sync=True):
            if stopwatch is not None:
                if sync:
                    torch.cuda.synchronize()
                stopwatch.time(key)
        if self.learner is not None and self.num_update % a.num_update_between_sync == 0:
            self.learner.sync_target_with_online()
        if self.num_update % a.actor_sync_freq == 0:
            self.update_actor_model()
        mark("sync and updating")
        # --draw_ahead 1 (default): this update's batch was drawn at the end of the previous one -- behind its priority write-back, in front
        # of the rollout step issued since -- so that the draw (a chain of small dependent launches, 0.1 ms) runs next to that step instead
        # of waiting for its flush.  The reference's sampler thread draws `prefetch` batches ahead of the training thread the same way
        # (rela/prioritized_replay.h:229-237); 0 = draw here (the strictly alternating order of rounds 1-4).
        res, self._drawn = self._drawn, None
        if res is None:
            res = self.sharded.sample(a.batchsize)
        elif self._drawn_ev is not None:          # drawn on the draw stream: the update reads it behind that
            torch.cuda.current_stream(self._drawn_ev_dev).wait_event(self._drawn_ev)
            self._drawn_ev = None
        mark("sample data", sync=False)
        batch, weight, seq_len = self.batch_of(res)
        if self.draw_stream is not None:
            dev = self.draw_stream.device
            main = torch.cuda.current_stream(dev)

            def between(loss_, priority_):
                ev = torch.cuda.Event()
                ev.record(main)                   # behind the forward half: loss and priorities are final
                self.draw_stream.wait_event(ev)
                with torch.cuda.stream(self.draw_stream):
                    prio = aggregate_priority(priority_, seq_len, a.eta)
                    self.sharded.update_priority(prio)
                    self._drawn = self.sharded.sample(a.batchsize)
                    self._drawn_ev, self._drawn_ev_dev = torch.cuda.Event(), dev
                    self._drawn_ev.record(self.draw_stream)
                # allocator bookkeeping: each side's blocks are in use on the other stream too
                priority_.record_stream(self.draw_stream)
                seq_len.record_stream(self.draw_stream)
                for t in _tensors_of(self._drawn):
                    t.record_stream(main)
            loss, priority = self.learner.loss(batch, weight, a.pred_weight, between=between)
            mark("forward & backward")
            g_norm = self.learner.optimizer_step()
            mark("update model")
            mark("updating priority", sync=False)
            self.num_update += 1
            return (loss * weight).mean(), g_norm
        loss, priority = self.learner.loss(batch, weight, a.pred_weight)
        prio = aggregate_priority(priority, seq_len, a.eta)
        mark("forward & backward")
        g_norm = self.learner.optimizer_step()
        mark("update model")
        self.sharded.update_priority(prio)
        if getattr(a, "draw_ahead", 0):
            self._drawn = self.sharded.sample(a.batchsize)
        mark("updating priority", sync=False)
        self.num_update += 1
        return (loss * weight).mean(), g_norm

    def batch_of(self, res):
        """a sampled / assembled batch -> the learner's input dict (selfplay.py:222-224 + the VDN reshape), weight, seq_len"""
        (f, reward, terminal, bootstrap, seq_len), weight = res
        pk = "priv_s_bf16" if self.bf16_batch else "priv_s"       # bf16: [T, B, P | 1, in_dim_padded] as sampled
        if self.vdn:    # [T, B, P*w] -> [T, B, P, w]
            P_, v4 = self.env.P, lambda t: t.view(t.shape[0], t.shape[1], self.env.P, -1)
            batch = {pk: v4(f["priv_s"]), "legal_move": v4(f["legal_move"]), "a": f["a"], "reward": reward,
                     "bootstrap": bootstrap, "seq_len": seq_len, "own_hand": v4(f["own_hand"])}
        else:
            batch = {pk: f["priv_s"], "legal_move": f["legal_move"], "a": f["a"].squeeze(2), "reward": reward,
                     "bootstrap": bootstrap, "seq_len": seq_len, "own_hand": f["own_hand"]}
        return batch, weight, seq_len


def _tensors_of(x):
    """every tensor inside a nested tuple / list / dict (a sampled batch)"""
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _tensors_of(v)
    elif isinstance(x, (tuple, list)):
        for v in x:
            yield from _tensors_of(v)


def parse_args(argv=None):
    """flags of pyhanabi/selfplay.py:26-86 (same names; defaults of the formal run scripts tools/*.sh where the two differ),
    plus the few this stack adds (--num_game, --act_steps_per_update, --num_update, --precision, --stopwatch, --dist_backend)"""
    p = argparse.ArgumentParser(description="R2D2 self-play on the device pipeline (flags as in pyhanabi/selfplay.py)")
    p.add_argument("--save_dir", type=str, default="", help="checkpoints + train.log go here; empty = do not save / tee")
    p.add_argument("--method", type=str, default="iql", choices=["iql", "vdn"])
    p.add_argument("--shuffle_obs", type=int, default=0)
    p.add_argument("--shuffle_color", type=int, default=0)
    p.add_argument("--pred_weight", type=float, default=0.0)
    p.add_argument("--num_eps", type=int, default=80)
    p.add_argument("--load_model", type=str, default="")
    p.add_argument("--seed", type=int, default=10001)
    p.add_argument("--gamma", type=float, default=0.999)
    p.add_argument("--eta", type=float, default=0.9)
    p.add_argument("--train_bomb", type=int, default=0)
    p.add_argument("--eval_bomb", type=int, default=0)
    p.add_argument("--sad", type=int, default=1)
    p.add_argument("--num_player", type=int, default=2)
    p.add_argument("--hand_size", type=int, default=5)
    p.add_argument("--lr", type=float, default=6.25e-5)
    p.add_argument("--eps", type=float, default=1.5e-5)
    p.add_argument("--grad_clip", type=float, default=5.0)
    p.add_argument("--num_lstm_layer", type=int, default=2)
    p.add_argument("--rnn_hid_dim", type=int, default=512)
    p.add_argument("--train_device", type=str, default="cuda:0", help="accepted for compatibility: rank r trains / acts on its own GPU")
    p.add_argument("--act_device", type=str, default="", help="accepted for compatibility (see --train_device)")
    p.add_argument("--batchsize", type=int, default=128)
    p.add_argument("--num_epoch", type=int, default=0, help="epochs of --epoch_len updates, each followed by Tachometer / "
                   "statistics / evaluation / top-k save; 0 = the plain --num_update loop")
    p.add_argument("--epoch_len", type=int, default=1000)
    p.add_argument("--num_update", type=int, default=100, help="updates of the plain loop (when --num_epoch 0)")
    p.add_argument("--num_update_between_sync", type=int, default=2500)
    p.add_argument("--multi_step", type=int, default=3)
    p.add_argument("--burn_in_frames", type=int, default=2000)
    p.add_argument("--replay_buffer_size", type=int, default=32768)
    p.add_argument("--priority_exponent", type=float, default=0.9)
    p.add_argument("--priority_weight", type=float, default=0.6)
    p.add_argument("--max_len", type=int, default=80)
    p.add_argument("--prefetch", type=int, default=3)
    p.add_argument("--num_thread", type=int, default=0, help="with --num_game_per_thread: concurrent games = their product")
    p.add_argument("--num_game_per_thread", type=int, default=0)
    p.add_argument("--num_game", type=int, default=4096, help="concurrent games on this GPU (num_thread*num_game_per_thread)")
    p.add_argument("--act_base_eps", type=float, default=0.1)
    p.add_argument("--act_eps_alpha", type=float, default=7)
    p.add_argument("--actor_sync_freq", type=int, default=10)
    p.add_argument("--act_steps_per_update", type=int, default=1)
    p.add_argument("--num_eval_game", type=int, default=1000)
    p.add_argument("--early_draw", type=int, default=1, help="with --draw_ahead and the composite learner: priority write-back and the next draw are "
                   "issued between the forward half and the BPTT of an update on a stream of their own (0: behind the optimizer step on the caller's stream)")
    p.add_argument("--draw_ahead", type=int, default=1, help="1: the batch of update u + 1 is drawn at the end of update u (behind its priority "
                   "write-back), in front of the next rollout step, as the reference's prefetching sampler thread does; 0: at the start of update u + 1")
    p.add_argument("--overlap_rollout", type=int, default=1, help="1: the rollout steps of a one-GPU job issue on a stream of their own, "
                   "next to the update (like the reference's actor threads next to its training thread); 0: one stream, strictly alternating")
    p.add_argument("--stopwatch", type=int, default=0, help="1 = time the reference's five learner sections (adds device syncs)")
    p.add_argument("--native_actor", type=int, default=1,
                   help="1: the actor-loop body is the library's hsad_actor_step (one C call per step); 0: the same body in Python "
                   "(actor.DeviceActor.step; the priority cross-check of the tests and contract models use it)")
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"],
                   help="bf16 = production kernels (bf16 MFMA operands, fp32 accumulate/state); fp32 = the exact mode (the "
                        "reference's arithmetic type, for validation: ~20x slower)")
    p.add_argument("--python_schedule", type=int, default=0, help="1 = drive the bf16 kernels from r2d2.py instead of the "
                   "library's composite entry points (same kernels, same order; for A/B tests)")
    p.add_argument("--rounds_ahead", type=int, default=3, help="multi-GPU: how many exchange rounds the learner keeps open (a batch is requested "
                   "this many updates before it is trained on; the reference's replay prefetch, selfplay.py prefetch = 3)")
    p.add_argument("--dist_backend", type=str, default="nccl", help="nccl (= RCCL, one GPU per rank) | gloo (smoke runs "
                   "with several ranks sharing a GPU: tensors are staged through host memory)")
    args = p.parse_args(argv)
    if args.num_thread > 0 and args.num_game_per_thread > 0:
        args.num_game = args.num_thread * args.num_game_per_thread
    if args.shuffle_obs:
        raise SystemExit("--shuffle_obs is not supported (selfplay.py:175 asserts it off)")
    if not 1 <= args.num_lstm_layer <= 3:
        raise SystemExit("--num_lstm_layer: 1, 2 or 3 (nn.LSTM(num_layers), pyhanabi/selfplay.py:50)")
    if args.num_lstm_layer != 2 and getattr(args, "python_schedule", 0):
        raise SystemExit("--python_schedule 1 (the A/B twin of the composite entry points) is written for 2 LSTM layers")
    return args


# ---- several GPUs: a dedicated learner rank and free-running actor ranks joined by dist.ReplayLink -------------------------------
def make_link(tr, args):
    from .dist import ReplayLink
    n = tr.act_online.flat.numel() if hasattr(tr.act_online, "flat") else sum(v.numel() for v in tr.act_online.w.values())
    # rounds are opened `rounds_ahead` updates before their batch is trained on (the reference's sampler prefetches 3 batches the same way:
    # selfplay.py prefetch = 3, rela/prioritized_replay.h:229-237): an actor answers a round 1.7-2.6 ms after it was opened, an update takes 1.4
    return ReplayLink(tr.replay, args.batchsize, args.priority_weight, tr.device, learner_rank=0, param_numel=2 * n,
                      ahead=max(1, int(getattr(args, "rounds_ahead", 3))))


def _flat_of(net):
    return net.flat if hasattr(net, "flat") else torch.cat([net.w[k].reshape(-1) for k in param_order(*arch_of(net.w))])


def _load_flat(net, flat):
    if hasattr(net, "flat"):
        net.flat.copy_(flat)
    else:
        off = 0
        for k in param_order(*arch_of(net.w)):
            n = net.w[k].numel()
            net.w[k].copy_(flat[off:off + n].view_as(net.w[k]))
            off += n
    net.refresh()


def run_link_actor(tr, args, link):
    """an actor rank: burn its shard in, then step its games for as long as the learner runs -- between two steps it asks the store
    whether a round is open and serves it from its own stream (statistics, its share of the draw, late priorities, sometimes new
    parameters).  It never waits for the learner's compute.  (The reference's actor threads free-run the same way while the
    training thread trains: rela/context.h:43-50, selfplay.py:208-244.)"""
    share = max(args.batchsize, args.burn_in_frames // max(link.world - 1, 1))
    announced = False
    tr.actor.set_run_ahead(2)      # a round is served behind at most two queued steps (the host issues ~10x faster than the device runs;
                                   # two is the smallest bound that never lets the device run dry)
    n = tr.act_online.flat.numel() if hasattr(tr.act_online, "flat") else link.bucket.numel() // 2
    while True:
        tr.actor.step()
        if not announced and tr.actor.num_act % (50 * tr.actor.N) == 0 and tr.replay.size() >= share:
            link.store.set("hsad/link/acts/%d" % link.rank, "%d %d %d" % (tr.actor.num_act, tr.replay.num_add(), tr.replay.size()))
            link.store.add("hsad/link/burned_in", 1)
            announced = True
        flags = link.poll()
        if flags is None:
            continue
        stop = link.serve(flags)
        if link.served % 64 == 0 or stop:     # Tachometer input for the learner (replay counters are read from the device: rarely)
            link.store.set("hsad/link/acts/%d" % link.rank, "%d %d %d" % (tr.actor.num_act, tr.replay.num_add(), tr.replay.size()))
        if flags & link.PARAMS:          # BatchRunner::updateModel: the learner's snapshot [online | target], stream-ordered
            _load_flat(tr.act_online, link.bucket[:n])
            _load_flat(tr.act_target, link.bucket[n:])
        if stop:
            break
    torch.cuda.synchronize()
    tr.env.check_errors()
    tr.replay.check_errors()


def run_link_learner(tr, args, link, num_update, on_update=None, stop=True):
    """the learner rank: update u runs on the compute stream while round u+1 (the draw of batch u+1, the priorities of update u-1,
    sometimes the parameters) runs on the exchange stream; neither side's host waits for the other's GPU"""
    import time as _time
    a, L = args, tr.learner
    while int(link.store.add("hsad/link/burned_in", 0)) < link.world - 1:     # every actor shard holds its share of the burn-in
        _time.sleep(0.05)
    first = tr.num_update == 0
    if first:
        link.stage_params(_flat_of(L.online), _flat_of(L.target))
        for i in range(link.ahead):               # the batches of the first `ahead` updates are requested up front
            link.begin(None, params=(i == 0))
        tr._cur, tr._prios = link.finish(), []
    for u in range(num_update):
        if tr.num_update % a.num_update_between_sync == 0:
            L.sync_target_with_online()
        params = tr.num_update % a.actor_sync_freq == 0 and tr.num_update > 0
        if params:
            link.stage_params(_flat_of(L.online), _flat_of(L.target))
        link.begin(tr._prios.pop(0) if tr._prios else None, params=params)
        batch, weight, seq_len = tr.batch_of(tr._cur)
        if os.environ.get("HSAD_LINK_DEBUG"):
            bad = [k for k, v in list(batch.items()) + [("weight", weight), ("seq_len", seq_len)] if v.is_floating_point() and not bool(torch.isfinite(v.float()).all())]
            if bad or float(seq_len.min()) < 1:
                print("LINK_DEBUG update %d: non-finite %s  weight min/max %s %s  seq_len min %s  timings %s" % (
                    tr.num_update, bad, float(weight.min()), float(weight.max()), float(seq_len.min()), link.timings()), flush=True)
        loss, priority = L.loss(batch, weight, a.pred_weight)
        if os.environ.get("HSAD_LINK_DEBUG") and not bool(torch.isfinite(loss).all()):
            print("LINK_DEBUG update %d: loss non-finite; params finite: %s" % (tr.num_update, bool(torch.isfinite(_flat_of(L.online)).all())), flush=True)
        tr._prios.append(aggregate_priority(priority, seq_len, a.eta))
        g_norm = L.optimizer_step()
        if on_update is not None:
            on_update(u, (loss * weight).mean(), g_norm)
        tr._cur = link.finish()
        tr.num_update += 1
    if stop:
        while link._rounds:                       # batches requested ahead that will not be trained on any more
            link.finish()
        link.begin(tr._prios.pop(0) if tr._prios else None, stop=True)
        link.finish()
    torch.cuda.synchronize()
    if hasattr(L, "check_sync"):
        L.check_sync()
    check_sync()


class LinkCounters:
    """what the actor ranks last published (acts, replay adds, replay size), summed: the Tachometer's actors / replay_buffer on the
    learner rank of a multi-GPU job"""

    def __init__(self, link):
        self.link = link

    def _sum(self, col):
        return sum(int(self.link.store.get("hsad/link/acts/%d" % r).decode().split()[col])
                   for r in range(self.link.world) if r != self.link.learner)

    def num_act(self):
        return self._sum(0)

    def num_add(self):
        return self._sum(1)

    def size(self):
        return self._sum(2)


def run_epochs(tr, args, rank=0, link=None):
    """the epoch loop of selfplay.py:201-281: Tachometer / Stopwatch / MultiCounter output per epoch, then evaluation of the
    online net on fresh games (eval.py:19-66) and the top-k / every-50-epochs saves"""
    from .common import MultiCounter, Stopwatch, Tachometer, TopkSaver
    from .eval import evaluate
    saver = TopkSaver(args.save_dir, 5) if (args.save_dir and rank == 0) else None
    stat, tach, sw = MultiCounter(args.save_dir or None), Tachometer(), Stopwatch()
    history = []
    rows = torch.zeros(args.epoch_len, 2, dtype=torch.float32, device=tr.device)   # (loss, grad_norm) per update, read once per epoch
    factor = args.num_player if args.method == "vdn" else 1

    counters = LinkCounters(link) if link is not None else None

    class _ActCount:       # Tachometer reads sum(actor.num_act()): DeviceActor counts P acts per game step (utils.py:345-352)
        def num_act(self_inner):
            return (counters.num_act() if counters else tr.actor.num_act) // factor
    for epoch in range(args.num_epoch):
        if rank == 0:
            print("beginning of epoch: ", epoch)
        tach.start()
        stat.reset()
        sw.reset()
        if link is not None:     # several GPUs: the actor ranks free-run; this rank only trains (and evaluates below)
            def record(b, loss, g_norm):
                rows[b, 0], rows[b, 1] = loss, g_norm
            run_link_learner(tr, args, link, args.epoch_len, on_update=record, stop=False)
        for b in range(args.epoch_len if link is None else 0):
            tr.act_step(args.act_steps_per_update)
            loss, g_norm = tr.learner_update(sw if args.stopwatch else None)
            if loss is not None:
                rows[b, 0], rows[b, 1] = loss, g_norm
        tr.join_rollout()
        check_sync()
        if tr.learner is not None and hasattr(tr.learner, "check_sync"):
            tr.learner.check_sync()
        tr.env.check_errors()
        tr.replay.check_errors()
        for l_, g_ in rows.cpu().tolist():
            stat["loss"].feed(l_)
            stat["grad_norm"].feed(g_)
        print("EPOCH: %d" % epoch)
        tach.lap([_ActCount()], counters if counters else tr.replay, args.epoch_len * args.batchsize, factor)
        if args.stopwatch:
            sw.summary()
        stat.summary(epoch)
        # context.pause() has no counterpart: actors and learner alternate on this GPU, nothing runs while we evaluate
        eval_seed = (9917 + epoch * 999999) % 7777777
        score, perfect, _, _ = evaluate(tr.learner.online.w, args.num_eval_game, eval_seed, args.eval_bomb, args.sad,
                                        num_player=args.num_player, hand_size=args.hand_size, device=str(tr.device))
        saved = False
        if saver is not None:
            force = "model_epoch%d" % epoch if (epoch > 0 and epoch % 50 == 0) else None
            sd = {k: tr.learner.online.w[k].detach().cpu().clone() for k in tr.param_names}
            saved = saver.save(None, sd, score, force_save_name=force)
        print("epoch %d, eval score: %.4f, perfect: %.2f, model saved: %s" % (epoch, score, perfect * 100, saved))
        print("==========")
        history.append((score, perfect, saved))
    return history


def main(argv=None):
    import pprint
    import sys
    from .common import Logger, set_all_seeds
    from .dist import rank_world
    args = parse_args(argv)
    rank, world = rank_world()
    if world > 1:
        import os
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend)
    dev = "cuda:%d" % torch.cuda.current_device()
    if args.save_dir and rank == 0:
        import os
        os.makedirs(args.save_dir, exist_ok=True)
        sys.stdout = Logger(os.path.join(args.save_dir, "train.log"))
    set_all_seeds(args.seed)
    if rank == 0:
        pprint.pprint(vars(args))       # first thing in train.log: doubles as the saved configuration (utils.py:87-116)
    if args.method == "vdn":
        # selfplay.py:103-106: a VDN transition holds every player's row, so batch / replay / burn-in count games
        args.batchsize = int(np.round(args.batchsize / args.num_player))
        args.replay_buffer_size //= args.num_player
        args.burn_in_frames //= args.num_player
    tr = Trainer(args, dev, rank, world)
    if args.load_model and tr.learner is not None:
        from .checkpoint import load_weight
        print("*****loading pretrained model*****")
        load_weight(tr.learner.online.w, args.load_model)        # online net only, like selfplay.py:143-146
        tr.learner.online.refresh()
        # the reference loads BEFORE ActGroup clones the agent (selfplay.py:143-171): burn-in data and its priorities come from the loaded
        # model.  Here the acting nets already exist: bring the target in line like at update 0 (selfplay.py:209-210) and hand both to
        # the actors now (several GPUs: the first parameter round carries them before the actors' burn-in, run_link_learner)
        tr.learner.sync_target_with_online()
        tr.update_actor_model()
        print("*****done*****")
    elif args.load_model:
        # an actor rank of a multi-GPU job: it reads the file itself, so that its burn-in runs on the loaded model as well instead of
        # on the random initialisation until the first parameter round
        from .checkpoint import load_weight
        for net in (tr.act_online, tr.act_target):
            load_weight(net.w, args.load_model, verbose=False)
            net.refresh()
    if world > 1:
        link = make_link(tr, args)
        if rank != 0:
            run_link_actor(tr, args, link)
        elif args.num_epoch > 0:
            run_epochs(tr, args, rank, link)
            run_link_learner(tr, args, link, 0, stop=True)
        else:
            t0 = time.time()

            def show(u, loss, g_norm):
                if u % 20 == 0:
                    print("update %d loss %.4f grad_norm %.3f" % (u, float(loss), float(g_norm)))
            run_link_learner(tr, args, link, args.num_update, on_update=show)
            dt, c = time.time() - t0, LinkCounters(link)
            print("Speed: train: %.1f, act: %.1f, buffer_size: %d" % (args.num_update * args.batchsize / dt, c.num_act() / dt, c.size()))
            print("exchange per round (ms), transport %s: %s" % (link.transport, ", ".join("%s %.3f" % kv for kv in sorted(link.timings().items()))))
        dist.barrier()
        link.close()
        dist.destroy_process_group()
        return
    t0 = time.time()
    while tr.replay.size() < max(args.batchsize, args.burn_in_frames // world):   # per-shard share of the burn-in
        tr.act_step(10)
    tr.env.check_errors()
    print("burn-in done: replay %d sequences after %d acts in %.1fs" % (tr.replay.size(), tr.actor.num_act, time.time() - t0))
    if args.num_epoch > 0:
        run_epochs(tr, args, rank)
        torch.cuda.synchronize()
        return
    t0, acts0 = time.time(), tr.actor.num_act
    for u in range(args.num_update):
        tr.act_step(args.act_steps_per_update)
        loss, g_norm = tr.learner_update()
        if u % 20 == 0 and loss is not None:
            print("update %d loss %.4f grad_norm %.3f replay %d" % (u, float(loss), float(g_norm), tr.replay.size()))
        if u % 200 == 199:
            check_sync()      # a persistent recurrence that gave up waiting for a sibling leaves garbage: stop, do not train on it
    torch.cuda.synchronize()
    dt = time.time() - t0
    tr.env.check_errors()
    tr.replay.check_errors()
    check_sync()
    if tr.learner is not None and hasattr(tr.learner, "check_sync"):
        tr.learner.check_sync()
    # Tachometer definitions (pyhanabi/utils.py:229-240)
    print("Speed: train: %.1f, act: %.1f, buffer_size: %d" % (args.num_update * args.batchsize / dt,
                                                             (tr.actor.num_act - acts0) / dt, tr.replay.size()))


if __name__ == "__main__":
    main()
