"""Developer tool: where one selfplay iteration (actor step + learner update) spends its time at 16,384 games."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.selfplay import Trainer, parse_args
from hanabi_sad_amd.rela import aggregate_priority
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
args = parse_args(["--num_game", str(G), "--replay_buffer_size", "65536"])
tr = Trainer(args, "cuda:0")
while tr.replay.size() < 4096: tr.actor.step()
def timed(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def host_only(f, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t = (time.perf_counter() - t0) / n * 1e3; torch.cuda.synchronize(); return t
print("actor.step            %.2f ms (host issue %.2f)" % (timed(tr.actor.step), host_only(tr.actor.step)))
print("learner_update        %.2f ms (host issue %.2f)" % (timed(tr.learner_update), host_only(tr.learner_update)))
def samp():
    tr.sharded.sample(args.batchsize); tr.sharded.update_priority(torch.ones(args.batchsize, device="cuda:0"))
print("  sample+update_prio  %.2f ms (host issue %.2f)" % (timed(samp), host_only(samp)))
res = tr.sharded.sample(args.batchsize); tr.sharded.update_priority(torch.ones(args.batchsize, device="cuda:0"))
(f, reward, terminal, bootstrap, seq_len), weight = res
batch = {"priv_s": f["priv_s"], "legal_move": f["legal_move"], "a": f["a"].squeeze(2), "reward": reward, "bootstrap": bootstrap,
         "seq_len": seq_len, "own_hand": f["own_hand"]}
def upd():
    loss, p = tr.learner.loss(batch, weight, 0.0); aggregate_priority(p, seq_len, args.eta); tr.learner.optimizer_step()
print("  loss+step           %.2f ms (host issue %.2f)" % (timed(upd), host_only(upd)))
def both():
    tr.actor.step(); tr.learner_update()
print("actor + learner       %.2f ms (host issue %.2f)" % (timed(both), host_only(both)))
# GPU time of each half inside the interleaved loop (events on the main stream)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(30)]
for e in ev:
    e[0].record(); tr.actor.step(); e[1].record(); tr.learner_update(); e[2].record()
torch.cuda.synchronize()
a = sum(e[0].elapsed_time(e[1]) for e in ev[5:]) / 25; l = sum(e[1].elapsed_time(e[2]) for e in ev[5:]) / 25
print("interleaved: actor %.2f ms, learner %.2f ms (main-stream event time)" % (a, l))
