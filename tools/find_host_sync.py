"""which part of Trainer.learner_update blocks the host? (host time of each call with the GPU kept busy)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.selfplay import Trainer, parse_args
from hanabi_sad_amd.replay import aggregate_priority
args = parse_args(["--num_game", "16384", "--replay_buffer_size", "65536", "--sad", "1"])
tr = Trainer(args, "cuda:0")
for _ in range(120):
    tr.actor.step()
for _ in range(5):
    tr.learner_update()
torch.cuda.synchronize()
acc = {}
def T(name, f):
    t0 = time.perf_counter(); r = f(); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
for it in range(50):
    res = T("sample", lambda: tr.sharded.sample(128))
    batch, weight, seq_len = T("batch_of", lambda: tr.batch_of(res))
    loss, prio = T("loss", lambda: tr.learner.loss(batch, weight, 0.0))
    p = T("aggregate", lambda: aggregate_priority(prio, seq_len, 0.9))
    g = T("optimizer_step", lambda: tr.learner.optimizer_step())
    T("update_priority", lambda: tr.sharded.update_priority(p))
    T("mean", lambda: (loss * weight).mean())
torch.cuda.synchronize()
for k, v in acc.items():
    print("%-16s %.3f ms" % (k, v / 50 * 1e3))
# inside sample(): allocation vs the library call
import ctypes as C
from hanabi_sad_amd import _lib
from hanabi_sad_amd.replay import _ptr_array, _stream
rep = tr.replay
acc2 = {}
def T2(name, f):
    t0 = time.perf_counter(); r = f(); acc2[name] = acc2.get(name, 0) + time.perf_counter() - t0; return r
for it in range(30):
    tr.learner.loss(batch, weight, 0.0); tr.learner.optimizer_step()      # keep the GPU busy ~2 ms
    outs = T2("alloc_fields", lambda: rep._alloc_outs(128))
    d, Tn = rep.device, rep.T
    sc = T2("alloc_scalars", lambda: [torch.empty(Tn, 128, dtype=torch.float32, device=d), torch.empty(Tn, 128, dtype=torch.uint8, device=d),
                                      torch.empty(Tn, 128, dtype=torch.float32, device=d), torch.empty(128, dtype=torch.float32, device=d),
                                      torch.empty(128, dtype=torch.float32, device=d)])
    T2("lib_sample", lambda: _lib.check(rep.lib.hsad_replay_sample(rep.h, 128, _ptr_array(outs), sc[0].data_ptr(), sc[1].data_ptr(), sc[2].data_ptr(),
                                                                   sc[3].data_ptr(), sc[4].data_ptr(), _stream(d))))
    T2("bool", lambda: sc[1].bool())
    T2("update", lambda: rep.update_priority(torch.ones(128, device=d)))
torch.cuda.synchronize()
for k, v in acc2.items():
    print("  %-16s %.3f ms" % (k, v / 30 * 1e3))
