"""VERDICT r4 weak 6, measured on ONE GPU: what resident communication kernels do to the learner's whole-chip persistent launches.
A posted RCCL receive whose peer has not sent yet is a kernel that sits on k CUs; here k workgroups (256 threads, 16 KB LDS) spin on a
pinned host word on a second stream for `hold` ms, issued before an update the way ReplayLink._star_open posts its receives (rounds
ahead), and the update is timed next to them.
    python tools/resident_probe.py"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
lib = _lib.load_library()
dev = torch.device("cuda:0")
F, H, A, T, B = 838, 512, 21, 80, 128
W = init_weights(F, H, A, 5, 0)
seq_len = torch.randint(40, 81, (B,)).float().to(dev)
mask = (torch.arange(T, device=dev).unsqueeze(1) < seq_len.unsqueeze(0)).float()
legal = (torch.rand(T, B, A, device=dev) < 0.4).float(); legal[..., 0] = 1
a = torch.multinomial(legal.view(-1, A), 1).view(T, B)
batch = {"priv_s": (torch.rand(T, B, F, device=dev) < 0.15).float() * mask.unsqueeze(2), "legal_move": legal * mask.unsqueeze(2),
         "a": a * mask.long(), "reward": (torch.rand(T, B, device=dev) < 0.05).float() * mask, "bootstrap": mask.clone(),
         "seq_len": seq_len, "own_hand": torch.zeros(T, B, 15, device=dev)}
weight = torch.ones(B, device=dev)
lr = CompositeLearner(W, W, 3, 0.999, device=dev)
side = torch.cuda.Stream(dev)
flag = torch.zeros(16, dtype=torch.int32).pin_memory()


def upd():
    lr.loss(batch, weight, 0.0); lr.optimizer_step()


for _ in range(10): upd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): upd()
torch.cuda.synchronize()
base = (time.perf_counter() - t0) / 50
print("no residents: %.3f ms / update" % (base * 1e3))
for k in (1, 4, 16):
    for hold_ms in (0.5, 2.5):
        flag.zero_()
        ts = []
        for it in range(12):
            flag[0] = 0
            _lib.check(lib.hsad_debug_resident_kernel(k, 256, 16384, C.c_void_p(flag.data_ptr()), int(hold_ms * 1000), C.c_void_p(side.cuda_stream)))
            torch.cuda.synchronize() if False else None
            time.sleep(0.0005)                      # the residents are on their CUs before the update is issued
            t0 = time.perf_counter()
            upd()
            torch.cuda.current_stream(dev).synchronize()
            ts.append(time.perf_counter() - t0)
            flag[0] = 1
            torch.cuda.synchronize()
        ts.sort()
        try:
            lr.check_sync(); to = "no"
        except Exception as e:
            to = "YES"
            _lib.check(lr.lib.hsad_r2d2_learner_inject_timeout(lr.h, 0))
        print("%2d resident workgroups held %.1f ms: update %.3f ms median (%.3f min, %.3f max), x %.2f; sibling-wait timeout: %s"
              % (k, hold_ms, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3, ts[len(ts) // 2] / base, to))
