"""Developer tool: R2D2 learner update time at BASELINE configs[2] (2p SAD IQL, H=512, B=128, T=80)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd.r2d2 import R2D2Learner
DEV = "cuda:0"
torch.manual_seed(0)
F, H, A, T, B = 838, 512, 21, 80, 128
lin = lambda o, i: (torch.rand(o, i) * 2 - 1) / i ** 0.5
W = {"net.0.weight": lin(H, F), "net.0.bias": lin(H, 1).squeeze(1), "fc_v.weight": lin(1, H), "fc_v.bias": torch.zeros(1),
     "fc_a.weight": lin(A, H), "fc_a.bias": torch.zeros(A), "pred.weight": lin(15, H), "pred.bias": torch.zeros(15)}
for l in range(2):
    W["lstm.weight_ih_l%d" % l] = lin(4 * H, H); W["lstm.weight_hh_l%d" % l] = lin(4 * H, H)
    W["lstm.bias_ih_l%d" % l] = lin(4 * H, 1).squeeze(1); W["lstm.bias_hh_l%d" % l] = lin(4 * H, 1).squeeze(1)
lr = R2D2Learner(W, W, 3, 0.999, device=DEV)
lr.chunks = int(os.environ.get("CHUNKS", lr.chunks))
lr.wgrad_split = int(os.environ.get("WSPLIT", lr.wgrad_split))
seq_len = torch.randint(40, 81, (B,)).float().to(DEV)
mask = (torch.arange(T, device=DEV).unsqueeze(1) < seq_len.unsqueeze(0)).float()
legal = (torch.rand(T, B, A, device=DEV) < 0.4).float(); legal[..., 0] = 1
batch = {"priv_s": (torch.rand(T, B, F, device=DEV) < 0.15).float() * mask.unsqueeze(2), "legal_move": legal * mask.unsqueeze(2),
         "a": torch.zeros(T, B, dtype=torch.int64, device=DEV), "reward": (torch.rand(T, B, device=DEV) < 0.05).float() * mask,
         "bootstrap": mask.clone(), "seq_len": seq_len, "own_hand": torch.zeros(T, B, 15, device=DEV)}
weight = torch.ones(B, device=DEV)
def upd():
    loss, prio = lr.loss(batch, weight, 0.0)
    lr.optimizer_step()
for _ in range(5): upd()
n, dt = 20, 1e9
for rep in range(4):   # best of 4 x 20 updates (box-to-box and run-to-run noise is several %)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): upd()
    torch.cuda.synchronize(); dt = min(dt, (time.perf_counter() - t0) / n)
print("chunks=%d" % lr.chunks, end="  "); print("HIP learner update: %.2f ms  -> %.0f sequences/s  (%.1f TFLOP/s of 380.3 GFLOP/update)" % (dt * 1e3, B / dt, 380.3e9 / dt / 1e12))
t0 = time.perf_counter()
for _ in range(0 if os.environ.get("TRACE") else n): lr.loss(batch, weight, 0.0, compute_grad=False)
torch.cuda.synchronize(); print("  forward only (online+target+TD): %.2f ms" % ((time.perf_counter() - t0) / n * 1e3))
from hanabi_sad_amd.r2d2 import check_sync, _SYNC; check_sync()
import ctypes
from hanabi_sad_amd import _lib
lib = _lib.load_library(); buf = (ctypes.c_uint64 * 16)()
lib.hsad_lstm_debug_timing(None, 1); upd(); upd(); torch.cuda.synchronize(); lib.hsad_lstm_debug_timing(buf, 0)
names = ["wait", "h loads", "mfma", "cell", "publish", "state stores", "", "", "wait", "loads+mfma", "cell bwd", "publish"]
# the timed workgroup (row block 0, unit block 0) exists once per recurrence: 4 fwd recurrences x 80 steps, 2 bwd x 80
print("  fwd per step (us):", ", ".join("%s %.2f" % (names[i], buf[i] / 100.0 / 640) for i in range(6)), " sum %.2f" % (sum(buf[:6]) / 100.0 / 640))
print("  bwd per step (us):", ", ".join("%s %.2f" % (names[i], buf[i] / 100.0 / 320) for i in range(8, 12)), " sum %.2f" % (sum(buf[8:12]) / 100.0 / 320))
for k, b in _SYNC.items():   # XCD placement words of the last launch on each buffer: how many groups were co-located?
    if "m/" not in k[2]: continue
    words = b[:64].cpu().numpy().view("uint64")
    groups = [[int((w >> (6 * i)) & 63) for i in range(8)] for w in words[:16] if w]
    groups = [g for g in groups if sum(g) == 16]     # the words after the last group of a launch are step counters
    ok = sum(max(g) == sum(g) for g in groups)
    print("  %s: %d/%d groups on one XCD" % (k[2], ok, len(groups)), groups[:2] if ok == len(groups) else groups)
if os.environ.get("NO_TORCH"): sys.exit(0)
# torch eager baseline of the same math (nn.LSTM / MIOpen fp32) -- "what you get without hand-written kernels"
import torch.nn as nn
class Net(nn.Module):
    def __init__(s):
        super().__init__(); s.net = nn.Sequential(nn.Linear(F, H), nn.ReLU()); s.lstm = nn.LSTM(H, H, 2); s.fc_v = nn.Linear(H, 1); s.fc_a = nn.Linear(H, A)
    def forward(s, priv, legal, a):
        o, _ = s.lstm(s.net(priv)); av = s.fc_a(o); v = s.fc_v(o); la = av * legal; q = v + la - la.mean(2, keepdim=True)
        return q.gather(2, a.unsqueeze(2)).squeeze(2), ((1 + q - q.min()) * legal).argmax(2)
on, tg = Net().to(DEV), Net().to(DEV)
opt = torch.optim.Adam(on.parameters(), lr=6.25e-5, eps=1.5e-5)
def upd_t():
    qa, g = on(batch["priv_s"], legal, batch["a"])
    with torch.no_grad():
        tq, _ = tg(batch["priv_s"], legal, g); tq = torch.cat([tq[3:], tq[:3]], 0); tq[-3:] = 0
        target = batch["reward"] + batch["bootstrap"] * 0.997 * tq
    err = (target - qa) * mask
    l = (nn.functional.smooth_l1_loss(err, torch.zeros_like(err), reduction="none").sum(0) * weight).mean()
    opt.zero_grad(); l.backward(); nn.utils.clip_grad_norm_(on.parameters(), 5); opt.step()
for _ in range(3): upd_t()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): upd_t()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("torch eager fp32 (MIOpen LSTM) update: %.2f ms -> %.0f sequences/s" % (dt * 1e3, B / dt))
