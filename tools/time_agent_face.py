"""The learner through the reference's agent API (`import r2d2`: nn.Module R2D2Agent, autograd loss, torch.optim.Adam + clip_grad_norm_, or the
fused HsadAdam) next to the composite learner it wraps, at configs[2] (2-player SAD, H = 512, T = 80, B = 128): ms per update and the ratio.
    python tools/time_agent_face.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r2d2
import rela
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
dev = "cuda:0"
F, H, A, T, B = 838, 512, 21, 80, 128
W = init_weights(F, H, A, 5, 0)
seq_len = torch.randint(40, 81, (B,)).float().to(dev)
mask = (torch.arange(T, device=dev).unsqueeze(1) < seq_len.unsqueeze(0)).float()
legal = (torch.rand(T, B, A, device=dev) < 0.4).float(); legal[..., 0] = 1
a = torch.multinomial(legal.view(-1, A), 1).view(T, B)
obs = {"priv_s": (torch.rand(T, B, F, device=dev) < 0.15).float() * mask.unsqueeze(2), "legal_move": legal * mask.unsqueeze(2),
       "own_hand": torch.zeros(T, B, 15, device=dev)}
batch = rela.RNNTransition(obs, {"a": a * mask.long()}, (torch.rand(T, B, device=dev) < 0.05).float() * mask, torch.zeros(T, B, device=dev), mask.clone(), seq_len)
weight = torch.ones(B, device=dev)
flat = dict(obs, a=batch.action["a"], reward=batch.reward, bootstrap=batch.bootstrap, seq_len=seq_len)


def timed(upd, n=50):
    for _ in range(8): upd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): upd()
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, t_issue


cl = CompositeLearner(W, W, 3, 0.999, device=dev)
def upd_c():
    cl.loss(flat, weight, 0.0); cl.optimizer_step()
base, bi = timed(upd_c)
print("composite learner (library calls)                        %.3f ms/update (host issue %.3f)  %.1f k sequences/s" % (base * 1e3, bi * 1e3, B / base / 1e3))
sd = {"online_net." + k: v for k, v in W.items()}; sd.update({"target_net." + k: v for k, v in W.items()})
for name, fused in (("r2d2.R2D2Agent + torch.optim.Adam + clip_grad_norm_    ", False), ("r2d2.R2D2Agent + r2d2.HsadAdam (clip + Adam + zero_grad)", True)):
    agent = r2d2.R2D2Agent(False, 3, 0.999, 0.9, dev, F, H, A, 2, 5, False)
    agent.load_state_dict(sd)
    optim = r2d2.HsadAdam(agent.online_net.parameters(), agent, lr=6.25e-5, eps=1.5e-5, max_grad_norm=5.0) if fused else \
        torch.optim.Adam(agent.online_net.parameters(), lr=6.25e-5, eps=1.5e-5)
    def upd():
        loss, priority = agent.loss(batch, 0.0, None)
        loss = (loss * weight).mean()
        loss.backward()
        if not fused:
            torch.nn.utils.clip_grad_norm_(agent.online_net.parameters(), 5.0)
        optim.step()
        optim.zero_grad()
    dt, ti = timed(upd)
    print("%s %.3f ms/update (host issue %.3f)  %.1f k sequences/s  = %.0f %% of the composite learner's rate" % (name, dt * 1e3, ti * 1e3, B / dt / 1e3, 100 * base / dt))
