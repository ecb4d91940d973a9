#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json): Hanabi env-steps/sec.

`python bench.py --gpus N --steps K --warmup W`; for N>1 launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU).

A "step" = one iteration of the reference thread-loop body (cpp/thread_loop.h:46-72) over all G games
of this rank: reset-terminated -> random-legal policy -> env step (+observe), i.e. G env-steps
(Tachometer `act` unit: pyhanabi/utils.py:229-236).  Workload = BASELINE.json configs[1]:
65,536 concurrent 2-player games per GPU, random-action policy, synthetic (random-policy) play.
Games are independent, so N GPUs run N shards with no data-path collective (weak scaling).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: env step) and, at N=1,
`cpu_baseline` (the CPU oracle timed on this box's host cores — a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
GAMES_PER_GPU = 65536
PLAYERS, HAND = 2, 5
EPS = [0.1 ** (1 + 7 * i / 79) for i in range(80)]  # utils.generate_explore_eps(0.1, 7, 80)


def algorithmic_bytes_per_step(P, F, A, H, sad):
    """obs outputs f32 + reward/terminal + actions in + nominal 128-B state r/w."""
    return P * (F + A + 3 * H + 1) * 4 + 5 + P * 8 * (1 + int(sad)) + 2 * 128


# committed rocprofv3 PMC summaries, newest round first (profiles/, collected with separate --pmc WRITE_SIZE / FETCH_SIZE passes and
# corrected per MI355X_MICROARCH.md §HBM by tools/pmc_summarize.py); a leg is read from the newest file that holds it
PMC_FILES = ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json")


def pmc_leg(leg):
    for name in PMC_FILES:
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name))).get(leg)
            if rec:
                return rec, name
        except Exception:
            pass
    return {}, None


def pmc_kernel(leg, needle, field="hbm_bytes_per_launch"):
    """`field` of the kernels of a leg whose name contains `needle`, averaged (None when the leg was never measured)"""
    rec, _ = pmc_leg(leg)
    v = [x[field] for k, x in rec.items() if needle in k and isinstance(x, dict) and field in x]
    return sum(v) / len(v) if v else None


def measured_traffic_bytes(G, mode, chunk=0):
    """HBM bytes per launch of env_kernel<mode,2,5> (chunk > 0: of the persistent env_rollout_kernel<2,5> running `chunk`
    iterations per launch).  Only valid for the configuration it was measured at; None otherwise."""
    if G != 65536:
        return None
    if chunk > 0:
        v = pmc_kernel("env", "env_rollout_kernel<2,5>", "hbm_bytes_per_iteration")
        return None if v is None else v * chunk                                    # chunk = iterations per launch
    return pmc_kernel("env", "env_kernel<%d,2,5>" % mode)


def gemm_traffic_bytes():
    """HBM bytes per launch of the learner's LSTM input-projection GEMM (10240x2048x512)"""
    return pmc_kernel("gemm", "gemm8_kernel") or pmc_kernel("gemm", "gemm_nt_bf16_kernel")


def fused_traffic_bytes(which="fwd"):
    """HBM bytes per launch of the fused forward / BPTT recurrence kernel inside a learner update"""
    # (the PMC file must be of the round whose kernel runs: the BPTT launch is lstm_bptt_wide_kernel since round 6)
    return pmc_kernel("learner", "lstm_bptt_wide_kernel" if which == "bwd" else "lstm_fused_%s_kernel" % which)


def cell_traffic_bytes():
    """HBM bytes per launch of the fused cell kernel inside an acting step (one launch = the online net's cell, which also writes the fp32
    state, and the target net's)"""
    return pmc_kernel("actor", "gemm8_kernel<2") or pmc_kernel("actor", "lstm_cell_pp_kernel")


def env5_traffic_bytes(games, chunk, sad):
    """HBM bytes per launch of env_rollout_kernel<5,4> at configs[4]'s per-GPU size (the leg "env5" is the SAD variant, "env5_literal"
    the configuration as BASELINE.json states it)"""
    if games != 16384:
        return None
    v = pmc_kernel("env5" if sad else "env5_literal", "env_rollout_kernel<5,4>", "hbm_bytes_per_iteration")
    return None if v is None else v * chunk


def mfma_counters():
    """MFMA-busy COUNTER figures of the three MFMA kernels from the committed rocprofv3 PMC pass (profiles/rNN_mfma_util.json, written by
    tools/mfma_util.sh): {kernel: {mfma_busy, flop_frac, avg_duration_us}}; {} when never collected"""
    for name in ("r06_mfma_util.json", "r05_mfma_util.json", "r04_mfma_util.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        gui = "mfma_busy_frac_if_gui_is_summed_over_8_xcds"      # (what the calibration leg of tools/mfma_util.sh shows GRBM_GUI_ACTIVE to be)
        out = {}
        for leg in ("learner", "actor"):
            for k, r in rec.get(leg, {}).items():
                key = k.split("<")[0]
                out[key] = {"mfma_busy_at_2p4GHz_wall": r.get("mfma_busy_frac_at_2p4GHz_wall"), "mfma_busy_over_gui_active": r.get(gui),
                            "flop_frac": r.get("flop_frac"), "avg_duration_us": r.get("avg_duration_us_uninstrumented"), "source": "profiles/" + name}
        return out
    return {}


def learner_bench(dev, updates=20, warmup=3, gemm_probe=True):
    """Second half of BASELINE.json's metric: R2D2 learner samples/sec at configs[2] (2p SAD IQL, F=838, A=21,
    H=512, 2-layer LSTM, B=128, T=80, n=3): sample-shaped synthetic batch -> loss fwd (online+target) -> BPTT ->
    clip+Adam, all on the hand-written HIP kernels (bf16 MFMA operands, fp32 accumulate/state)."""
    import hanabi_sad_amd.r2d2 as r2d2
    from hanabi_sad_amd.r2d2 import R2D2Learner, gemm_nt
    torch.manual_seed(0)
    F, H, A, T, B = 838, 512, 21, 80, 128
    lin = lambda o, i: (torch.rand(o, i) * 2 - 1) / i ** 0.5
    W = {"net.0.weight": lin(H, F), "net.0.bias": lin(H, 1).squeeze(1), "fc_v.weight": lin(1, H),
         "fc_v.bias": torch.zeros(1), "fc_a.weight": lin(A, H), "fc_a.bias": torch.zeros(A),
         "pred.weight": lin(15, H), "pred.bias": torch.zeros(15)}
    for l in range(2):
        W["lstm.weight_ih_l%d" % l] = lin(4 * H, H)
        W["lstm.weight_hh_l%d" % l] = lin(4 * H, H)
        W["lstm.bias_ih_l%d" % l] = lin(4 * H, 1).squeeze(1)
        W["lstm.bias_hh_l%d" % l] = lin(4 * H, 1).squeeze(1)
    lr = R2D2Learner(W, W, 3, 0.999, lr=6.25e-5, eps=1.5e-5, grad_clip=5.0, device=dev)
    seq_len = torch.randint(40, 81, (B,)).float().to(dev)
    mask = (torch.arange(T, device=dev).unsqueeze(1) < seq_len.unsqueeze(0)).float()
    legal = (torch.rand(T, B, A, device=dev) < 0.4).float()
    legal[..., 0] = 1
    a = torch.multinomial(legal.view(-1, A), 1).view(T, B)
    batch = {"priv_s": (torch.rand(T, B, F, device=dev) < 0.15).float() * mask.unsqueeze(2),
             "legal_move": legal * mask.unsqueeze(2), "a": a * mask.long(),
             "reward": (torch.rand(T, B, device=dev) < 0.05).float() * mask, "bootstrap": mask.clone(),
             "seq_len": seq_len, "own_hand": torch.zeros(T, B, 15, device=dev)}
    weight = torch.ones(B, device=dev)

    def upd():
        lr.loss(batch, weight, 0.0)
        lr.optimizer_step()
    for _ in range(warmup):
        upd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(updates):
        upd()
    torch.cuda.synchronize()
    dt_py = (time.perf_counter() - t0) / updates
    r2d2.check_sync()
    # the product path: the library's composite entry points (hsad_r2d2_loss_fwd / _loss_bwd / _optimizer_step: one C call each),
    # forward recurrences fused (round 3); the Python-orchestrated learner above drives the chunk-pipelined schedule of rounds 1-2
    # and is the A/B reference (python_schedule_ms_per_update)
    from hanabi_sad_amd.composite import CompositeLearner
    cl = CompositeLearner(W, W, 3, 0.999, lr=6.25e-5, eps=1.5e-5, grad_clip=5.0, device=dev)
    for _ in range(warmup):
        cl.loss(batch, weight, 0.0)
        cl.optimizer_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(updates):
        cl.loss(batch, weight, 0.0)
        cl.optimizer_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / updates
    cl.check_sync()
    cl.close()
    flop = 380.3e9  # online fwd + target fwd + online bwd
    M, N, K = T * B, 4 * H, H
    if not gemm_probe:
        return {"value": B / dt, "unit": "sequences/s", "ms_per_update": dt * 1e3}
    # spread: repeat the timed 20-update block until >= 0.5 s of GPU time
    blocks = []
    cl = CompositeLearner(W, W, 3, 0.999, lr=6.25e-5, eps=1.5e-5, grad_clip=5.0, device=dev)
    for _ in range(warmup):
        cl.loss(batch, weight, 0.0)
        cl.optimizer_step()
    torch.cuda.synchronize()
    while sum(blocks) < 0.5 or len(blocks) < 5:
        t0 = time.perf_counter()
        for _ in range(updates):
            cl.loss(batch, weight, 0.0)
            cl.optimizer_step()
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
    per = sorted(x / updates * 1e3 for x in blocks)
    # The LSTM GEMM FLOPs of the forward pass -- [x | h] [W_ih | W_hh]^T of both layers of both nets -- live in ONE persistent
    # kernel since round 3 (lstm_fused_fwd_kernel: projection inside the recurrence, weights register-resident; the stand-alone
    # x W_ih^T GEMM of rounds 1-2 is gone from the update).  Timed WHERE IT RUNS: HIP events around its launches inside five more
    # updates, on the stream it is launched on (hsad_lstm_fused_timing).
    import ctypes as C
    from hanabi_sad_amd import _lib
    lib = _lib.load_library()
    _lib.check(lib.hsad_lstm_fused_timing(1))
    for _ in range(5):
        cl.loss(batch, weight, 0.0)
        cl.optimizer_step()
    f_ms, f_fl, f_n = C.c_double(0), C.c_double(0), C.c_int32(0)
    b_ms, b_fl, b_n = C.c_double(0), C.c_double(0), C.c_int32(0)
    _lib.check(lib.hsad_lstm_fused_timing_read_kind(0, C.byref(f_ms), C.byref(f_fl), C.byref(f_n)))
    _lib.check(lib.hsad_lstm_fused_timing_read_kind(1, C.byref(b_ms), C.byref(b_fl), C.byref(b_n)))
    _lib.check(lib.hsad_lstm_fused_timing(0))
    # ... and the chunk-pipelined schedule of rounds 1-2 (stand-alone projection GEMMs + chunked recurrences) on the same learner
    cl.set_fused(False)
    for _ in range(warmup):
        cl.loss(batch, weight, 0.0)
        cl.optimizer_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(updates):
        cl.loss(batch, weight, 0.0)
        cl.optimizer_step()
    torch.cuda.synchronize()
    dt_chunked = (time.perf_counter() - t0) / updates
    _lib.check(lib.hsad_gemm_timing(1))
    for _ in range(5):
        cl.loss(batch, weight, 0.0)
        cl.optimizer_step()
    g_ms, g_n, g_np = C.c_double(0), C.c_int32(0), C.c_int32(1)
    _lib.check(lib.hsad_gemm_timing_read(M, N, K, C.byref(g_ms), C.byref(g_n), C.byref(g_np)))
    _lib.check(lib.hsad_gemm_timing(0))
    cl.check_sync()
    cl.close()
    gemm_ms = g_ms.value / max(g_np.value, 1)
    gemm_tf = 2.0 * M * N * K / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    fused_tf = f_fl.value / (f_ms.value * 1e-3) / 1e12 if f_ms.value > 0 else 0.0
    bptt_tf = b_fl.value / (b_ms.value * 1e-3) / 1e12 if b_ms.value > 0 else 0.0
    counters = mfma_counters()
    # the same update through the reference's agent API (`import r2d2`: nn.Module R2D2Agent over the kernels, autograd loss, torch.optim.Adam +
    # clip_grad_norm_ -- the calls of pyhanabi/selfplay.py:218-241 -- and with the fused HsadAdam)
    face = {}
    try:
        import r2d2 as r2d2_face
        import rela
        rb = rela.RNNTransition({"priv_s": batch["priv_s"], "legal_move": batch["legal_move"], "own_hand": batch["own_hand"]}, {"a": batch["a"]},
                                batch["reward"], torch.zeros_like(batch["reward"]), batch["bootstrap"], batch["seq_len"])
        sd = {"online_net." + k: v for k, v in W.items()}
        sd.update({"target_net." + k: v for k, v in W.items()})
        for key, fused_opt in (("torch_adam_ms_per_update", False), ("hsad_adam_ms_per_update", True)):
            ag = r2d2_face.R2D2Agent(False, 3, 0.999, 0.9, str(dev), F, H, A, 2, 5, False)
            ag.load_state_dict(sd)
            opt = r2d2_face.HsadAdam(ag.online_net.parameters(), ag, lr=6.25e-5, eps=1.5e-5, max_grad_norm=5.0) if fused_opt else \
                torch.optim.Adam(ag.online_net.parameters(), lr=6.25e-5, eps=1.5e-5)

            def upd_face():
                l, _ = ag.loss(rb, 0.0, None)
                (l * weight).mean().backward()
                if not fused_opt:
                    torch.nn.utils.clip_grad_norm_(ag.online_net.parameters(), 5.0)
                opt.step()
                opt.zero_grad()
            for _ in range(warmup + 3):
                upd_face()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(updates):
                upd_face()
            torch.cuda.synchronize()
            face[key] = (time.perf_counter() - t0) / updates * 1e3
            del ag, opt
        face["share_of_composite_rate"] = {k.replace("_ms_per_update", ""): dt * 1e3 / v for k, v in face.items() if k.endswith("_ms_per_update")}
    except Exception as e:      # (the face is a convenience layer: its absence must not cost the learner's number)
        face = {"error": "%s: %s" % (type(e).__name__, e)}
    # the fp32-exact mode (the reference learner's own arithmetic type, pyhanabi/selfplay.py:149: fp32 operands on v_mfma_f32_32x32x2_f32,
    # csrc/hsad_r2d2_f32.hip; parity mode, orchestrated step by step from Python -- not the product path): the same update, a few repeats
    try:
        l32 = R2D2Learner(W, W, 3, 0.999, lr=6.25e-5, eps=1.5e-5, grad_clip=5.0, device=dev, precision="fp32")
        for _ in range(2):
            l32.loss(batch, weight, 0.0)
            l32.optimizer_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n32 = 5
        for _ in range(n32):
            l32.loss(batch, weight, 0.0)
            l32.optimizer_step()
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / n32
        fp32_leg = {"value": B / dt32, "unit": "sequences/s", "ms_per_update": dt32 * 1e3, "updates_timed": n32,
                    "dtype": "fp32 operands and accumulate (v_mfma_f32_32x32x2_f32), the reference learner's arithmetic type",
                    "update_tflops": flop / dt32 / 1e12, "peak": 157.3, "unit_peak": "TFLOP/s (dense fp32 MFMA)", "frac": flop / dt32 / 1e12 / 157.3,
                    "note": "parity mode (golden-vector tolerance 1e-6 against the reference's r2d2.py): one GEMM + one cell launch per time step "
                            "issued from Python, no persistent recurrence -- reported next to the bf16-operand product path, not tuned",
                    "bf16_path_speedup": dt32 / dt}
        del l32
    except Exception as e:
        fp32_leg = {"error": "%s: %s" % (type(e).__name__, e)}
    return {
        "value": B / dt, "unit": "sequences/s", "ms_per_update": dt * 1e3, "python_schedule_ms_per_update": dt_py * 1e3,
        "reference_agent_api": face,
        "learner_fp32": fp32_leg,
        "chunk_pipelined_schedule_ms_per_update": dt_chunked * 1e3,
        "repeats": {"blocks": len(blocks), "updates_per_block": updates, "ms_per_update_median": per[len(per) // 2], "ms_per_update_min": per[0],
                    "ms_per_update_max": per[-1]},
        "dtype": "bf16 MFMA operands, fp32 accumulate",
        "config": {"workload": "BASELINE configs[2]: 2p SAD IQL learner update, F=838 A=21 H=512 L=2 B=128 T=80 n=3, "
                               "synthetic batch, random-init nets, loss fwd + BPTT + clip + Adam"},
        "update_tflops": flop / dt / 1e12,
        # the DOMINANT kernel of the update: the BPTT launch, timed where it runs
        "roofline": {"bound": "mfma",
                     "kernel": "lstm_bptt_wide_kernel<64> (the BPTT of an update: both LSTM layers of the online net, the dO = dG1 W_ih1 projection "
                               "stage and the input layer's dx = dG0 W_ih0 sink stage as four pipeline stages x %d steps in one persistent launch; "
                               "round 6: 16 rows x 64 units per workgroup, weight slices in registers -- lstm_fused_bwd_kernel<64> is the 32 x 32 "
                               "blocking of rounds 3-5)" % T,
                     "achieved": bptt_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": bptt_tf / 2500.0,
                     "traffic": fused_traffic_bytes("bwd"), "avg_launch_ms": b_ms.value, "in_update_launches_timed": b_n.value,
                     "algorithmic_flop_per_launch": b_fl.value, "share_of_update": b_ms.value / (dt * 1e3) if dt > 0 else None,
                     "mfma_busy_counter": counters.get("lstm_bptt_wide_kernel"),
                     "note": "a recurrence over B = 128 rows is latency-bound by construction (each of the 80 steps needs the previous one: its "
                             "step time is the cross-workgroup exchange, not MFMA issue); the fraction is reported as what it is"},
        "roofline_forward": {"bound": "mfma",
                     "kernel": "lstm_fused_fwd_kernel<16> (the forward LSTM of an update: 2 nets x 2 layers x %d steps, [x_t | h_t-1] [W_ih | W_hh]^T "
                               "inside the persistent recurrence; one launch per update)" % T,
                     "achieved": fused_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": fused_tf / 2500.0,
                     "traffic": fused_traffic_bytes("fwd"), "avg_launch_ms": f_ms.value, "in_update_launches_timed": f_n.value,
                     "algorithmic_flop_per_launch": f_fl.value, "share_of_update": f_ms.value / (dt * 1e3) if dt > 0 else None,
                     "mfma_busy_counter": counters.get("lstm_fused_fwd_kernel")},
        "roofline_projection_gemm_rounds_1_2": {
            "bound": "mfma", "kernel": "gemm8_kernel<G8_F32>, the 256 x 256 core since round 5 (LSTM input projection %dx%dx%d of the chunk-pipelined schedule; online + target "
                                       "net = one launch of two problems, avg_launch_ms is per problem)" % (M, N, K),
            "achieved": gemm_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": gemm_tf / 2500.0, "traffic": gemm_traffic_bytes(),
            "avg_launch_ms": gemm_ms, "in_update_launches_timed": g_n.value, "problems_per_launch": g_np.value,
            "algorithmic_flop_per_launch": 2.0 * M * N * K, "algorithmic_bytes_per_launch": M * K * 2 + N * K * 2 + M * N * 4},
    }


def gemm_core_bench(dev, n=8192, reps=12):
    """The 256 x 256 phase-interleaved MFMA core every GEMM-shaped kernel of the hot path now runs on (gemm8_kernel: the fused cell of an acting
    step, the learner's weight gradients, the input layer), as a PLAIN bf16 GEMM of exactly known FLOP count on random operands (normal and
    uniform [-1, 1): operand data moves the sustained clock) -- timed here, with HIP events on the launch stream; the MFMA-busy COUNTER of the
    same GEMM is the `calib` leg of profiles/rNN_mfma_util.json."""
    from hanabi_sad_amd.r2d2 import gemm_nt
    out = {}
    for fill in ("normal", "uniform"):
        mk = (lambda *s: torch.randn(*s, device=dev)) if fill == "normal" else (lambda *s: torch.rand(*s, device=dev) * 2 - 1)
        A, B = mk(n, n).to(torch.bfloat16), mk(n, n).to(torch.bfloat16)
        for kind in ("fp32", "bf16"):
            Cm = torch.empty(n, n, device=dev, dtype=torch.float32 if kind == "fp32" else torch.bfloat16)
            f = (lambda: gemm_nt(A, B, n, n, n, out32=Cm)) if kind == "fp32" else (lambda: gemm_nt(A, B, n, n, n, out16=Cm))
            for _ in range(4):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out["%s_operands_%s_out" % (fill, kind)] = {"avg_launch_ms": ms, "tflops": 2.0 * n ** 3 / (ms * 1e-3) / 1e12}
            del Cm
        del A, B
    best = out["normal_operands_fp32_out"]["tflops"]
    torch.cuda.empty_cache()
    return {"config": {"workload": "C = A B^T, %d^3, bf16 operands, fp32 accumulate, one launch of gemm8_kernel (256 workgroups, persistent over 1,024 tiles)" % n},
            "runs": out,
            "roofline": {"bound": "mfma", "kernel": "gemm8_kernel<G8_F32> (fp32 output through wave-private LDS, full 128-byte rows), normal-distributed operands",
                         "achieved": best, "peak": 2500.0, "unit": "TFLOP/s", "frac": best / 2500.0, "traffic": None,
                         "algorithmic_flop_per_launch": 2.0 * n ** 3, "avg_launch_ms": out["normal_operands_fp32_out"]["avg_launch_ms"],
                         "mfma_busy_counter": (mfma_calib() or None)}}


def mfma_calib():
    """the calibration GEMM's counter figures from the committed PMC pass (profiles/rNN_mfma_util.json, leg `calib`)"""
    for name in ("r06_mfma_util.json", "r05_mfma_util.json", "r04_mfma_util.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name))).get("calib", {})
        except Exception:
            continue
        for k, r in rec.items():
            return {"kernel": k, "flop_frac": r.get("flop_frac"), "mfma_busy_over_gui_active": r.get("mfma_busy_frac_if_gui_is_summed_over_8_xcds"),
                    "avg_duration_us": r.get("avg_duration_us_uninstrumented"), "source": "profiles/" + name}
    return None


def actor_bench(dev, games=16384, steps=160, warmup=120):
    """The rollout with the agent in the loop (SURVEY.md §8 rows a/f; what the reference's actor threads + BatchRunner do):
    one DeviceActor.step() = reset finished games -> observe -> R2D2 act (eps-greedy, SAD greedy action) -> env step ->
    n-step pop -> compute_priority (online + target nets) -> sequence push -> flush finished sequences into the
    prioritized replay.  2-player SAD IQL, H=512, 2-layer LSTM; acts = games x players per step (utils.py:229-236).
    Warm-up runs past the first episode ends (all games start together) and the timed window spans two max-length
    episodes, so the per-step average includes the steady-state rate of finished sequences being flushed."""
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    args = parse_args(["--num_game", str(games), "--replay_buffer_size", "65536", "--sad", "1"])
    tr = Trainer(args, str(dev))
    for _ in range(warmup):
        tr.actor.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.actor.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tr.env.check_errors()
    tr.replay.check_errors()
    # host issue time per step: the CPU time of the calls themselves (no device wait inside; the queue is drained first and the
    # 40 steps stay well inside the HIP queue depth), for the library's actor body and for the Python body of rounds 1-2
    def issue_us(actor_obj, k=40):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(k):
            actor_obj.step()
        h = (time.perf_counter() - t) / k * 1e6
        torch.cuda.synchronize()
        return h
    host_issue_us = issue_us(tr.actor)
    # the dominant kernel of a step (four launches: two LSTM layers x online / target net) where it runs: HIP events around every
    # launch of the fused GEMM + cell kernel during 40 more steps, on the stream it is launched on
    import ctypes as C
    from hanabi_sad_amd import _lib
    lib = _lib.load_library()
    _lib.check(lib.hsad_lstm_cell_timing(1))
    for _ in range(40):
        tr.actor.step()
    ms, fl, nl = C.c_double(0), C.c_double(0), C.c_int32(0)
    _lib.check(lib.hsad_lstm_cell_timing_read(C.byref(ms), C.byref(fl), C.byref(nl)))
    _lib.check(lib.hsad_lstm_cell_timing(0))
    cell_tf = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    # one whole learner iteration on the sequences the rollout just produced (selfplay.py:208-244): prioritized sample out of the
    # bit-packed replay (observation expanded straight to the bf16 GEMM operand) -> loss fwd + BPTT -> clip + Adam -> priorities
    # aggregated and written back
    it_ms = None
    if tr.replay.size() >= args.batchsize:
        for _ in range(3):
            tr.learner_update()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            tr.learner_update()
        torch.cuda.synchronize()
        it_ms = (time.perf_counter() - t1) / 20 * 1e3
        tr.learner.check_sync()
        tr.replay.check_errors()
    out = {"value": games * 2 / dt, "unit": "acts/s", "ms_per_step": dt * 1e3, "game_steps_per_sec": games / dt,
           "loop_body": "hsad_actor_step (C ABI, one call per step)" if tr.actor.c_actor is not None else "python (actor.DeviceActor.step)",
           "host_issue_us_per_step": host_issue_us,
           "learner_iteration_ms_on_rollout_data": it_ms, "replay_bytes": tr.replay.bytes(),
           "roofline": {"bound": "mfma", "kernel": "gemm8_kernel<G8_CELL> (fused [x | h] [W_ih | W_hh]^T GEMM + LSTM cell update, %d x %d x %d, 256 x 256 tiles, "
                                                   "phase-interleaved k loop; the online and the target net's cell of a layer are ONE launch of two problems: 2 launches per step)" % (games * 2, 2048, 1024),
                        "achieved": cell_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": cell_tf / 2500.0, "traffic": cell_traffic_bytes(),
                        "mfma_busy_counter": mfma_counters().get("gemm8_kernel") or mfma_counters().get("lstm_cell_pp_kernel"),
                        "algorithmic_bytes_per_launch": 2 * (2 * games * 2 * 512 * 2 + 2048 * 1024 * 2 + games * 2 * 512 * 4 + games * 2 * 512 * 2) + 2 * games * 2 * 512 * 4,
                        "avg_launch_ms": ms.value, "in_step_launches_timed": nl.value, "algorithmic_flop_per_launch": fl.value,
                        "launches_per_step": nl.value / 40.0, "share_of_step": nl.value / 40.0 * ms.value / (dt * 1e3)},
           "observation_path": "packed (bit words + bf16 rows from the env kernel)" if tr.actor.packed_obs else "float32",
           "config": {"workload": "%d concurrent 2-player SAD games, IQL R2D2 agent (F=838 A=21 H=512 L=2) in the loop, n-step 3, "
                                  "max_len 80, priorities from online+target nets, finished sequences flushed into a "
                                  "65,536-sequence device replay" % games}}
    del tr
    torch.cuda.empty_cache()
    return out


def training_bench(dev, games=6400, iterations=150, warmup=20, extra_args=()):
    """One-GPU self-play training as `python -m hanabi_sad_amd.selfplay` runs it (the configuration of the committed convergence run: 6,400 games,
    one rollout step per update on the rollout stream, B = 128 sequences drawn from the replay the rollout fills, draw ahead): rollout step ->
    update (loss fwd + BPTT, clip + Adam, priorities written back) -> next draw, timed over whole iterations with the host running ahead."""
    from hanabi_sad_amd.selfplay import Trainer, parse_args
    args = parse_args(["--num_game", str(games), "--replay_buffer_size", "131072", "--sad", "1"] + list(extra_args))
    tr = Trainer(args, str(dev))
    tr.act_step(130)                                       # past the first episode ends: the replay holds > 2 batches
    tr.join_rollout()

    def iteration():
        tr.act_step(1)
        tr.learner_update()
    for _ in range(warmup):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iterations):
        iteration()
    t_issue = time.perf_counter() - t0
    tr.join_rollout()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iterations
    # what the host needs to ISSUE an iteration when nothing holds it back (empty queue: device drained in front of every iteration) -- the
    # figure above includes the back-pressure of a full queue (staging-slot ring, run-ahead bound), i.e. it reads ~ the device's iteration time
    # whenever the device is the limiter
    t_free = 0.0
    for _ in range(30):
        tr.join_rollout()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        iteration()
        t_free += time.perf_counter() - t1
    tr.join_rollout()
    torch.cuda.synchronize()
    tr.learner.check_sync()
    tr.env.check_errors()
    tr.replay.check_errors()
    out = {"value": args.batchsize / dt, "unit": "sequences/s", "ms_per_iteration": dt * 1e3, "acts_per_sec": games * 2 / dt,
           "host_issue_ms_per_iteration": t_issue / iterations * 1e3, "host_issue_ms_per_iteration_on_an_empty_queue": t_free / 30 * 1e3,
           "iterations": iterations,
           "config": {"workload": "%d concurrent 2-player SAD games, one rollout step (reset + observe + act + env step + n-step / priority / sequence push + "
                                  "flush into the prioritized replay) per learner update of %d sequences x 80 steps drawn from that replay; rollout on its own "
                                  "stream, priority write-back and the draw of the next batch on a third one behind the forward half of the update (selfplay "
                                  "--overlap_rollout 1 --draw_ahead 1 --early_draw 1)" % (games, args.batchsize)}}
    del tr
    torch.cuda.empty_cache()
    return out


EXCHANGE_TIMEOUT_S = int(os.environ.get("HSAD_BENCH_EXCHANGE_TIMEOUT", "180"))


def exchange_bench(dev, rank, world, rounds=60, batch=128, mode="star"):
    """--gpus N > 1: the learner <-> actor exchange of a multi-GPU self-play job (hanabi_sad_amd/dist.py ReplayLink) over RCCL, timed
    per section with HIP events on the learner's exchange stream.  Rank 0 is the dedicated learner (empty shard), every other rank
    an actor whose shard holds 2,048 synthetic 80-step sequences in the real transition layout (2p SAD: bit-packed 838-plane
    observation); a parameter round ([online | target] = 2 x 4.9 M floats) every 10th round.  Collective: every rank calls it."""
    from types import SimpleNamespace
    from hanabi_sad_amd.actor import transition_fields
    from hanabi_sad_amd.dist import ReplayLink
    from hanabi_sad_amd.replay import DeviceReplay
    T, n_seq, n_param = 80, 2048, 2 * 4_883_000
    fields = transition_fields(SimpleNamespace(P=2, F=838, A=21, H=5, knowledge_mode=0), vdn=False)
    shard = DeviceReplay(4096, 100 + rank, 0.9, 0.6, 0, T, fields, dev)
    if rank != 0:
        g = torch.Generator(device=dev).manual_seed(rank)
        for lo in range(0, n_seq, 256):
            n = 256
            f = {"priv_s": (torch.rand(n, T, 838, device=dev, generator=g) < 0.15).float(),
                 "legal_move": (torch.rand(n, T, 21, device=dev, generator=g) < 0.4).float(),
                 "eps": torch.rand(n, T, 1, device=dev, generator=g), "own_hand": torch.zeros(n, T, 15, device=dev),
                 "a": torch.zeros(n, T, 1, dtype=torch.int64, device=dev), "greedy_a": torch.zeros(n, T, 1, dtype=torch.int64, device=dev)}
            z = torch.zeros(n, T, device=dev)
            shard.add(f, z, z.to(torch.uint8), z + 1, torch.full((n,), float(T), device=dev), torch.rand(n, device=dev, generator=g) + 0.1)
    shard.set_field_output("priv_s", "bf16", 896)
    # star rounds are pipelined like selfplay.run_link_learner runs them: three rounds open at once (the reference's prefetch depth)
    ahead = int(os.environ.get("HSAD_BENCH_ROUNDS_AHEAD", "3")) if mode == "star" else 1
    link = ReplayLink(shard, batch, 0.6, dev, learner_rank=0, depth=2, param_numel=n_param, mode=mode, name="bench_" + mode, ahead=ahead)
    torch.cuda.synchronize()
    out = None
    if rank == 0:
        prios = []
        link.stage_params(torch.zeros(n_param, device=dev))
        t0 = None
        for _ in range(ahead - 1):
            link.begin(None)
        lag = 2 if ahead == 1 else 1        # (one round open at a time: the priorities of batch r leave with round r + 2, as in rounds 1-3)
        for r in range(rounds + 5):
            if r == 5:                      # five warm-up rounds (communicator set-up, first-touch allocations)
                torch.cuda.synchronize()
                link.timer, link.timer_down = type(link.timer)(dev), type(link.timer)(dev)
                link.wait_ms, link.wait_n = 0.0, 0
                t0 = time.perf_counter()
            link.begin(prios.pop(0) if len(prios) >= lag else None, params=(r % 10 == 5), stop=(r == rounds + 4))
            (f, *_), w = link.finish()
            prios.append(torch.rand(batch, device=dev) + 0.05)
        torch.cuda.synchronize()
        wall_end = time.perf_counter()
        while link._rounds:
            link.finish()
        torch.cuda.synchronize()
        wall = (wall_end - t0) / rounds * 1e3
        out = {"world": world, "rounds": rounds, "batch": batch, "round_shape": link.mode, "transport": link.transport, "transport_decision": link.transport_decision,
               "rounds_open_at_once": ahead,
               "wire_bytes_per_sequence": shard.wire_bytes(),
               "batch_bytes_per_rank_message": shard.wire_bytes() * batch, "param_bucket_bytes": n_param * 4,
               "round_wall_ms": wall, "per_round_ms": link.timings(),
               "note": "sections are HIP-event times on the learner's exchange stream, averaged over all rounds (param_send_ms / "
                       "param_bcast_ms: the parameter rounds' time spread over all rounds); actors serve a round between two polls "
                       "of the store; round_shape star = point-to-point learner <-> actor messages only (HSAD_LINK_MODE=collective: "
                       "the world-wide collectives of rounds 1-2)"}
    else:
        while True:
            flags = link.poll()
            if flags is None:
                time.sleep(0.0002)        # an actor step would run here; keeps N - 1 ranks from hammering the rendezvous store
                continue
            if link.serve(flags):
                break
        torch.cuda.synchronize()
    shard.check_errors()
    link.close()             # ipc transport: unmap the peers' arenas, free the landing ring (two links per A/B run)
    return out


def env_config4_bench(dev, games=16384, steps=100, warmup=50, chunk=50, sad=False):
    """BASELINE configs[4] at its per-GPU size (131,072 games over 8 GPUs = 16,384 per GPU): 5 players, hand 4, colour shuffle --
    literally (no SAD: F = 1380, A = 49; SURVEY §8d "Config 5") or as the SAD variant (F = 1439) -- through the same persistent
    rollout kernel, timed with events like the headline run"""
    from hanabi_sad_amd import BatchedHanabiEnv
    env = BatchedHanabiEnv(games, players=5, hand_size=4, seed=7, eps_list=EPS, max_len=80, sad=sad, shuffle_color=True, device=dev,
                           track_deck_history=False)
    env.set_rollout_chunk(chunk)
    env.rollout_random(warmup, 99)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.rollout_random(steps, 99)
    e1.record()
    torch.cuda.synchronize()
    env.check_errors()
    it_ms = e0.elapsed_time(e1) / steps
    bps = algorithmic_bytes_per_step(env.P, env.F, env.A, env.H, sad)
    gbs = bps * games / (it_ms * 1e-3) / 1e9
    out = {"value": games / (it_ms * 1e-3), "unit": "env-steps/s", "iteration_ms": it_ms, "games": games,
           "config": {"workload": "BASELINE configs[4] per GPU: %d concurrent 5-player games (hand 4, %s, colour shuffle; F=%d A=%d), "
                                  "random-legal policy, persistent fused kernel, %d iterations per launch"
                                  % (games, "SAD" if sad else "no SAD", env.F, env.A, chunk),
                      "games_per_workgroup": env.games_per_workgroup},
           "roofline": {"bound": "hbm", "kernel": "env_rollout_kernel<5,4>", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "traffic": env5_traffic_bytes(games, chunk, sad), "algorithmic_bytes_per_env_step": bps}}
    del env
    torch.cuda.empty_cache()
    return out


def cpu_baseline(seconds=8.0):
    """The CPU oracle (port of the reference algorithm; the reference binary is unbuildable here: HLE
    submodule absent) in the reference's config-1 shape: 1 thread, 80 games, max_len 80, random policy."""
    from oracle.oracle import OracleVecEnv
    import multiprocessing as mp

    def run(seed, secs, q=None):
        v = OracleVecEnv(80, seed, fast=True, players=PLAYERS, hand_size=HAND, eps_list=EPS, max_len=80)
        v.rollout(20, 3)  # warm-up
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < secs:
            n += v.rollout(50, 3)
        rate = n / (time.perf_counter() - t0)
        if q is not None:
            q.put(rate)
        return rate, n

    one, n_one = run(1, seconds)
    ncores = os.cpu_count() or 1
    q = mp.Queue()
    procs = [mp.Process(target=run, args=(1000 * (i + 1), seconds, q)) for i in range(ncores)]
    for p in procs:
        p.start()
    rates = [q.get() for _ in procs]
    for p in procs:
        p.join()
    return {
        "value": one, "unit": "env-steps/s", "cores": 1, "kind": "port",
        "sample": "oracle/hanabi_oracle.cc (-O3 -march=native), 1 thread x 80 games (configs[0] shape), "
                  "%d env-steps in %.0f s wall" % (n_one, seconds),
        "all_cores_value": sum(rates), "all_cores": ncores,
    }



def device_state(local_dev=0):
    """clocks, power and partition mode of this rank's device as rocm-smi reports them (the GPU box's own tool; None where it is missing or
    a field is not reported): recorded in front of and behind the timed region so that a reader can tell a slow box / a power-capped or
    partitioned device from a slow kernel"""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        raw = subprocess.run([exe, "-c", "-P", "-p", "--showmaxpower", "--showcomputepartition", "--showmemorypartition", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        doc = json.loads(raw[raw.index("{"):])
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
    card = doc.get("card%d" % local_dev) or (list(doc.values())[0] if doc else {})
    want = ("sclk", "mclk", "fclk", "socclk", "power", "performance level", "partition", "temperature")
    return {k: v for k, v in card.items() if any(w in k.lower() for w in want)}


class DeviceSampler:
    """rocm-smi polled from a thread WHILE a region runs (clocks, power, temperatures as the tool prints them): the state the device is
    actually in under the load, next to the before / after readings"""

    def __init__(self, local_dev=0):
        import threading
        self.dev, self.samples, self._stop = local_dev, [], False
        self.t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import shutil
        import subprocess
        exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        while not self._stop:
            try:
                raw = subprocess.run([exe, "-c", "-P", "-t", "--json"], capture_output=True, text=True, timeout=10).stdout
                doc = json.loads(raw[raw.index("{"):])
                card = doc.get("card%d" % self.dev) or (list(doc.values())[0] if doc else {})
                self.samples.append({k: v for k, v in card.items() if any(w in k.lower() for w in ("sclk clock speed", "mclk clock speed", "fclk clock speed", "power (w)", "temperature"))})
            except Exception:
                return

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self.t.join(timeout=15)
        return False


def streaming_ceilings(dev, nbytes=448 * 1024 * 1024, reps=15):
    """what THIS device writes / copies when nothing but a streaming kernel runs: a fill of `nbytes` (the env kernel's traffic per iteration
    at the headline shape is 448 MB, 98 % of it writes) and a copy of the same size, one HIP event pair each, median of `reps`.  The env
    kernel's fraction of the 8 TB/s spec figure next to its fraction of this in-run write ceiling separates the box from the kernel."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    b = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out = {}
    for name, fn, moved in (("write", lambda: a.fill_(1.0), nbytes), ("copy", lambda: b.copy_(a), 2 * nbytes)):
        fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        ms.sort()
        out[name] = moved / (ms[len(ms) // 2] * 1e-3) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return out

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-learner", action="store_true", help="skip the R2D2 learner samples/sec measurement")
    ap.add_argument("--no-actor", action="store_true", help="skip the agent-in-the-loop actor measurement")
    ap.add_argument("--kernel-samples", type=int, default=50)
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, one GPU per rank; what the driver uses) | gloo (smoke-testing the multi-rank path with "
                         "several ranks sharing one GPU)")
    ap.add_argument("--chunk", type=int, default=50,
                    help="persistent rollout: iterations of every game per launch (hsad_env_set_rollout_chunk); 0 = one launch "
                         "per iteration and partition (--partitions / --lock-us).  Results are bit-identical either way")
    ap.add_argument("--partitions", type=int, default=3,
                    help="independent game ranges the rollout runs on private HIP streams (hsad_env_set_partitions); results "
                         "are bit-identical for any value")
    ap.add_argument("--lock-us", type=int, default=30,
                    help="phase lock between the partition chains (hsad_env_set_rollout_stagger): partition k starts each "
                         "launch this long after partition k-1, so one partition's logic phase overlaps the others' HBM stream")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one GPU per rank with the RCCL backend: fewer devices than ranks is said in ONE JSON line (rank 0 / the spawning parent) with rc 2 --
    # not as a rank dying in hipSetDevice and the launcher's traceback
    if args.gpus > 1 and args.dist_backend == "nccl" and torch.cuda.device_count() < args.gpus:
        if rank == 0:
            print(json.dumps({"metric": "hanabi_env_steps_per_sec", "value": None, "unit": "env-steps/s", "n_gpus": args.gpus,
                              "error": "--gpus %d with the nccl (= RCCL) backend needs %d devices, this node shows %d; "
                                       "--dist-backend gloo smoke-tests the multi-rank path with ranks sharing a GPU"
                                       % (args.gpus, args.gpus, torch.cuda.device_count())}), flush=True)
        raise SystemExit(2)
    stage = ["start"]           # where the run is: what a watchdog line names
    if world > 1:
        # per-rank watchdog: a rank that is wedged (a peer that never arrives at a barrier, a collective that never completes) ends with a
        # JSON line carrying `error` on rank 0 -- and every rank with an exit code -- instead of sitting in the driver's timeout
        import threading
        limit = int(os.environ.get("HSAD_BENCH_TIMEOUT", "900"))

        def wedged():
            if rank == 0:
                print(json.dumps({"metric": "hanabi_env_steps_per_sec", "value": None, "unit": "env-steps/s", "n_gpus": world,
                                  "error": "rank 0 gave up after %d s in stage '%s' (HSAD_BENCH_TIMEOUT)" % (limit, stage[0])}), flush=True)
            else:
                print("bench.py: rank %d gave up after %d s in stage '%s'" % (rank, limit, stage[0]), file=sys.stderr, flush=True)
            os._exit(3)
        run_watchdog = threading.Timer(limit, wedged)
        run_watchdog.daemon = True
        run_watchdog.start()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the ranks ourselves (one process per GPU, torch.distributed.run on 127.0.0.1) and
        # pass their output through -- rank 0 prints the JSON line
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local_dev = local_rank % max(1, torch.cuda.device_count()) if args.dist_backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = "cuda:%d" % local_dev
    dist = None
    stage[0] = "init_process_group"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from hanabi_sad_amd import BatchedHanabiEnv
    from hanabi_sad_amd.dist import shard_range, shard_seed
    G = args.games  # per GPU (weak scaling): rank r owns global games [r*G, (r+1)*G)
    begin, _ = shard_range(G * world, rank, world)
    env = BatchedHanabiEnv(G, players=PLAYERS, hand_size=HAND, seed=shard_seed(1, begin), eps_list=EPS, max_len=80,
                           sad=False, device=dev, track_deck_history=False)
    persistent = args.chunk > 0
    if persistent:
        env.set_rollout_chunk(args.chunk)
        env.set_rollout_stagger(0)
    else:
        env.set_partitions(args.partitions)
        env.set_rollout_stagger(args.lock_us if args.partitions > 1 else 0)
    policy_seed = 12345 + rank

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    state_before = device_state(local_dev) if rank == 0 else None
    stage[0] = "warm-up + barrier"
    if os.environ.get("HSAD_BENCH_WEDGE_RANK") == str(rank) and world > 1:      # (fault injection for the watchdog's test: this rank never arrives)
        time.sleep(10 ** 6)
    env.rollout_random(args.warmup, policy_seed)
    barrier()
    stage[0] = "timed region"

    # the timed region launches the fused kernel (env_kernel<3,P,H>: reset-terminated + random-legal policy + step +
    # observe) once per iteration and partition; HIP events on the launch stream(s) give its average duration over exactly
    # this region
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    k0.record()
    env.rollout_random(args.steps, policy_seed)
    k1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    iter_ms = k0.elapsed_time(k1) / args.steps                      # all partitions of one iteration (they overlap)
    # spread (not part of the contract's timed region): the same K-step region again and again until >= 0.5 s of GPU time
    rep_ms = []
    sampler = DeviceSampler(local_dev) if rank == 0 else None
    if sampler is not None:
        sampler.__enter__()
    while sum(rep_ms) < (1500.0 if rank == 0 else 500.0) or len(rep_ms) < 5:      # (rank 0: long enough for a few rocm-smi readings under load)
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        env.rollout_random(args.steps, policy_seed)
        r1.record()
        torch.cuda.synchronize()
        rep_ms.append(r0.elapsed_time(r1))
        if len(rep_ms) >= 2000:
            break
    if sampler is not None:
        sampler.__exit__()
    rep_sorted = sorted(x / args.steps for x in rep_ms)
    state_after = device_state(local_dev) if rank == 0 else None
    ceil = streaming_ceilings(dev) if rank == 0 else None
    if persistent:
        # one launch = args.chunk iterations of all G games (the last one shorter if steps is not a multiple), back to back
        # on the caller's stream between the two events: average launch duration = region / launches
        K = 1
        n_launch = (args.steps + args.chunk - 1) // args.chunk
        fused_ms = k0.elapsed_time(k1) / n_launch
        iters_per_launch = args.steps / n_launch
    else:
        K = max(1, args.partitions)
        part_ms = env.last_rollout_ms() if K > 1 else [iter_ms]     # events on the partition streams themselves
        fused_ms = sum(part_ms) / len(part_ms)
        n_launch, iters_per_launch = args.steps * K, 1.0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    env.check_errors()

    # the API's separate step kernel (actions from HBM) for reference: one event pair per launch.  The two records put
    # barrier / timestamp packets around the kernel; what an EMPTY pair measures under the same conditions is subtracted.
    def pairs(with_step):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.kernel_samples)]
        for e0, e1 in ev:
            env.reset()
            a, g = env.policy_random(policy_seed)
            e0.record()
            if with_step:
                env.step(a, g)
            e1.record()
        torch.cuda.synchronize()
        return sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev)
    pairs(True)
    step_raw_ms, pair_overhead_ms = pairs(True), pairs(False)
    step_ms = step_raw_ms - pair_overhead_ms
    env.check_errors()
    bytes_per_step = algorithmic_bytes_per_step(env.P, env.F, env.A, env.H, False)
    achieved_step = bytes_per_step * G / (step_ms * 1e-3) / 1e9
    achieved = bytes_per_step * G / (iter_ms * 1e-3) / 1e9        # all concurrent launches together = the chip's rate
    per_launch = bytes_per_step * (G / K) * iters_per_launch / (fused_ms * 1e-3) / 1e9

    out = None
    if rank == 0:
        out = {
            "metric": "hanabi_env_steps_per_sec",
            "value": world * G * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": {"regions": len(rep_ms), "steps_per_region": args.steps, "gpu_seconds": sum(rep_ms) / 1e3,
                        "ms_per_step_median": rep_sorted[len(rep_sorted) // 2], "ms_per_step_min": rep_sorted[0], "ms_per_step_max": rep_sorted[-1],
                        "value_at_median_per_gpu": G / (rep_sorted[len(rep_sorted) // 2] * 1e-3),
                        "note": "the timed region repeated (HIP events, this rank) until >= 0.5 s of GPU time; `value` is the contract's single region"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: %d concurrent 2-player Hanabi games per GPU, random-legal policy, fused "
                            "reset-terminated + policy + step + observe kernel, %s, fp32 obs [G,2,783] written to HBM every "
                            "step" % (G, ("persistent launches of %d iterations each" % args.chunk) if persistent else
                                      ("%d phase-locked stream partition(s) per iteration" % K)),
                "rollout_chunk": args.chunk if persistent else 0,
                "partitions": K, "phase_lock_us": args.lock_us if (K > 1 and not persistent) else 0,
                "games_per_gpu": G, "players": PLAYERS, "hand_size": HAND, "feature_size": env.F,
                "num_action": env.A, "max_len": 80, "sharding": "games sharded across ranks, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("env_rollout_kernel<2,5> (persistent fused reset-terminated + policy + step + observe; the only kernel "
                           "in the timed region: one launch = %d iterations of all %d games)" % (args.chunk, G)) if persistent else
                          ("env_kernel<3,2,5> (fused reset-terminated + policy + step + observe; the only kernel in the timed "
                           "region: %d launch(es) per iteration on %d stream partition(s), overlapping in time)" % (K, K)),
                # achieved = algorithmic bytes of the timed region / its duration (HIP events on the caller's stream) = what the
                # chip sustains; per_launch_* = one launch against its own duration -- the number rocprofv3's AverageNs for
                # this kernel must agree with (persistent: the same thing, launches run back to back on one stream)
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic_bytes(G, 3, iters_per_launch if persistent else 0),
                "algorithmic_bytes_per_env_step": bytes_per_step,
                "algorithmic_bytes_per_launch": bytes_per_step * G / K * iters_per_launch, "avg_launch_ms": fused_ms,
                "launches": n_launch, "iterations_per_launch": iters_per_launch,
                "launches_per_iteration": K if not persistent else 1.0 / iters_per_launch, "iteration_ms": iter_ms,
                "per_launch_achieved": per_launch,
                # spread of the timed region on this device (the `repeats` block above, kept here where a parser that only keeps the
                # contract's keys still finds it) and this device's own streaming ceilings measured in this run
                "ms_per_step_median": rep_sorted[len(rep_sorted) // 2], "ms_per_step_min": rep_sorted[0], "ms_per_step_max": rep_sorted[-1],
                "regions_repeated": len(rep_ms),
                "achieved_at_median": bytes_per_step * G / (rep_sorted[len(rep_sorted) // 2] * 1e-3) / 1e9,
                "write_ceiling_gbs": ceil["write"], "copy_ceiling_gbs": ceil["copy"],
                "frac_of_write_ceiling": achieved / ceil["write"],
                "frac_of_write_ceiling_at_median": bytes_per_step * G / (rep_sorted[len(rep_sorted) // 2] * 1e-3) / 1e9 / ceil["write"],
                "ceiling_note": "write_ceiling = torch fill_ of 448 MB (one iteration's traffic at this shape, 98 % writes), copy_ceiling = "
                                "copy_ of the same size (read + write bytes), median of 15 event pairs each, measured in this run on this "
                                "device right behind the timed region",
                "launch_boundary_note": "one launch = %d iterations; a %d-step region is %d launch(es): short regions pay the launch "
                                        "boundary (pipeline fill + tail) once per launch -- DESIGN 3a measures 80 us / iteration at 10 "
                                        "iterations per launch against 68 us at >= 50" % (args.chunk, args.steps, n_launch) if persistent else None,
                "device_state_before": state_before, "device_state_after": state_after,
                "device_state_during_repeats": sampler.samples[:6] if sampler is not None else None,
            },
            "roofline_step_kernel": {
                "bound": "hbm", "kernel": "env_kernel<1,2,5> (HanabiEnv::step + observe with actions from HBM, the "
                                          "hsad_env_step entry point)",
                "achieved": achieved_step, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_step / HBM_PEAK_GBS,
                "traffic": measured_traffic_bytes(G, 1), "avg_launch_ms": step_ms,
                "event_pair_ms": step_raw_ms, "empty_event_pair_ms": pair_overhead_ms,
            },
        }
    if world > 1:
        # who is in the job: every rank's device and what the collective backend itself reports
        info = [None] * world
        import torch.distributed as _d
        _d.all_gather_object(info, {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(),
                                    "pid": os.getpid()})
        if rank == 0:
            out["ranks"] = info
            out["backend"] = args.dist_backend + (" (= RCCL)" if args.dist_backend == "nccl" else "")
            out["rccl_ranks"] = _d.get_world_size() if args.dist_backend == "nccl" else None
    stage[0] = "exchange leg"
    if world > 1 and not os.environ.get("HSAD_BENCH_NO_EXCHANGE"):
        # the exchange leg is collective: a rank that never arrives would leave the others inside a collective for good and the
        # headline number, already measured, unprinted.  A watchdog prints the line without the leg and ends the process instead.
        import threading

        def give_up():
            if rank == 0:
                out["exchange" if "exchange" not in out else "exchange_ab"] = {
                    "error": "the exchange leg did not finish within %d s" % EXCHANGE_TIMEOUT_S}
                print(json.dumps(out), flush=True)
            os._exit(0)
        watchdog = threading.Timer(EXCHANGE_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        del env
        torch.cuda.empty_cache()
        # the product's round shape first (point-to-point), then its A/B twin (world-wide collectives); a leg that finished is in
        # `out` before the next one starts, so the watchdog's line keeps it
        for key, mode in (("exchange", os.environ.get("HSAD_LINK_MODE", "star")), ("exchange_ab", None)):
            if mode is None:
                mode = "collective" if out_mode == "star" else "star"
            out_mode = mode
            try:
                exchange = exchange_bench(dev, rank, world, mode=mode)
            except Exception as e:               # (a failure on ONE rank: the others meet the watchdog)
                exchange = {"error": "%s: %s" % (type(e).__name__, e)}
            if rank == 0:
                out[key] = exchange
            torch.cuda.empty_cache()
        watchdog.cancel()
    if rank == 0:
        if world == 1 and not args.no_actor:
            out["env_configs4"] = env_config4_bench(dev, sad=False)
            out["env_configs4_sad"] = env_config4_bench(dev, sad=True)
        if world == 1 and not args.no_learner:
            out["gemm_core"] = gemm_core_bench(dev)
            out["learner"] = learner_bench(dev)
        if world == 1 and not args.no_actor:
            out["actor"] = actor_bench(dev)
            out["one_gpu_training"] = training_bench(dev, extra_args=tuple(os.environ.get("HSAD_BENCH_TRAINING_ARGS", "").split()))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        # what in this line is NOT measured by this run: the HBM-traffic and MFMA-busy counter figures are read from the committed rocprofv3
        # PMC passes (a PMC pass cannot run inside the driver's command); every time, rate and fraction is measured here
        out["profile_sources"] = {"every `traffic` field": "profiles/" + (pmc_leg("env")[1] or "-") + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/collect_profiles.sh)",
                                  "every `mfma_busy_counter` field": "profiles/r06_mfma_util.json or older (rocprofv3 --pmc SQ_* pass, tools/mfma_util.sh; the `source` key of each says which)",
                                  "everything else": "measured in this run"}
        print(json.dumps(out))
    stage[0] = "final barrier"
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
