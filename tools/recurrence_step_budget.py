"""Developer tool: per-step critical path of the two persistent learner recurrences inside real updates.

hsad_lstm_debug_enable(2) makes thread 0 of every workgroup of row block 0 stamp s_memrealtime (100 MHz, one clock for the chip) at its
phase boundaries of every step.  From the stamps of all 16 unit blocks of a (net, layer) group this prints / writes, as medians over the
steps of the sequence and the group's members (and over the traced updates):
  forward  : last producer's signal -> consumer has seen the counter -> h DMA issued (+ second half of the X product) -> tile landed ->
             h MFMAs done -> cell update staged -> stores drained + own signal -> background window done
  backward : last producer's signal -> seen -> tile loaded + MFMAs done -> K-split reduction + cell backward staged -> hand-off stores
             issued -> drained + signalled -> transposed copy done
usage: python tools/recurrence_step_budget.py [out.json] [updates] [fused flags]"""
import os, sys, json, ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hanabi_sad_amd import _lib
from hanabi_sad_amd.composite import CompositeLearner
from hanabi_sad_amd.selfplay import init_weights
from tests.test_r2d2_kernels_gpu import _rand_batch

out_path = sys.argv[1] if len(sys.argv) > 1 else None
NUPD = int(sys.argv[2]) if len(sys.argv) > 2 else 8
F, H, A, T, B = 838, 512, 21, 80, 128
lib = _lib.load_library()
W = init_weights(F, H, A, 5, 1)
L = CompositeLearner(W, W, 3, 0.999, device="cuda:0")
if len(sys.argv) > 3:
    L.set_fused(int(sys.argv[3], 0))
batch, weight = _rand_batch(T, B, F, A)
for _ in range(3):
    L.loss(batch, weight, 0.0); L.optimizer_step()
torch.cuda.synchronize()
KREC, KNB, KT, KK = 6, 16, 96, 12
NW = 2 * KREC * KNB * KT * KK
buf = (C.c_uint64 * NW)()
_lib.check(lib.hsad_lstm_debug_enable(2))
_lib.check(lib.hsad_lstm_debug_trace(buf, NW))
traces = []
for _ in range(NUPD):
    L.loss(batch, weight, 0.0); L.optimizer_step()
    _lib.check(lib.hsad_lstm_debug_trace(buf, NW))
    traces.append(np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(2, KREC, KNB, KT, KK) * 0.01)   # us
_lib.check(lib.hsad_lstm_debug_enable(0))
L.check_sync()
# untraced update time next to it
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.loss(batch, weight, 0.0); L.optimizer_step()
e1.record(); torch.cuda.synchronize()
ms_update = e0.elapsed_time(e1) / 20

def med(x):
    x = np.asarray(x, dtype=np.float64)
    return round(float(np.median(x)), 3) if x.size else None

def forward_budget(rec):
    """rec: record index net * 2 + layer of the forward launch"""
    rows = {k: [] for k in ("signal_skew", "last_signal_to_seen", "seen_to_dma_issued_plus_x_half", "dma_issued_to_landed", "h_mfma", "cell_update_staged",
                            "stores_drained_signalled", "background_window", "step")}
    for tr in traces:
        s = tr[0, rec]                     # [nb][t][k]
        if not s[:, 1:T, 6].all():
            continue
        for t in range(2, T - 1):
            sig = s[:, t - 1, 6]
            rows["signal_skew"].append(sig.max() - sig.min())
            rows["last_signal_to_seen"] += list(s[:, t, 1] - sig.max())
            rows["seen_to_dma_issued_plus_x_half"] += list(s[:, t, 2] - s[:, t, 1])
            rows["dma_issued_to_landed"] += list(s[:, t, 3] - s[:, t, 2])
            rows["h_mfma"] += list(s[:, t, 4] - s[:, t, 3])
            rows["cell_update_staged"] += list(s[:, t, 5] - s[:, t, 4])
            rows["stores_drained_signalled"] += list(s[:, t, 6] - s[:, t, 5])
            rows["background_window"] += list(s[:, t, 7] - s[:, t, 6])
            rows["step"] += list(s[:, t, 6] - s[:, t - 1, 6])
    mean_step = float(np.mean(rows["step"])) if rows["step"] else None
    out = {k: med(v) for k, v in rows.items()}
    out["step_mean"] = round(mean_step, 3) if mean_step else None
    # which members publish late: mean offset of each unit block's signal behind the group's first, steps 20..T-2
    offs = [np.mean([tr[0, rec][:, t, 6] - tr[0, rec][:, t, 6].min() for t in range(20, T - 1)], axis=0) for tr in traces if tr[0, rec][:, 1:T, 6].all()]
    if offs:
        out["member_signal_offset_us"] = [round(float(x), 2) for x in np.mean(offs, axis=0)]
        # and where a late member loses it: mean of each phase per member
        ph = {}
        for name, k0, k1 in (("bg", 6, 7), ("h_mfma", 3, 4), ("cell", 4, 5), ("publish", 5, 6)):
            ph[name] = [round(float(x), 2) for x in np.mean([np.mean([tr[0, rec][:, t, k1] - tr[0, rec][:, t, k0] for t in range(20, T - 1)], axis=0) for tr in traces], axis=0)]
        ph["step_top_to_landed"] = [round(float(x), 2) for x in np.mean([np.mean([tr[0, rec][:, t, 3] - tr[0, rec][:, t, 0] for t in range(20, T - 1)], axis=0) for tr in traces], axis=0)]
        out["member_phase_us"] = ph
    return out

def backward_budget(jrec):
    """jrec: internal record of the BPTT launch (0 top layer, 1 projection stage, 2 lower layer, 3 sink with the default schedule)"""
    rows = {k: [] for k in ("signal_skew", "last_signal_to_poll_satisfied", "poll_satisfied_to_seen_by_all_waves", "last_signal_to_seen", "seen_to_first_quarter_in_registers",
                            "first_quarter_to_whole_tile_in_registers", "whole_tile_to_mfma_done", "seen_to_tile_loaded_mfma_done", "reduction_cell_backward_staged", "handoff_stores_issued",
                            "stores_drained_signalled", "transposed_copy", "loop_top_to_seen", "step")}
    for tr in traces:
        s = tr[1, jrec]
        s = s[s[:, T // 2, 5] > 0]          # the members that ran (the 16-row x 64-unit blocking has 8 unit blocks per group, the 32 x 32 one 16)
        if not len(s) or not s[:, 1:T - 1, 5].all():
            continue
        for t in range(T - 3, 1, -1):       # step t consumes the tiles of step t + 1
            sig = s[:, t + 1, 5]
            rows["signal_skew"].append(sig.max() - sig.min())
            rows["last_signal_to_seen"] += list(s[:, t, 2] - sig.max())
            rows["last_signal_to_poll_satisfied"] += list(s[:, t, 8] - sig.max())
            rows["poll_satisfied_to_seen_by_all_waves"] += list(s[:, t, 2] - s[:, t, 8])
            rows["seen_to_first_quarter_in_registers"] += list(s[:, t, 9] - s[:, t, 2])
            rows["first_quarter_to_whole_tile_in_registers"] += list(s[:, t, 10] - s[:, t, 9])
            rows["whole_tile_to_mfma_done"] += list(s[:, t, 3] - s[:, t, 10])
            rows["seen_to_tile_loaded_mfma_done"] += list(s[:, t, 3] - s[:, t, 2])
            rows["reduction_cell_backward_staged"] += list(s[:, t, 4] - s[:, t, 3])
            rows["handoff_stores_issued"] += list(s[:, t, 7] - s[:, t, 4])
            rows["stores_drained_signalled"] += list(s[:, t, 5] - s[:, t, 7])
            rows["transposed_copy"] += list(s[:, t, 6] - s[:, t, 5])
            rows["loop_top_to_seen"] += list(s[:, t, 2] - s[:, t, 0])
            rows["step"] += list(s[:, t, 5] - s[:, t + 1, 5])
    mean_step = float(np.mean(rows["step"])) if rows["step"] else None
    out = {k: med(v) for k, v in rows.items()}
    out["step_mean"] = round(mean_step, 3) if mean_step else None
    return out

def stage_budget(jrec):
    rows = {"seen_to_product_published": [], "step": []}
    for tr in traces:
        s = tr[1, jrec]
        s = s[s[:, T // 2, 5] > 0]
        if not len(s) or not s[:, 1:T - 1, 5].all():
            continue
        for t in range(T - 3, 1, -1):
            rows["seen_to_product_published"] += list(s[:, t, 5] - s[:, t, 2])
            rows["step"] += list(s[:, t, 5] - s[:, t + 1, 5])
    return {k: med(v) for k, v in rows.items()}

res = {
    "what": "per-step critical path of the persistent learner recurrences inside real updates: medians (us) over steps 2..T-2, the 16 workgroups of "
            "row block 0 of a (net, layer) group and %d traced updates; s_memrealtime stamps (10 ns resolution) of thread 0" % len(traces),
    "config": {"F": F, "H": H, "A": A, "T": T, "B": B, "fused_flags": sys.argv[3] if len(sys.argv) > 3 else "default"},
    "ms_per_update_untraced": round(ms_update, 4),
    "forward": {"online_layer0": forward_budget(0), "online_layer1": forward_budget(1), "target_layer0": forward_budget(2), "target_layer1": forward_budget(3)},
    "backward": {"top_layer": backward_budget(0), "lower_layer": backward_budget(2), "projection_stage": stage_budget(1), "sink_stage": stage_budget(3)},
}
txt = json.dumps(res, indent=1)
print(txt)
if out_path:
    open(out_path, "w").write(txt + "\n")
