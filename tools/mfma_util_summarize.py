"""tools/mfma_util.sh -> profiles/<round>_mfma_util.json.

  python tools/mfma_util_summarize.py out.json <leg>:<counter_collection.csv>:<kernel_stats.csv> ...

Per kernel: the raw SQ counters (mean per dispatch, summed over the chip by rocprofv3), the un-instrumented average duration of the same
command (kernel_stats of a --stats pass), the kernel's FLOP per launch and two utilisation figures:
  flop_frac   = FLOP / duration / 2.5 PFLOP/s                       (what bench.py's roofline objects report)
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x elapsed shader cycles), elapsed cycles = GRBM_GUI_ACTIVE of the dispatch
                -- the share of SIMD-cycles in which the matrix pipe was busy, a COUNTER, independent of any FLOP bookkeeping
The `calib` leg (a GEMM of exactly known MFMA count) pins the counter's meaning in this tool chain: busy cycles per wave-level
v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md: 32 = 8 passes x 4 cycles) -- if that does not come out, the summary says so
instead of quoting utilisations."""
import csv, json, re, sys
from collections import defaultdict

PEAK = 2.5e15
CUS, SIMDS = 256, 4
_MH = 80 * 128 * 512
FLOPS = {   # FLOP per launch (DESIGN section 3d / VERDICT r3's arithmetic)
    "calib": [("gemm8_kernel<1", 2.0 * 8192 ** 3), ("gemm_nt_bf16_kernel<128, 128>", 2.0 * 8192 ** 3)],     # the 256 x 256 core (round 5), or the 128 x 128 kernel (HSAD_GEMM_PP=0)
    # fused forward: 2 nets x 2 layers x 80 steps x (128 rows x 2048 gates x (512 + 512) k) x 2
    "learner": [("lstm_fused_fwd_kernel", 2 * 2 * 80 * 128 * 2048 * 1024 * 2.0),
                # BPTT: per step dh = dG W_hh^T (both layers), dO = dG1 W_ih1 (projection stage), dx = dG0 W_ih0 (sink): 4 stages x 128 x 2048 x 512 x 2
                ("lstm_bptt_wide_kernel", 4 * 80 * 128 * 2048 * 512 * 2.0),      # round 6 blocking (16 rows x 64 units)
                ("lstm_fused_bwd_kernel", 4 * 80 * 128 * 2048 * 512 * 2.0)],
    # one launch = online + target net's cell of one layer: 2 x 32768 rows x 2048 x 1024 x 2
    "actor": [("gemm8_kernel<2", 2 * 32768 * 2048 * 1024 * 2.0)],      # gemm8_kernel<G8_CELL>
}


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
    return out


out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                  "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -- python tools/mfma_probe.py <leg>; durations from a separate "
                  "--kernel-trace --stats pass of the same command", "peak_flops": PEAK}
for spec in sys.argv[2:]:
    leg, cpath, spath = spec.split(":")
    cnt, dur = counters(cpath), durations(spath)
    rec = {}
    for needle, flop in FLOPS[leg]:
        names = [k for k in cnt if needle in k]
        if not names:
            continue
        name = max(names, key=lambda k: len(cnt[k].get("SQ_WAVE_CYCLES", [])))
        c = {k: sum(v) / len(v) for k, v in cnt[name].items()}
        dn = [k for k in dur if needle in k]
        avg_ns, calls = dur[max(dn, key=lambda k: dur[k][1])] if dn else (None, 0)
        r = {"kernel": name[:160], "dispatches_counted": len(cnt[name].get("SQ_WAVE_CYCLES", [])), "counters_mean_per_dispatch": c,
             "flop_per_launch": flop, "avg_duration_us_uninstrumented": None if avg_ns is None else avg_ns / 1e3, "calls_timed": calls}
        if avg_ns:
            r["flop_frac"] = flop / (avg_ns * 1e-9) / PEAK
        gui, busy = c.get("GRBM_GUI_ACTIVE"), c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if gui and busy is not None:
            # GRBM_GUI_ACTIVE: cycles the dispatch kept the GPU busy; rocprofv3 may report it summed over the 8 XCDs -- both normalisations
            # are given, `calib` below says which one is consistent with the known MFMA count
            r["mfma_busy_frac_if_gui_is_per_chip"] = busy / (SIMDS * CUS * gui)
            r["mfma_busy_frac_if_gui_is_summed_over_8_xcds"] = busy / (SIMDS * CUS * gui / 8.0)
            r["n_mfma_32x32x16_equiv"] = flop / (2.0 * 32 * 32 * 16)
            r["busy_cycles_per_mfma_32x32x16_equiv"] = busy / r["n_mfma_32x32x16_equiv"]
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            r["wave_cycle_shares"] = {k: c[k] / w for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in c}
        rec[needle] = r
    out[leg] = rec
# the calibration's verdict
cal = out.get("calib", {}).get("gemm_nt_bf16_kernel<128, 128>")
if cal and "busy_cycles_per_mfma_32x32x16_equiv" in cal:
    per = cal["busy_cycles_per_mfma_32x32x16_equiv"]
    out["calibration"] = {"busy_cycles_per_v_mfma_f32_32x32x16_bf16": per, "guide_value": 32.0,
                          "consistent": bool(0.9 * 32 <= per <= 1.1 * 32),
                          "note": "with 32 busy cycles per instruction, 100 % MFMA-busy on 1,024 SIMDs = 1024 / 32 x 32768 FLOP per cycle = "
                                  "2.5 PFLOP/s at 2.4 GHz: mfma_busy = FLOP-equivalent utilisation AT THE CLOCK THE KERNEL RAN AT, "
                                  "flop_frac = the same against the 2.4 GHz peak"}
    # utilisation from busy cycles alone, no GRBM needed: busy / (1024 SIMDs x duration x clock) is clock-dependent, so report per-kernel
    # the implied sustained clock instead: clock = busy_frac-independent check  flop_frac / mfma_busy
    for leg in ("learner", "actor", "calib"):
        for k, r in out.get(leg, {}).items():
            if r.get("avg_duration_us_uninstrumented") and "counters_mean_per_dispatch" in r:
                busy = r["counters_mean_per_dispatch"].get("SQ_VALU_MFMA_BUSY_CYCLES")
                if busy:
                    r["mfma_busy_frac_at_2p4GHz_wall"] = busy / (SIMDS * CUS * r["avg_duration_us_uninstrumented"] * 1e-6 * 2.4e9)
json.dump(out, open(sys.argv[1], "w"), indent=1)
for leg, rec in out.items():
    if isinstance(rec, dict) and leg not in ("calibration",):
        for k, r in rec.items():
            if isinstance(r, dict):
                print(leg, k, {x: (round(r[x], 4) if isinstance(r[x], float) else r[x]) for x in r if x.startswith(("flop_frac", "mfma_busy", "busy_cycles", "avg_dur"))})
print("calibration", out.get("calibration"))
