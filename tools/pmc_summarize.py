"""Summarise rocprofv3 PMC passes (WRITE_SIZE and FETCH_SIZE, each collected in its OWN run with --kernel-trace only, as
MI355X_MICROARCH.md §HBM prescribes) of tools/pmc_probe.py legs into profiles/<round>_pmc_hbm_traffic.json.

  python tools/pmc_summarize.py <out.json> <leg>:<write_counter_collection.csv>:<fetch_counter_collection.csv> ...

Corrections (the guide: FETCH_SIZE reports 1/2 of a wide coalesced read on gfx950; WRITE_SIZE and other access widths are
UNCALIBRATED -- calibrate on a known byte count in the same run): every leg ends with a streaming fill and a copy of exactly
CAL_BYTES; write_factor = CAL_BYTES / WRITE_SIZE(fill), fetch_factor = CAL_BYTES / FETCH_SIZE(copy).  The factors must come
out near 1 (WRITE_SIZE in KiB) and near 2: anything else means the calibration kernel was not found or the counter changed
meaning, and this script FAILS instead of guessing."""
import csv, json, re, sys
from collections import defaultdict

CAL_BYTES = 65536 * 2 * 783 * 4
G, P, F, A, H = 65536, 2, 783, 21, 5
ALGO_ENV = (P * (F + A + 3 * H + 1) * 4 + 5 + P * 8 + 256) * G     # SURVEY.md §8(d) bytes per env-step x G
GEMM_ALGO = 10240 * 512 * 2 + 2048 * 512 * 2 + 10240 * 2048 * 4    # A + B read (bf16), C written (fp32)
ALGO_ENV5 = (5 * (1439 + 49 + 12 + 1) * 4 + 5 + 5 * 8 * 2 + 256) * 16384        # configs[4] per GPU: 5p hand 4 SAD, 16,384 games
ALGO_ENV5_LITERAL = (5 * (1380 + 49 + 12 + 1) * 4 + 5 + 5 * 8 + 256) * 16384     # configs[4] literally: no SAD
# fused forward recurrence of a learner update (T=80, B=128, H=512, 2 nets x 2 layers): weights 8 x 2 MB, inputs 2 x 10 MB, outputs: bf16 h
# of 4 recurrences, fp32 gates + c of the online net's 2 layers (the h tiles exchanged between workgroups stay in L2: not algorithmic HBM bytes)
_MH = 80 * 128 * 512
FUSED_FWD_ALGO = 8 * 2048 * 512 * 2 + 2 * _MH * 2 + 4 * _MH * 2 + 2 * (_MH * 4 * 4 + _MH * 4)
# fused BPTT (online net, 2 layers): weights 3 x 2 MB, saved gates + c read (2 layers), dO of the top layer (fp32), dG written (bf16, 2 layers)
# fused inference cell, 32,768 rows, H = 512: x and h_prev bf16 read once, the [2048 x 1024] weight panel once, c_prev fp32 read; written: the
# bf16 layer output (+ c and h fp32 for the online pass)
_NH = 32768 * 512
CELL_ALGO_NOSTATE = 2 * _NH * 2 + 2048 * 1024 * 2 + _NH * 4 + _NH * 2
CELL_ALGO_STATE = CELL_ALGO_NOSTATE + 2 * _NH * 4
# default schedule (four pipeline stages): four weight slices; per layer the saved gates (fp32 x 4) and c of two steps; dO of the top layer,
# the ReLU mask of the sink; written: dG transposed (2 layers, bf16), the written-through second tile copies of both layers, the fp32 dO rows
# of the projection stage (written once, read once), d x1 transposed
FUSED_BWD_ALGO = (4 * 2048 * 512 * 2 + 2 * (_MH * 4 * 4 + 2 * _MH * 4) + _MH * 4 + _MH * 2 + 2 * _MH * 4 * 2 + 2 * _MH * 4 * 2 + 2 * _MH * 4 + _MH * 2)


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def pick(acc, pattern):
    out = []
    for k, v in acc.items():
        if re.search(pattern, k):
            out += v
    return out


def calibration(w, f):
    fill = [v for v in pick(w, r"FillFunctor<float>") if v > 0.5 * CAL_BYTES / 1024]      # only the CAL_BYTES fills
    copy = [v for v in pick(f, r"direct_copy_kernel|copyBuffer") if v > 0.2 * CAL_BYTES / 1024]
    if not fill or not copy:
        raise SystemExit("calibration kernels not found (fill: %d, copy: %d dispatches of the right size)" % (len(fill), len(copy)))
    wf = (CAL_BYTES / 1024.0) / (sum(fill) / len(fill))
    ff = (CAL_BYTES / 1024.0) / (sum(copy) / len(copy))
    if not (0.9 < wf < 1.1) or not (1.8 < ff < 2.2):
        raise SystemExit("implausible calibration: write_factor %.3f (want ~1), fetch_factor %.3f (want ~2)" % (wf, ff))
    return {"cal_bytes": CAL_BYTES, "fill_WRITE_SIZE_KiB": sum(fill) / len(fill), "write_factor": wf,
            "copy_FETCH_SIZE_KiB": sum(copy) / len(copy), "fetch_factor": ff,
            "note": "factors = known bytes / counter, measured in this run on a %d-byte streaming fill (WRITE_SIZE, KiB) and copy "
                    "(FETCH_SIZE, KiB; wide coalesced reads count half on gfx950)" % CAL_BYTES}


def mean(v):
    return sum(v) / len(v) if v else None


KERNELS = {   # leg -> [(label, name regex, algorithmic bytes per launch or None, note)]
    "env": [("env_rollout_kernel<2,5> persistent fused reset+policy+step+observe, G=65536, 50 iterations per launch",
             r"env_rollout_kernel<2, 5>", ALGO_ENV * 50, "iterations_per_launch=50"),
            ("env_kernel<3,2,5> fused reset+policy+step+observe (rollout), G=65536 in 3 partition launches", r"env_kernel<3, 2, 5>", ALGO_ENV / 3, ""),
            ("env_kernel<1,2,5> step+observe, G=65536", r"env_kernel<1, 2, 5>", ALGO_ENV, ""),
            ("env_kernel<0,2,5> reset-terminated, G=65536", r"env_kernel<0, 2, 5>", None, "")],
    "env5": [("env_rollout_kernel<5,4> persistent fused rollout, configs[4] per GPU: G=16384 5-player hand-4 SAD + colour shuffle, 50 iterations per launch",
              r"env_rollout_kernel<5, 4>", ALGO_ENV5 * 50, "iterations_per_launch=50")],
    "env5_literal": [("env_rollout_kernel<5,4> persistent fused rollout, configs[4] literally: G=16384 5-player hand-4 colour shuffle, no SAD, "
                      "50 iterations per launch", r"env_rollout_kernel<5, 4>", ALGO_ENV5_LITERAL * 50, "iterations_per_launch=50")],
    "gemm": [("gemm8_kernel<G8_F32> (the 256 x 256 core) LSTM input projection 10240x2048x512, fp32 output", r"gemm8_kernel<1", GEMM_ALGO, ""),
             ("gemm_nt_bf16_kernel<128,128> LSTM input projection 10240x2048x512, fp32 output", r"gemm_nt_bf16_kernel<128, 128>", GEMM_ALGO, "")],
    "learner": [("learner update: lstm_fused_fwd_kernel<16> (2 nets x 2 layers x 80 steps per launch)", r"lstm_fused_fwd_kernel<16>", FUSED_FWD_ALGO, ""),
                ("learner update: lstm_bptt_wide_kernel<64> (round 6: 2 layers + projection + sink stage x 80 steps per launch, 16 rows x 64 units per workgroup)", r"lstm_bptt_wide_kernel<64>", FUSED_BWD_ALGO, ""),
                ("learner update: lstm_fused_bwd_kernel<64> (the 32 x 32 blocking of rounds 3-5; 2 layers x 80 steps per launch)", r"lstm_fused_bwd_kernel<64>", FUSED_BWD_ALGO, ""),
                ("learner update: loss_tail_kernel", r"loss_tail_kernel", None, ""),
                ("learner update: gemm_nt_bf16_kernel<128,128> (all shapes of an update)", r"gemm_nt_bf16_kernel<128, 128>", None, ""),
                ("learner update: gemm_nt_bf16_kernel<128,64>", r"gemm_nt_bf16_kernel<128, 64>", None, ""),
                ("learner update: lstm_seq_fwd_kernel<16> (4 recurrences x 20 steps per launch)", r"lstm_seq_fwd_kernel<16>", None, ""),
                ("learner update: lstm_seq_bwd_kernel<64> (2 recurrences x 20 steps per launch)", r"lstm_seq_bwd_kernel<64>", None, ""),
                ("learner update: transpose_bf16_kernel", r"transpose_bf16_kernel", None, ""),
                ("learner update: sum_slabs_kernel", r"(?<!g8_)sum_slabs_kernel", None, ""),
                ("learner update: gemm8_kernel<G8_F32> (the five weight-gradient problems, one grouped split-K launch)", r"gemm8_kernel<1", None, ""),
                ("learner update: gemm8_kernel<G8_BF16> (input layer, online + target)", r"gemm8_kernel<0", None, ""),
                ("learner update: g8_sum_slabs_kernel", r"g8_sum_slabs_kernel", None, ""),
                ("learner update: adam_kernel", r"adam_kernel", None, "")],
    "actor": [("actor step: gemm8_kernel<G8_CELL> (one launch = the online AND the target net's cell of a layer: 2 x 32,768 rows x 2048 x 1024; "
               "online pass writes fp32 state + bf16 output, target pass bf16 output only)", r"gemm8_kernel<2", CELL_ALGO_STATE + CELL_ALGO_NOSTATE, ""),
              ("actor step: gemm8_kernel<G8_BF16> (input layer, online + target net in one launch)", r"gemm8_kernel<0", None, ""),
              ("actor step: gemm_nt_bf16_kernel<128,64> (heads)", r"gemm_nt_bf16_kernel<128, 64>", None, ""),
              ("actor step: env_kernel<1,2,5> G=16384", r"env_kernel<1, 2, 5>", None, ""),
              ("actor step: cast_pad_bf16_vec8_kernel", r"cast_pad_bf16_vec8_kernel", None, ""),
              ("actor step: pack_rows_kernel", r"pack_rows_kernel", None, ""),
              ("actor step: seq_flush_copy_kernel", r"seq_flush_copy_kernel", None, "")],
}

out = {"command": "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python tools/pmc_probe.py <leg> ; the same with "
                  "--pmc FETCH_SIZE (separate passes; no other trace domains)"}
for spec in sys.argv[2:]:
    leg, wpath, fpath = spec.split(":")
    w, f = per_kernel(wpath, "WRITE_SIZE"), per_kernel(fpath, "FETCH_SIZE")
    cal = calibration(w, f)
    rec = {"calibration": cal}
    for label, pat, algo, note in KERNELS[leg]:
        wv, fv = pick(w, pat), pick(f, pat)
        if not wv:
            continue
        hbm = (mean(wv) * cal["write_factor"] + (mean(fv) or 0.0) * cal["fetch_factor"]) * 1024.0
        r = {"dispatches": len(wv), "WRITE_SIZE_KiB": mean(wv), "FETCH_SIZE_KiB_raw": mean(fv), "hbm_bytes_per_launch": hbm,
             "hbm_write_bytes_per_launch": mean(wv) * cal["write_factor"] * 1024.0,
             "hbm_read_bytes_per_launch": (mean(fv) or 0.0) * cal["fetch_factor"] * 1024.0}
        if algo:
            r["algorithmic_bytes_per_launch"] = algo
            r["traffic_over_algorithmic"] = hbm / algo
        if note.startswith("iterations_per_launch="):
            n = int(note.split("=")[1])
            r["iterations_per_launch"] = n
            r["hbm_bytes_per_iteration"] = hbm / n
        rec[label] = r
    out[leg] = rec
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({leg: {k: (v.get("traffic_over_algorithmic"), v["hbm_bytes_per_launch"]) for k, v in rec.items() if k != "calibration"}
                  for leg, rec in out.items() if leg != "command"}, indent=1))
