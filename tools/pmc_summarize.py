"""Summarise two rocprofv3 PMC passes (WRITE_SIZE, FETCH_SIZE; each collected in its own run with --kernel-trace only,
as MI355X_MICROARCH.md §HBM prescribes) of tools/pmc_probe.py into profiles/<round>_pmc_hbm_traffic.json.

  python tools/pmc_summarize.py <write_counter_collection.csv> <fetch_counter_collection.csv> <out.json>

Corrections: WRITE_SIZE is reported in KiB and is calibrated against a streaming fill of known size in the same run;
FETCH_SIZE counts wide coalesced 128-B requests as 64 B on gfx950 (x2), calibrated against a known copy."""
import csv, json, re, sys
from collections import defaultdict

G, P, F, A, H = 65536, 2, 783, 21, 5
ALGO = (P * (F + A + 3 * H + 1) * 4 + 5 + P * 8 + 256) * G     # SURVEY.md §8(d) bytes per env-step x G
FILL_BYTES = G * P * F * 4


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def pick(acc, pattern):
    for k, v in acc.items():
        if re.search(pattern, k):
            return v
    return []


def mean(v):
    return sum(v) / len(v) if v else None


w, f = per_kernel(sys.argv[1], "WRITE_SIZE"), per_kernel(sys.argv[2], "FETCH_SIZE")
fill_w = max(pick(w, "fillBufferAligned|FillFunctor|fill") or [0])          # the calibration fill of FILL_BYTES
copy_f = max(pick(f, "copyBuffer|copy") or [0])
write_factor = (FILL_BYTES / 1024.0) / fill_w if fill_w else None
fetch_factor = (FILL_BYTES / 1024.0) / copy_f if copy_f else None
out = {
    "command": "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python tools/pmc_probe.py ; "
               "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/pmc_probe.py  (separate passes)",
    "calibration": {"fill_bytes": FILL_BYTES, "fill_WRITE_SIZE_KiB": fill_w, "write_factor": write_factor,
                    "copy_FETCH_SIZE_KiB": copy_f, "fetch_factor": fetch_factor,
                    "note": "factors = known bytes / counter; MI355X_MICROARCH.md §HBM: WRITE_SIZE exact in KiB, "
                            "FETCH_SIZE undercounts wide coalesced reads 2x on gfx950"},
}
PARTS = 3    # the launch-per-iteration rollout runs as 3 stream partitions: one launch covers G/3 games
CHUNK = 50   # the persistent rollout kernel: one launch = CHUNK iterations of all G games (tools/pmc_probe.py)
wk, fk = mean(pick(w, r"env_rollout_kernel<2, 5>")), mean(pick(f, r"env_rollout_kernel<2, 5>"))
if wk is not None:
    wf = write_factor if write_factor and abs(write_factor - 1) < 0.05 else 1.0
    ff = fetch_factor if fetch_factor and 1.5 < fetch_factor < 2.5 else 2.0
    hbm = (wk * wf + (fk or 0.0) * ff) * 1024.0
    out["env_rollout_kernel<2,5> persistent fused reset+policy+step+observe, G=65536, %d iterations per launch" % CHUNK] = {
        "WRITE_SIZE_KiB": wk, "FETCH_SIZE_KiB_raw": fk, "dispatches": len(pick(w, r"env_rollout_kernel<2, 5>")),
        "iterations_per_launch": CHUNK, "hbm_bytes_per_launch": hbm, "hbm_bytes_per_iteration": hbm / CHUNK,
        "algorithmic_bytes_per_launch": ALGO * CHUNK, "traffic_over_algorithmic": hbm / (ALGO * CHUNK)}
for mode, label in ((3, "env_kernel<3,2,5> fused reset+policy+step+observe (rollout), G=65536 in 3 partition launches"),
                    (1, "env_kernel<1,2,5> step+observe, G=65536"),
                    (0, "env_kernel<0,2,5> reset-terminated, G=65536")):
    pat = r"env_kernel<%d, 2, 5>" % mode
    wk, fk = mean(pick(w, pat)), mean(pick(f, pat))
    if wk is None:
        continue
    wf = write_factor if write_factor and abs(write_factor - 1) < 0.05 else 1.0
    ff = fetch_factor if fetch_factor and 1.5 < fetch_factor < 2.5 else 2.0
    hbm = (wk * wf + (fk or 0.0) * ff) * 1024.0
    rec = {"WRITE_SIZE_KiB": wk, "FETCH_SIZE_KiB_raw": fk, "dispatches": len(pick(w, pat)), "hbm_bytes_per_launch": hbm}
    if mode in (1, 3):
        algo = ALGO / PARTS if mode == 3 else ALGO
        rec["algorithmic_bytes_per_launch"] = algo
        rec["traffic_over_algorithmic"] = hbm / algo
    out[label] = rec
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("env_")}, indent=1))
