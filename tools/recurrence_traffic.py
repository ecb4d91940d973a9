"""python tools/recurrence_traffic.py <dir of counters_b<rows>_set<k>.csv> <out.json>: the memory-side traffic of lstm_fused_fwd_kernel / lstm_fused_bwd_kernel
per launch as a function of the batch rows (tools/recurrence_traffic.sh collects the passes).  FETCH_SIZE is reported in KiB with 128-byte requests
tallied at 64 (MI355X_MICROARCH.md, HBM): bytes are bracketed -- `lo` = counter x 1 (every request narrow), `hi` = counter x 2 (every request wide) --
and, where the request counters exist, the 32-byte requests are taken out (TCC_EA0_RDREQ_32B).  NOTE: that figure equals `lo` by construction
(FETCH_SIZE = requests x 64 B) -- the counters cannot tell a 64-byte request from a 128-byte one tallied at 64; what does is the fit below: a read
side whose per-row slope is under the bytes the kernel must at least fetch per row proves that requests are wide (`hi` applies).  A line through
the four batch sizes splits every figure into a part per launch (weights: once per row block's XCD, counter blocks) and a part per batch row."""
import csv, glob, json, os, re, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
KERNELS = {"forward": r"lstm_fused_fwd_kernel", "bptt": r"lstm_bptt_wide_kernel|lstm_fused_bwd_kernel"}
T, H = 80, 512
# algorithmic bytes per launch as tools/pmc_summarize.py counts them, split the same way (per launch / per batch row)
ALGO = {"forward": {"const": 8 * 2048 * H * 2, "per_row": T * H * (2 * 2 + 4 * 2 + 2 * (4 * 4 + 4)),
                    "read_const": 8 * 2048 * H * 2, "read_per_row": T * H * 2 * 2},
        "bptt": {"const": 4 * 2048 * H * 2, "per_row": T * H * (2 * (4 * 4 + 2 * 4) + 4 + 2 + 2 * 4 * 2 + 2 * 4 * 2 + 2 * 4 + 2),
                 "read_const": 4 * 2048 * H * 2, "read_per_row": T * H * (2 * (4 * 4 + 2 * 4) + 4 + 2 + 4)}}      # saved gates + c, dO, ReLU mask, the stage's dO rows back
data = defaultdict(lambda: defaultdict(dict))          # kernel -> rows -> counter -> mean per dispatch
for f in sorted(glob.glob(os.path.join(src, "counters_b*_set*.csv"))):
    rows = int(re.search(r"counters_b(\d+)_set", f).group(1))
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k, pat in KERNELS.items():
            if re.search(pat, r["Kernel_Name"]):
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        v = v[1:] if len(v) > 2 else v                 # the first launch of a process also reads what later ones find on die
        data[k][rows][c] = sum(v) / len(v)


def fit(points):                                       # least squares y = a + b x
    n = len(points)
    if n < 2:
        return None
    sx = sum(x for x, _ in points); sy = sum(y for _, y in points)
    sxx = sum(x * x for x, _ in points); sxy = sum(x * y for x, y in points)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    return {"per_launch": (sy - b * sx) / n, "per_row": b}


res = {"command": "tools/recurrence_traffic.sh (rocprofv3 --pmc <one set per pass> --kernel-trace -- python tools/pmc_probe.py learner_b<rows>)", "T": T, "H": H}
for k in KERNELS:
    per_rows, series = {}, defaultdict(list)
    for rows in sorted(data[k]):
        c = data[k][rows]
        e = {"counters_mean_per_launch": c}
        if "WRITE_SIZE" in c:
            e["write_bytes"] = c["WRITE_SIZE"] * 1024
            series["write_bytes"].append((rows, e["write_bytes"]))
        if "FETCH_SIZE" in c:
            e["read_bytes_lo"], e["read_bytes_hi"] = c["FETCH_SIZE"] * 1024, c["FETCH_SIZE"] * 2048
            series["read_bytes_lo"].append((rows, e["read_bytes_lo"])); series["read_bytes_hi"].append((rows, e["read_bytes_hi"]))
        if "TCC_EA0_RDREQ_sum" in c:
            n, n32 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            e["read_requests"], e["read_requests_32B"] = n, n32
            e["read_bytes_if_others_64B"] = n32 * 32 + (n - n32) * 64
            series["read_bytes_if_others_64B"].append((rows, e["read_bytes_if_others_64B"]))
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        a = ALGO[k]
        e["algorithmic_bytes"] = a["const"] + a["per_row"] * rows
        e["algorithmic_read_bytes"] = a["read_const"] + a["read_per_row"] * rows
        per_rows[str(rows)] = e
    res[k] = {"by_batch_rows": per_rows, "fit": {name: fit(pts) for name, pts in series.items()}, "algorithmic": ALGO[k]}
json.dump(res, open(out, "w"), indent=1)
for k in KERNELS:
    print(k)
    for rows, e in res[k]["by_batch_rows"].items():
        print("  rows %4s: " % rows + "  ".join("%s %.1f MB" % (n, e[n] / 1e6) for n in ("write_bytes", "read_bytes_lo", "read_bytes_hi", "read_bytes_if_others_64B",
                                                                                          "algorithmic_bytes", "algorithmic_read_bytes") if n in e)
              + ("  L2 hit %.3f" % e["l2_hit_rate"] if "l2_hit_rate" in e else ""))
    for name, f in res[k]["fit"].items():
        if f:
            print("  %-26s per launch %8.1f MB + %7.3f MB per batch row" % (name, f["per_launch"] / 1e6, f["per_row"] / 1e6))
    a = ALGO[k]
    print("  %-26s per launch %8.1f MB + %7.3f MB per batch row  (reads: %.1f + %.3f)" % ("algorithmic", a["const"] / 1e6, a["per_row"] / 1e6, a["read_const"] / 1e6,
                                                                                          a["read_per_row"] / 1e6))
