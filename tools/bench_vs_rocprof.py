"""Same-lease reconciliation of the bench line with the rocprofv3 kernel stats filed next to it: the env kernel's
average launch duration from `bench.py` (HIP events) against AverageNs of the same kernel in the --kernel-trace --stats pass of the SAME
command on the SAME box, and the bench line's own repeat spread.  Writes a small json; exit code 1 when they differ by more than 5 %.
usage: python tools/bench_vs_rocprof.py <bench.json> <bench_under_rocprof.json> <kernel_stats.csv> <out.json>"""
import csv, json, sys

def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)

bench, under, stats_csv, out_path = sys.argv[1:5]
b, u = last_json(bench), last_json(under)
avg_ns = calls = None
for row in csv.DictReader(open(stats_csv)):
    if "env_rollout_kernel" in row.get("Name", ""):
        avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])
        break
r = b["roofline"]
res = {
    "same_lease": True,
    "bench_avg_launch_ms": r["avg_launch_ms"], "bench_iterations_per_launch": r["iterations_per_launch"],
    "bench_under_rocprof_avg_launch_ms": u["roofline"]["avg_launch_ms"],
    "rocprof_AverageNs": avg_ns, "rocprof_calls": calls,
    "rocprof_over_bench": (avg_ns / 1e6) / r["avg_launch_ms"] if avg_ns else None,
    "rocprof_over_bench_under_rocprof": (avg_ns / 1e6) / u["roofline"]["avg_launch_ms"] if avg_ns else None,
    "bench_ms_per_step": b["ms_per_step"], "bench_ms_per_step_median_min_max": [r.get("ms_per_step_median"), r.get("ms_per_step_min"), r.get("ms_per_step_max")],
    "bench_frac": r["frac"], "bench_frac_of_write_ceiling": r.get("frac_of_write_ceiling"), "write_ceiling_gbs": r.get("write_ceiling_gbs"),
    "device_state_before": r.get("device_state_before"),
    "note": "rocprof's AverageNs covers every call of the kernel in the profiled process (warm-up and the repeat regions included: all 50-iteration "
            "launches of the same shape); the bench figure is the timed region's launches only",
}
ok = avg_ns is not None and abs(res["rocprof_over_bench"] - 1.0) <= 0.05
res["agree_within_5_percent"] = bool(ok)
open(out_path, "w").write(json.dumps(res, indent=1) + "\n")
print(json.dumps(res, indent=1))
sys.exit(0 if ok else 1)
