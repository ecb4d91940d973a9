"""python tools/update_timeline.py <kernel_trace.csv>: the kernels of ONE steady-state learner update (the last adam_kernel-to-adam_kernel
span of a rocprofv3 --kernel-trace csv of tools/learner_update_breakdown.py; optional second argument k: the k-th last span) as a timeline:
start offset, duration, name"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
Q = {}
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = adam[-k - 1] + 1, adam[-k]
t0 = ev[lo][0]
print("update span %.1f us, %d kernels" % ((ev[hi][1] - t0) / 1e3, hi - lo + 1))
for s, e, name, q, wgs in ev[lo:hi + 1]:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("%8.1f  %7.1f  q%-3s %6d wg  %s" % ((s - t0) / 1e3, (e - s) / 1e3, Q.setdefault(q, len(Q)), wgs, short))
