"""python tools/actor_timeline.py <kernel_trace.csv>: the kernels of ONE steady-state actor step (between the last two env_kernel<0,..> launches
of a rocprofv3 --kernel-trace csv of tools/actor_step_breakdown.py) as a timeline: start offset, duration, queue, workgroups, name"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
Q = {}
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))) for r in rows)
marks = [i for i, e in enumerate(ev) if "env_kernel<0" in e[2]]
lo, hi = marks[-3], marks[-2]
t0 = ev[lo][0]
print("step span %.1f us, %d kernels" % ((ev[hi][0] - t0) / 1e3, hi - lo))
for s, e, name, q, wgs in ev[lo:hi]:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("%8.1f  %7.1f  q%-3s %6d wg  %s" % ((s - t0) / 1e3, (e - s) / 1e3, Q.setdefault(q, len(Q)), wgs, short))
