"""Developer tool: print the kernel timeline of the last learner update found in a rocprofv3 kernel-trace csv."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ev.sort()
# an update ends with adam_kernel: take the window between the last two adam kernels
adam = [i for i, e in enumerate(ev) if "adam" in e[2]]
lo, hi = adam[-2] + 1, adam[-1] + 1
t0 = ev[lo][0]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "").replace("at::native::", "")[:60]
for s, e, n, q in ev[lo:hi]:
    print("%8.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short(n)))
print("update span %.1f us" % ((ev[hi - 1][1] - t0) / 1e3))
