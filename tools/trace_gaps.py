"""python tools/trace_gaps.py <kernel_trace.csv> <n_updates>: GPU busy time (union of kernel intervals), idle gaps and the kernels that
follow the largest gaps, from a rocprofv3 --kernel-trace csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows)
# keep the last 60 % of the trace (steady state)
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 0.4
ev = [e for e in ev if e[0] >= t_lo]
busy, gaps, cur_end = 0, [], ev[0][0]
for s, e, name in ev:
    if s > cur_end:
        gaps.append((s - cur_end, name))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
span = ev[-1][1] - ev[0][0]
print("span %.1f us, busy %.1f us (%.1f %%), idle %.1f us in %d gaps" % (span / 1e3, busy / 1e3, 100 * busy / span, (span - busy) / 1e3, len(gaps)))
from collections import defaultdict
by = defaultdict(lambda: [0, 0])
for g, name in gaps:
    by[name][0] += g
    by[name][1] += 1
for name, (tot, cnt) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  idle before %-60s total %8.1f us in %5d gaps (avg %.2f us)" % (name, tot / 1e3, cnt, tot / cnt / 1e3))
