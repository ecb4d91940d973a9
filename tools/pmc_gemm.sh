#!/bin/bash
# PMC passes over tools/gemm_probe.py; prints the mean of each counter for the GEMM kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for pass in "MfmaUtil LdsUtil" "LdsBankConflict LdsLatency" "VmemLatency MemUnitStalled" "OccupancyPercent MeanOccupancyPerActiveCU" "FetchSize" "WriteSize"; do
  rm -rf /tmp/pg
  timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pg -o g -- python $R/tools/gemm_probe.py "$@" > /dev/null 2>&1
  f=$(find /tmp/pg -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-42s %-26s mean %.4g (n=%d)" % (k[0], k[1], sum(v) / len(v), len(v)))
PY
done
