#!/bin/bash
# Run ON THE GPU BOX (gpurun): memory-side traffic of the learner's two fused recurrence launches at B = 32 / 64 / 96 / 128 batch rows, one counter
# set per pass (where the bytes above the algorithmic count come from).  tools/recurrence_traffic.py turns the csv files
# into profiles/r06_recurrence_traffic.json: per launch a constant part (weights, per row block) and a part per batch row, read and write
# side apart, request sizes where the counters exist.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/rectraffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "TCC_[A-Z0-9_]+(\[[0-9]+\])?|FETCH_SIZE|WRITE_SIZE" | sort -u > $O/avail_tcc.txt
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum")
for B in ${ROWS:-128 96 64 32}; do
  i=0
  for set in "${SETS[@]}"; do
    rm -rf /tmp/rt_${B}_$i
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/rt_${B}_$i -o c -- python $R/tools/pmc_probe.py learner_b$B > $O/log_${B}_$i.txt 2>&1
    f=$(find /tmp/rt_${B}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f $O/counters_b${B}_set$i.csv
    i=$((i+1))
  done
done
python $R/tools/recurrence_traffic.py $O $R/gpurun_out/r06_recurrence_traffic.json > $O/summary.txt 2>&1
cat $O/summary.txt
