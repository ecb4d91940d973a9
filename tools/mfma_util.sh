#!/bin/bash
# Run ON THE GPU BOX (gpurun): MFMA utilisation of the three MFMA-bound kernels of the hot path as COUNTERS (north_star: "evidenced by
# rocprof ... MFMA utilisation"): one rocprofv3 pass per leg with --pmc <SQ counters> + --kernel-trace only (no other trace domains),
# a second pass with --kernel-trace --stats for the un-instrumented durations of the same command, then tools/mfma_util_summarize.py.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
RN=${ROUND:-r06}
O=$R/gpurun_out/mfma
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
CNT_MIN="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SPEC=""
for leg in calib learner actor; do
  rm -rf /tmp/mu_$leg /tmp/mt_$leg
  timeout 400 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/mu_$leg -o c -- python $R/tools/mfma_probe.py $leg > $O/pmc_$leg.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mt_$leg -o t -- python $R/tools/mfma_probe.py $leg > $O/time_$leg.log 2>&1
  C=$(find /tmp/mu_$leg -name "*counter_collection.csv" | head -1)
  if [ -z "$C" ]; then   # the wide pass was refused: the three counters the utilisation needs, alone
    rm -rf /tmp/mu_$leg
    timeout 400 rocprofv3 --pmc $CNT_MIN --kernel-trace --output-format csv -d /tmp/mu_$leg -o c -- python $R/tools/mfma_probe.py $leg > $O/pmc_min_$leg.log 2>&1
    C=$(find /tmp/mu_$leg -name "*counter_collection.csv" | head -1)
  fi; T=$(find /tmp/mt_$leg -name "*kernel_stats.csv" | head -1)
  cp $C $O/counters_$leg.csv 2>/dev/null; cp $T $O/stats_$leg.csv 2>/dev/null
  SPEC="$SPEC $leg:$O/counters_$leg.csv:$O/stats_$leg.csv"
done
python $R/tools/mfma_util_summarize.py $O/${RN}_mfma_util.json $SPEC > $O/summary.txt 2>&1
cp $O/${RN}_mfma_util.json $R/profiles/ 2>/dev/null
cat $O/summary.txt
