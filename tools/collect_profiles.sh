#!/bin/bash
# Run ON THE GPU BOX (gpurun): PMC passes per leg -> traffic json -> bench line -> rocprofv3 kernel stats of the same command.
# Outputs land in gpurun_out/ ; the summaries are also written into profiles/ (copy them back from gpurun_out/ afterwards).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
RN=${ROUND:-r06}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SPEC=""
for leg in ${LEGS:-env env5 env5_literal gemm learner actor}; do
  rm -rf /tmp/pw_$leg /tmp/pf_$leg
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$leg -o w -- python $R/tools/pmc_probe.py $leg > $O/pmc_write_$leg.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$leg -o f -- python $R/tools/pmc_probe.py $leg > $O/pmc_fetch_$leg.log 2>&1
  W=$(find /tmp/pw_$leg -name "*counter_collection.csv" | head -1); F=$(find /tmp/pf_$leg -name "*counter_collection.csv" | head -1)
  SPEC="$SPEC $leg:$W:$F"
done
python $R/tools/pmc_summarize.py $O/${RN}_pmc_hbm_traffic.json $SPEC > $O/pmc_summary.txt 2>&1
cp $O/${RN}_pmc_hbm_traffic.json $R/profiles/${RN}_pmc_hbm_traffic.json
cd $R && timeout 900 python bench.py > $O/${RN}_bench.json 2> $O/bench.err
# kernel stats of the timed env region alone (the learner / actor legs launch the same env kernels at other sizes) ...
rm -rf /tmp/ps /tmp/ps2
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o r1 -- python $R/bench.py --no-cpu-baseline --no-learner --no-actor > $O/${RN}_bench_under_rocprof.json 2> $O/rocprof_stats.err
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $O/${RN}_bench_kernel_stats.csv
# same lease, same command: the line's avg_launch_ms against rocprof's AverageNs (exit code 1 = more than 5 % apart)
python $R/tools/bench_vs_rocprof.py $O/${RN}_bench.json $O/${RN}_bench_under_rocprof.json $O/${RN}_bench_kernel_stats.csv $O/${RN}_bench_vs_rocprof.json > $O/bench_vs_rocprof.txt 2>&1; echo "bench vs rocprof: rc $?"
cp $O/${RN}_bench_vs_rocprof.json $R/profiles/ 2>/dev/null
# ... and of the whole default command (env + learner + actor legs)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps2 -o r2 -- python $R/bench.py --no-cpu-baseline > $O/${RN}_bench_full_under_rocprof.json 2> $O/rocprof_stats_full.err
cp $(find /tmp/ps2 -name "*kernel_stats.csv" | head -1) $O/${RN}_bench_full_kernel_stats.csv
# one learner update, kernel by kernel (what the in-update figures of bench.py can be recomputed from without bench.py)
rm -rf /tmp/pl
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o l -- python $R/tools/learner_update_breakdown.py 20 > $O/learner_breakdown.log 2>&1
cp $(find /tmp/pl -name "*kernel_stats.csv" | head -1) $O/${RN}_learner_update_kernel_stats.csv
python $R/tools/update_timeline.py $(find /tmp/pl -name "*kernel_trace.csv" | head -1) > $O/${RN}_learner_update_timeline.txt 2>&1
cp $O/${RN}_learner_update_kernel_stats.csv $O/${RN}_learner_update_timeline.txt $R/profiles/ 2>/dev/null
# 200 steady-state actor steps, kernel by kernel (per-step figures = TotalDurationNs / 200)
rm -rf /tmp/pa
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o a -- python $R/tools/actor_step_breakdown.py 200 > $O/actor_breakdown.log 2>&1
cp $(find /tmp/pa -name "*kernel_stats.csv" | head -1) $O/${RN}_actor_200_steps_kernel_stats.csv
cp $O/${RN}_actor_200_steps_kernel_stats.csv $R/profiles/ 2>/dev/null
# the fused cell kernel's phase timers and ablations
cd $R && mkdir -p tools/bin && { [ -x tools/bin/gemm8_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm8_probe.hip -o tools/bin/gemm8_probe; }
cd $R && (tools/bin/gemm8_probe; HSAD_CELL_PP=1 timeout 100 python tools/cell_phases.py 32768 1; HSAD_CELL_PP=1 timeout 100 python tools/cell_phases.py 32768 0; HSAD_CELL_PP=0 timeout 100 python tools/cell_phases.py 32768 1) 2>&1 | grep -v amdgpu.ids > $O/${RN}_cell_kernel_ablations.txt
cp $O/${RN}_cell_kernel_ablations.txt $R/profiles/ 2>/dev/null
tail -c 2500 $O/${RN}_bench.json; echo; head -8 $O/${RN}_bench_kernel_stats.csv | cut -c1-160; cat $O/pmc_summary.txt | head -60
