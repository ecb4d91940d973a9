# Round-6 evidence in ONE lease: step budgets (wide / 32 x 32), shape + determinism check, bench + rocprof stats + PMC traffic
# (tools/collect_profiles.sh), MFMA counters, update timeline.  Outputs land in gpurun_out/ and profiles/ (copied back by hand).
set -u
O=gpurun_out; mkdir -p $O
export ROUND=r06
timeout 200 python tools/recurrence_step_budget.py $O/r06_recurrence_step_budget.json 8 > $O/r06_step_budget.txt 2>&1
timeout 200 python tools/recurrence_step_budget.py $O/r06_recurrence_step_budget_32x32_blocks.json 8 $(( 0x39 | (1<<8) | (1<<25) )) > $O/r06_step_budget_32.txt 2>&1
for c in 80x128 80x64 24x96; do timeout 120 python -u tools/jobs/r06_wide_shapes.py $c 2>&1 | grep -v amdgpu.ids; done > $O/r06_bptt_blocking_ab_and_determinism.txt
LEGS="env learner actor" timeout 2400 bash tools/collect_profiles.sh > $O/r06_collect.log 2>&1
tail -40 $O/r06_collect.log
timeout 900 bash tools/mfma_util.sh > $O/r06_mfma.log 2>&1
tail -15 $O/r06_mfma.log
