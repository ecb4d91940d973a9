set -u
O=gpurun_out; mkdir -p $O
timeout 600 python bench.py > $O/r06a_bench.json 2> $O/r06a_bench.err
timeout 200 python tools/recurrence_step_budget.py $O/r06a_step_budget.json 8 > $O/r06a_step_budget.txt 2>&1
HSAD_FWD_EARLY=0 HSAD_FWD_KEEP_AUX=0 HSAD_BWD_ROT=0 timeout 200 python tools/recurrence_step_budget.py $O/r06a_step_budget_r5sched.json 8 > $O/r06a_step_budget_r5sched.txt 2>&1
timeout 300 python -m pytest tests/test_composite_abi_gpu.py -x -q > $O/r06a_composite_tests.txt 2>&1
tail -3 $O/r06a_composite_tests.txt
tail -c 3000 $O/r06a_bench.json
