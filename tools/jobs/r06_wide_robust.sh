#!/bin/bash
# after the straight-line rewrite of the wide BPTT kernel's tile product: the two diagnostics that showed the stale fragments, the
# suites that exercise the kernel, the step budget
mkdir -p gpurun_out/robust
O=gpurun_out/robust
python tools/jobs/r06_diag_first_eval.py > $O/diag_first_eval.txt 2>&1
python tools/jobs/r06_diag_bf16.py > $O/diag_bf16.txt 2>&1
timeout 900 python -m pytest tests/test_composite_abi_gpu.py tests/test_driver_gpu.py tests/test_compiled_boundary_gpu.py tests/test_r2d2_kernels_gpu.py -x -q > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 300 python tools/recurrence_step_budget.py $O/step_budget.json > $O/step_budget.out 2>&1
timeout 200 python tools/jobs/r06_trace_dump.py > $O/trace_dump.txt 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
cat $O/diag_first_eval.txt | tail -5; cat $O/diag_bf16.txt | tail -4
