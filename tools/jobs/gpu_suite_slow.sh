# the tests marked slow (larger twins of comparisons that run by default; the ten-second stress of the persistent launches): their own gpurun
set -u
mkdir -p gpurun_out
timeout 1400 python -m pytest tests -m "gpu and slow" --runslow -q --durations=20 > gpurun_out/gpu_suite_slow.txt 2>&1
tail -30 gpurun_out/gpu_suite_slow.txt
