set -u
O=gpurun_out; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline > $O/r06f_bench.json 2> $O/r06f_bench.err
python - <<'PY'
import json
b=json.loads(open("gpurun_out/r06f_bench.json").read().strip().splitlines()[-1])
print("env", b["value"], b["roofline"]["frac"])
L=b["learner"]
print({k:v for k,v in L.items() if not isinstance(v,(dict,list))})
print("bwd", L["roofline"]["avg_launch_ms"], L["roofline"]["frac"], "fwd", L["roofline_forward"]["avg_launch_ms"], L["roofline_forward"]["frac"])
print("actor", b["actor"]["ms_per_step"], b["actor"]["roofline"]["frac"])
print("one_gpu", {k:v for k,v in b["one_gpu_training"].items() if k!="config"})
PY
timeout 900 python -m pytest tests/test_composite_abi_gpu.py tests/test_r2d2_kernels_gpu.py tests/test_driver_gpu.py -x -q > $O/r06f_tests.txt 2>&1; tail -3 $O/r06f_tests.txt
