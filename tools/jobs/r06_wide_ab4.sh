set -u
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_composite_abi_gpu.py -x -q -k "fused_recurrences" > $O/r06e_wide_tests.txt 2>&1
tail -5 $O/r06e_wide_tests.txt
timeout 200 python tools/recurrence_step_budget.py $O/r06e_step_budget_wide.json 8 > $O/r06e_step_budget_wide.txt 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06e_step_budget_wide.json"))
print("ms/update", d["ms_per_update_untraced"])
for k,v in d["backward"].items(): print("  ", k, v)
PY
