set -u
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_composite_abi_gpu.py -x -q -k "fused_recurrences" > $O/r06b_wide_tests.txt 2>&1
tail -15 $O/r06b_wide_tests.txt
timeout 200 python tools/recurrence_step_budget.py $O/r06b_step_budget_wide.json 8 > $O/r06b_step_budget_wide.txt 2>&1
timeout 200 python tools/recurrence_step_budget.py $O/r06b_step_budget_32x32.json 8 $(( 1 | 8 | 16 | 32 | (1<<8) | (1<<25) )) > $O/r06b_step_budget_32x32.txt 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/r06b_step_budget_wide.json","gpurun_out/r06b_step_budget_32x32.json"):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ms/update", d["ms_per_update_untraced"])
    for k,v in d["backward"].items(): print("  ", k, v)
PY
