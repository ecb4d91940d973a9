set -u
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_composite_abi_gpu.py -x -q -k "fused_recurrences_equal" > $O/r06c_wide_tests.txt 2>&1
tail -5 $O/r06c_wide_tests.txt
timeout 200 python tools/recurrence_step_budget.py $O/r06c_step_budget_wide.json 8 > $O/r06c_step_budget_wide.txt 2>&1
HSAD_BPTT_WIDE_DEV=1 timeout 200 python tools/recurrence_step_budget.py $O/r06c_step_budget_wide_dev1.json 8 > $O/r06c_dev1.txt 2>&1
HSAD_BPTT_WIDE_DEV=2 timeout 200 python tools/recurrence_step_budget.py $O/r06c_step_budget_wide_dev2.json 8 > $O/r06c_dev2.txt 2>&1
HSAD_BPTT_WIDE_DEV=3 timeout 200 python tools/recurrence_step_budget.py $O/r06c_step_budget_wide_dev3.json 8 > $O/r06c_dev3.txt 2>&1
python - <<'PY'
import json
for f in ("wide","wide_dev1","wide_dev2","wide_dev3"):
    try:
        d=json.load(open("gpurun_out/r06c_step_budget_%s.json"%f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ms/update", d["ms_per_update_untraced"])
    for k,v in d["backward"].items(): print("  ", k, v)
PY
