set -u
O=gpurun_out; mkdir -p $O
for d in 0 4 8 16; do
  echo "=== HSAD_BPTT_WIDE_DEV=$d"
  HSAD_BPTT_WIDE_DEV=$d timeout 200 python tools/jobs/r06_trace_dump.py 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-250
  HSAD_BPTT_WIDE_DEV=$d timeout 200 python tools/recurrence_step_budget.py $O/r06d_dev$d.json 4 > $O/r06d_dev$d.txt 2>&1
done
python - <<'PY'
import json
for f in (0,4,8,16):
    try:
        d=json.load(open("gpurun_out/r06d_dev%d.json"%f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print("dev",f, "ms/update", d["ms_per_update_untraced"], {k:(v or {}).get("step") for k,v in d["backward"].items()})
PY
