#!/bin/bash
# LDS counters of the persistent recurrence launches inside learner updates (a PMC pass of its own, kernel trace only):
# is the forward launch's MFMA rate held by its LDS reads (every wave reads the whole 32 KB tile per product)?
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pl_$tag
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pl_$tag -o c -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py learner > $O/lds_$tag.log 2>&1
done
python - <<'PY' > $O/r06_recurrence_lds_counters.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("/tmp/pl_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "lstm_fused_fwd_kernel" if "lstm_fused_fwd" in k else "lstm_bptt_wide_kernel" if "lstm_bptt_wide" in k else "gemm8_kernel<1" if "gemm8_kernel<1" in k else None
        if k is None: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("per launch (sums over the chip as rocprofv3 reports them), learner updates at configs[2]")
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-32s %16.0f" % (c, x / max(n[k][c], 1)))
PY
cat $O/r06_recurrence_lds_counters.txt; tail -2 $O/lds_SQ_INSTS_LDS.log
