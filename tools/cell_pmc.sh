R=$PWD; O=$R/gpurun_out/cellpmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/tools/cell_probe.py > $O/time.log 2>&1
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pc$i -o p -- python $R/tools/cell_probe.py 32768 4 > $O/run$i.log 2>&1
  cp $(find /tmp/pc$i -name "*counter_collection.csv" | head -1) $O/pmc$i.csv 2>/dev/null
done
cat $O/time.log
