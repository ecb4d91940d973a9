O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o t -- python $GRAFT_REPO_ROOT/tools/jobs/r06_training_timeline.py > $O/training_timeline.log 2>&1
python $GRAFT_REPO_ROOT/tools/update_timeline.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) 5 > $O/r06_training_iteration_timeline.txt 2>&1
cat $O/r06_training_iteration_timeline.txt
