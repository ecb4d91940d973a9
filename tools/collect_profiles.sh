#!/bin/bash
# Run ON THE GPU BOX (gpurun): PMC passes -> traffic json -> bench line -> rocprofv3 kernel stats of the same command.
# Outputs land in gpurun_out/ ; copy the summaries into profiles/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw /tmp/pf /tmp/ps
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o w -- python $R/tools/pmc_probe.py > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o f -- python $R/tools/pmc_probe.py > $O/pmc_fetch.log 2>&1
W=$(find /tmp/pw -name "*counter_collection.csv" | head -1); F=$(find /tmp/pf -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_summarize.py $W $F $R/profiles/r01_pmc_hbm_traffic.json > $O/pmc_summary.txt 2>&1
cp $R/profiles/r01_pmc_hbm_traffic.json $O/r01_pmc_hbm_traffic.json
cd $R && timeout 900 python bench.py > $O/r01_bench.json 2> $O/bench.err
# kernel stats of the timed env region alone (the learner / actor legs launch the same env kernels at other sizes) ...
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o r1 -- python $R/bench.py --no-cpu-baseline --no-learner --no-actor > $O/r01_bench_under_rocprof.json 2> $O/rocprof_stats.err
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $O/r01_bench_kernel_stats.csv
# ... and of the whole default command (env + learner + actor legs)
rm -rf /tmp/ps2
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps2 -o r2 -- python $R/bench.py --no-cpu-baseline > $O/r01_bench_full_under_rocprof.json 2> $O/rocprof_stats_full.err
cp $(find /tmp/ps2 -name "*kernel_stats.csv" | head -1) $O/r01_bench_full_kernel_stats.csv
tail -c 1500 $O/r01_bench.json; echo; head -8 $O/r01_bench_kernel_stats.csv | cut -c1-160; cat $O/pmc_summary.txt | head -40
