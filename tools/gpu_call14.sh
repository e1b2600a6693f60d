#!/bin/bash
# (1) LDS pass-partition microbenchmark, (2) rocprofv3 kernel stats of the 4-clips-per-step workload (configs[2] per-GPU shape)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 60 tools/lds_phase_bench > $O/r03_lds_phase.log 2>&1; cat $O/r03_lds_phase.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03b4_prof -- python $R/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --no-exact --no-graph --no-soak > $O/r03b4_rocprof.log 2>&1
cd $R
ST=$(find $O/r03b4_prof -name "*kernel_stats.csv" | head -1); [ -n "$ST" ] && cp $ST $O/r03_rocprofv3_kernel_stats_bench_b4.csv
find $O/r03b4_prof -name "*.db" -delete 2>/dev/null; find $O/r03b4_prof -name "*kernel_trace.csv" -delete 2>/dev/null
head -8 $O/r03_rocprofv3_kernel_stats_bench_b4.csv | cut -c1-160
