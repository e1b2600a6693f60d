#!/bin/bash
# rocprofv3 kernel stats of the batch-4 shape (configs[2] per GPU): the authority behind the batch4 figures of the bench line
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_b4 -- python $R/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --no-exact --no-batch4 --no-graph --no-soak > $O/${TAG}_rocprof_b4.log 2>&1)
cd $R
ST=$(find $O/${TAG}_prof_b4 -name "*kernel_stats.csv" | head -1); [ -n "$ST" ] && cp $ST $O/${TAG}_rocprofv3_kernel_stats_bench_b4.csv
find $O/${TAG}_prof_b4 -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null; find $O -name "*.db" -size +20M -delete 2>/dev/null
head -8 $O/${TAG}_rocprofv3_kernel_stats_bench_b4.csv | cut -c1-200
