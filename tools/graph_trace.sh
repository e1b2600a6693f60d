#!/bin/bash
# rocprofv3 kernel trace of the CAPTURED step (bench.py in graph mode): per-kernel durations inside the graph and the idle gaps between
# consecutive kernels.  usage (GPU box): tools/graph_trace.sh <tag>
TAG=${1:-rX}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_graphtrace -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-exact --no-batch4 --no-two-streams --no-configs34 --no-soak > $O/${TAG}_graphtrace.log 2>&1
cd $R
python tools/graph_trace_summary.py $(find $O/${TAG}_graphtrace -name "*kernel_trace.csv" | head -1) > $O/${TAG}_graph_trace_summary.txt 2>&1
find $O/${TAG}_graphtrace -name "*.csv" -size +8M -delete 2>/dev/null; find $O/${TAG}_graphtrace -name "*.db" -delete 2>/dev/null
cat $O/${TAG}_graph_trace_summary.txt | head -80
