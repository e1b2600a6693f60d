#!/bin/bash
# GPU-box helper: bench lines (+ per-launch tables), rocprofv3 kernel stats and PMC passes -> gpurun_out/<tag>_*, summaries -> profiles/
# (copy them back from gpurun_out/ afterwards: profiles/ on the GPU box is not merged).
# usage: tools/gpu_profile.sh <tag> [pmc|nopmc] [extras...]   extras: b4stats (rocprofv3 stats of the batch-4 shape), cfg3stats (of BASELINE configs[3])
TAG=${1:-rX}; PMC=${2:-nopmc}; shift 2 2>/dev/null
EXTRAS="$*"
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 600 python bench.py --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -16) > $O/${TAG}_bench_b1.log 2>&1
(timeout 300 python bench.py --config 2 --steps 100 --no-cpu-baseline --no-exact --breakdown $O/${TAG}_per_launch_events_b4.json 2>&1 | tail -3) > $O/${TAG}_bench_b4.log 2>&1
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact --no-batch4 --no-two-streams --no-configs34 --no-graph --no-soak"
stats() {   # stats <suffix> <bench args...>: rocprofv3 --kernel-trace --stats of one bench command, summary csv copied beside the logs
  local sfx=$1; shift
  (timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$sfx -- $BENCH "$@" > $O/${TAG}_rocprof_$sfx.log 2>&1)
  local st=$(find $O/${TAG}_prof_$sfx -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp $st $O/${TAG}_rocprofv3_kernel_stats_$sfx.csv
  find $O/${TAG}_prof_$sfx -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
}
stats bench_b1
for e in $EXTRAS; do
  [ "$e" = "b4stats" ] && stats bench_b4 --config 2
  [ "$e" = "cfg3stats" ] && stats cfg3 --config 3 --steps 5
done
if [ "$PMC" = "pmc" ]; then
  (timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_sq -- $BENCH > $O/${TAG}_pmc_sq.log 2>&1)
  (timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- $BENCH > $O/${TAG}_pmc_fetch.log 2>&1)
  (timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- $BENCH > $O/${TAG}_pmc_write.log 2>&1)
  mkdir -p $O/${TAG}_summaries
  python $R/tools/pmc_aggregate.py $O $TAG $O/${TAG}_summaries > $O/${TAG}_pmc_aggregate.log 2>&1
fi
cd $R
find $O -name "*.db" -size +20M -delete 2>/dev/null
find $O/${TAG}_pmc_sq $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
for f in $O/${TAG}_bench_b1.log $O/${TAG}_bench_b4.log; do [ -f $f ] && tail -1 $f | cut -c1-400; done
ls $O | grep "^${TAG}" | tail -24
