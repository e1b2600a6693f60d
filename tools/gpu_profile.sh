#!/bin/bash
# GPU-box helper: tests, bench (+ablations), rocprofv3 kernel stats and PMC passes -> gpurun_out/<tag>_*
# usage: tools/gpu_profile.sh <tag> [tests|notests] [ablate|noablate] [pmc|nopmc]
TAG=${1:-rX}; TESTS=${2:-tests}; ABL=${3:-noablate}; PMC=${4:-nopmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
if [ "$TESTS" = "tests" ]; then
  (timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -30) > $O/${TAG}_tests.log 2>&1
fi
(timeout 400 python bench.py --steps 20 --warmup 3 --breakdown $O/${TAG}_breakdown.json 2>&1 | tail -12) > $O/${TAG}_bench.log 2>&1
(timeout 300 python bench.py --steps 10 --warmup 3 --batch-per-gpu 4 --no-cpu-baseline --breakdown $O/${TAG}_breakdown_b4.json 2>&1 | tail -3) > $O/${TAG}_bench_b4.log 2>&1
(timeout 300 python bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline --breakdown $O/${TAG}_breakdown_fp32.json 2>&1 | tail -3) > $O/${TAG}_bench_fp32.log 2>&1
if [ "$ABL" = "ablate" ]; then
  for a in 1 2 4 8 15; do
    (MV2_ABLATE=$a timeout 200 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --breakdown $O/${TAG}_breakdown_ab$a.json 2>&1 | tail -2) > $O/${TAG}_bench_ab$a.log 2>&1
  done
fi
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $O/${TAG}_rocprof.log 2>&1)
if [ "$PMC" = "pmc" ]; then
  rocprofv3 -L > $O/${TAG}_counters.txt 2>&1
  (timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_sq -- env MV2_BENCH_NO_SPIN=1 python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-graph > $O/${TAG}_pmc_sq.log 2>&1)
  (timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- env MV2_BENCH_NO_SPIN=1 python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-graph > $O/${TAG}_pmc_fetch.log 2>&1)
  (timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- env MV2_BENCH_NO_SPIN=1 python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-graph > $O/${TAG}_pmc_write.log 2>&1)
fi
cd $R
# keep the merge small: drop bulky traces, keep stats / counter csv
find $O -name "*.db" -size +20M -delete 2>/dev/null
for f in $O/${TAG}_tests.log $O/${TAG}_bench.log $O/${TAG}_bench_b4.log $O/${TAG}_bench_fp32.log; do [ -f $f ] && tail -6 $f; done
ls -la $O | tail -30
