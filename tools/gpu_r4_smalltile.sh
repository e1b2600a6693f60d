#!/bin/bash
# conv_kernel (the gathered convs: down-sampling, time-down, TokenShift-ed Linears) on 64 x 64 tiles for launches with fewer than N tiles of
# 128 x 128 per CU (product: N = 2), in the step graph (experiments library, MV2_SMALL_TILE=N)
TAG=${1:-r04q}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_small_tile_ab.log; : > $L
for pass in 0 1; do
  for n in 2 3 6; do
    echo "#### B1 pass $pass MV2_SMALL_TILE=$n" >> $L
    (MV2_SMALL_TILE=$n timeout 200 python tools/bench_exp.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact --no-batch4 --no-soak 2>&1 | tail -1 | cut -c1-230) >> $L 2>&1
  done
  for n in 2 4 8 16 64; do
    echo "#### B4 pass $pass MV2_SMALL_TILE=$n" >> $L
    (MV2_SMALL_TILE=$n timeout 200 python tools/bench_exp.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline --no-exact --no-batch4 --no-soak 2>&1 | tail -1 | cut -c1-230) >> $L 2>&1
  done
done
grep -o "####.*\|ms_per_step\": [0-9.]*" $L | paste - -
