#!/bin/bash
# GPU-box helper: A/B timing of env-switched variants through bench.py --breakdown -> gpurun_out/<tag>_*
# usage: tools/gpu_ab.sh <tag> "<VAR=1 ...>" "<VAR=1 ...>" ...   (first variant "" = defaults)
TAG=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
i=0
for v in "$@"; do
  (env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown $O/${TAG}_v${i}.json 2>&1 | grep "timed region") > $O/${TAG}_v${i}.log 2>&1
  echo "variant $i [$v]: $(cat $O/${TAG}_v${i}.log)"
  i=$((i+1))
done
