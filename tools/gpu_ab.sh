#!/bin/bash
# Generic in-graph A/B on the GPU box (replaces the one-shot gpu_r4_* lease scripts): the step graph timed under each value of ONE environment
# knob of the experiments library (tools/bench_exp.py loads libmv2_hip_exp.so), interleaved passes so that box drift shows as spread.
# usage: tools/gpu_ab.sh <tag> <ENVVAR> "<values>" [passes] [bench args...]     e.g. tools/gpu_ab.sh r05x MV2_SMALL_TILE "2 3 6" 2 --config 2
TAG=${1:?tag}; VAR=${2:?env var}; VALS=${3:?values}; PASSES=${4:-2}; shift 4 2>/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_ab_${VAR}.log; : > $L
for pass in $(seq 1 $PASSES); do
  for v in $VALS; do
    echo "#### pass $pass $VAR=$v $*" >> $L
    (env $VAR="$v" timeout 240 python tools/bench_exp.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact --no-batch4 --no-two-streams --no-configs34 --no-soak "$@" 2>&1 | tail -1 | cut -c1-230) >> $L 2>&1
  done
done
grep -o "####.*\|ms_per_step\": [0-9.]*" $L | paste - -
