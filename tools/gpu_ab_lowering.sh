#!/bin/bash
# In-graph A/B of lowering options on one box: interleaved bench runs (bench.py --lowering ...), B = 1 and optionally the batch-4 shape.
# usage: tools/gpu_ab_lowering.sh <tag> "<opts A>" "<opts B>" [passes] [extra bench args]     ("" = the product's defaults)
TAG=${1:?tag}; A=${2-}; B=${3-}; PASSES=${4:-2}; shift 4 2>/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_ab_lowering.log; : > $L
for pass in $(seq 1 $PASSES); do
  for opt in "$A" "$B"; do
    echo "#### pass $pass lowering='$opt' $*" >> $L
    (timeout 240 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact --no-two-streams --no-configs34 ${opt:+--lowering $opt} "$@" 2>&1 | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); b4 = d.get('batch4') or {}
        print('ms_per_step', d['ms_per_step'], 'batch4', b4.get('ms_per_step'), 'nodes', d['roofline']['graph_nodes'])
    except Exception: print(l[:300])
") >> $L 2>&1
  done
done
cat $L
