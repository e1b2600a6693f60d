#!/bin/bash
# Tile sweep of the LDS-DMA pointwise kernel (conv_pw_dma_kernel templated on its tile): per launch in the harness (every variant compared
# element-wise with the generic kernel), then in the step graph with candidate rules (experiments library, MV2_PW_DMA_RULE).
TAG=${1:-r04p}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
LIB=magvit2_pytorch_amd/libmv2_hip_exp.so
timeout 400 tools/k3_bench $LIB pwtile > $O/${TAG}_pw_tile_sweep.log 2>&1; echo "sweep rc=$?"; grep -c BAD $O/${TAG}_pw_tile_sweep.log; head -24 $O/${TAG}_pw_tile_sweep.log | cut -c1-220
RULES=("256:128x128" "1024:128x128,256:128x64" "256:128x64" "256:64x64" "1024:128x128,256:128x64,96:64x64" "256:128x64,96:64x64")
: > $O/${TAG}_pw_rule_ab.log
for pass in 0 1; do
  for r in "${RULES[@]}"; do
    echo "#### pass $pass rule $r" >> $O/${TAG}_pw_rule_ab.log
    (MV2_PW_DMA_RULE="$r" timeout 200 python tools/bench_exp.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact --no-batch4 --no-soak 2>&1 | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l); print('ms_per_step', d['ms_per_step'], 'value', d['value'])
    except Exception: print(l[:300])
") >> $O/${TAG}_pw_rule_ab.log 2>&1
  done
done
cat $O/${TAG}_pw_rule_ab.log
# batch 4 with the two extremes
for r in "256:128x128" "1024:128x128,256:128x64" "256:128x64"; do
  echo "#### B4 rule $r" >> $O/${TAG}_pw_rule_ab_b4.log
  (MV2_PW_DMA_RULE="$r" timeout 200 python tools/bench_exp.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline --no-exact --no-batch4 --no-soak 2>&1 | tail -1 | cut -c1-200) >> $O/${TAG}_pw_rule_ab_b4.log 2>&1
done
cat $O/${TAG}_pw_rule_ab_b4.log
