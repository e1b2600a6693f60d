#!/bin/bash
# in-graph A/B of the LDS-DMA pointwise kernel on ONE box: off / tile threshold 128 (product rule) / 256 / 64, two rounds, B = 1; one round B = 4
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r03n_bench_ab.log; : > $L
for round in 1 2; do
  for v in 0 128 256 64; do
    echo "== round $round MV2_PW_DMA=$v (B=1)" >> $L
    MV2_PW_DMA=$v timeout 200 python tools/bench_exp.py --no-cpu-baseline --no-exact --no-soak --steps 200 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('   step %.4f ms  %.1f f/s | sum of launches %.4f ms | conv launches %.4f ms' % (d['ms_per_step'], d['value'], r['all_kernels_ms_per_step'], r['all_conv_launches']['ms_per_step']))" >> $L 2>&1
  done
done
for v in 0 128; do
  echo "== MV2_PW_DMA=$v (B=4)" >> $L
  MV2_PW_DMA=$v timeout 200 python tools/bench_exp.py --config 2 --no-cpu-baseline --no-exact --no-soak --steps 100 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('   step %.4f ms  %.1f f/s | sum of launches %.4f ms | conv launches %.4f ms' % (d['ms_per_step'], d['value'], r['all_kernels_ms_per_step'], r['all_conv_launches']['ms_per_step']))" >> $L 2>&1
done
cat $L
