#!/bin/bash
# conv_kernel with four register sets of gathers in flight on long-K launches (experiments library, MV2_NSET4_MINSTEPS), in the step graph
TAG=${1:-r04n}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_nset4_ab.log; : > $L
for pass in 0 1; do
  for n in 0 16 8; do
    echo "#### B1 pass $pass MV2_NSET4_MINSTEPS=$n" >> $L
    (MV2_NSET4_MINSTEPS=$n timeout 200 python tools/bench_exp.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact --no-batch4 --no-soak --breakdown $O/${TAG}_events_n${n}_p${pass}.json 2>&1 | tail -1 | cut -c1-230) >> $L 2>&1
  done
done
for n in 0 16; do
  echo "#### B4 MV2_NSET4_MINSTEPS=$n" >> $L
  (MV2_NSET4_MINSTEPS=$n timeout 200 python tools/bench_exp.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline --no-exact --no-batch4 --no-soak 2>&1 | tail -1 | cut -c1-230) >> $L 2>&1
done
grep -o "####.*\|ms_per_step\": [0-9.]*" $L | paste - -
python - <<PY
import json
a = {n: json.load(open('$O/${TAG}_events_n%d_p1.json' % n))['rows'] for n in (0, 16, 8)}
for i, r in enumerate(a[0]):
    if r['what'] in ('mv2_conv', 'mv2_conv_rms'):
        t = [a[n][i]['ms'] * 1e3 for n in (0, 16, 8)]
        if max(t) - min(t) > 0.4: print('%-34s off %6.1f  >=16 %6.1f  >=8 %6.1f' % (r['label'], *t))
PY
