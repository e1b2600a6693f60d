#!/bin/bash
TAG=${1:-r04o}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "lockstep_full_channel or two_launch_se or three_launch_se or lockstep_every_kernel or bf16_mode or round_trip" > $O/${TAG}_tests_subset.log 2>&1; tail -4 $O/${TAG}_tests_subset.log | cut -c1-300
for e in 1 0 1 0; do
  if [ $e = 1 ]; then export MV2_SE_TWO=1; else unset MV2_SE_TWO; fi
  (timeout 300 python bench.py --no-cpu-baseline --no-exact --no-batch4 --breakdown $O/${TAG}_events_two$e.json 2>&1 | tail -1 | cut -c1-220) 
done
unset MV2_SE_TWO
python - <<PY
import json
for e in (1, 0):
    d = json.load(open('gpurun_out/${TAG}_events_two%d.json' % e))
    rows = d['rows']
    se = [r for r in rows if r['what'] in ('mv2_se_rec_fc1', 'mv2_se_fc2_apply', 'mv2_se_rec_apply')]
    print('MV2_SE_TWO' if e else 'one launch', len(rows), 'launches; SE record-form launches', len(se), 'sum us', round(sum(r['ms'] for r in se) * 1000, 1), 'all kernels ms', round(sum(r['ms'] for r in rows), 4), [round(r['ms'] * 1000, 1) for r in se][:8])
PY
