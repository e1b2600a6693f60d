#!/bin/bash
TAG=${1:-r04h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_thirdparty_pins_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "taylor or lockstep_every_kernel or lockstep_full_channel or bf16_drift" > $O/${TAG}_tests_subset.log 2>&1; tail -4 $O/${TAG}_tests_subset.log | cut -c1-300
(timeout 300 python bench.py --no-cpu-baseline --no-exact --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -3) > $O/${TAG}_bench_b1.log 2>&1; tail -1 $O/${TAG}_bench_b1.log | cut -c1-300
python - <<'PY'
import json
d = json.load(open('gpurun_out/%s_per_launch_events_b1.json' % '${TAG}'))
for r in d['rows']:
    if 'taylor' in r['what'] or 'fn.q' in r['label'] or 'fn.kv' in r['label']: print(r['i'], r['label'], round(r['ms']*1000, 1))
print('sum', round(sum(r['ms'] for r in d['rows']), 4), len(d['rows']))
PY
