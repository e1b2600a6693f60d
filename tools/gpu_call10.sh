#!/bin/bash
TAG=${1:-r03k}; SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "HEAD sha of this run: $SHA" > $O/${TAG}_gpu_tests.log
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "conv_in or conv_out or lockstep_full_channel or fp32_mode_vs_reference_golden or bf16_mode or round_trip" --durations=5 >> $O/${TAG}_gpu_tests.log 2>&1
tail -14 $O/${TAG}_gpu_tests.log | cut -c1-300
(timeout 400 python bench.py --no-cpu-baseline --no-exact --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -1) > $O/${TAG}_bench_b1.log 2>&1; tail -1 $O/${TAG}_bench_b1.log | cut -c1-330
(timeout 300 python bench.py --config 2 --steps 100 --no-cpu-baseline --no-exact --breakdown $O/${TAG}_per_launch_events_b4.json 2>&1 | tail -1 | cut -c1-330)
python - <<PY
import json
for t in ('b1','b4'):
    R=json.load(open('$O/${TAG}_per_launch_events_%s.json'%t))['rows']
    print(t, round(sum(r['ms'] for r in R),3), {r['label']: round(r['ms']*1000,1) for r in R if r['label'] in ('conv_in','conv_out')})
PY
