#!/bin/bash
TAG=${1:-r03h}; SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
MV2_TILE= timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so generic 2>&1 | awk '/^---- tile 64,64/{exit} {print}' > $O/${TAG}_generic_default.log
timeout 120 tools/k3_bench magvit2_pytorch_amd/libmv2_hip.so quick > $O/${TAG}_k3_product_quick.log 2>&1
echo "HEAD sha of this run: $SHA" > $O/${TAG}_gpu_tests.log
timeout 900 python -m pytest tests -m gpu -q -k "lockstep or thirdparty or taylor or conv_out or round_trip or fp32_mode or bf16_drift or headless or random_specs_lockstep" --durations=5 >> $O/${TAG}_gpu_tests.log 2>&1
tail -12 $O/${TAG}_gpu_tests.log | cut -c1-300
(timeout 400 python bench.py --no-cpu-baseline --no-exact --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -1) > $O/${TAG}_bench_b1.log 2>&1; tail -1 $O/${TAG}_bench_b1.log | cut -c1-330
(timeout 300 python bench.py --config 2 --steps 100 --no-cpu-baseline --no-exact 2>&1 | tail -1 | cut -c1-330)
grep -h "^== generic\|prof" $O/${TAG}_generic_default.log | cut -c1-200
grep -h "FAIL\|bad [1-9]" $O/${TAG}_k3_product_quick.log | head -5
python - <<PY
import json
from collections import defaultdict
R=json.load(open('$O/${TAG}_per_launch_events_b1.json'))['rows']
g=defaultdict(lambda:[0,0.0])
for r in R: g[r['what']][0]+=1; g[r['what']][1]+=r['ms']*1000
print(round(sum(r['ms'] for r in R),3), len(R))
for k,(n,ms) in sorted(g.items(), key=lambda kv:-kv[1][1]): print(f'{k:22s} n={n:3d} {ms:8.1f} us  avg {ms/n:6.1f}')
PY
