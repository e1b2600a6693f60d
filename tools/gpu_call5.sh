#!/bin/bash
# conv3 tap-loop ablations (timing only): which component sets the solo rate (the T = 5 level runs one workgroup per CU)
TAG=${1:-r03f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in a4 a5 a6 a7 a8; do
  K3_DIST=1 timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp_$v.so levels > $O/${TAG}_k3_$v.log 2>&1
done
K3_DIST=1 timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels > $O/${TAG}_k3_default.log 2>&1
for v in default a4 a5 a6 a7 a8; do echo "== $v"; grep -h "prof plain" $O/${TAG}_k3_$v.log | cut -c1-170; grep -h "time new plain" $O/${TAG}_k3_$v.log | cut -c1-60 | tr '\n' ' '; echo; done
for b in 512 256 128; do (MV2_TAYLOR_BLOCKS=$b timeout 200 python bench.py --no-cpu-baseline --no-exact --steps 100 --breakdown $O/${TAG}_events_taylor$b.json 2>&1 | tail -1 | cut -c1-200); python - <<PY
import json
R=json.load(open('$O/${TAG}_events_taylor$b.json'))['rows']
print('taylor blocks $b:', {k: round(sum(r['ms'] for r in R if r['what']==k)*1000,1) for k in ('mv2_taylor_kv','mv2_taylor_apply')})
PY
done
