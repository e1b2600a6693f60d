#!/bin/bash
TAG=${1:-r03l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for m in 512 256 128 64; do
  MV2_PWSE_MIN64=$m timeout 100 tools/se_bench magvit2_pytorch_amd/libmv2_hip_exp.so b4 > $O/${TAG}_se_min64_$m.log 2>&1
  echo "== min64 $m"; grep -h "^== tail\|time:" $O/${TAG}_se_min64_$m.log | tail -20 | paste - - | awk '{print $3,$4,$5,"|",$0}' | sed 's/.*time: pw_se/ pw_se/' | cut -c1-60
done
grep -h "FAIL\|SOME" $O/${TAG}_se_min64_*.log | head -3
