#!/bin/bash
TAG=${1:-r03i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in hw7 hw6 hw3; do
  L=magvit2_pytorch_amd/libmv2_hip_exp_$v.so; [ -f $L ] || continue
  K3_DIST=1 timeout 60 tools/k3_bench $L levels > $O/${TAG}_k3_$v.log 2>&1
done
K3_DIST=1 timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels > $O/${TAG}_k3_default.log 2>&1
timeout 120 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp_hw7.so quick > $O/${TAG}_k3_hw7_quick.log 2>&1
for v in default hw3 hw6 hw7; do echo "== $v"; grep -h "prof plain" $O/${TAG}_k3_$v.log | awk '{print $11,$12,$13}' | tr '\n' ';'; echo; grep -h "time new plain\|time new fused" $O/${TAG}_k3_$v.log | awk '{print $5}' | tr '\n' ' '; echo; done
grep -h "FAIL\|bad [1-9]" $O/${TAG}_k3_hw7_quick.log | head -3
