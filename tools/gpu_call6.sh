#!/bin/bash
TAG=${1:-r03g}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in ht1r9 ht1; do
  L=magvit2_pytorch_amd/libmv2_hip_exp_$v.so; [ -f $L ] || continue
  timeout 120 tools/k3_bench $L quick > $O/${TAG}_k3_${v}_quick.log 2>&1
  K3_DIST=1 timeout 60 tools/k3_bench $L levels > $O/${TAG}_k3_$v.log 2>&1
  timeout 60 tools/k3_bench $L levels4 > $O/${TAG}_k3_${v}_levels4.log 2>&1
done
K3_DIST=1 timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels > $O/${TAG}_k3_default.log 2>&1
timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels4 > $O/${TAG}_k3_default_levels4.log 2>&1
for v in default ht1 ht1r9; do echo "== $v"; grep -h "prof plain" $O/${TAG}_k3_$v.log | awk '{print $9,$10,$11,$12,$13, "|", $NF, $(NF-2)}' | tr '\n' ';'; echo; grep -h "time new plain\|time new fused" $O/${TAG}_k3_$v.log | awk '{print $5}' | tr '\n' ' '; echo; grep -h "time new plain\|time new fused" $O/${TAG}_k3_${v}_levels4.log | awk '{print $5}' | tr '\n' ' '; echo; done
grep -h "FAIL\|bad [1-9]" $O/${TAG}_k3_*_quick.log | head
