#!/bin/bash
# conv3 with one 8-wave workgroup per CU (two teams on one tile): correctness on the small shapes, timing at the level shapes
TAG=${1:-r03j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=magvit2_pytorch_amd/libmv2_hip_exp.so
MV2_C3_TEAMS=3 timeout 100 tools/k3_bench $L quick > $O/${TAG}_k3_teams2_quick.log 2>&1; echo "quick rc=$?"
MV2_C3_TEAMS=2 K3_DIST=1 timeout 60 tools/k3_bench $L levels > $O/${TAG}_k3_teams2.log 2>&1; echo "levels rc=$?"
MV2_C3_TEAMS=2 timeout 60 tools/k3_bench $L levels4 > $O/${TAG}_k3_teams2_levels4.log 2>&1
K3_DIST=1 timeout 60 tools/k3_bench $L levels > $O/${TAG}_k3_teams1.log 2>&1
timeout 60 tools/k3_bench $L levels4 > $O/${TAG}_k3_teams1_levels4.log 2>&1
for v in teams1 teams2; do echo "== $v"; grep -h "prof plain" $O/${TAG}_k3_$v.log | cut -c1-175; grep -h "time new plain\|time new fused" $O/${TAG}_k3_$v.log | awk '{print $5}' | tr '\n' ' '; echo; grep -h "time new plain\|time new fused" $O/${TAG}_k3_${v}_levels4.log | awk '{print $5}' | tr '\n' ' '; echo; done
grep -h "new vs old\|vs CPU\|FAIL\|bad" $O/${TAG}_k3_teams2.log | cut -c1-120 | head -8
grep -h "FAIL\|bad [1-9]\|ok\|OK" $O/${TAG}_k3_teams2_quick.log | cut -c1-140 | tail -12
