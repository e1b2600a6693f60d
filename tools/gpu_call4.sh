#!/bin/bash
# conv3 skewed ranges inside a CU's pair of workgroups (MV2_C3_SKEW, per mille; experiments library reads it per launch)
TAG=${1:-r03e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for s in 0 100 160 220 280 340; do
  MV2_C3_SKEW=$s K3_DIST=1 timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels > $O/${TAG}_k3_skew$s.log 2>&1
done
MV2_C3_SKEW=220 timeout 120 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so quick > $O/${TAG}_k3_skew220_quick.log 2>&1
MV2_C3_SKEW=220 timeout 120 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels4 > $O/${TAG}_k3_skew220_levels4.log 2>&1
MV2_C3_SKEW=0 timeout 120 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels4 > $O/${TAG}_k3_skew0_levels4.log 2>&1
for s in 0 100 160 220 280 340; do echo "== skew $s"; grep -h "time new plain\|time new fused\|bad [1-9]" $O/${TAG}_k3_skew$s.log | cut -c1-100 | tr '\n' ' '; echo; done
grep -h "FAIL\|bad [1-9]" $O/${TAG}_k3_skew220_quick.log | head
for s in 0 220; do echo "== levels4 skew $s"; grep -h "time new plain\|time new fused" $O/${TAG}_k3_skew${s}_levels4.log | cut -c1-100 | tr '\n' ' '; echo; done
grep -h "prof plain" $O/${TAG}_k3_skew220.log | cut -c1-250 | head -4
(timeout 400 python bench.py --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -14) > $O/${TAG}_bench_b1.log 2>&1; grep "cpu baseline" $O/${TAG}_bench_b1.log | cut -c1-300; tail -1 $O/${TAG}_bench_b1.log | cut -c1-3500
