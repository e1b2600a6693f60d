#!/bin/bash
# Round-4 GPU call: the position-tile conv3 kernel (conv3p.hip) in the harness against the CPU reference / the generic kernel, A/B against
# the round-3 K-split kernel on the level shapes, then (only if the harness is green) the GPU suite and the bench line.
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
LIB=magvit2_pytorch_amd/libmv2_hip_exp.so
MV2_C3_KERNEL=new timeout 400 tools/k3_bench $LIB > $O/${TAG}_k3_new_full.log 2>&1; RC=$?
grep -E "^==|FAIL|ok$|time new|ALL OK|SOME FAILED|rc=" $O/${TAG}_k3_new_full.log | cut -c1-200 | tail -80
MV2_C3_KERNEL=new K3_DIST=1 timeout 200 tools/k3_bench $LIB levels > $O/${TAG}_k3_new_levels_dist.log 2>&1
MV2_C3_KERNEL=old timeout 300 tools/k3_bench $LIB levels4 > $O/${TAG}_k3_old_levels4.log 2>&1
MV2_C3_KERNEL=new timeout 300 tools/k3_bench $LIB levels4 > $O/${TAG}_k3_new_levels4.log 2>&1
echo "---- old"; grep -E "^==|time new" $O/${TAG}_k3_old_levels4.log | cut -c1-160
echo "---- new"; grep -E "^==|time new|prof" $O/${TAG}_k3_new_levels4.log | cut -c1-260
if [ $RC -eq 0 ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.log 2>&1; tail -5 $O/${TAG}_gpu_tests.log | cut -c1-300
  (timeout 400 python bench.py --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -14) > $O/${TAG}_bench_b1.log 2>&1; tail -1 $O/${TAG}_bench_b1.log | cut -c1-600
fi
