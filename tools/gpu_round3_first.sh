#!/bin/bash
# First GPU call of a round: cheap C++ harness runs first (seconds), then the FULL GPU suite (no -x: every failure is listed), then the
# bench line.  Everything lands in gpurun_out/.  Build before the call (no GPU needed): tools/prep_round3.sh
# usage (on the GPU box): tools/gpu_round3_first.sh [tag] [sha]
TAG=${1:-r03}; SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
[ -x tools/mfma_cadence ] && timeout 60 tools/mfma_cadence > $O/${TAG}_mfma_cadence.log 2>&1
[ -x tools/se_bench ] && timeout 120 tools/se_bench magvit2_pytorch_amd/libmv2_hip.so > $O/${TAG}_se_bench.log 2>&1
if [ -x tools/k3_bench ]; then
  K3_DIST=1 timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels > $O/${TAG}_k3_default.log 2>&1
  for v in magvit2_pytorch_amd/libmv2_hip_exp_*.so; do [ -f "$v" ] && K3_DIST=1 timeout 60 tools/k3_bench $v levels > $O/${TAG}_k3_$(basename $v .so | sed s/libmv2_hip_exp_//).log 2>&1; done
  timeout 120 tools/k3_bench magvit2_pytorch_amd/libmv2_hip.so quick > $O/${TAG}_k3_product_quick.log 2>&1
  MV2_TILE= timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so generic 2>&1 | awk '/^---- tile 64,64/{exit} {print}' > $O/${TAG}_generic_default.log
  [ -f magvit2_pytorch_amd/libmv2_hip_exp_cvg.so ] && MV2_TILE= timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp_cvg.so generic 2>&1 | awk '/^---- tile 64,64/{exit} {print}' > $O/${TAG}_generic_agpr.log
fi
echo "HEAD sha of this run: $SHA" > $O/${TAG}_gpu_tests.log
timeout 1500 python -m pytest tests -m gpu -q --durations=15 >> $O/${TAG}_gpu_tests.log 2>&1; tail -40 $O/${TAG}_gpu_tests.log | cut -c1-300
(timeout 400 python bench.py --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -12) > $O/${TAG}_bench_b1.log 2>&1; tail -2 $O/${TAG}_bench_b1.log | cut -c1-3000
grep -h "cycles per MFMA" $O/${TAG}_mfma_cadence.log 2>/dev/null | cut -c1-200
grep -h "^==\|time:\|FAIL\|ALL OK\|SOME" $O/${TAG}_se_bench.log 2>/dev/null | cut -c1-220
grep -h "time new plain\|^==" $O/${TAG}_k3_*.log 2>/dev/null | cut -c1-120 | head -60
