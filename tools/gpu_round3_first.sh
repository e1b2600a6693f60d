#!/bin/bash
# First GPU call of round 3: cheap C++ harness runs first (seconds), then the full GPU suite, then the bench /
# rocprof set.  Everything lands in gpurun_out/.  Build the variant libraries BEFORE the call (no GPU needed):
#     tools/build_c3_variants.sh p3 "-DMV2_C3_PRIO=3"; hipcc -w -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_cadence.cpp -o tools/mfma_cadence
#     hipcc -w -O2 -std=c++17 tools/k3_bench.cpp -o tools/k3_bench -ldl
# usage (on the GPU box): tools/gpu_round3_first.sh [tag]
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
[ -x tools/mfma_cadence ] && timeout 60 tools/mfma_cadence > $O/${TAG}_mfma_cadence.log 2>&1
[ -x tools/se_bench ] && timeout 120 tools/se_bench magvit2_pytorch_amd/libmv2_hip.so > $O/${TAG}_se_bench.log 2>&1
if [ -x tools/k3_bench ]; then
  K3_DIST=1 timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels > $O/${TAG}_k3_default.log 2>&1
  for v in magvit2_pytorch_amd/libmv2_hip_exp_*.so; do [ -f "$v" ] && K3_DIST=1 timeout 60 tools/k3_bench $v levels > $O/${TAG}_k3_$(basename $v .so | sed s/libmv2_hip_exp_//).log 2>&1; done
  timeout 120 tools/k3_bench magvit2_pytorch_amd/libmv2_hip.so quick > $O/${TAG}_k3_product_quick.log 2>&1
  # generic kernel (feed-forward / projection GEMM shapes): default tile rule only, VGPR-form against AGPR-form MFMAs
  MV2_TILE= timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so generic 2>&1 | awk '/^---- tile 64,64/{exit} {print}' > $O/${TAG}_generic_default.log
  [ -f magvit2_pytorch_amd/libmv2_hip_exp_cvg.so ] && MV2_TILE= timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp_cvg.so generic 2>&1 | awk '/^---- tile 64,64/{exit} {print}' > $O/${TAG}_generic_agpr.log
fi
timeout 2400 python -m pytest tests -m gpu -q -x > $O/${TAG}_gpu_tests.log 2>&1; tail -3 $O/${TAG}_gpu_tests.log
timeout 1500 python -m pytest tests/gpu_extra_random_specs.py -m gpu -q > $O/${TAG}_gpu_random_specs.log 2>&1; tail -3 $O/${TAG}_gpu_random_specs.log
bash tools/gpu_profile.sh $TAG nopmc
grep -h "cycles per MFMA" $O/${TAG}_mfma_cadence.log 2>/dev/null | cut -c1-200
grep -h "^==\|time:\|FAIL\|ALL OK\|SOME" $O/${TAG}_se_bench.log 2>/dev/null | cut -c1-220
grep -h "time new plain\|^==" $O/${TAG}_k3_*.log 2>/dev/null | cut -c1-120 | head -60
