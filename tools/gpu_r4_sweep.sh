#!/bin/bash
# Round-4 GPU call: variants of the position-tile conv3 kernel in the harness (tools/build_c3_variants.sh builds them):
# interleave form, K parts per tile (MV2_C3P_S), workgroups per CU (MV2_C3P_LDS).  usage: tools/gpu_r4_sweep.sh <tag> "<variant tags>" "<S values>"
TAG=${1:-r04b}; VARS=${2:-"ilv0 ilv1"}; SVALS=${3:-"1 2 3 4 6 8"}; BEST=${4:-ilv1}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export MV2_C3_KERNEL=new
summ() { grep -E "^== C|time new|prof|epilogue parts|FAIL|ALL OK|SOME FAILED" $1 | sed -e 's/class 64co x 8x16 (fused tiles [0-9x]*, //' | cut -c1-250; }
for v in $VARS; do
  timeout 300 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp_$v.so quick > $O/${TAG}_k3_${v}_quick.log 2>&1
  echo "######## $v"; summ $O/${TAG}_k3_${v}_quick.log | grep -v "^== C.*T=[1-5] \|ok$" 
done
L=magvit2_pytorch_amd/libmv2_hip_exp_$BEST.so
for s in $SVALS; do
  MV2_C3P_S=$s timeout 200 tools/k3_bench $L levels > $O/${TAG}_k3_${BEST}_S$s.log 2>&1
  echo "######## $BEST S=$s"; grep -E "^== C|time new plain|FAIL" $O/${TAG}_k3_${BEST}_S$s.log | sed -e 's/class.*//' | paste - - | cut -c1-120
done
MV2_C3P_LDS=60000 timeout 200 tools/k3_bench $L levels > $O/${TAG}_k3_${BEST}_2wg.log 2>&1
echo "######## $BEST two workgroups per CU"; grep -E "^== C|time new|FAIL" $O/${TAG}_k3_${BEST}_2wg.log | sed -e 's/class.*//' | paste - - | cut -c1-160
