#!/bin/bash
# Closing GPU run of a round: the FULL GPU suite on the committed sources (the sha is written into the log), smoke(), then the
# bench / rocprofv3 / PMC set.  usage (on the GPU box): tools/gpu_closing.sh <tag> <sha> [pmc|nopmc]  (also the rocprofv3 stats of the batch-4 shape and of BASELINE configs[3])
TAG=${1:-r03}; SHA=${2:-unknown}; PMC=${3:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "HEAD sha of this run: $SHA" > $O/${TAG}_gpu_tests.log
timeout 1500 python -m pytest tests -m gpu -q --durations=10 >> $O/${TAG}_gpu_tests.log 2>&1; tail -25 $O/${TAG}_gpu_tests.log | cut -c1-300
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4) > $O/${TAG}_smoke.log; cat $O/${TAG}_smoke.log
bash tools/gpu_profile.sh $TAG $PMC b4stats cfg3stats
# round 6 additions: the id-exact 'split' mode's encoder per launch, and the zero-operand DVFS probe of the conv3 kernel (same binary, same
# instruction stream: what the clock allows when the data path is quiet)
(timeout 300 python tools/split_breakdown.py $O/${TAG}_split_mode_encoder_per_launch.json 2>&1 | grep -v Warn | tail -70) > $O/${TAG}_split_mode_breakdown.log 2>&1
if [ -x tools/k3_bench ] && [ -f magvit2_pytorch_amd/libmv2_hip_exp.so ]; then
  for z in 0 1 0 1; do if [ $z = 1 ]; then export K3_ZERO=1; else unset K3_ZERO; fi; echo "#### K3_ZERO=$z (tools/k3_bench levels4: C=64 plain, fused, C=128, 256, 512 T=20/10/5, batch 4 C=64/128/256)"; timeout 200 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels4 2>&1 | grep "time new" | sed "s/time new//" | tr "\n" " "; echo; done > $O/${TAG}_k3_dvfs_probe.log 2>&1
  unset K3_ZERO
fi
