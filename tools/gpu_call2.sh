#!/bin/bash
# GPU call: conv3 HTAB variants (harness: correctness on the small shapes + timing at the level shapes), mv2_ru_tail kernel tests,
# the lockstep tests that now contain mv2_ru_tail, the bf16 drift study, and a bench line with the per-launch table.
TAG=${1:-r03b}; SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in ht1 ht2 ht1p; do
  L=magvit2_pytorch_amd/libmv2_hip_exp_$v.so
  [ -f $L ] || continue
  timeout 120 tools/k3_bench $L quick > $O/${TAG}_k3_${v}_quick.log 2>&1
  K3_DIST=1 timeout 60 tools/k3_bench $L levels > $O/${TAG}_k3_${v}.log 2>&1
done
K3_DIST=1 timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so levels > $O/${TAG}_k3_default.log 2>&1
echo "HEAD sha of this run: $SHA" > $O/${TAG}_gpu_tests.log
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -s -k "ru_tail or lockstep_full_channel or lockstep_every_kernel or bf16_drift or bf16_mode or round_trip" --durations=8 >> $O/${TAG}_gpu_tests.log 2>&1
tail -30 $O/${TAG}_gpu_tests.log | cut -c1-400
(timeout 400 python bench.py --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -12) > $O/${TAG}_bench_b1.log 2>&1; tail -1 $O/${TAG}_bench_b1.log | cut -c1-1500
(MV2_NO_RU_TAIL=1 timeout 400 python bench.py --no-cpu-baseline --no-exact 2>&1 | tail -1) > $O/${TAG}_bench_b1_no_rutail.log 2>&1; tail -1 $O/${TAG}_bench_b1_no_rutail.log | cut -c1-400
grep -h "FAIL\|bad [1-9]" $O/${TAG}_k3_*_quick.log | head -10
grep -h "time new plain\|time new fused\|^==" $O/${TAG}_k3_ht*.log $O/${TAG}_k3_default.log 2>/dev/null | grep -v quick | cut -c1-120 | head -80
grep -h "bf16 drift" $O/${TAG}_gpu_tests.log | cut -c1-600
