#!/bin/bash
# GPU call: SE tail kernels + generic GEMM harness (correctness vs CPU + timing), lockstep / drift tests, bench line with per-launch table
TAG=${1:-r03c}; SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 tools/se_bench magvit2_pytorch_amd/libmv2_hip.so > $O/${TAG}_se_bench.log 2>&1
MV2_TILE= timeout 60 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so generic 2>&1 | awk '/^---- tile 64,64/{exit} {print}' > $O/${TAG}_generic_default.log
echo "HEAD sha of this run: $SHA" > $O/${TAG}_gpu_tests.log
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -s -k "lockstep or bf16_drift or bf16_mode or round_trip or pw_se or fp32_mode" --durations=5 >> $O/${TAG}_gpu_tests.log 2>&1
tail -25 $O/${TAG}_gpu_tests.log | cut -c1-400
(timeout 400 python bench.py --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -12) > $O/${TAG}_bench_b1.log 2>&1; tail -1 $O/${TAG}_bench_b1.log | cut -c1-1200
grep -h "^==\|time:\|FAIL\|ALL OK\|SOME" $O/${TAG}_se_bench.log 2>/dev/null | tail -14 | cut -c1-220
grep -h "^== generic" $O/${TAG}_generic_default.log | cut -c1-150
grep -h "bf16 drift" $O/${TAG}_gpu_tests.log | cut -c1-700
