#!/bin/bash
# round 3, LDS-DMA pointwise kernel: A/B against conv_kernel (experiments library), the new parity test, the end-to-end tests that
# run the full-size plan, and a bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 200 tools/k3_bench magvit2_pytorch_amd/libmv2_hip_exp.so pwdma > $O/r03m_pwdma_ab.log 2>&1; grep "^##" $O/r03m_pwdma_ab.log | cut -c1-200
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "pointwise_dma or lockstep_every_kernel or bf16_mode_within or drift_within" > $O/r03m_tests.log 2>&1; tail -6 $O/r03m_tests.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-exact > $O/r03m_bench_b1.log 2>&1; tail -1 $O/r03m_bench_b1.log | cut -c1-400
timeout 300 python bench.py --config 2 --no-cpu-baseline --no-exact > $O/r03m_bench_b4.log 2>&1; tail -1 $O/r03m_bench_b4.log | cut -c1-300
