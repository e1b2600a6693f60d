#!/bin/bash
# Round-4 GPU call: the tests added / changed this round, smoke() (with the reference copy), the bench line with its new legs.
TAG=${1:-r04f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "c_consumer_launches or full_size_configs or bf16_mode or lockstep_full_channel or odd_shapes" > $O/${TAG}_tests_subset.log 2>&1; tail -4 $O/${TAG}_tests_subset.log | cut -c1-300
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > $O/${TAG}_smoke.log; cat $O/${TAG}_smoke.log
(timeout 600 python bench.py --breakdown $O/${TAG}_per_launch_events_b1.json 2>&1 | tail -16) > $O/${TAG}_bench_b1.log 2>&1; tail -12 $O/${TAG}_bench_b1.log | cut -c1-1500
