#!/bin/bash
# Closing GPU run of a round: the FULL GPU suite on the committed sources (the sha is written into the log), smoke(), then the
# bench / rocprofv3 / PMC set.  usage (on the GPU box): tools/gpu_closing.sh <tag> <sha> [pmc|nopmc]  (also the rocprofv3 stats of the batch-4 shape and of BASELINE configs[3])
TAG=${1:-r03}; SHA=${2:-unknown}; PMC=${3:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "HEAD sha of this run: $SHA" > $O/${TAG}_gpu_tests.log
timeout 1500 python -m pytest tests -m gpu -q --durations=10 >> $O/${TAG}_gpu_tests.log 2>&1; tail -25 $O/${TAG}_gpu_tests.log | cut -c1-300
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4) > $O/${TAG}_smoke.log; cat $O/${TAG}_smoke.log
bash tools/gpu_profile.sh $TAG $PMC b4stats cfg3stats
