#!/bin/bash
# Mid-round GPU call: a test subset (-k expression), then the default bench line with its per-launch table.
# usage: tools/gpu_check.sh <tag> "<pytest -k expression or 'none'>" [bench args...]
TAG=${1:-rXc}; KEXPR=${2:-none}; shift 2 2>/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
if [ "$KEXPR" != "none" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x -k "$KEXPR" > $O/${TAG}_tests_subset.log 2>&1; tail -6 $O/${TAG}_tests_subset.log | cut -c1-300
fi
(timeout 600 python bench.py --breakdown $O/${TAG}_per_launch_events_b1.json "$@" 2>&1 | tail -16) > $O/${TAG}_bench_b1.log 2>&1; tail -12 $O/${TAG}_bench_b1.log | cut -c1-3000
