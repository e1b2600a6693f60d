#!/bin/bash
# Lists VGPR count, scratch bytes and spill count of every kernel in magvit2_pytorch_amd/csrc (gfx950 ISA, no GPU
# needed).  A non-zero scratch / spill column on a hot kernel is a bug.
# usage: tools/check_spills.sh [file.hip ...]        (default: the product sources)
cd "$(dirname "$0")/../magvit2_pytorch_amd/csrc" || exit 1
T=$(mktemp -d)
for f in ${@:-conv3p.hip conv3.hip convin.hip convout.hip pwse.hip ff.hip conv.hip misc.hip attn.hip quant.hip}; do
  b=$(basename "$f" .hip)
  hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -I../../include "$b.hip" -o "$T/$b.s" 2>/dev/null || { echo "$b: compile failed"; continue; }
  grep "\.name: \+_Z\|private_segment_fixed_size\|\.vgpr_count\|vgpr_spill" "$T/$b.s" |
    awk -v F="$b" '/\.name:/{n=$2} /private_segment_fixed_size/{p=$2} /\.vgpr_count/{v=$2} /vgpr_spill/{printf "%-12s %-70s vgpr=%-4s scratch=%-6s spills=%s\n", F, substr(n,1,70), v, p, $2}'
done
rm -rf "$T"
