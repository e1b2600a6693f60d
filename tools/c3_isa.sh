#!/bin/bash
# Compile conv3.hip to gfx950 ISA (no GPU needed) and print code size, VGPR count and spill counts per kernel.
# usage: tools/c3_isa.sh [out.s]
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-/tmp/conv3.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -I"$R/include" "$R/magvit2_pytorch_amd/csrc/conv3.hip" -o "$O" 2>&1 | grep -v hip-link
/opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$O" -o "$O.elf" 2>/dev/null && /opt/rocm/lib/llvm/bin/llvm-readelf -s "$O.elf" | grep FUNC | awk '{print $3, $8}'
grep -E "^\s+\.name: +_Z|\.vgpr_count|vgpr_spill|sgpr_spill" "$O" | awk '{printf "%s ", $0} /vgpr_spill/{print ""}'
