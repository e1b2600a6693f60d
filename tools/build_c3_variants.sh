#!/bin/bash
# Builds experiment variants of the conv3 kernels (conv3.hip and conv3p.hip are both recompiled under the defines; the spill report is conv3p's)
# Builds experiment variants of the conv3 kernel (tools/k3_bench A/B): libmv2_hip_exp_<tag>.so = the experiments objects with conv3.hip
# recompiled under the given defines.  usage: tools/build_c3_variants.sh tag "-DMV2_C3_RING=9 -DMV2_C3_HSPLIT=1" [tag defs ...]
# Prints VGPR / spill counts of each variant (a spill inside the tap loop is a bug).
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/magvit2_pytorch_amd
python3 -m magvit2_pytorch_amd.build --experiments >/dev/null || exit 1
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DMV2_EXPERIMENTS $defs -c $P/csrc/conv3.hip -o $P/csrc/_obj_exp/conv3_$tag.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DMV2_EXPERIMENTS $defs -c $P/csrc/conv3p.hip -o $P/csrc/_obj_exp/conv3p_$tag.o || exit 1
  extra=""; skip="/conv3"
  if echo "$defs" | grep -q MV2_CONV_; then      # defines of the generic kernel: conv.hip is recompiled too
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DMV2_EXPERIMENTS $defs -c $P/csrc/conv.hip -o $P/csrc/_obj_exp/convg_$tag.o || exit 1
    extra=$P/csrc/_obj_exp/convg_$tag.o; skip="/conv3\|/conv\.o\|/convg_"
  fi
  objs=$(ls $P/csrc/_obj_exp/*.o | grep -v "$skip" | grep -v "/conv3_\|/conv3p_\|/convg_" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $P/csrc/_obj_exp/conv3_$tag.o $P/csrc/_obj_exp/conv3p_$tag.o $extra -o $P/libmv2_hip_exp_$tag.so || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -DMV2_EXPERIMENTS $defs -I$R/include $P/csrc/conv3p.hip -o /tmp/c3_$tag.s 2>/dev/null
  echo "== $tag ($defs)"; grep -E "^\s+\.name: +_Z|\.vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed" /tmp/c3_$tag.s | awk '{printf "%s ", $0} /vgpr_spill/{print ""}'
done
