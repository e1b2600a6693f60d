#!/bin/bash
# A/B libraries for tools/k3_bench: libmv2_hip_exp_<tag>.so = the experiments objects with conv3p.o replaced by a variant kept under
# tools/experiments/:
#   tools/build_c3p_variant.sh r5 tools/experiments/conv3p_r5_oneshot.hip            round 5's kernel with its ablation knobs (MV2_C3P_ILV / PRIO / RING / ABL / WDEAD: pass -D... as 3rd argument)
#   tools/build_c3p_variant.sh persist tools/experiments/conv3p_persistent.hip "-DCONV3P_QUEUE_WORDS=8 -mllvm -amdgpu-atomic-optimizer-strategy=None"
#                                                                                    round 6's persistent-workgroup form (needs the queue words behind the tickets: MV2_CONV3_EXTRA_COUNTERS=8 in the environment of the harness)
TAG=${1:?tag}; SRC=${2:?source}; DEFS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/magvit2_pytorch_amd
python3 -m magvit2_pytorch_amd.build --experiments >/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DMV2_EXPERIMENTS $DEFS -I $P/csrc -c $R/$SRC -o $P/csrc/_obj_exp/conv3p_$TAG.o || exit 1
objs=$(ls $P/csrc/_obj_exp/*.o | grep -v "/conv3p\.o\|/conv3p_\|/conv3_\|/convg_" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $P/csrc/_obj_exp/conv3p_$TAG.o -o $P/libmv2_hip_exp_$TAG.so || exit 1
echo built $P/libmv2_hip_exp_$TAG.so
