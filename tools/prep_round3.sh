#!/bin/bash
# Run in the authoring container (no GPU needed) before the first gpurun call of round 3: builds the product + experiments
# libraries, the conv3 experiment variants, the kernel harness and the MFMA cadence micro-benchmark.  Then:
#     gpurun --timeout 3000 -- 'bash tools/gpu_round3_first.sh r03'
cd "$(dirname "$0")/.." || exit 1
python3 -m magvit2_pytorch_amd.build || exit 1
bash tools/build_c3_variants.sh \
    p3  "-DMV2_C3_PRIO=3" \
    g2  "-DMV2_C3_AGPR=2 -DMV2_C3_HPRO=1" \
    g2h "-DMV2_C3_AGPR=2 -DMV2_C3_HPRO=1 -DMV2_C3_HSPLIT=1 -DMV2_C3_PSETS=2" \
    g2l "-DMV2_C3_AGPR=2 -DMV2_C3_HPRO=1 -DMV2_C3_LATEPF=2" \
    hpro "-DMV2_C3_HPRO=1" \
    oq "-DMV2_C3_OWNQ=1" \
    cvg "-DMV2_CONV_AGPR=1" \
    ps2 "-DMV2_C3_PSETS=2" | grep -E "^==|BF16Lb0" | cut -c1-200
/opt/rocm/bin/hipcc -w -O2 -std=c++17 tools/k3_bench.cpp -o tools/k3_bench -ldl || exit 1
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_cadence.cpp -o tools/mfma_cadence || exit 1
/opt/rocm/bin/hipcc -w -O2 -std=c++17 tools/se_bench.cpp -o tools/se_bench -ldl || exit 1
ls -la tools/k3_bench tools/mfma_cadence tools/se_bench magvit2_pytorch_amd/*.so | awk '{print $5, $9}'
