// tools/mfma_cadence.cpp — how many shader cycles does ONE wave need per v_mfma_f32_32x32x16_bf16 when it issues them
// back to back, and what changes it?  Written after the round-2 finding (profiles/r02_k3_findings.md) that the conv3 tap loop
// runs at ~50 cycles per MFMA with one wave per SIMD — even with every load and LDS read removed — while two waves per SIMD
// reach ~33.6 (the pipe's 32-cycle cadence).  The variants below separate the candidate causes:
//   acc    : number of independent accumulators the stream cycles through (8 = conv3's wave tile, 4, 2, 1 = dependent chain)
//   regs   : __launch_bounds__ second argument (1 -> 512-register budget: accumulators may go to AGPRs; 2 -> 256, all arch VGPRs)
//   waves  : waves per SIMD actually resident (blocks per CU x waves per block / 4)
//   filler : independent VALU instructions between two MFMAs (0, 1, 2)
// Build (no GPU needed): hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_cadence.cpp -o tools/mfma_cadence
// Run on the GPU box:    tools/mfma_cadence            (prints one line per variant, ~1 s)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// One wave issues ITERS x NACC MFMAs over NACC independent accumulators; FILL independent v_fma between two MFMAs.
// AGPR = 1 forces the accumulators into the accumulator file through an inline-asm "+a" constraint (under a 256-register budget the
// compiler otherwise picks the VGPR form of the MFMA: C / D in arch VGPRs, .agpr_count 0)
template <int NACC, int FILL, int MINB, int AGPR = 0>
__global__ __launch_bounds__(256, MINB) void cadence_kernel(unsigned long long* out, float* sink, int iters) {
    extern __shared__ char lds_pad[];      // dynamic LDS only pins the residency: 96 / 64 / 36 KB per block -> exactly 1 / 2 / 4 blocks per CU
    if (iters < 0) sink[1] = lds_pad[threadIdx.x];
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = (float)(threadIdx.x + a + r) * 1e-9f;
    bf16x8 va, vb;
#pragma unroll
    for (int k = 0; k < 8; ++k) { va[k] = (__bf16)(0.001f * (float)((threadIdx.x + k) & 7)); vb[k] = (__bf16)(0.002f * (float)((threadIdx.x * 3 + k) & 7)); }
    float f0 = (float)threadIdx.x, f1 = 1.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[a]) : "v"(va), "v"(vb));
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[a], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < FILL; ++q) f0 = __builtin_fmaf(f0, f1, 0.5f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = f0;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) sink[0] = s;                     // keeps the work alive
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int FILL, int MINB, int AGPR = 0>
static int run(const char* name, int blocks_per_cu, int cus, unsigned long long* d_out, float* d_sink) {
    const int iters = 2000, blocks = blocks_per_cu * cus;
    const int lds = blocks_per_cu == 1 ? 96 * 1024 : (blocks_per_cu == 2 ? 64 * 1024 : 36 * 1024);
    CK(hipFuncSetAttribute((const void*)cadence_kernel<NACC, FILL, MINB, AGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipMemset(d_out, 0, (size_t)blocks * 4 * sizeof(unsigned long long)));
    hipLaunchKernelGGL((cadence_kernel<NACC, FILL, MINB, AGPR>), dim3(blocks), dim3(256), lds, 0, d_out, d_sink, 16);      // warm-up
    hipLaunchKernelGGL((cadence_kernel<NACC, FILL, MINB, AGPR>), dim3(blocks), dim3(256), lds, 0, d_out, d_sink, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0, mx = 0; for (auto v : h) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
    const double per = sum / h.size() / ((double)iters * NACC), per_max = mx / ((double)iters * NACC);
    printf("%-52s acc %d filler %d launch_bounds(256,%d) blocks/CU %d: %6.1f cycles per MFMA per wave (slowest wave %6.1f)\n", name, NACC, FILL, MINB, blocks_per_cu, per, per_max);
    return 0;
}

int main() {
    int dev = 0, cus = 256; CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    unsigned long long* d_out; float* d_sink;
    CK(hipMalloc((void**)&d_out, (size_t)8 * cus * 4 * sizeof(unsigned long long))); CK(hipMalloc((void**)&d_sink, 64));
    // one wave per SIMD
    run<8, 0, 1>("1 wave/SIMD, 512-register budget", 1, cus, d_out, d_sink);
    run<8, 0, 2>("1 wave/SIMD, 256-register budget (arch VGPR acc)", 1, cus, d_out, d_sink);
    run<8, 0, 2, 1>("1 wave/SIMD, 256-register budget, AGPR acc (asm)", 1, cus, d_out, d_sink);
    run<4, 0, 2>("1 wave/SIMD, 4 accumulators", 1, cus, d_out, d_sink);
    run<2, 0, 2>("1 wave/SIMD, 2 accumulators", 1, cus, d_out, d_sink);
    run<1, 0, 2>("1 wave/SIMD, dependent chain", 1, cus, d_out, d_sink);
    run<8, 1, 2>("1 wave/SIMD, one VALU filler per MFMA", 1, cus, d_out, d_sink);
    run<8, 2, 2>("1 wave/SIMD, two VALU fillers per MFMA", 1, cus, d_out, d_sink);
    // two waves per SIMD (two workgroups per CU, as conv3 runs)
    run<8, 0, 2>("2 waves/SIMD (arch VGPR acc)", 2, cus, d_out, d_sink);
    run<8, 0, 2, 1>("2 waves/SIMD, AGPR acc (asm)", 2, cus, d_out, d_sink);
    run<8, 1, 2>("2 waves/SIMD, one VALU filler per MFMA", 2, cus, d_out, d_sink);
    // four waves per SIMD: needs <= 128 registers -> 4 accumulators
    run<4, 0, 4>("4 waves/SIMD, 4 accumulators", 4, cus, d_out, d_sink);
    hipFree(d_out); hipFree(d_sink);
    return 0;
}
