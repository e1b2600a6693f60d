// tools/experiments/pw_dma_bench.cpp — PROTOTYPE, not part of the library: does LDS-DMA staging (buffer_load ... lds) pay for the pointwise
// GEMMs of the path?  (profiles/r03_findings.md: the generic kernel's K loop runs at 27 % of the MFMA rate with the
// LDS-write path and the L1s both half busy and a barrier per step.)
//
//   C[m][n] = sum_k X[m][k] W[n][k]        X: M x K bf16 rows (activations), W: N x K bf16 rows (weights), C: M x N bf16
//
// Kernel: BM x BN tile, 4 waves (2 x 2), K step = 64 channels (128-byte rows), NS LDS stages.  Both operands go global -> LDS directly:
// a wave-instruction fills 8 rows x 128 bytes = 1 KB of a stage (lane l -> row l / 8, 16-byte slot l % 8); the XOR swizzle that keeps
// the MFMA fragment reads conflict-free is applied on the SOURCE side (lane l fetches chunk (l % 8) ^ ((row >> 1) & 7)).  No register
// staging, no ds_write; waits are hand-placed (s_waitcnt vmcnt(N) never 0 in the loop), one barrier per step.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/experiments/pw_dma_bench.cpp -o tools/pw_dma_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct P { const void* x; const void* w; void* c; int M, N, K; uint32_t xbytes, wbytes; };

template <int N_> struct WaitVm;
#define WAITVM(n) template <> struct WaitVm<n> { static __device__ __forceinline__ void run() { asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); } };
WAITVM(0) WAITVM(2) WAITVM(3) WAITVM(4) WAITVM(6) WAITVM(8) WAITVM(9) WAITVM(12) WAITVM(16) WAITVM(18) WAITVM(24)

__device__ __forceinline__ unsigned short f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); }

// BM rows of X and BN rows of W per tile; WM x WN MFMA tiles (32 x 32) per wave, 2 x 2 waves
template <int BM, int BN, int NS>
__global__ __launch_bounds__(256, 2) void pw_dma_kernel(const P p) {
    constexpr int WM = BM / 64, WN = BN / 64;                    // MFMA tiles per wave along M (positions) and N (channels)
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int LPS = (BM + BN) * 8 / 256;                     // DMA loads per thread and step
    static_assert(NS >= 2 && NS * STAGE <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int ntn = (p.N + BN - 1) / BN;
    const int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nsteps = p.K / 64;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
    // per-thread source offsets: load i covers rows 8 (wave + 4 i) + lane / 8 of the stage image [W rows BN | X rows BM]
    uint32_t voff[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int row = 8 * (wave + 4 * i) + (lane >> 3);        // row inside the stage image
        const int slot = lane & 7, chunk = slot ^ ((row >> 1) & 7);
        if (row < BN) { const int n = n0 + row; voff[i] = n < p.N ? (uint32_t)n * (uint32_t)p.K * 2u + chunk * 16u : p.wbytes; }
        else { const int m = m0 + row - BN; voff[i] = m < p.M ? (uint32_t)m * (uint32_t)p.K * 2u + chunk * 16u : p.xbytes; }
    }
    auto issue = [&](int s, int stage) {                         // every load unconditional: steps past the end read out of range -> zeros
        const bool live = s < nsteps;                             // past the end: out-of-range VGPR offset (the range check does not see soffset)
        const uint32_t so = live ? (uint32_t)s * 128u : 0u;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int rowbase = 8 * (wave + 4 * i);
            __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(smem + stage * STAGE + rowbase * 128);
            if (rowbase < BN) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, (int)(live ? voff[i] : p.wbytes), (int)so, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, (int)(live ? voff[i] : p.xbytes), (int)so, 0, 0);
        }
    };
    f32x16 acc[WN][WM];
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const uint32_t swz = (uint32_t)((l31 >> 1) & 7);
    const uint32_t rd_w = (uint32_t)(wn * 32 * WN + l31) * 128u, rd_x = (uint32_t)(BN + wm * 32 * WM + l31) * 128u;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s);                // prologue: NS - 1 steps in flight
    for (int s = 0; s < nsteps; ++s) {
        WaitVm<(NS - 2) * LPS>::run();                           // this thread's loads of step s have landed
        __builtin_amdgcn_s_barrier();                                       // ... everybody's, and everybody is done reading stage (s - 1) % NS
        issue(s + NS - 1, (s + NS - 1) % NS);
        const char* base = smem + (s % NS) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t ck = (uint32_t)(((ks * 2 + hi) ^ swz) << 4);
            i32x4 fw[WN], fx[WM];
#pragma unroll
            for (int a = 0; a < WN; ++a) fw[a] = *(const i32x4*)(base + rd_w + a * 32 * 128 + ck);
#pragma unroll
            for (int b = 0; b < WM; ++b) fx[b] = *(const i32x4*)(base + rd_x + b * 32 * 128 + ck);
#pragma unroll
            for (int a = 0; a < WN; ++a)
#pragma unroll
                for (int b = 0; b < WM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[a]), __builtin_bit_cast(bf16x8, fx[b]), acc[a][b], 0, 0, 0);
        }
    }
    WaitVm<0>::run();
    __syncthreads();
    // epilogue: acc[a][b][4 g + j] = C[m = .. + l31][n = .. + 8 g + 4 hi + j]: bf16 tile in LDS, 16-byte row stores
    constexpr int PITCH = BN * 2 + 16;
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WM; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = (wn * WN + a) * 32 + 8 * g + 4 * hi, row = (wm * WM + b) * 32 + l31;
                i32x2 v;
                v[0] = (int)((unsigned)f2bf(acc[a][b][4 * g]) | ((unsigned)f2bf(acc[a][b][4 * g + 1]) << 16));
                v[1] = (int)((unsigned)f2bf(acc[a][b][4 * g + 2]) | ((unsigned)f2bf(acc[a][b][4 * g + 3]) << 16));
                *(i32x2*)(smem + row * PITCH + col * 2) = v;
            }
    __syncthreads();
    constexpr int CPR = BN / 8;
    for (int q = tid; q < BM * CPR; q += 256) {
        const int row = q / CPR, ch = q % CPR, m = m0 + row, n = n0 + ch * 8;
        if (m < p.M && n < p.N) *(i32x4*)((char*)p.c + ((size_t)m * p.N + n) * 2) = *(const i32x4*)(smem + row * PITCH + ch * 16);
    }
}

// ---- variant: K step = 32 channels (64-byte rows), so that NS stages cost half the LDS: NS = 4 is 64 KB (two workgroups per CU, as NS = 2
// above) with a prefetch distance of three half-steps instead of two.  A wave-instruction fills 16 rows x 64 bytes; swizzle over the four
// 16-byte slots of a row: slot p of row r receives chunk p ^ ((r >> 2) & 3) (conflict-free for the 16-lane groups of ds_read_b128).
template <int BM, int BN, int NS>
__global__ __launch_bounds__(256, 2) void pw_dma32_kernel(const P p) {
    constexpr int WM = BM / 64, WN = BN / 64;
    constexpr int STAGE = (BM + BN) * 64;
    constexpr int LPS = (BM + BN) * 4 / 256;                     // DMA loads per thread and half-step
    static_assert(NS >= 2 && NS * STAGE <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int ntn = (p.N + BN - 1) / BN;
    const int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nhalf = p.K / 32;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
    uint32_t voff[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int row = 16 * (wave + 4 * i) + (lane >> 2);       // row inside the stage image [W rows BN | X rows BM]
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        if (row < BN) { const int n = n0 + row; voff[i] = n < p.N ? (uint32_t)n * (uint32_t)p.K * 2u + chunk * 16u : p.wbytes; }
        else { const int m = m0 + row - BN; voff[i] = m < p.M ? (uint32_t)m * (uint32_t)p.K * 2u + chunk * 16u : p.xbytes; }
    }
    auto issue = [&](int t, int stage) {
        const bool live = t < nhalf;
        const uint32_t so = live ? (uint32_t)t * 64u : 0u;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int rowbase = 16 * (wave + 4 * i);
            __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(smem + stage * STAGE + rowbase * 64);
            if (rowbase < BN) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, (int)(live ? voff[i] : p.wbytes), (int)so, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, (int)(live ? voff[i] : p.xbytes), (int)so, 0, 0);
        }
    };
    f32x16 acc[WN][WM];
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const uint32_t swz = (uint32_t)((l31 >> 2) & 3);
    const uint32_t rd_w = (uint32_t)(wn * 32 * WN + l31) * 64u, rd_x = (uint32_t)(BN + wm * 32 * WM + l31) * 64u;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t, t);
    int stage = 0;
    for (int t = 0; t < nhalf; ++t) {
        WaitVm<(NS - 2) * LPS>::run();
        __builtin_amdgcn_s_barrier();
        issue(t + NS - 1, stage == 0 ? NS - 1 : stage - 1);
        const char* base = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t ck = (uint32_t)(((ks * 2 + hi) ^ swz) << 4);
            i32x4 fw[WN], fx[WM];
#pragma unroll
            for (int a = 0; a < WN; ++a) fw[a] = *(const i32x4*)(base + rd_w + a * 32 * 64 + ck);
#pragma unroll
            for (int b = 0; b < WM; ++b) fx[b] = *(const i32x4*)(base + rd_x + b * 32 * 64 + ck);
#pragma unroll
            for (int a = 0; a < WN; ++a)
#pragma unroll
                for (int b = 0; b < WM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[a]), __builtin_bit_cast(bf16x8, fx[b]), acc[a][b], 0, 0, 0);
        }
        stage = stage == NS - 1 ? 0 : stage + 1;
    }
    WaitVm<0>::run();
    __syncthreads();
    constexpr int PITCH = BN * 2 + 16;
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WM; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = (wn * WN + a) * 32 + 8 * g + 4 * hi, row = (wm * WM + b) * 32 + l31;
                i32x2 v;
                v[0] = (int)((unsigned)f2bf(acc[a][b][4 * g]) | ((unsigned)f2bf(acc[a][b][4 * g + 1]) << 16));
                v[1] = (int)((unsigned)f2bf(acc[a][b][4 * g + 2]) | ((unsigned)f2bf(acc[a][b][4 * g + 3]) << 16));
                *(i32x2*)(smem + row * PITCH + col * 2) = v;
            }
    __syncthreads();
    constexpr int CPR = BN / 8;
    for (int q = tid; q < BM * CPR; q += 256) {
        const int row = q / CPR, ch = q % CPR, m = m0 + row, n = n0 + ch * 8;
        if (m < p.M && n < p.N) *(i32x4*)((char*)p.c + ((size_t)m * p.N + n) * 2) = *(const i32x4*)(smem + row * PITCH + ch * 16);
    }
}

static uint32_t rng = 12345u;
static inline float frand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f; }
static inline uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float h_bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int BM, int BN, int NS, bool HALF = false>
static void run(const char* name, int M, int N, int K, int reps) {
    std::vector<uint16_t> x((size_t)M * K), w((size_t)N * K), c((size_t)M * N);
    for (auto& v : x) v = h_f2bf(frand());
    for (auto& v : w) v = h_f2bf(frand() * 0.1f);
    void *dx, *dw, *dc;
    CK(hipMalloc(&dx, x.size() * 2)); CK(hipMalloc(&dw, w.size() * 2)); CK(hipMalloc(&dc, c.size() * 2));
    CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dc, 0xff, c.size() * 2));
    P p{dx, dw, dc, M, N, K, (uint32_t)(x.size() * 2), (uint32_t)(w.size() * 2)};
    constexpr int STAGE = (BM + BN) * (HALF ? 64 : 128), EPI = BM * (BN * 2 + 16);
    const int lds = NS * STAGE > EPI ? NS * STAGE : EPI;
    void (*kern)(const P);
    if constexpr (HALF) kern = pw_dma32_kernel<BM, BN, NS>; else kern = pw_dma_kernel<BM, BN, NS>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c.data(), dc, c.size() * 2, hipMemcpyDeviceToHost));
    // check 4096 sampled elements against fp64
    int bad = 0; double maxd = 0;
    for (int t = 0; t < 4096; ++t) {
        rng = rng * 1664525u + 1013904223u; const int m = (rng >> 4) % M;
        rng = rng * 1664525u + 1013904223u; const int n = (rng >> 4) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)h_bf2f(x[(size_t)m * K + k]) * h_bf2f(w[(size_t)n * K + k]);
        const double d = fabs(ref - h_bf2f(c[(size_t)m * N + n]));
        if (d > maxd) maxd = d;
        if (d > 0.02 + 0.01 * fabs(ref)) ++bad;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, p);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    printf("%-28s tile %3dx%-3d %s stages %d  M=%-6d K=%-5d N=%-5d  %7.1f us  %6.0f TF/s  max|d| %.3e bad %d  (%d blocks, %d KB LDS)\n", name, BM, BN, HALF ? "k32" : "k64", NS, M, K, N, us,
           2.0 * M * N * K / us / 1e6, maxd, bad, blocks, lds / 1024);
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dc));
}

int main() {
    // shapes of the path (batch 1 and 4): the generic kernel's times on the same shapes are in profiles/r03_generic_default_phase_clocks.log
    struct S { const char* name; int M, N, K; } shapes[] = {
        {"ff1 C=512 T=20 (32.8 us)", 5120, 2816, 512}, {"ff2 C=512 T=20 (21.7 us)", 5120, 512, 1408}, {"ff1 C=256 T=20 (41.4 us)", 20480, 1408, 256},
        {"ff2 C=256 T=20 (22.1 us)", 20480, 256, 704}, {"qkv C=512 T=20 (13.8 us)", 5120, 768, 512}, {"out C=512 T=20 (8.9 us)", 5120, 512, 256},
        {"kv C=256 T=20 (13.2 us)", 20480, 256, 256}, {"up 128->256 (46 us)", 81920, 256, 128}, {"ff1 B=4 C=512 (110 us)", 20480, 2816, 512},
        {"ff2 B=4 C=512 (59.5 us)", 20480, 512, 1408}};
    for (const S& s : shapes) {
        run<128, 128, 2>(s.name, s.M, s.N, s.K, 50);
        run<128, 128, 3>(s.name, s.M, s.N, s.K, 50);
        run<128, 128, 3, true>(s.name, s.M, s.N, s.K, 50);
        run<128, 128, 4, true>(s.name, s.M, s.N, s.K, 50);
        run<128, 128, 6, true>(s.name, s.M, s.N, s.K, 50);
        run<64, 64, 4>(s.name, s.M, s.N, s.K, 50);
        run<64, 64, 8, true>(s.name, s.M, s.N, s.K, 50);
    }
    return 0;
}
