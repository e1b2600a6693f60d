// tools/experiments/c3_halo_dma_bench.cpp — MICRO-BENCHMARK for the next round, not part of the library: what would conv3's tap loop gain if
// the halo of a tap group went global -> LDS by buffer_load ... lds instead of global -> VGPR -> ds_write_b128?
//
// It reproduces the loop structure of conv3_kernel, nothing else: 64 output channels x 8 x 16 positions per workgroup,
// 4 waves, K split across waves (wave w owns the 16-byte chunk 2 w + hi of every 128-byte pixel and the whole 64 x 128 accumulator tile),
// per tap and wave 2 weight fragments global -> VGPR (three taps ahead), 4 position fragments from the LDS halo, 8 MFMAs; per group of nine
// taps one (8 + 2) x (16 + 2) halo of 128-byte pixels, double buffered, one barrier per group.
//   variant 0 (the product's form): halo pixels padded to 144 bytes, staged through registers, taps are immediate offsets;
//   variant 1: halo in whole 1 KB wave-rows (8 pixels x 128 bytes) written by DMA, XOR swizzle chunk ^ ((pixel >> 1) & 7) applied on the source
//              side, so every fragment address needs the pixel-dependent XOR (four VALU operations per fragment and tap).
// Both variants read the same data and must produce the same accumulators (checked); the result is clocks per tap and workgroup.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/experiments/c3_halo_dma_bench.cpp -o tools/c3_halo_dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int NPX = 180, HW = 18;
struct P { const void* halo; const void* wfrag; float* out; long long* clocks; uint32_t halo_bytes; int groups; };

template <int V> struct L {
    static constexpr int PITCH = V == 0 ? 144 : 128;
    static constexpr int BUF = ((NPX * PITCH + 1023) / 1024) * 1024 + (V == 1 ? 1024 : 0);      // DMA writes whole 1 KB rows: 23 rows cover 184 pixels
};

// halo source: group g, pixel p at byte (g * 184 + p) * 128 (184 = 23 wave-rows of 8 pixels); weights: [group][tap][wave][a 2][64 lanes][16 B]
template <int V>
__global__ __launch_bounds__(256, 2) void tap_loop(const P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.halo, 0, p.halo_bytes, 0x00020000);
    const uint32_t gstride = 184u * 128u;
    const uint32_t blk_off = (uint32_t)(blockIdx.x % 7) * gstride;                          // workgroups read different groups first (same data volume)
    // ---- staging: 23 wave-rows of 8 pixels per group; wave w takes rows w, w + 4, ... (6 per wave, the last partly out of range)
    uint32_t s_off[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = wave + 4 * i, px = row * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = V == 1 ? (slot ^ ((px >> 1) & 7)) : slot;                          // variant 1: the swizzle on the source side
        s_off[i] = row < 23 ? (uint32_t)(px * 128 + chunk * 16) : p.halo_bytes;              // past the halo: out of range (zeros, no traffic)
    }
    i32x4 hv[6];
    auto stage_issue = [&](int g, int buf) {
        const uint32_t so = (uint32_t)(g % 7) * gstride + blk_off;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if constexpr (V == 1) {
                __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(smem + buf * L<V>::BUF + (wave + 4 * i) * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)s_off[i], (int)so, 0, 0);      // unconditional (row 23: out of range, zeros into the spare row): a load in a branch makes the wait-count pass drain the queue
            } else {
                hv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)s_off[i], (int)so, 0);
            }
        }
    };
    auto stage_write = [&](int buf) {                                                          // variant 0 only: registers -> padded pixels
        if constexpr (V == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int row = wave + 4 * i, px = row * 8 + (lane >> 3);
                if (px < NPX) *(i32x4*)(smem + buf * L<V>::BUF + px * 144 + (lane & 7) * 16) = hv[i];
            }
        }
    };
    // ---- position fragments: tile b = positions 32 b + l31 -> halo pixel of tap (0, 0)
    uint32_t px0[4], base0[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        // odd tile rows rotated by 14 columns, as conv3's pos_of: conflict-free for BOTH layouts under the measured ds_read_b128 pass partition
        const int r = 2 * b + (l31 >> 4), x = ((l31 & 15) + 14 * (l31 >> 4)) & 15;
        px0[b] = (uint32_t)(r * HW + x);
        base0[b] = px0[b] * 144u + (uint32_t)(2 * wave + hi) * 16u;
    }
    const uint32_t my_chunk = (uint32_t)(2 * wave + hi);
    // ---- weights
    const char* wbase = (const char*)p.wfrag + (size_t)wave * 2048 + (size_t)lane * 16;
    auto wfrag = [&](int gt, int a) { return *(const i32x4*)(wbase + ((size_t)(gt % 63) * 4 * 2 + a) * 1024); };    // gt = running tap index
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    i32x4 wf[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t) { wf[t][0] = wfrag(t, 0); wf[t][1] = wfrag(t, 1); }
    stage_issue(0, 0);
    if constexpr (V == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage_write(0);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < p.groups; ++g) {
        const int buf = g & 1;
        stage_issue(g + 1, buf ^ 1);
        const char* hb = smem + buf * L<V>::BUF;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ih = tap / 3, iw = tap - ih * 3, toff = ih * HW + iw;
            i32x4 bf[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if constexpr (V == 0) {
                    bf[b] = *(const i32x4*)(hb + base0[b] + toff * 144);
                } else {
                    const uint32_t px = px0[b] + (uint32_t)toff;
                    bf[b] = *(const i32x4*)(hb + px * 128u + ((my_chunk ^ ((px >> 1) & 7u)) << 4));
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[tap % 3][a]), __builtin_bit_cast(bf16x8, bf[b]), acc[a][b], 0, 0, 0);
            wf[tap % 3][0] = wfrag(g * 9 + tap + 3, 0); wf[tap % 3][1] = wfrag(g * 9 + tap + 3, 1);        // three taps ahead, behind the last use
            __builtin_amdgcn_sched_barrier(0);                                                   // keep the reload HERE (the scheduler sinks it to its use: one tap of flight)
        }
        if constexpr (V == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                 // the DMA rows (older than the six weight fragments in flight)
        stage_write(buf ^ 1);
        if constexpr (V == 1) __builtin_amdgcn_s_barrier(); else __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) p.clocks[blockIdx.x] = t1 - t0;
    // checksum of the accumulators (same for both variants)
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) s += acc[a][b][q];
    p.out[(size_t)blockIdx.x * 256 + tid] = s;
}

static uint32_t rng = 12345u;
static inline float frand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f; }
static inline uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <int V>
static std::vector<float> run(const void* dh, uint32_t hbytes, const void* dw, int blocks, int groups, double* clk_per_tap, double* us) {
    float* dout; long long* dclk;
    CK(hipMalloc(&dout, (size_t)blocks * 256 * 4)); CK(hipMalloc(&dclk, blocks * 8));
    P p{dh, dw, dout, dclk, hbytes, groups};
    const int lds = 2 * L<V>::BUF;
    CK(hipFuncSetAttribute((const void*)tap_loop<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(tap_loop<V>, dim3(blocks), dim3(256), lds, 0, p); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tap_loop<V>, dim3(blocks), dim3(256), lds, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); *us = ms * 100.0;
    std::vector<long long> c(blocks); CK(hipMemcpy(c.data(), dclk, blocks * 8, hipMemcpyDeviceToHost));
    double s = 0; for (long long v : c) s += (double)v;
    *clk_per_tap = s / blocks / (groups * 9.0);
    std::vector<float> o((size_t)blocks * 256); CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
    hipFree(dout); hipFree(dclk);
    return o;
}

int main() {
    const int groups = 36;
    std::vector<uint16_t> halo((size_t)16 * 184 * 64), w((size_t)64 * 4 * 2 * 64 * 8);
    for (auto& v : halo) v = h_f2bf(frand());
    for (auto& v : w) v = h_f2bf(frand() * 0.05f);
    void *dh, *dw; CK(hipMalloc(&dh, halo.size() * 2)); CK(hipMalloc(&dw, w.size() * 2));
    CK(hipMemcpy(dh, halo.data(), halo.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    for (int blocks : {512, 256}) {
        double c0, c1, u0, u1;
        const std::vector<float> a = run<0>(dh, (uint32_t)(halo.size() * 2), dw, blocks, groups, &c0, &u0);
        const std::vector<float> b = run<1>(dh, (uint32_t)(halo.size() * 2), dw, blocks, groups, &c1, &u1);
        size_t diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
        printf("%d workgroups (%d per CU), %d groups of 9 taps: registers + ds_write (padded pixels) %.0f clocks per tap, %.1f us | DMA + swizzle %.0f clocks per tap, %.1f us | "
               "accumulator checksums differing: %zu of %zu; ideal 256 clocks per tap and wave (8 MFMAs)\n", blocks, blocks / 256, groups, c0, u0, c1, u1, diff, a.size());
    }
    return 0;
}
