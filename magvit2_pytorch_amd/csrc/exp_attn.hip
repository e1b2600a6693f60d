// EXPERIMENTS LIBRARY ONLY: taylor_kv_frame_kernel (attn.hip, verbatim) reading a HEAD-MAJOR k / v tensor [frame][head][token][k 8 | v 8]
// instead of the projection GEMM's rows [frame][token][k of 16 heads | v of 16 heads].  profiles/r04_findings.md section 10: the product
// kernel fetches 82 MB for a 10.5 MB tensor (16 + 16 bytes of every 512-byte row per (frame, head) workgroup); this variant measures what the
// state kernel would cost if the projection's epilogue wrote the head-major layout (tools/taylor_hm_bench.cpp transposes on the host).
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/mv2.h"
#define TAY_F 73
#define TAY_S (73 * 9)

template <typename E>
__global__ __launch_bounds__(256) void taylor_kv_frame_hm_kernel(const void* __restrict__ kv, float* __restrict__ state, int n, int heads) {
    __shared__ float red[4][8][92];
    const int f = blockIdx.y, h = blockIdx.x;
    const int tid = threadIdx.x, i = tid & 7, tg = tid >> 3, wave = tid >> 6;
    constexpr int NC = 8 / E::kVec, U = 4;
    float a2[8][9], a1[9], a0[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) { a1[e] = 0.f; a0[e] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) a2[jj][e] = 0.f; }
    const float r2 = 0.70710678118654752440f;
    for (int t0 = tg; t0 < n; t0 += 32 * U) {
        i32x4 kr[U][NC], vr[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(t0 + 32 * u, n - 1);                  // clamped: the tail trip re-reads a valid row and is masked below
            const int64_t row = (((int64_t)f * heads + h) * n + t) * 16;      // head-major: [frame][head][token][k 8 | v 8]
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                kr[u][c] = *(const i32x4*)((const char*)kv + row * E::kBytes + c * 16);
                vr[u][c] = *(const i32x4*)((const char*)kv + (row + 8) * E::kBytes + c * 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float k[8], v[9];
#pragma unroll
            for (int c = 0; c < NC; ++c) { Chunk<E>::unpack(kr[u][c], k + c * E::kVec); Chunk<E>::unpack(vr[u][c], v + c * E::kVec); }
            const bool live = t0 + 32 * u < n;
            v[8] = live ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = live ? v[e] : 0.f;
            float ki = k[0];
#pragma unroll
            for (int jj = 1; jj < 8; ++jj) ki = (i == jj) ? k[jj] : ki;
#pragma unroll
            for (int e = 0; e < 9; ++e) { a0[e] += v[e]; a1[e] += ki * v[e]; }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float ph = (ki * k[jj]) * r2;
#pragma unroll
                for (int e = 0; e < 9; ++e) a2[jj][e] += ph * v[e];
            }
        }
    }
    // the eight token groups of a wave: lanes with equal i
    auto wred = [&](float x) { x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64); return x; };
#pragma unroll
    for (int e = 0; e < 9; ++e) { a0[e] = wred(a0[e]); a1[e] = wred(a1[e]);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) a2[jj][e] = wred(a2[jj][e]); }
    if ((tid & 63) < 8) {
        float* r = red[wave][i];
#pragma unroll
        for (int e = 0; e < 9; ++e) { r[e] = a0[e]; r[9 + e] = a1[e];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) r[18 + jj * 9 + e] = a2[jj][e]; }
    }
    __syncthreads();
    float* dst = state + ((int64_t)f * heads + h) * TAY_S;
    for (int x = tid; x < 8 * 90; x += 256) {
        const int ii = x / 90, r = x - ii * 90;
        const float s = ((red[0][ii][r] + red[1][ii][r]) + red[2][ii][r]) + red[3][ii][r];
        if (r < 9) { if (ii == 0) dst[r] = s; }                                   // row 0: sum of [v, 1] (every i holds a copy)
        else if (r < 18) dst[(1 + ii) * 9 + (r - 9)] = s;                          // row 1 + i
        else dst[(9 + 8 * ii + (r - 18) / 9) * 9 + (r - 18) % 9] = s;             // row 9 + 8 i + j
    }
}


extern "C" int mv2x_taylor_kv_head_major(int dtype, const void* kv_hm, float* state, int frames, int n, int heads, void* stream) {
    if (!kv_hm || !state || frames <= 0 || n <= 0 || heads <= 0) return MV2_EINVAL;
    dim3 grid((unsigned)heads, (unsigned)frames);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_kv_frame_hm_kernel<BF16>), grid, dim3(256), 0, st, kv_hm, state, n, heads);
    else if (dtype == MV2_F32) hipLaunchKernelGGL((taylor_kv_frame_hm_kernel<F32>), grid, dim3(256), 0, st, kv_hm, state, n, heads);
    else return MV2_EINVAL;
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
