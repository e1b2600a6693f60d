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

// ------------------------------------------------------------------------------------------------------------------------------------
// EXPERIMENTS ONLY: the Taylor state on the matrix pipe (bf16 inputs).  Per (frame, head):
//     State[f][e] = sum_n phi_f(k_n) * ve_e(n),   f in [0, 73), ve = [v_0 .. v_7, 1]
// as D[32 feature rows][32 columns (9 used)] += A[32 x 16 tokens] * B[16 tokens x 32] with v_mfma_f32_32x32x16_bf16, three feature tiles.
// Both operands need the TOKENS along k, i.e. 8 consecutive tokens of one component in 16 bytes, and the projection writes tokens as the
// slow axis: a chunk of 1024 tokens is transposed into LDS once (kT[8][1024], vT[10][1024] bf16: row 8 = 1 for live tokens, row 9 = 0).
// A lane builds its A fragment from two 16-byte reads: phi_f = (k_a * k_b) * scale with (a, b, scale) fixed per lane and tile - the
// constant and the linear features read a row of ones for the missing factor, rows past 72 the row of zeros - so the code is branch-free.
// SPLIT = 2: phi in two bf16 terms (hi + lo, two MFMAs): the state then carries ~16 significant bits of phi instead of 8 (v is bf16
// already, exact).  Four waves take 256 tokens of a chunk each; the partial tiles meet through LDS in a fixed order.
// reference: TaylorSeriesLinearAttn (third party, restated in oracle/thirdparty.py) via LinearAttention.forward magvit2_pytorch.py:421-430.
template <int SPLIT>
__global__ __launch_bounds__(256) void taylor_kv_mfma_kernel(const void* __restrict__ kv, float* __restrict__ state, int n, int heads) {
    constexpr int CH = 1024, ROWB = CH * 2;                      // tokens per chunk; bytes per transposed row
    constexpr int ONES = 16 * ROWB, ZERO = 17 * ROWB;            // rows: 0..7 kT, 8..15 vT, 16 ones (live tokens), 17 zeros
    __shared__ __attribute__((aligned(16))) char smem[18 * ROWB];                  // 36 KB; the cross-wave sum reuses it (3 x 12 KB)
    const int f = blockIdx.y, h = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = 2 * heads * 8;
    // per lane and tile: the two rows whose product is its feature, and the scale
    int rowA[3], rowB[3]; float sc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int ft = 32 * m + l31;
        if (ft == 0) { rowA[m] = ONES; rowB[m] = ONES; sc[m] = 1.f; }
        else if (ft <= 8) { rowA[m] = (ft - 1) * ROWB; rowB[m] = ONES; sc[m] = 1.f; }
        else if (ft < 73) { rowA[m] = ((ft - 9) >> 3) * ROWB; rowB[m] = ((ft - 9) & 7) * ROWB; sc[m] = 0.70710678118654752440f; }
        else { rowA[m] = ZERO; rowB[m] = ZERO; sc[m] = 0.f; }
    }
    const int rowV = (l31 < 8 ? 8 + l31 : (l31 == 8 ? 16 : 17)) * ROWB;            // B operand: column e = l31 (v_e, the ones, zeros)
    f32x16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    for (int c0 = 0; c0 < n; c0 += CH) {
        if (c0) __syncthreads();                                 // everybody has read the previous chunk
        // ---- transpose the chunk: thread -> token PAIRS tid, tid + 256 (tokens 2 p, 2 p + 1): one 4-byte LDS write per row and pair
#pragma unroll
        for (int q = 0; q < CH / 512; ++q) {
            const int pp = tid + 256 * q, t0 = c0 + 2 * pp;
            i32x4 kr[2], vr[2];
            unsigned one[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const bool live = t0 + x < n;
                const int64_t row = ((int64_t)f * n + (live ? t0 + x : n - 1)) * ld;
                kr[x] = *(const i32x4*)((const char*)kv + (row + h * 8) * 2);
                vr[x] = *(const i32x4*)((const char*)kv + (row + (heads + h) * 8) * 2);
                if (!live) { kr[x] = i32x4{0, 0, 0, 0}; vr[x] = i32x4{0, 0, 0, 0}; }
                one[x] = live ? 0x3f80u : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ka = (unsigned)kr[0][j], kb = (unsigned)kr[1][j], va = (unsigned)vr[0][j], vb = (unsigned)vr[1][j];
                *(unsigned*)(smem + (2 * j) * ROWB + pp * 4) = (ka & 0xffffu) | (kb << 16);
                *(unsigned*)(smem + (2 * j + 1) * ROWB + pp * 4) = (ka >> 16) | (kb & 0xffff0000u);
                *(unsigned*)(smem + (8 + 2 * j) * ROWB + pp * 4) = (va & 0xffffu) | (vb << 16);
                *(unsigned*)(smem + (9 + 2 * j) * ROWB + pp * 4) = (va >> 16) | (vb & 0xffff0000u);
            }
            *(unsigned*)(smem + ONES + pp * 4) = one[0] | (one[1] << 16);
            *(unsigned*)(smem + ZERO + pp * 4) = 0u;
        }
        __syncthreads();
        // ---- wave w: tokens [256 w, 256 w + 256) of the chunk, 16 tokens per MFMA step, two steps per trip: the fourteen 16-byte fragment
        // reads of a trip are issued before its math (one LDS latency per trip instead of one per tile and step)
        for (int s = 0; s < 16; s += 2) {
            const int tb0 = 256 * wave + 16 * s;
            if (c0 + tb0 >= n) break;                            // wave-uniform: nothing live from here on (a half-dead trip multiplies zeros)
            i32x4 bfr[2], ar[2][3], br[2][3];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int tbo = (tb0 + 16 * x + 8 * hi) * 2;     // byte offset of this lane's 8 tokens inside a row
                bfr[x] = *(const i32x4*)(smem + rowV + tbo);
#pragma unroll
                for (int m = 0; m < 3; ++m) { ar[x][m] = *(const i32x4*)(smem + rowA[m] + tbo); br[x][m] = *(const i32x4*)(smem + rowB[m] + tbo); }
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    float ka[8], kb[8], p[8];
                    Chunk<BF16>::unpack(ar[x][m], ka);
                    Chunk<BF16>::unpack(br[x][m], kb);
#pragma unroll
                    for (int u = 0; u < 8; ++u) p[u] = (ka[u] * kb[u]) * sc[m];
                    const i32x4 ahi = Chunk<BF16>::pack(p);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ahi), __builtin_bit_cast(bf16x8, bfr[x]), acc[m], 0, 0, 0);
                    if constexpr (SPLIT == 2) {
                        float ph[8];
                        Chunk<BF16>::unpack(ahi, ph);
#pragma unroll
                        for (int u = 0; u < 8; ++u) p[u] -= ph[u];
                        const i32x4 alo = Chunk<BF16>::pack(p);
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, alo), __builtin_bit_cast(bf16x8, bfr[x]), acc[m], 0, 0, 0);
                    }
                }
        }
    }
    // ---- cross-wave sum (waves 1..3 -> LDS, wave 0 adds in wave order) and the state rows
    __syncthreads();
    float* red = (float*)smem;                                   // [wave - 1][tile][r][lane]
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave - 1) * 3 + m) * 16 + r) * 64 + lane] = acc[m][r];
    }
    __syncthreads();
    if (wave == 0) {
        float* dst = state + ((int64_t)f * heads + h) * TAY_S;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][r];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[((w * 3 + m) * 16 + r) * 64 + lane];
                const int ft = 32 * m + 8 * (r >> 2) + 4 * hi + (r & 3);      // D row of this register; column = l31
                if (ft < TAY_F && l31 < 9) dst[ft * 9 + l31] = v;
            }
    }
}

extern "C" int mv2x_taylor_kv_mfma(const void* kv, float* state, int frames, int n, int heads, int split, void* stream) {
    if (!kv || !state || frames <= 0 || n <= 0 || heads <= 0 || (split != 1 && split != 2)) return MV2_EINVAL;
    dim3 grid((unsigned)heads, (unsigned)frames);
    hipStream_t st = (hipStream_t)stream;
    if (split == 1) hipLaunchKernelGGL((taylor_kv_mfma_kernel<1>), grid, dim3(256), 0, st, kv, state, n, heads);
    else hipLaunchKernelGGL((taylor_kv_mfma_kernel<2>), grid, dim3(256), 0, st, kv, state, n, heads);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// EXPERIMENTS ONLY: taylor_kv_frame_kernel with the frame's tokens cut into nchunk ranges (grid z): nchunk partial states per (frame,
// head) in taylor_kv_kernel's layout.  320 workgroups on 256 CUs take two rounds (a lone workgroup needs 13.5 us, the launch 22.3);
// 640 / 1280 half- or quarter-length workgroups are all resident at once.  The partial states would be summed where mv2_taylor_apply
// stages the state (a few KB per block), not by a reduce launch.
template <typename E>
__global__ __launch_bounds__(256) void taylor_kv_frame_split_kernel(const void* __restrict__ kv, float* __restrict__ state, int n, int heads, int nchunk) {
    __shared__ float red[4][8][92];
    const int f = blockIdx.y, h = blockIdx.x, z = blockIdx.z;
    const int per = (n + nchunk - 1) / nchunk, tbeg = z * per, tend = min(n, tbeg + per);      // this workgroup's token range
    const int tid = threadIdx.x, i = tid & 7, tg = tid >> 3, wave = tid >> 6;
    const int ld = 2 * heads * 8;
    constexpr int NC = 8 / E::kVec, U = 4;
    float a2[8][9], a1[9], a0[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) { a1[e] = 0.f; a0[e] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) a2[jj][e] = 0.f; }
    const float r2 = 0.70710678118654752440f;
    for (int t0 = tbeg + tg; t0 < tend; t0 += 32 * U) {
        i32x4 kr[U][NC], vr[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(t0 + 32 * u, tend - 1);                  // clamped: the tail trip re-reads a valid row and is masked below
            const int64_t row = ((int64_t)f * n + t) * ld;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                kr[u][c] = *(const i32x4*)((const char*)kv + (row + h * 8) * E::kBytes + c * 16);
                vr[u][c] = *(const i32x4*)((const char*)kv + (row + (heads + h) * 8) * E::kBytes + c * 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float k[8], v[9];
#pragma unroll
            for (int c = 0; c < NC; ++c) { Chunk<E>::unpack(kr[u][c], k + c * E::kVec); Chunk<E>::unpack(vr[u][c], v + c * E::kVec); }
            const bool live = t0 + 32 * u < tend;
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
    float* dst = state + (((int64_t)f * nchunk + z) * heads + h) * TAY_S;      // the partial-state layout of taylor_kv_kernel
    for (int x = tid; x < 8 * 90; x += 256) {
        const int ii = x / 90, r = x - ii * 90;
        const float s = ((red[0][ii][r] + red[1][ii][r]) + red[2][ii][r]) + red[3][ii][r];
        if (r < 9) { if (ii == 0) dst[r] = s; }                                   // row 0: sum of [v, 1] (every i holds a copy)
        else if (r < 18) dst[(1 + ii) * 9 + (r - 9)] = s;                          // row 1 + i
        else dst[(9 + 8 * ii + (r - 18) / 9) * 9 + (r - 18) % 9] = s;             // row 9 + 8 i + j
    }
}


extern "C" int mv2x_taylor_kv_split(const void* kv, float* part, int frames, int n, int heads, int nchunk, void* stream) {
    if (!kv || !part || frames <= 0 || n <= 0 || heads <= 0 || nchunk <= 0 || nchunk > 16) return MV2_EINVAL;
    dim3 grid((unsigned)heads, (unsigned)frames, (unsigned)nchunk);
    hipLaunchKernelGGL((taylor_kv_frame_split_kernel<BF16>), grid, dim3(256), 0, (hipStream_t)stream, kv, part, n, heads, nchunk);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
