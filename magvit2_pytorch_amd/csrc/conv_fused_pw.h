// Fused second stage of the ResidualUnit convolution kernels (shared by conv3.hip and the experimental conv_halo.hip).
#pragma once
#include "conv_common.h"

// ---- fused epilogue of mv2_conv_pw (bf16): out = act2(W2 . act1(conv3x3 + b1) + b2) without leaving the CU.
// Valid when the block owns every output channel of its positions (one co tile, no split-K).  The finished
// first-stage tile is dropped into LDS as [position][channel] bf16 (pitch chosen so that the MFMA B-operand
// reads of 32 consecutive rows are bank-conflict free), the C x C pointwise weights are staged next to it
// in the usual swizzled [row][8 x 16 B] image, a second MFMA GEMM runs out of LDS, and the result replaces
// the tile and is streamed out with whole-row 16-byte stores.  Replaces: nn.ELU + nn.Conv3d(dim, dim, 1) +
// nn.ELU of ResidualUnit (reference magvit2_pytorch.py:938-940) and one full write + read of the tensor.
template <int BCO, int BPOS, int WCO, int WPOS, int NT, typename RowFn>
__device__ __forceinline__ void fused_pointwise_epilogue(const ConvParams& p, char* lds, f32x16 (&acc)[WCO][WPOS], int wco, int wpos,
                                                         int tid, int64_t tile_index, RowFn row_of) {
    typedef BF16 E;
    const mv2_conv_desc& d = p.d;
    constexpr int PITCH = StageCfg<E, BCO, BPOS>::kPitch;
    constexpr int TILE = ((StageCfg<E, BCO, BPOS>::kBytes + 255) / 256) * 256;
    constexpr int NS2 = BCO * 2 / 128;                 // K steps of the pointwise conv (64 bf16 each)
    char* const t1 = lds;
    char* const w2 = lds + TILE;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    // ---- stage 1 values -> tile
#pragma unroll
    for (int b = 0; b < WPOS; ++b) {
        const int lp = wpos * 32 * WPOS + b * 32 + l31;
        char* rowp = t1 + lp * PITCH;
#pragma unroll
        for (int a = 0; a < WCO; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wco * 32 * WCO + a * 32 + 8 * g + 4 * hi;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[a][b][4 * g + j];
                    if (p.bias) x += p.bias[col + j];
                    if (d.act == MV2_ACT_ELU) x = Act<E>::elu(x);
                    else if (d.act == MV2_ACT_SILU) x = Act<E>::silu(x);
                    v[j] = x;
                }
                lds_put4<E>(rowp + col * 2, v);
            }
    }
    // ---- pointwise weights -> LDS (swizzled rows)
    for (int q = tid; q < NS2 * BCO * 8; q += NT) {
        const int st = q / (BCO * 8), rem = q - st * BCO * 8;
        const int row = rem >> 3, c = rem & 7;
        const i32x4 v = *(const i32x4*)((const char*)p.wp2 + ((size_t)st * d.co_pad + row) * 128 + c * 16);
        *(i32x4*)(w2 + (st * BCO + row) * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
    }
    __syncthreads();
    // ---- second GEMM out of LDS
#pragma unroll
    for (int a = 0; a < WCO; ++a)
#pragma unroll
        for (int b = 0; b < WPOS; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const uint32_t swz = (uint32_t)((l31 >> 1) & 7);
#pragma unroll
    for (int st = 0; st < NS2; ++st)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t ck = (uint32_t)(ks * 2 + hi);
            i32x4 fw[WCO], fp[WPOS];
#pragma unroll
            for (int a = 0; a < WCO; ++a) fw[a] = *(const i32x4*)(w2 + (st * BCO + wco * 32 * WCO + a * 32 + l31) * 128 + ((ck ^ swz) << 4));
#pragma unroll
            for (int b = 0; b < WPOS; ++b) fp[b] = *(const i32x4*)(t1 + (wpos * 32 * WPOS + b * 32 + l31) * PITCH + st * 128 + ck * 16);
#pragma unroll
            for (int a = 0; a < WCO; ++a)
#pragma unroll
                for (int b = 0; b < WPOS; ++b) Mma<E>::run(fw[a], fp[b], acc[a][b]);
        }
    __syncthreads();      // every wave is done reading the stage-1 tile
    // ---- stage 2 values -> tile, then coalesced rows
#pragma unroll
    for (int b = 0; b < WPOS; ++b) {
        const int lp = wpos * 32 * WPOS + b * 32 + l31;
        char* rowp = t1 + lp * PITCH;
#pragma unroll
        for (int a = 0; a < WCO; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wco * 32 * WCO + a * 32 + 8 * g + 4 * hi;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[a][b][4 * g + j];
                    if (p.bias2) x += p.bias2[col + j];
                    if (p.act2 == MV2_ACT_ELU) x = Act<E>::elu(x);
                    else if (p.act2 == MV2_ACT_SILU) x = Act<E>::silu(x);
                    v[j] = x;
                }
                lds_put4<E>(rowp + col * 2, v);
            }
    }
    __syncthreads();
    constexpr int cpr = BCO * 2 / 16;
    for (int q = tid; q < BPOS * cpr; q += NT) {
        const int lp = q / cpr, ch = q - lp * cpr;
        const int n = ch * 8;
        if (n >= d.co) continue;
        const int64_t m = row_of(lp);
        if (m >= p.M) continue;
        *(i32x4*)((char*)p.out + (m * d.out_c + n) * 2) = *(const i32x4*)(t1 + lp * PITCH + ch * 16);
    }
    // ---- optional: SqueezeExcite pooling partials of this tile (reference SqueezeExcite.forward M:226-231; same
    // record as mv2_se_pool: {max logit, sum exp, exp-weighted channel sums}), computed from the bf16 tile that is
    // still in LDS -- saves the separate pooling pass over the tensor.  The pointwise weights are dead: scratch.
    if (p.se_part) {
        float* lg = (float*)w2;            // [BPOS] logits, then exp(logit - max)
        float* red = lg + BPOS;            // [32]
        float* slab = red + 32;            // [NT]
        constexpr int TPR = NT / BPOS;     // threads per position row (1 or 2)
        constexpr int CH = BCO / TPR;      // channels per thread
        static_assert(NT % BPOS == 0 && (TPR == 1 || TPR == 2) && NT % BCO == 0 && CH % 8 == 0, "pooling thread map");
        const int lane = tid & 63, wave = tid >> 6;
        const int lp = tid / TPR, half = tid % TPR;
        float dot = 0.f;
#pragma unroll
        for (int c8 = 0; c8 < CH / 8; ++c8) {
            float xv[8];
            Chunk<E>::unpack(*(const i32x4*)(t1 + lp * PITCH + (half * CH + c8 * 8) * 2), xv);
#pragma unroll
            for (int k = 0; k < 8; ++k) dot += xv[k] * p.se_wk[half * CH + c8 * 8 + k];
        }
        if (TPR == 2) dot += __shfl_xor(dot, 1, 64);
        const bool valid = row_of(lp) < p.M;
        const float logit = valid ? dot + p.se_bk : -INFINITY;
        const float mx = wave_max(logit);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        float M = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) M = fmaxf(M, red[w]);
        const float e = (valid && half == 0) ? expf(logit - M) : 0.f;     // every tile holds at least one valid row
        if (half == 0) lg[lp] = e;
        const float sw = wave_sum(e);
        if (lane == 0) red[16 + wave] = sw;
        __syncthreads();
        constexpr int SL = NT / BCO, RPS = BPOS / SL;       // position slices x rows per slice
        const int c = tid % BCO, sl = tid / BCO;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
        for (int r = sl * RPS; r < (sl + 1) * RPS; r += 2) {
            a0 += lg[r] * bf2f(*(const unsigned short*)(t1 + r * PITCH + c * 2));
            a1 += lg[r + 1] * bf2f(*(const unsigned short*)(t1 + (r + 1) * PITCH + c * 2));
        }
        slab[sl * BCO + c] = a0 + a1;
        __syncthreads();
        float* dst = p.se_part + tile_index * (BCO + 2);
        if (tid < BCO) {
            float o = 0.f;
#pragma unroll
            for (int q = 0; q < SL; ++q) o += slab[q * BCO + tid];
            dst[2 + tid] = o;
        }
        if (tid == 0) {
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) S += red[16 + w];
            dst[0] = M; dst[1] = S;
        }
    }
}
