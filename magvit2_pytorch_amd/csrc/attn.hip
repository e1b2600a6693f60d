// Axial softmax attention with memory key/values, and Taylor-series linear attention.
// bf16 space attention runs on MFMA (attn_mfma_kernel: QK^T and PV as v_mfma_f32_32x32x16_bf16); the fp32
// (id-parity) mode and the tiny causal time attention use fp32 VALU kernels.
#include <cstdlib>
#include "common.h"

struct AttnParams {
    const void* qkv; const float* mem_kv; void* out;
    int nseq, n, heads, n_mem, causal, seq_inner;
    int64_t seq_outer_stride, tok_stride;
};

__device__ __forceinline__ int64_t attn_row(const AttnParams& p, int seq, int i) {
    return (int64_t)(seq / p.seq_inner) * p.seq_outer_stride + (seq % p.seq_inner) + (int64_t)i * p.tok_stride;
}

// dim_head^-0.5 as the correctly rounded fp32 constant (what torch computes on the host: q * dim_head ** -0.5)
template <int D> __host__ __device__ constexpr float attn_scale() {
    static_assert(D == 8 || D == 16 || D == 32 || D == 64, "dim_head of the VALU attention kernels");
    return D == 8 ? 0.35355339059327376220f : D == 16 ? 0.25f : D == 32 ? 0.17677669529663687f : 0.125f;
}

template <typename E, int D = 32>
__device__ __forceinline__ void load32(const void* base, int64_t elem, float* f) {      // D elements (the name is round 1's: D was 32)
    const char* ptr = (const char*)base + elem * E::kBytes;
#pragma unroll
    for (int c = 0; c < D / E::kVec; ++c) {
        const i32x4 v = *(const i32x4*)(ptr + c * 16);
        Chunk<E>::unpack(v, f + c * E::kVec);
    }
}
template <typename E, int D = 32>
__device__ __forceinline__ void store32(void* base, int64_t elem, const float* f) {
    char* ptr = (char*)base + elem * E::kBytes;
#pragma unroll
    for (int c = 0; c < D / E::kVec; ++c) *(i32x4*)(ptr + c * 16) = Chunk<E>::pack(f + c * E::kVec);
}

// ---- small sequences (time attention): one thread per (seq, head, query), K/V straight from L2.  D = dim_head (8 / 16 / 32 / 64: the
// reference takes any attn_dim_head, M:1076; 32 is its default and the only width the MFMA kernel below is built for)
template <typename E, int D>
__global__ __launch_bounds__(256) void attn_small_kernel(const AttnParams p) {
    const int64_t total = (int64_t)p.nseq * p.heads * p.n;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % p.n);
    const int h = (int)((idx / p.n) % p.heads);
    const int seq = (int)(idx / ((int64_t)p.n * p.heads));
    const int ld = 3 * p.heads * D;
    float q[D], o[D];
    load32<E, D>(p.qkv, attn_row(p, seq, i) * ld + h * D, q);
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int klen = p.n_mem + (p.causal ? i + 1 : p.n);
    constexpr float scale = attn_scale<D>();
    for (int j = 0; j < klen; ++j) {
        float k[D], v[D];
        if (j < p.n_mem) {
            const float* mk = p.mem_kv + ((int64_t)h * p.n_mem + j) * D;
            const float* mv = p.mem_kv + (((int64_t)p.heads + h) * p.n_mem + j) * D;
#pragma unroll
            for (int d = 0; d < D; ++d) { k[d] = mk[d]; v[d] = mv[d]; }
        } else {
            const int64_t r = attn_row(p, seq, j - p.n_mem) * ld;
            load32<E, D>(p.qkv, r + (p.heads + h) * D, k);
            load32<E, D>(p.qkv, r + (2 * p.heads + h) * D, v);
        }
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) s += q[d] * k[d];
        s *= scale;
        const float mn = fmaxf(m, s);
        const float a = expf(m - mn), e = expf(s - mn);
        l = l * a + e;
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] = o[d] * a + e * v[d];
        m = mn;
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] *= inv;
    store32<E, D>(p.out, attn_row(p, seq, i) * (p.heads * D) + h * D, o);
}

// ---- long sequences (space attention): block = 256 queries of one (seq, head); K/V tiles of 64
// keys staged as fp32 in LDS and broadcast-read by every lane.
template <typename E, int D>
__global__ __launch_bounds__(256) void attn_tile_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[64][D];
    __shared__ __attribute__((aligned(16))) float Vs[64][D];
    constexpr int DP = D >= 32 ? D / 4 : 8;          // dims a staging thread owns (whole 16-byte chunks in either dtype) ...
    constexpr int NP = D / DP;                       // ... and staging threads per key (4 from D = 32 on)
    const int seq = blockIdx.z, h = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool active = i < p.n;
    const int ld = 3 * p.heads * D;
    float q[D], o[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { q[d] = 0.f; o[d] = 0.f; }
    if (active) load32<E, D>(p.qkv, attn_row(p, seq, i) * ld + h * D, q);
    float m = -INFINITY, l = 0.f;
    constexpr float scale = attn_scale<D>();
    const int klen = p.n_mem + p.n;
    const int key_l = threadIdx.x >> 2, part = threadIdx.x & 3;
    for (int j0 = 0; j0 < klen; j0 += 64) {
        __syncthreads();
        if (part < NP) {   // stage: thread -> (key, DP dims)
            const int j = j0 + key_l;
            float kf[DP], vf[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) { kf[d] = 0.f; vf[d] = 0.f; }
            if (j < klen) {
                if (j < p.n_mem) {
                    const float* mk = p.mem_kv + ((int64_t)h * p.n_mem + j) * D + part * DP;
                    const float* mv = p.mem_kv + (((int64_t)p.heads + h) * p.n_mem + j) * D + part * DP;
#pragma unroll
                    for (int d = 0; d < DP; ++d) { kf[d] = mk[d]; vf[d] = mv[d]; }
                } else {
                    const int64_t r = attn_row(p, seq, j - p.n_mem) * ld;
                    const char* kp = (const char*)p.qkv + (r + (p.heads + h) * D + part * DP) * E::kBytes;
                    const char* vp = (const char*)p.qkv + (r + (2 * p.heads + h) * D + part * DP) * E::kBytes;
#pragma unroll
                    for (int c = 0; c < DP / E::kVec; ++c) {
                        Chunk<E>::unpack(*(const i32x4*)(kp + c * 16), kf + c * E::kVec);
                        Chunk<E>::unpack(*(const i32x4*)(vp + c * 16), vf + c * E::kVec);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < DP; ++d) { Ks[key_l][part * DP + d] = kf[d]; Vs[key_l][part * DP + d] = vf[d]; }
        }
        __syncthreads();
        const int jn = min(64, klen - j0);
        for (int jj = 0; jj < jn; ++jj) {
            const int j = j0 + jj;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s += q[d] * Ks[jj][d];
            s *= scale;
            const bool vis = !p.causal || j < p.n_mem || (j - p.n_mem) <= i;
            if (vis) {
                const float mn = fmaxf(m, s);
                const float a = expf(m - mn), e = expf(s - mn);
                l = l * a + e;
#pragma unroll
                for (int d = 0; d < D; ++d) o[d] = o[d] * a + e * Vs[jj][d];
                m = mn;
            }
        }
    }
    if (active) {
        const float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] *= inv;
        store32<E, D>(p.out, attn_row(p, seq, i) * (p.heads * D) + h * D, o);
    }
}

// ---- MFMA flash-style attention (bf16, non-causal, n >= 64: the space-attention case).
// Block = 4 waves x 32 queries of one (sequence, head); keys (memory kv first, then tokens) are staged in
// LDS chunks of up to AT_KC keys: K row-major [key][32 d] (64-byte rows, chunk ^ ((key>>2)&3) swizzle) and
// V TRANSPOSED [32 dv][key] (pitch chosen bank-conflict free).  Per 32-key tile a wave computes
//     S^T[key][query] = K . Q^T        (2 x v_mfma_f32_32x32x16_bf16; "swapped QK^T": lane = query)
// so the online-softmax statistics of a query live in one lane pair, and
//     O^T[dv][query] += V^T . P^T      (2 MFMAs)
// where the P^T operand is exactly the lane's own 8 exponentiated scores per k-step (the k order of the
// contraction is permuted identically on the V^T side), so no cross-lane movement or LDS round trip of P.
#define AT_KC 320
#define AT_VPITCH (AT_KC * 2 + 8)      // bytes per V^T row: 162 dwords -> 32 lanes x 2 dwords hit 64 distinct banks

__device__ __forceinline__ unsigned pack_bf2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

__global__ __launch_bounds__(256) void attn_mfma_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char Ks[AT_KC * 64];
    __shared__ __attribute__((aligned(16))) char Vt[32 * AT_VPITCH];
    const int seq = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int q_idx = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_ok = q_idx < p.n;
    const int ld = 3 * p.heads * 32;
    const int KL = p.n_mem + p.n;
    // Q fragments (B operand of S^T): this lane's query row, d slots [step*16 + hi*8, +8)
    i32x4 qf[2];
    {
        const char* qp = (const char*)p.qkv + (attn_row(p, seq, q_ok ? q_idx : 0) * ld + h * 32) * 2;
        qf[0] = *(const i32x4*)(qp + hi * 16);
        qf[1] = *(const i32x4*)(qp + 32 + hi * 16);
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float scale = 0.17677669529663687f;

    for (int k0 = 0; k0 < KL; k0 += AT_KC) {
        const int nk = min(AT_KC, KL - k0);
        __syncthreads();
        // ---- stage: item = (key, which); K rows swizzled, V transposed
        // (keys of the last partial 32-tile are zero-filled: their scores are masked to -inf, but a stale
        //  NaN bit pattern in V^T would still poison the PV MFMA through 0 * NaN)
        const int nk32 = ((nk + 31) / 32) * 32;
        for (int it = tid; it < nk32 * 2; it += 256) {
            const int key = it >> 1, which = it & 1;
            const int gk = k0 + key;
            i32x4 v[4];
            if (key >= nk) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) { v[c4][0] = 0; v[c4][1] = 0; v[c4][2] = 0; v[c4][3] = 0; }
            } else if (gk < p.n_mem) {
                const float* src = p.mem_kv + (((int64_t)which * p.heads + h) * p.n_mem + gk) * 32;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    v[c4][0] = (int)pack_bf2(src[c4 * 8 + 0], src[c4 * 8 + 1]); v[c4][1] = (int)pack_bf2(src[c4 * 8 + 2], src[c4 * 8 + 3]);
                    v[c4][2] = (int)pack_bf2(src[c4 * 8 + 4], src[c4 * 8 + 5]); v[c4][3] = (int)pack_bf2(src[c4 * 8 + 6], src[c4 * 8 + 7]);
                }
            } else {
                const char* src = (const char*)p.qkv + (attn_row(p, seq, gk - p.n_mem) * ld + ((1 + which) * p.heads + h) * 32) * 2;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) v[c4] = *(const i32x4*)(src + c4 * 16);
            }
            if (which == 0) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) *(i32x4*)(Ks + key * 64 + ((c4 ^ ((key >> 2) & 3)) << 4)) = v[c4];
            } else {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned u = (unsigned)v[c4][w];
                        const int dv = c4 * 8 + w * 2;
                        *(unsigned short*)(Vt + dv * AT_VPITCH + key * 2) = (unsigned short)(u & 0xffffu);
                        *(unsigned short*)(Vt + (dv + 1) * AT_VPITCH + key * 2) = (unsigned short)(u >> 16);
                    }
            }
        }
        __syncthreads();
        const int ntile = (nk + 31) / 32;
        for (int kb = 0; kb < ntile; ++kb) {
            // S^T tile: rows = keys kb*32 + l31 (A operand from LDS), cols = this wave's 32 queries
            const int krow = kb * 32 + l31;
            const int ksw = (krow >> 2) & 3;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const i32x4 kf = *(const i32x4*)(Ks + krow * 64 + (((st * 2 + hi) ^ ksw) << 4));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[st]), s, 0, 0, 0);
            }
            // reg r <-> key_local = 8*(r>>2) + 4*hi + (r&3); lane <-> query
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = kb * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
                s[r] = kl < nk ? s[r] * scale : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);
            const float alpha = __expf(m - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mn); ps += s[r]; }
            l = l * alpha + ps;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
            // O^T += V^T . P^T : k-step j uses this lane's scores r = 8j..8j+7, i.e. keys 16j + {0..3, 8..11} + 4*hi
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                i32x4 pf;
                pf[0] = (int)pack_bf2(s[8 * j + 0], s[8 * j + 1]); pf[1] = (int)pack_bf2(s[8 * j + 2], s[8 * j + 3]);
                pf[2] = (int)pack_bf2(s[8 * j + 4], s[8 * j + 5]); pf[3] = (int)pack_bf2(s[8 * j + 6], s[8 * j + 7]);
                const char* vrow = Vt + l31 * AT_VPITCH + (kb * 32 + 16 * j + 4 * hi) * 2;
                const i32x2 va = *(const i32x2*)(vrow), vb = *(const i32x2*)(vrow + 16);
                i32x4 vf; vf[0] = va[0]; vf[1] = va[1]; vf[2] = vb[0]; vf[3] = vb[1];
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), o, 0, 0, 0);
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (q_ok) {
        const float inv = 1.f / l;
        char* op = (char*)p.out + (attn_row(p, seq, q_idx) * (p.heads * 32) + h * 32) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {      // reg r <-> dv = 8*(r>>2) + 4*hi + (r&3)
            i32x2 w;
            w[0] = (int)pack_bf2(o[4 * g + 0] * inv, o[4 * g + 1] * inv);
            w[1] = (int)pack_bf2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            *(i32x2*)(op + (8 * g + 4 * hi) * 2) = w;
        }
    }
}

extern "C" int mv2_attention(int dtype, const void* qkv, const float* mem_kv, void* out, int nseq, int n, int heads,
                             int dim_head, int n_mem, int causal, int seq_inner, int64_t seq_outer_stride,
                             int64_t tok_stride, void* stream) {
    if (!qkv || !mem_kv || !out || nseq <= 0 || n <= 0 || heads <= 0 || n_mem < 0 || seq_inner <= 0) return MV2_EINVAL;
    if ((dim_head != 8 && dim_head != 16 && dim_head != 32 && dim_head != 64) || (dtype != MV2_BF16 && dtype != MV2_F32)) return MV2_EINVAL;
    AttnParams p{qkv, mem_kv, out, nseq, n, heads, n_mem, causal, seq_inner, seq_outer_stride, tok_stride};
    hipStream_t st = (hipStream_t)stream;
#ifdef MV2_EXPERIMENTS
    static const bool no_mfma = getenv("MV2_ATTN_VALU") != nullptr;
#else
    constexpr bool no_mfma = false;
#endif
    if (dtype == MV2_BF16 && !causal && n >= 64 && dim_head == 32 && !no_mfma) {
        if (nseq > 65535) return MV2_EINVAL;
        dim3 grid((unsigned)((n + 127) / 128), (unsigned)heads, (unsigned)nseq);
        hipLaunchKernelGGL(attn_mfma_kernel, grid, dim3(256), 0, st, p);
        MV2_LAUNCH_CHECK();
        return MV2_OK;
    }
    // every other case (fp32, the causal time attention, short sequences, dim_head 8 / 16 / 64) on the VALU kernels
    if (n >= 64 && nseq > 65535) return MV2_EINVAL;
#define MV2_ATTN_VALU(Dv)                                                                                                   \
    do {                                                                                                                    \
        if (n < 64) {                                                                                                       \
            const int64_t total = (int64_t)nseq * heads * n;                                                                \
            const unsigned blocks = (unsigned)((total + 255) / 256);                                                        \
            if (dtype == MV2_BF16) hipLaunchKernelGGL((attn_small_kernel<BF16, Dv>), dim3(blocks), dim3(256), 0, st, p);    \
            else hipLaunchKernelGGL((attn_small_kernel<F32, Dv>), dim3(blocks), dim3(256), 0, st, p);                       \
        } else {                                                                                                            \
            dim3 grid((unsigned)((n + 255) / 256), (unsigned)heads, (unsigned)nseq);                                        \
            if (dtype == MV2_BF16) hipLaunchKernelGGL((attn_tile_kernel<BF16, Dv>), grid, dim3(256), 0, st, p);             \
            else hipLaunchKernelGGL((attn_tile_kernel<F32, Dv>), grid, dim3(256), 0, st, p);                                \
        }                                                                                                                   \
    } while (0)
    if (dim_head == 8) MV2_ATTN_VALU(8);
    else if (dim_head == 16) MV2_ATTN_VALU(16);
    else if (dim_head == 32) MV2_ATTN_VALU(32);
    else MV2_ATTN_VALU(64);
#undef MV2_ATTN_VALU
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// ------------------------------------------------------------------------------ Taylor attention
// phi(x) (d = 8): index 0 -> 1 ; 1 + i -> x_i ; 9 + 8 i + j -> x_i x_j sqrt(0.5).   73 features.
// KV state per (frame, head): [73][9] fp32, column 8 = sum of phi(k) (the normaliser).
#define TAY_F 73
#define TAY_S (73 * 9)
// Where a (frame, head, token) entry of q / k / v sits, in elements.  Row layout (mv2_taylor_kv / mv2_taylor_apply): kv rows [token][k (heads x 8)
// | v (heads x 8)], q rows [token][heads x 8].  Fused head-major layout (mv2_taylor_kv_fused / mv2_taylor_apply_fused): ONE tensor
// [frame][head][token][q 8 | k 8 | v 8] written by the projection GEMM's depth-to-time store (MV2_STORE_SHUFFLE with rt = heads): a
// (frame, head) workgroup of the state kernel then streams 48 contiguous bytes per token instead of two 16-byte pieces out of a 512-byte row
// (80 MB fetched for a 10.5 MB kv tensor in the PMC pass of round 5).
struct TayLay { int64_t frame; int64_t head; int tok; int a_off; int b_off; };
__host__ __device__ inline TayLay tay_rows_kv(int n, int heads) { return TayLay{(int64_t)n * 2 * heads * 8, 8, 2 * heads * 8, 0, heads * 8}; }
__host__ __device__ inline TayLay tay_rows_q(int n, int heads) { return TayLay{(int64_t)n * heads * 8, 8, heads * 8, 0, 0}; }
__host__ __device__ inline TayLay tay_fused(int n, int heads) { return TayLay{(int64_t)heads * n * 24, (int64_t)n * 24, 24, 8, 16}; }

template <typename E>
__global__ __launch_bounds__(256) void taylor_kv_kernel(const void* __restrict__ kv, float* __restrict__ part, int n, int heads,
                                                        int nchunk, const TayLay L) {
    const int f = blockIdx.y, j = blockIdx.x;
    const int h = threadIdx.x >> 3, i = threadIdx.x & 7;
    if (h >= heads) return;
    const int per = (n + nchunk - 1) / nchunk;
    const int t0 = j * per, t1 = min(n, t0 + per);
    float a2[8][9], a1[9], a0[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) { a1[e] = 0.f; a0[e] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) a2[jj][e] = 0.f; }
    const float r2 = 0.70710678118654752440f;
    for (int t = t0; t < t1; ++t) {
        const int64_t row = (int64_t)f * L.frame + (int64_t)h * L.head + (int64_t)t * L.tok;
        float k[8], v[9];
#pragma unroll
        for (int c = 0; c < 8 / E::kVec; ++c) {
            Chunk<E>::unpack(*(const i32x4*)((const char*)kv + (row + L.a_off) * E::kBytes + c * 16), k + c * E::kVec);
            Chunk<E>::unpack(*(const i32x4*)((const char*)kv + (row + L.b_off) * E::kBytes + c * 16), v + c * E::kVec);
        }
        v[8] = 1.f;
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
    float* dst = part + (((int64_t)f * nchunk + j) * heads + h) * TAY_S;
    if (i == 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) dst[e] = a0[e];
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) dst[(1 + i) * 9 + e] = a1[e];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
        for (int e = 0; e < 9; ++e) dst[(9 + 8 * i + jj) * 9 + e] = a2[jj][e];
}

// One workgroup per (frame, head) walks ALL tokens of the frame and writes the finished state: no partial states in HBM (the chunked
// form above wrote 26 partial 73 x 9 states per frame and head - 21 MB per launch at the 32 x 32 level - and a second launch read them
// back: profiles/r03_findings.md section 4).  thread = (token group tg of 32, feature row i of 8): it keeps the rows [1 + i] and
// [9 + 8 i .. 9 + 8 i + 7] of the state (81 accumulators, + row 0) over the tokens tg, tg + 32, ..., four tokens in flight per trip; the
// 32 partial states meet through three xor-shuffles inside a wave (lanes l, l ^ 8, l ^ 16, l ^ 32 hold the same i) and one LDS round
// across the four waves, summed in a fixed order (bitwise reproducible).
template <typename E>
__global__ __launch_bounds__(256) void taylor_kv_frame_kernel(const void* __restrict__ kv, float* __restrict__ state, int n, int heads,
                                                              const TayLay L) {
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
            const int64_t row = (int64_t)f * L.frame + (int64_t)h * L.head + (int64_t)t * L.tok;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                kr[u][c] = *(const i32x4*)((const char*)kv + (row + L.a_off) * E::kBytes + c * 16);
                vr[u][c] = *(const i32x4*)((const char*)kv + (row + L.b_off) * E::kBytes + c * 16);
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

__global__ __launch_bounds__(256) void taylor_reduce_kernel(const float* __restrict__ part, float* __restrict__ kvsum, int heads,
                                                            int nchunk) {
    const int fh = blockIdx.x;             // frame * heads + head
    const int f = fh / heads, h = fh % heads;
    for (int e = threadIdx.x; e < TAY_S; e += 256) {
        float a = 0.f;
        int j = 0;
        for (; j + 8 <= nchunk; j += 8) {                            // eight independent loads in flight, summed in chunk order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(((int64_t)f * nchunk + j + u) * heads + h) * TAY_S + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; j < nchunk; ++j) a += part[(((int64_t)f * nchunk + j) * heads + h) * TAY_S + e];
        kvsum[(int64_t)fh * TAY_S + e] = a;
    }
}

// block: 64 tokens of one frame; thread -> (token slot, head) with FOUR tokens per thread (slots ts, ts + 16, ...): every LDS read
// of the state (rows padded to 12 floats = three ds_read_b128) feeds 4 x 9 FMAs instead of one read per FMA.
#define TAY_LP 12
template <typename E>
__global__ __launch_bounds__(256) void taylor_apply_kernel(const void* __restrict__ q, const float* __restrict__ kvsum,
                                                           void* __restrict__ out, int n, int heads, const TayLay L) {
    extern __shared__ __attribute__((aligned(16))) float kvs[];   // heads * 73 * 12
    const int f = blockIdx.y;
    const int tot = heads * TAY_S;
    for (int e = threadIdx.x; e < tot; e += 256) {
        const int hh = e / TAY_S, r = e - hh * TAY_S;
        kvs[(hh * TAY_F + r / 9) * TAY_LP + r % 9] = kvsum[(int64_t)f * tot + e];
    }
    __syncthreads();
    const int tpb = 256 / heads;           // token slots per pass
    const int h = threadIdx.x % heads, ts = threadIdx.x / heads;
    const float* S = kvs + h * TAY_F * TAY_LP;
    const float scale = 0.35355339059327376220f;   // 8^-0.5
    const float r2 = 0.70710678118654752440f;
    const int tile0 = blockIdx.x * 64;
    constexpr int NT = 4, NC = 8 / E::kVec;
    for (int base = ts; base < 64; base += tpb * NT) {
        float qv[NT][8], acc[NT][9];
        int tok[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            tok[u] = tile0 + base + u * tpb;
            const int tc = min(tok[u], n - 1);
            const int64_t qe = (int64_t)f * L.frame + (int64_t)h * L.head + (int64_t)tc * L.tok + L.a_off;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                Chunk<E>::unpack(*(const i32x4*)((const char*)q + qe * E::kBytes + c * 16), qv[u] + c * E::kVec);
#pragma unroll
            for (int d = 0; d < 8; ++d) qv[u][d] *= scale;
        }
        // The state rows are read in BATCHES of eight (rows 1..8, then the eight rows of every i): the 24 LDS reads of a batch are in flight
        // together, so a batch pays one LDS latency.  (A compiler barrier per ROW kept the unrolled code from hoisting all 73 rows into
        // registers, but exposed 73 LDS round trips per trip of a wave that is alone on its SIMD; the barriers now sit between batches.)
        auto row9 = [&](int r, float* s9) {
            const f32x4 a = *(const f32x4*)(S + r * TAY_LP), b = *(const f32x4*)(S + r * TAY_LP + 4), c = *(const f32x4*)(S + r * TAY_LP + 8);
            s9[0] = a[0]; s9[1] = a[1]; s9[2] = a[2]; s9[3] = a[3]; s9[4] = b[0]; s9[5] = b[1]; s9[6] = b[2]; s9[7] = b[3]; s9[8] = c[0];
        };
        {
            float s9[9]; row9(0, s9);                              // phi_0 = 1
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int e = 0; e < 9; ++e) acc[u][e] = s9[e];
        }
        asm volatile("" ::: "memory");
        {
            float s8[8][9];
#pragma unroll
            for (int i = 0; i < 8; ++i) row9(1 + i, s8[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int e = 0; e < 9; ++e) acc[u][e] += qv[u][i] * s8[i][e];
        }
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {                              // a real loop: unrolled, the 256 products q_i q_j were all hoisted (spills)
            float qi[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                qi[u] = qv[u][0];
#pragma unroll
                for (int k = 1; k < 8; ++k) qi[u] = (i == k) ? qv[u][k] : qi[u];
                qi[u] *= r2;
            }
            asm volatile("" ::: "memory");
            float s8[8][9];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) row9(9 + 8 * i + jj, s8[jj]);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const float ph = qi[u] * qv[u][jj];
#pragma unroll
                    for (int e = 0; e < 9; ++e) acc[u][e] += ph * s8[jj][e];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if (tok[u] >= n || base + u * tpb >= 64) continue;
            const float inv = 1.f / fmaxf(acc[u][8], 1e-5f);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = acc[u][e] * inv;
            const int64_t row = (int64_t)f * n + tok[u];
#pragma unroll
            for (int c = 0; c < NC; ++c)
                *(i32x4*)((char*)out + (row * heads * 8 + h * 8) * E::kBytes + c * 16) = Chunk<E>::pack(o + c * E::kVec);
        }
    }
}

static int taylor_kv_launch(int dtype, const void* kv, float* part, int frames, int n, int heads, int nchunk, const TayLay& L, void* stream) {
    if (!kv || !part || frames <= 0 || n <= 0 || heads <= 0 || heads > 32 || nchunk <= 0 || nchunk > n) return MV2_EINVAL;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (nchunk == 1) {      // one workgroup per (frame, head) over all tokens: `part` receives the finished states [frames][heads][657]
        dim3 gridf((unsigned)heads, (unsigned)frames);
        if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_kv_frame_kernel<BF16>), gridf, dim3(256), 0, st, kv, part, n, heads, L);
        else hipLaunchKernelGGL((taylor_kv_frame_kernel<F32>), gridf, dim3(256), 0, st, kv, part, n, heads, L);
        MV2_LAUNCH_CHECK();
        return MV2_OK;
    }
    dim3 grid((unsigned)nchunk, (unsigned)frames);
    const unsigned threads = (unsigned)(heads * 8);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_kv_kernel<BF16>), grid, dim3(threads), 0, st, kv, part, n, heads, nchunk, L);
    else hipLaunchKernelGGL((taylor_kv_kernel<F32>), grid, dim3(threads), 0, st, kv, part, n, heads, nchunk, L);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

static int taylor_apply_launch(int dtype, const void* q, const float* part, void* out, float* kvsum_ws, int frames, int n, int heads, int nchunk,
                               const TayLay& L, void* stream) {
    if (!q || !part || !out || (nchunk > 1 && !kvsum_ws) || frames <= 0 || n <= 0 || heads <= 0 || 256 % heads != 0) return MV2_EINVAL;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    const size_t lds = (size_t)heads * TAY_F * TAY_LP * sizeof(float);
    if (lds > 64 * 1024) return MV2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const float* kvsum = kvsum_ws;
    if (nchunk == 1) kvsum = part;      // mv2_taylor_kv with nchunk == 1 wrote the finished states: nothing to reduce
    else {
        hipLaunchKernelGGL(taylor_reduce_kernel, dim3((unsigned)(frames * heads)), dim3(256), 0, st, part, kvsum_ws, heads, nchunk);
        MV2_LAUNCH_CHECK();
    }
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)frames);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_apply_kernel<BF16>), grid, dim3(256), lds, st, q, kvsum, out, n, heads, L);
    else hipLaunchKernelGGL((taylor_apply_kernel<F32>), grid, dim3(256), lds, st, q, kvsum, out, n, heads, L);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// ---- any dim_head (4 / 16; 8 has the kernels above): plain VALU kernels for the widths the reference accepts (linear_attn_dim_head is a
// free keyword, M:1079 -> 415-419) but does not default to.  phi has F = 1 + D + D^2 features, the state per (frame, head) is [F][D + 1]
// fp32 (column D = the normaliser).  Rows layout only (kv rows [token][k heads x D | v heads x D], q rows [token][heads x D]).
template <typename E> __device__ __forceinline__ float tay_ld(const void* base, int64_t elem) {
    if constexpr (E::kBytes == 2) { const unsigned short u = ((const unsigned short*)base)[elem]; return __uint_as_float((unsigned)u << 16); }
    else return ((const float*)base)[elem];
}
template <typename E> __device__ __forceinline__ void tay_st(void* base, int64_t elem, float v) {
    if constexpr (E::kBytes == 2) ((unsigned short*)base)[elem] = f2bf(v);
    else ((float*)base)[elem] = v;
}

// one workgroup per (frame, head): thread t owns the state entries t, t + 256, ... and walks all tokens (staged 64 at a time in LDS)
template <typename E, int D>
__global__ __launch_bounds__(256) void taylor_kv_any_kernel(const void* __restrict__ kv, float* __restrict__ state, int n, int heads) {
    constexpr int F = 1 + D + D * D, W = D + 1, NE = F * W, PER = (NE + 255) / 256, TT = 64;
    __shared__ float ks[TT][D];
    __shared__ float vs[TT][W];
    const int h = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    int ei[PER], ej[PER], ec[PER];
    float acc[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = min(tid + 256 * u, NE - 1), feat = e / W;
        ec[u] = e - feat * W;
        ei[u] = feat == 0 ? -1 : (feat <= D ? feat - 1 : (feat - 1 - D) / D);
        ej[u] = feat <= D ? -1 : (feat - 1 - D) % D;
        acc[u] = 0.f;
    }
    const float r2 = 0.70710678118654752440f;
    const int64_t ld = (int64_t)2 * heads * D;
    for (int t0 = 0; t0 < n; t0 += TT) {
        __syncthreads();
        for (int idx = tid; idx < TT * D; idx += 256) {
            const int tt = idx / D, d = idx - tt * D, t = t0 + tt;
            const int64_t row = ((int64_t)f * n + min(t, n - 1)) * ld;
            const bool live = t < n;
            ks[tt][d] = live ? tay_ld<E>(kv, row + h * D + d) : 0.f;
            vs[tt][d] = live ? tay_ld<E>(kv, row + (heads + h) * D + d) : 0.f;
            if (d == 0) vs[tt][D] = live ? 1.f : 0.f;
        }
        __syncthreads();
        for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const float ph = ei[u] < 0 ? 1.f : (ej[u] < 0 ? ks[tt][ei[u]] : (ks[tt][ei[u]] * ks[tt][ej[u]]) * r2);
                acc[u] += ph * vs[tt][ec[u]];
            }
        }
    }
    float* dst = state + ((int64_t)f * heads + h) * NE;
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (tid + 256 * u < NE) dst[tid + 256 * u] = acc[u];
}

// block = 256 tokens of one (frame, head): the state in LDS (broadcast reads), the scaled query of a thread's token in its own LDS row
template <typename E, int D>
__global__ __launch_bounds__(256) void taylor_apply_any_kernel(const void* __restrict__ q, const float* __restrict__ state, void* __restrict__ out,
                                                               int n, int heads) {
    constexpr int F = 1 + D + D * D, W = D + 1, NE = F * W;
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    float* S = tsm;                 // [F][W]
    float* qs = tsm + NE;           // [256][D]
    const int h = blockIdx.y, f = blockIdx.z, tid = threadIdx.x;
    const int tok = blockIdx.x * 256 + tid;
    const float* src = state + ((int64_t)f * heads + h) * NE;
    for (int e = tid; e < NE; e += 256) S[e] = src[e];
    const float scale = D == 4 ? 0.5f : (D == 8 ? 0.35355339059327376220f : 0.25f);          // D^-0.5
    const int64_t qrow = ((int64_t)f * n + min(tok, n - 1)) * heads * D + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) qs[tid * D + d] = tay_ld<E>(q, qrow + d) * scale;
    __syncthreads();
    const float r2 = 0.70710678118654752440f;
    float acc[W];
#pragma unroll
    for (int e = 0; e < W; ++e) acc[e] = S[e];                      // phi_0 = 1
    for (int i = 0; i < D; ++i) {
        const float qi = qs[tid * D + i];
        const float* r = S + (1 + i) * W;
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] += qi * r[e];
    }
    for (int i = 0; i < D; ++i) {
        const float qi = qs[tid * D + i];
        for (int j = 0; j < D; ++j) {
            const float ph = (qi * qs[tid * D + j]) * r2;
            const float* r = S + (1 + D + i * D + j) * W;
#pragma unroll
            for (int e = 0; e < W; ++e) acc[e] += ph * r[e];
        }
    }
    if (tok < n) {
        const float inv = 1.f / fmaxf(acc[D], 1e-5f);
        const int64_t orow = ((int64_t)f * n + tok) * heads * D + h * D;
#pragma unroll
        for (int e = 0; e < D; ++e) tay_st<E>(out, orow + e, acc[e] * inv);
    }
}

extern "C" size_t mv2_taylor_state_floats(int dim_head) {
    return (dim_head == 4 || dim_head == 8 || dim_head == 16) ? (size_t)(1 + dim_head + dim_head * dim_head) * (dim_head + 1) : 0;
}

extern "C" int mv2_taylor_kv_d(int dtype, const void* kv, float* state, int frames, int n, int heads, int dim_head, void* stream) {
    if (dim_head == 8) return taylor_kv_launch(dtype, kv, state, frames, n, heads, 1, tay_rows_kv(n, heads), stream);
    if (!kv || !state || frames <= 0 || n <= 0 || heads <= 0 || frames > 65535) return MV2_EINVAL;
    if ((dim_head != 4 && dim_head != 16) || (dtype != MV2_BF16 && dtype != MV2_F32)) return MV2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)heads, (unsigned)frames);
    if (dim_head == 4) {
        if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_kv_any_kernel<BF16, 4>), grid, dim3(256), 0, st, kv, state, n, heads);
        else hipLaunchKernelGGL((taylor_kv_any_kernel<F32, 4>), grid, dim3(256), 0, st, kv, state, n, heads);
    } else {
        if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_kv_any_kernel<BF16, 16>), grid, dim3(256), 0, st, kv, state, n, heads);
        else hipLaunchKernelGGL((taylor_kv_any_kernel<F32, 16>), grid, dim3(256), 0, st, kv, state, n, heads);
    }
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_taylor_apply_d(int dtype, const void* q, const float* state, void* out, int frames, int n, int heads, int dim_head, void* stream) {
    if (dim_head == 8) return taylor_apply_launch(dtype, q, state, out, nullptr, frames, n, heads, 1, tay_rows_q(n, heads), stream);
    if (!q || !state || !out || frames <= 0 || n <= 0 || heads <= 0 || frames > 65535 || heads > 65535) return MV2_EINVAL;
    if ((dim_head != 4 && dim_head != 16) || (dtype != MV2_BF16 && dtype != MV2_F32)) return MV2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)heads, (unsigned)frames);
    const size_t lds = (mv2_taylor_state_floats(dim_head) + (size_t)256 * dim_head) * sizeof(float);
    if (dim_head == 4) {
        if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_apply_any_kernel<BF16, 4>), grid, dim3(256), lds, st, q, state, out, n, heads);
        else hipLaunchKernelGGL((taylor_apply_any_kernel<F32, 4>), grid, dim3(256), lds, st, q, state, out, n, heads);
    } else {
        if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_apply_any_kernel<BF16, 16>), grid, dim3(256), lds, st, q, state, out, n, heads);
        else hipLaunchKernelGGL((taylor_apply_any_kernel<F32, 16>), grid, dim3(256), lds, st, q, state, out, n, heads);
    }
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_taylor_kv(int dtype, const void* kv, float* part, int frames, int n, int heads, int nchunk, void* stream) {
    return taylor_kv_launch(dtype, kv, part, frames, n, heads, nchunk, tay_rows_kv(n, heads), stream);
}

extern "C" int mv2_taylor_apply(int dtype, const void* q, const float* part, void* out, float* kvsum_ws, int frames, int n,
                                int heads, int nchunk, void* stream) {
    return taylor_apply_launch(dtype, q, part, out, kvsum_ws, frames, n, heads, nchunk, tay_rows_q(n, heads), stream);
}

// the same two launches on the FUSED projection's head-major tensor qkv [frame][head][token][q 8 | k 8 | v 8] (see TayLay)
extern "C" int mv2_taylor_kv_fused(int dtype, const void* qkv, float* part, int frames, int n, int heads, int nchunk, void* stream) {
    return taylor_kv_launch(dtype, qkv, part, frames, n, heads, nchunk, tay_fused(n, heads), stream);
}

extern "C" int mv2_taylor_apply_fused(int dtype, const void* qkv, const float* part, void* out, float* kvsum_ws, int frames, int n,
                                      int heads, int nchunk, void* stream) {
    TayLay L = tay_fused(n, heads);
    L.a_off = 0;                         // q leads the token's entry
    return taylor_apply_launch(dtype, qkv, part, out, kvsum_ws, frames, n, heads, nchunk, L, stream);
}
