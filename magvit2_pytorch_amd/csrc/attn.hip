// Axial softmax attention with memory key/values, and Taylor-series linear attention.
// Round-1 form: fp32 VALU arithmetic with LDS-staged K/V tiles (the contractions are <1% of the
// path's FLOPs at the benchmark shapes); the MFMA QK^T / PV form is the next step.
#include "common.h"

struct AttnParams {
    const void* qkv; const float* mem_kv; void* out;
    int nseq, n, heads, n_mem, causal, seq_inner;
    int64_t seq_outer_stride, tok_stride;
};

__device__ __forceinline__ int64_t attn_row(const AttnParams& p, int seq, int i) {
    return (int64_t)(seq / p.seq_inner) * p.seq_outer_stride + (seq % p.seq_inner) + (int64_t)i * p.tok_stride;
}

template <typename E>
__device__ __forceinline__ void load32(const void* base, int64_t elem, float* f) {
    const char* ptr = (const char*)base + elem * E::kBytes;
#pragma unroll
    for (int c = 0; c < 32 / E::kVec; ++c) {
        const i32x4 v = *(const i32x4*)(ptr + c * 16);
        Chunk<E>::unpack(v, f + c * E::kVec);
    }
}
template <typename E>
__device__ __forceinline__ void store32(void* base, int64_t elem, const float* f) {
    char* ptr = (char*)base + elem * E::kBytes;
#pragma unroll
    for (int c = 0; c < 32 / E::kVec; ++c) *(i32x4*)(ptr + c * 16) = Chunk<E>::pack(f + c * E::kVec);
}

// ---- small sequences (time attention): one thread per (seq, head, query), K/V straight from L2
template <typename E>
__global__ __launch_bounds__(256) void attn_small_kernel(const AttnParams p) {
    const int64_t total = (int64_t)p.nseq * p.heads * p.n;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % p.n);
    const int h = (int)((idx / p.n) % p.heads);
    const int seq = (int)(idx / ((int64_t)p.n * p.heads));
    const int ld = 3 * p.heads * 32;
    float q[32], o[32];
    load32<E>(p.qkv, attn_row(p, seq, i) * ld + h * 32, q);
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int klen = p.n_mem + (p.causal ? i + 1 : p.n);
    const float scale = 0.17677669529663687f;   // 32^-0.5
    for (int j = 0; j < klen; ++j) {
        float k[32], v[32];
        if (j < p.n_mem) {
            const float* mk = p.mem_kv + ((int64_t)h * p.n_mem + j) * 32;
            const float* mv = p.mem_kv + (((int64_t)p.heads + h) * p.n_mem + j) * 32;
#pragma unroll
            for (int d = 0; d < 32; ++d) { k[d] = mk[d]; v[d] = mv[d]; }
        } else {
            const int64_t r = attn_row(p, seq, j - p.n_mem) * ld;
            load32<E>(p.qkv, r + (p.heads + h) * 32, k);
            load32<E>(p.qkv, r + (2 * p.heads + h) * 32, v);
        }
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) s += q[d] * k[d];
        s *= scale;
        const float mn = fmaxf(m, s);
        const float a = expf(m - mn), e = expf(s - mn);
        l = l * a + e;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = o[d] * a + e * v[d];
        m = mn;
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] *= inv;
    store32<E>(p.out, attn_row(p, seq, i) * (p.heads * 32) + h * 32, o);
}

// ---- long sequences (space attention): block = 256 queries of one (seq, head); K/V tiles of 64
// keys staged as fp32 in LDS and broadcast-read by every lane.
template <typename E>
__global__ __launch_bounds__(256) void attn_tile_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[64][32];
    __shared__ __attribute__((aligned(16))) float Vs[64][32];
    const int seq = blockIdx.z, h = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool active = i < p.n;
    const int ld = 3 * p.heads * 32;
    float q[32], o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { q[d] = 0.f; o[d] = 0.f; }
    if (active) load32<E>(p.qkv, attn_row(p, seq, i) * ld + h * 32, q);
    float m = -INFINITY, l = 0.f;
    const float scale = 0.17677669529663687f;
    const int klen = p.n_mem + p.n;
    const int key_l = threadIdx.x >> 2, part = threadIdx.x & 3;
    for (int j0 = 0; j0 < klen; j0 += 64) {
        __syncthreads();
        {   // stage: thread -> (key, 8 dims)
            const int j = j0 + key_l;
            float kf[8], vf[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) { kf[d] = 0.f; vf[d] = 0.f; }
            if (j < klen) {
                if (j < p.n_mem) {
                    const float* mk = p.mem_kv + ((int64_t)h * p.n_mem + j) * 32 + part * 8;
                    const float* mv = p.mem_kv + (((int64_t)p.heads + h) * p.n_mem + j) * 32 + part * 8;
#pragma unroll
                    for (int d = 0; d < 8; ++d) { kf[d] = mk[d]; vf[d] = mv[d]; }
                } else {
                    const int64_t r = attn_row(p, seq, j - p.n_mem) * ld;
                    const char* kp = (const char*)p.qkv + (r + (p.heads + h) * 32 + part * 8) * E::kBytes;
                    const char* vp = (const char*)p.qkv + (r + (2 * p.heads + h) * 32 + part * 8) * E::kBytes;
#pragma unroll
                    for (int c = 0; c < 8 / E::kVec; ++c) {
                        Chunk<E>::unpack(*(const i32x4*)(kp + c * 16), kf + c * E::kVec);
                        Chunk<E>::unpack(*(const i32x4*)(vp + c * 16), vf + c * E::kVec);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < 8; ++d) { Ks[key_l][part * 8 + d] = kf[d]; Vs[key_l][part * 8 + d] = vf[d]; }
        }
        __syncthreads();
        const int jn = min(64, klen - j0);
        for (int jj = 0; jj < jn; ++jj) {
            const int j = j0 + jj;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s += q[d] * Ks[jj][d];
            s *= scale;
            const bool vis = !p.causal || j < p.n_mem || (j - p.n_mem) <= i;
            if (vis) {
                const float mn = fmaxf(m, s);
                const float a = expf(m - mn), e = expf(s - mn);
                l = l * a + e;
#pragma unroll
                for (int d = 0; d < 32; ++d) o[d] = o[d] * a + e * Vs[jj][d];
                m = mn;
            }
        }
    }
    if (active) {
        const float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] *= inv;
        store32<E>(p.out, attn_row(p, seq, i) * (p.heads * 32) + h * 32, o);
    }
}

extern "C" int mv2_attention(int dtype, const void* qkv, const float* mem_kv, void* out, int nseq, int n, int heads,
                             int dim_head, int n_mem, int causal, int seq_inner, int64_t seq_outer_stride,
                             int64_t tok_stride, void* stream) {
    if (!qkv || !mem_kv || !out || nseq <= 0 || n <= 0 || heads <= 0 || n_mem < 0 || seq_inner <= 0) return MV2_EINVAL;
    if (dim_head != 32 || (dtype != MV2_BF16 && dtype != MV2_F32)) return MV2_EINVAL;
    AttnParams p{qkv, mem_kv, out, nseq, n, heads, n_mem, causal, seq_inner, seq_outer_stride, tok_stride};
    hipStream_t st = (hipStream_t)stream;
    if (n < 64) {
        const int64_t total = (int64_t)nseq * heads * n;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (dtype == MV2_BF16) hipLaunchKernelGGL((attn_small_kernel<BF16>), dim3(blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_small_kernel<F32>), dim3(blocks), dim3(256), 0, st, p);
    } else {
        if (nseq > 65535) return MV2_EINVAL;
        dim3 grid((unsigned)((n + 255) / 256), (unsigned)heads, (unsigned)nseq);
        if (dtype == MV2_BF16) hipLaunchKernelGGL((attn_tile_kernel<BF16>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_tile_kernel<F32>), grid, dim3(256), 0, st, p);
    }
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// ------------------------------------------------------------------------------ Taylor attention
// phi(x) (d = 8): index 0 -> 1 ; 1 + i -> x_i ; 9 + 8 i + j -> x_i x_j sqrt(0.5).   73 features.
// KV state per (frame, head): [73][9] fp32, column 8 = sum of phi(k) (the normaliser).
#define TAY_F 73
#define TAY_S (73 * 9)

template <typename E>
__global__ __launch_bounds__(256) void taylor_kv_kernel(const void* __restrict__ kv, float* __restrict__ part, int n, int heads,
                                                        int nchunk) {
    const int f = blockIdx.y, j = blockIdx.x;
    const int h = threadIdx.x >> 3, i = threadIdx.x & 7;
    if (h >= heads) return;
    const int per = (n + nchunk - 1) / nchunk;
    const int t0 = j * per, t1 = min(n, t0 + per);
    const int ld = 2 * heads * 8;
    float a2[8][9], a1[9], a0[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) { a1[e] = 0.f; a0[e] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) a2[jj][e] = 0.f; }
    const float r2 = 0.70710678118654752440f;
    for (int t = t0; t < t1; ++t) {
        const int64_t row = ((int64_t)f * n + t) * ld;
        float k[8], v[9];
#pragma unroll
        for (int c = 0; c < 8 / E::kVec; ++c) {
            Chunk<E>::unpack(*(const i32x4*)((const char*)kv + (row + h * 8) * E::kBytes + c * 16), k + c * E::kVec);
            Chunk<E>::unpack(*(const i32x4*)((const char*)kv + (row + (heads + h) * 8) * E::kBytes + c * 16), v + c * E::kVec);
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

__global__ __launch_bounds__(256) void taylor_reduce_kernel(const float* __restrict__ part, float* __restrict__ kvsum, int heads,
                                                            int nchunk) {
    const int fh = blockIdx.x;             // frame * heads + head
    const int f = fh / heads, h = fh % heads;
    for (int e = threadIdx.x; e < TAY_S; e += 256) {
        float a = 0.f;
        for (int j = 0; j < nchunk; ++j) a += part[(((int64_t)f * nchunk + j) * heads + h) * TAY_S + e];
        kvsum[(int64_t)fh * TAY_S + e] = a;
    }
}

// block: 64 tokens of one frame; thread -> (token slot, head), loops 4 token slots
template <typename E>
__global__ __launch_bounds__(256) void taylor_apply_kernel(const void* __restrict__ q, const float* __restrict__ kvsum,
                                                           void* __restrict__ out, int n, int heads) {
    extern __shared__ __attribute__((aligned(16))) float kvs[];   // heads * 657
    const int f = blockIdx.y;
    const int tot = heads * TAY_S;
    for (int e = threadIdx.x; e < tot; e += 256) kvs[e] = kvsum[(int64_t)f * tot + e];
    __syncthreads();
    const int tpb = 256 / heads;           // tokens per pass
    const int h = threadIdx.x % heads, ts = threadIdx.x / heads;
    const float* S = kvs + h * TAY_S;
    const float scale = 0.35355339059327376220f;   // 8^-0.5
    const float r2 = 0.70710678118654752440f;
    const int tile0 = blockIdx.x * 64;
    for (int tt = ts; tt < 64; tt += tpb) {
        const int t = tile0 + tt;
        if (t >= n) break;
        const int64_t row = (int64_t)f * n + t;
        float qv[8];
#pragma unroll
        for (int c = 0; c < 8 / E::kVec; ++c)
            Chunk<E>::unpack(*(const i32x4*)((const char*)q + (row * heads * 8 + h * 8) * E::kBytes + c * 16), qv + c * E::kVec);
#pragma unroll
        for (int d = 0; d < 8; ++d) qv[d] *= scale;
        float acc[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) acc[e] = S[e];            // phi_0 = 1
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 9; ++e) acc[e] += qv[i] * S[(1 + i) * 9 + e];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float ph = (qv[i] * qv[jj]) * r2;
#pragma unroll
                for (int e = 0; e < 9; ++e) acc[e] += ph * S[(9 + 8 * i + jj) * 9 + e];
            }
        const float inv = 1.f / fmaxf(acc[8], 1e-5f);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = acc[e] * inv;
#pragma unroll
        for (int c = 0; c < 8 / E::kVec; ++c)
            *(i32x4*)((char*)out + (row * heads * 8 + h * 8) * E::kBytes + c * 16) = Chunk<E>::pack(o + c * E::kVec);
    }
}

extern "C" int mv2_taylor_kv(int dtype, const void* kv, float* part, int frames, int n, int heads, int nchunk, void* stream) {
    if (!kv || !part || frames <= 0 || n <= 0 || heads <= 0 || heads > 32 || nchunk <= 0 || nchunk > n) return MV2_EINVAL;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    dim3 grid((unsigned)nchunk, (unsigned)frames);
    const unsigned threads = (unsigned)(heads * 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_kv_kernel<BF16>), grid, dim3(threads), 0, st, kv, part, n, heads, nchunk);
    else hipLaunchKernelGGL((taylor_kv_kernel<F32>), grid, dim3(threads), 0, st, kv, part, n, heads, nchunk);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_taylor_apply(int dtype, const void* q, const float* part, void* out, float* kvsum_ws, int frames, int n,
                                int heads, int nchunk, void* stream) {
    if (!q || !part || !out || !kvsum_ws || frames <= 0 || n <= 0 || heads <= 0 || 256 % heads != 0) return MV2_EINVAL;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    const size_t lds = (size_t)heads * TAY_S * sizeof(float);
    if (lds > 64 * 1024) return MV2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(taylor_reduce_kernel, dim3((unsigned)(frames * heads)), dim3(256), 0, st, part, kvsum_ws, heads, nchunk);
    MV2_LAUNCH_CHECK();
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)frames);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((taylor_apply_kernel<BF16>), grid, dim3(256), lds, st, q, kvsum_ws, out, n, heads);
    else hipLaunchKernelGGL((taylor_apply_kernel<F32>), grid, dim3(256), lds, st, q, kvsum_ws, out, n, heads);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
