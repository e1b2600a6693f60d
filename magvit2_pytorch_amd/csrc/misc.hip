// Bandwidth-bound kernels of the hot path: API-edge layout change, RMSNorm statistics and the
// three SqueezeExcite passes.  All loads/stores are 16-byte vectors on NDHWC rows.
#include <algorithm>
#include "common.h"

// --------------------------------------------------------------------------- video -> padded NDHWC
template <typename E>
__global__ __launch_bounds__(256) void video_to_ndhwc_kernel(const float* __restrict__ in, typename E::T* __restrict__ out,
                                                             int B, int C, int T, int H, int W, int pad_t, int pad_hw,
                                                             int c_pad, int64_t npix, int64_t tail) {
    const int Tp = T + pad_t, Hp = H + 2 * pad_hw, Wp = W + 2 * pad_hw;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix + tail; i += stride) {
        if (i >= npix) {   // zero tail (read slack for the merged (dw, c) K segments of conv_in)
            for (int c = 0; c < c_pad; ++c) st_elem<E>(out, npix * c_pad + (i - npix) * c_pad + c, 0.f);
            continue;
        }
        const int w = (int)(i % Wp); int64_t r = i / Wp;
        const int h = (int)(r % Hp); r /= Hp;
        const int t = (int)(r % Tp); const int b = (int)(r / Tp);
        const int ti = t - pad_t, hi = h - pad_hw, wi = w - pad_hw;
        const bool inside = ti >= 0 && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        for (int c = 0; c < c_pad; ++c) {
            float v = 0.f;
            if (inside && c < C) v = in[(((int64_t)b * C + c) * T + ti) * ((int64_t)H * W) + (int64_t)hi * W + wi];
            st_elem<E>(out, i * c_pad + c, v);
        }
    }
}

extern "C" int mv2_video_to_ndhwc(int dtype, const float* in, void* out, int B, int C, int T, int H, int W,
                                  int pad_t, int pad_hw, int c_pad, int tail_elems, void* stream) {
    if (!in || !out || c_pad < C || B <= 0 || tail_elems % c_pad != 0) return MV2_EINVAL;
    const int64_t npix = (int64_t)B * (T + pad_t) * (H + 2 * pad_hw) * (W + 2 * pad_hw);
    const int64_t tail = tail_elems / c_pad;
    int64_t blocks = (npix + tail + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV2_BF16)
        hipLaunchKernelGGL((video_to_ndhwc_kernel<BF16>), dim3((unsigned)blocks), dim3(256), 0, st, in, (unsigned short*)out, B, C, T, H, W, pad_t, pad_hw, c_pad, npix, tail);
    else if (dtype == MV2_F32)
        hipLaunchKernelGGL((video_to_ndhwc_kernel<F32>), dim3((unsigned)blocks), dim3(256), 0, st, in, (float*)out, B, C, T, H, W, pad_t, pad_hw, c_pad, npix, tail);
    else return MV2_EINVAL;
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- RMSNorm statistics
// G lanes cooperate on one row (G = min(64, chunks per row rounded up to a power of two)); lane reads chunks
// lane, lane+G, ... (any chunk count); sum of squares reduced with xor shuffles inside the group.
template <typename E>
__global__ __launch_bounds__(256) void rms_rowscale_kernel(const void* __restrict__ x, float* __restrict__ rs, int64_t rows,
                                                           int C, int G, int shift_hw, int T) {
    const int nchunk = C / E::kVec;
    const int rows_per_block = 256 / G;
    const int g = threadIdx.x / G, l = threadIdx.x % G;
    for (int64_t row = (int64_t)blockIdx.x * rows_per_block + g; row < rows; row += (int64_t)gridDim.x * rows_per_block) {
        float ss = 0.f;
        bool prev_ok = true;
        if (shift_hw > 0) prev_ok = ((row / shift_hw) % T) > 0;
        for (int ch = l; ch < nchunk; ch += G) {
            int64_t src = row;
            bool zero = false;
            if (shift_hw > 0 && ch * E::kVec >= C / 2) { src = row - shift_hw; zero = !prev_ok; }
            if (!zero) {
                const i32x4 v = *(const i32x4*)((const char*)x + (src * C + (int64_t)ch * E::kVec) * E::kBytes);
                float f[8]; Chunk<E>::unpack(v, f);
#pragma unroll
                for (int j = 0; j < E::kVec; ++j) ss += f[j] * f[j];
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (l == 0) rs[row] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    }
}

// --------------------------------------------------------------------------- mv2_split_bf16
// one thread per 4 consecutive channels: 16 bytes in, 3 x 8 bytes out (HBM-bound: 4 + 6 bytes per element)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, int64_t n4, int c4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / c4; const int q = (int)(i - row * c4);
        const f32x4 v = *(const f32x4*)(x + i * 4);
        unsigned short h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = f2bf(v[j]); l[j] = f2bf(v[j] - bf2f(h[j])); }
        i32x2 hv, lv;
        hv[0] = (int)((unsigned)h[0] | ((unsigned)h[1] << 16)); hv[1] = (int)((unsigned)h[2] | ((unsigned)h[3] << 16));
        lv[0] = (int)((unsigned)l[0] | ((unsigned)l[1] << 16)); lv[1] = (int)((unsigned)l[2] | ((unsigned)l[3] << 16));
        unsigned short* o = out + row * (int64_t)(12 * c4) + q * 4;
        *(i32x2*)o = hv;
        *(i32x2*)(o + 4 * c4) = lv;
        *(i32x2*)(o + 8 * c4) = hv;
    }
}

extern "C" int mv2_split_bf16(const float* x, void* out, int64_t rows, int C, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || C % 4 != 0) return MV2_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 7)) return MV2_EINVAL;      // 16-byte loads, 8-byte stores
    const int64_t n4 = rows * (C / 4);
    int64_t blocks = (n4 + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)out, n4, C / 4);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int mv2_rms_rowscale(int dtype, const void* x, float* rowscale, int64_t rows, int C, int shift_hw, int T, void* stream) {
    if (!x || !rowscale || rows <= 0) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if (C % kvec != 0 || (shift_hw > 0 && (C / 2) % kvec != 0)) return MV2_EINVAL;
    const int nchunk = C / kvec;
    int G = 64;
    if (nchunk < 64) { G = 1; while (G < nchunk) G <<= 1; }      // next power of two: lanes >= nchunk find no chunk and add 0 (round 6: any width)
    const int rpb = 256 / G;
    int64_t blocks = (rows + rpb - 1) / rpb; if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV2_BF16) hipLaunchKernelGGL((rms_rowscale_kernel<BF16>), dim3((unsigned)blocks), dim3(256), 0, st, x, rowscale, rows, C, G, shift_hw, T);
    else if (dtype == MV2_F32) hipLaunchKernelGGL((rms_rowscale_kernel<F32>), dim3((unsigned)blocks), dim3(256), 0, st, x, rowscale, rows, C, G, shift_hw, T);
    else return MV2_EINVAL;
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- SqueezeExcite: pooling loop
// Online-softmax pooling of rows p_begin + slot, + R, ... < p_end of one frame by a group of G lanes (lane l owns
// channel chunks l + v*G).  U rows per trip: the U row loads are in flight together and the running state is
// rescaled once per trip (a one-row-per-trip loop is a chain of L2 latencies: measured 24 us for 16 trips).
template <typename E, int NV>
__device__ __forceinline__ void se_pool_rows(const char* __restrict__ base, int C, int G, int l, int p_begin, int p_end, int R,
                                             const float (&wreg)[NV][E::kVec], float bk, float& m_run, float& s_run,
                                             float (&acc)[NV][E::kVec], int nch = 1 << 30) {
    constexpr int KV = E::kVec;
    constexpr int U = 4;
    for (int p0 = p_begin; p0 < p_end; p0 += U * R) {
        i32x4 q[U][NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = min(p0 + u * R, p_end - 1);       // clamped duplicate rows are masked below
#pragma unroll
            for (int v = 0; v < NV; ++v) {       // lanes past the row's last chunk (G is the next power of two) read chunk 0 with zero weights
                const int ch = (l + v * G) < nch ? (l + v * G) : 0;
                q[u][v] = *(const i32x4*)(base + ((int64_t)p * C + (int64_t)ch * KV) * E::kBytes);
            }
        }
        float dot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dot[u] = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float xv[KV];
                Chunk<E>::unpack(q[u][v], xv);
#pragma unroll
                for (int k = 0; k < KV; ++k) dot[u] += xv[k] * wreg[v][k];
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < U; ++u) dot[u] += __shfl_xor(dot[u], o, 64);
        float m_new = m_run;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dot[u] = (p0 + u * R < p_end) ? dot[u] + bk : -INFINITY;
            m_new = fmaxf(m_new, dot[u]);
        }
        const float sc = expf(m_run - m_new);               // exp(-inf) = 0 on the first trip (row p0 is always valid)
        float e[U];
        s_run *= sc;
#pragma unroll
        for (int u = 0; u < U; ++u) { e[u] = expf(dot[u] - m_new); s_run += e[u]; }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int k = 0; k < KV; ++k) acc[v][k] *= sc;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float xv[KV];
                Chunk<E>::unpack(q[u][v], xv);
#pragma unroll
                for (int k = 0; k < KV; ++k) acc[v][k] += e[u] * xv[k];
            }
        }
        m_run = m_new;
    }
}

// --------------------------------------------------------------------------- SqueezeExcite: pool
// grid (nchunk, frames).  A block streams its slice of the frame's P rows once; each row slot keeps
// an online-softmax state (running max, sum of exp, exp-weighted channel sums) in registers; the
// slots are merged through LDS at the end.
template <typename E, int NV>
__global__ __launch_bounds__(256) void se_pool_kernel(const void* __restrict__ y, const float* __restrict__ wk, float bk,
                                                      float* __restrict__ part, int P, int C, int G, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KV = E::kVec;
    const int f = blockIdx.y, j = blockIdx.x;
    const int R = 256 / G;                      // row slots per block
    const int slot = threadIdx.x / G, l = threadIdx.x % G;
    const int rows_per_chunk = (P + nchunk - 1) / nchunk;
    const int p_begin = j * rows_per_chunk, p_end = min(P, p_begin + rows_per_chunk);

    const int nch = C / KV;                    // chunks of a row; G >= nch is a power of two when nch < 64 (masked lanes)
    float wreg[NV][KV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < KV; ++k) wreg[v][k] = (l + v * G) < nch ? wk[(l + v * G) * KV + k] : 0.f;

    float m_run = -INFINITY, s_run = 0.f;
    float acc[NV][KV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < KV; ++k) acc[v][k] = 0.f;

    const char* base = (const char*)y + ((int64_t)f * P) * C * E::kBytes;
    se_pool_rows<E, NV>(base, C, G, l, p_begin + slot, p_end, R, wreg, bk, m_run, s_run, acc, nch);
    // merge the R slots: sm[slot*(C+2)] = {m, s, acc[C]}
    float* mine = sm + slot * (C + 2);
    if (l == 0) { mine[0] = m_run; mine[1] = s_run; }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < KV; ++k) if ((l + v * G) < nch) mine[2 + (l + v * G) * KV + k] = acc[v][k];
    __syncthreads();
    float M = -INFINITY;
    for (int r = 0; r < R; ++r) M = fmaxf(M, sm[r * (C + 2)]);
    float* dst = part + ((int64_t)f * nchunk + j) * (C + 2);
    for (int c = threadIdx.x; c < C + 2; c += 256) {
        float o;
        if (c == 0) o = M;
        else {
            o = 0.f;
            for (int r = 0; r < R; ++r) {
                const float mr = sm[r * (C + 2)];
                const float w = (mr == -INFINITY) ? 0.f : expf(mr - M);
                o += sm[r * (C + 2) + c] * w;
            }
        }
        dst[c] = o;
    }
}

extern "C" int mv2_se_pool(int dtype, const void* y, const float* wk, float bk, float* part, int frames, int P, int C,
                           int nchunk, void* stream) {
    if (!y || !wk || !part || frames <= 0 || P <= 0 || nchunk <= 0 || nchunk > P) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    if (C % kvec != 0) return MV2_EINVAL;
    const int nch = C / kvec;
    int G = 64;
    if (nch < 64) { G = 1; while (G < nch) G <<= 1; }                 // next power of two; lanes >= nch are masked
    const int NV = (nch + G - 1) / G;                                    // row chunks per lane (the last group may be partly masked)
    if (NV != 1 && NV != 2 && NV != 4) return MV2_EINVAL;
    const size_t lds = (size_t)(256 / G) * (C + 2) * sizeof(float);
    if (lds > 64 * 1024) return MV2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nchunk, (unsigned)frames);
#define SE_POOL(EE, NVV) hipLaunchKernelGGL((se_pool_kernel<EE, NVV>), grid, dim3(256), lds, st, y, wk, bk, part, P, C, G, nchunk)
    if (dtype == MV2_BF16) { if (NV == 1) SE_POOL(BF16, 1); else if (NV == 2) SE_POOL(BF16, 2); else SE_POOL(BF16, 4); }
    else { if (NV == 1) SE_POOL(F32, 1); else if (NV == 2) SE_POOL(F32, 2); else SE_POOL(F32, 4); }
#undef SE_POOL
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- SqueezeExcite: gate
// one 1024-thread block per frame: merge chunk partials -> pooled[C]; hidden = lrelu(W1 pooled + b1);
// gate = sigmoid(W2 hidden + b2).  The work is tiny but latency-bound, so every reduction is split
// over thread slices (output unit x reduction slice) and combined through LDS.  Weights arrive
// TRANSPOSED (w1t[C][hidden], w2t[hidden][C]): lanes of a slice read consecutive output units.
__global__ __launch_bounds__(1024) void se_gate_kernel(const float* __restrict__ part, const float* __restrict__ w1t,
                                                       const float* __restrict__ b1, const float* __restrict__ w2t,
                                                       const float* __restrict__ b2, float* __restrict__ gate, int C,
                                                       int hidden, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pooled = sm;               // C
    float* hid = pooled + C;          // hidden
    float* wj = hid + hidden;         // nchunk: exp(m_j - M)
    float* red = wj + nchunk;         // 32 scratch
    float* slab = red + 32;           // 1024 partial sums
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pf = part + (int64_t)f * nchunk * (C + 2);
    float m = -INFINITY;
    for (int j = tid; j < nchunk; j += 1024) m = fmaxf(m, pf[(int64_t)j * (C + 2)]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    float M = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) M = fmaxf(M, red[w]);
    float s = 0.f;
    for (int j = tid; j < nchunk; j += 1024) {
        const float mj = pf[(int64_t)j * (C + 2)];
        const float w = (mj == -INFINITY) ? 0.f : expf(mj - M);
        wj[j] = w;
        s += pf[(int64_t)j * (C + 2) + 1] * w;
    }
    s = wave_sum(s);
    if (lane == 0) red[16 + wave] = s;
    __syncthreads();
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) S += red[16 + w];
    const float inv_s = 1.f / S;
    {   // pooled[c]: threads (c, j-slice)
        const int JS = max(1, 1024 / C);
        const int c = tid % C, js = tid / C;
        if (js < JS) {
            float a = 0.f;
            int j = js;
            for (; j + 7 * JS < nchunk; j += 8 * JS) {             // eight loads in flight (the plain loop was a chain of L2 latencies)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pf[(int64_t)(j + u * JS) * (C + 2) + 2 + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u] * wj[j + u * JS];
            }
            for (; j < nchunk; j += JS) a += pf[(int64_t)j * (C + 2) + 2 + c] * wj[j];
            slab[js * C + c] = a;
        }
        __syncthreads();
        if (tid < C) {
            float a = 0.f;
            for (int q = 0; q < JS; ++q) a += slab[q * C + tid];
            pooled[tid] = a * inv_s;
        }
        __syncthreads();
    }
    {   // hidden[k]: threads (k, c-slice)
        const int KS = max(1, 1024 / hidden);
        const int k = tid % hidden, cs = tid / hidden;
        const int per = (C + KS - 1) / KS;
        if (cs < KS) {
            const int c0 = cs * per, c1 = min(C, c0 + per);
            float a0 = 0.f, a1 = 0.f;
            int c = c0;
            for (; c + 16 <= c1; c += 16) {          // 16 independent loads in flight per thread
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = w1t[(int64_t)(c + u) * hidden + k];
#pragma unroll
                for (int u = 0; u < 16; u += 2) { a0 += w[u] * pooled[c + u]; a1 += w[u + 1] * pooled[c + u + 1]; }
            }
            if (c < c1) {                                            // tail (< 16): still one batch of independent loads
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = (c + u < c1) ? w1t[(int64_t)(c + u) * hidden + k] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) a0 += (c + u < c1) ? w[u] * pooled[c + u] : 0.f;
            }
            slab[cs * hidden + k] = a0 + a1;
        }
        __syncthreads();
        if (tid < hidden) {
            float a = b1[tid];
            for (int q = 0; q < KS; ++q) a += slab[q * hidden + tid];
            hid[tid] = a > 0.f ? a : 0.1f * a;
        }
        __syncthreads();
    }
    {   // gate[c]: threads (c, k-slice)
        const int GS = max(1, 1024 / C);
        const int c = tid % C, ks = tid / C;
        const int per = (hidden + GS - 1) / GS;
        if (ks < GS) {
            const int k0 = ks * per, k1 = min(hidden, k0 + per);
            float a0 = 0.f, a1 = 0.f;
            int k = k0;
            for (; k + 16 <= k1; k += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = w2t[(int64_t)(k + u) * C + c];
#pragma unroll
                for (int u = 0; u < 16; u += 2) { a0 += w[u] * hid[k + u]; a1 += w[u + 1] * hid[k + u + 1]; }
            }
            if (k < k1) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = (k + u < k1) ? w2t[(int64_t)(k + u) * C + c] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) a0 += (k + u < k1) ? w[u] * hid[k + u] : 0.f;
            }
            slab[ks * C + c] = a0 + a1;
        }
        __syncthreads();
        if (tid < C) {
            float a = b2[tid];
            for (int q = 0; q < GS; ++q) a += slab[q * C + tid];
            gate[(int64_t)f * C + tid] = 1.f / (1.f + expf(-a));
        }
    }
}

extern "C" int mv2_se_gate(const float* part, const float* w1t, const float* b1, const float* w2t, const float* b2,
                           float* gate, int frames, int C, int hidden, int nchunk, void* stream) {
    if (!part || !w1t || !b1 || !w2t || !b2 || !gate || frames <= 0 || nchunk <= 0) return MV2_EINVAL;
    if (C <= 0 || C > 1024 || hidden <= 0 || hidden > 1024) return MV2_EINVAL;
    const size_t lds = (size_t)(C + hidden + nchunk + 32 + 1024) * sizeof(float);
    if (lds > 64 * 1024) return MV2_EINVAL;
    hipLaunchKernelGGL(se_gate_kernel, dim3((unsigned)frames), dim3(1024), lds, (hipStream_t)stream, part, w1t, b1, w2t, b2, gate, C, hidden, nchunk);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- SqueezeExcite: two-launch form
// For frames that fit a CU's reach (P*C of a frame <= 512 KB: the 16x16 and 32x32 levels) the three launches
// above are latency chains on 5-20 CUs (measured 27 us per ResidualUnit at C = 512).  Two wide launches
// replace them:
//   pool_fc1 : grid (S, frames), 1024 threads.  Every block pools its WHOLE frame (one online-softmax pass,
//              redundant across the S slices -- the frame is L2 resident) and produces hidden/S units of
//              lrelu(W1 pooled + b1) with the reduction split over 1024/hs thread slices.
//   fc2_apply: grid (S2, frames), 1024 threads.  A block owns C/S2 channels: gate = sigmoid(W2 hid + b2) for
//              them (one round of 16 loads per thread), then out = gate * y + residual over the frame.
template <typename E, int NV>
__global__ __launch_bounds__(1024) void se_pool_fc1_kernel(const void* __restrict__ y, const float* __restrict__ wk, float bk,
                                                           const float* __restrict__ w1t, const float* __restrict__ b1,
                                                           float* __restrict__ hid, int P, int C, int G, int hidden, int hs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KV = E::kVec;
    const int f = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
    const int R = 1024 / G;
    const int slot = tid / G, l = tid % G;
    float wreg[NV][KV], acc[NV][KV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < KV; ++k) { wreg[v][k] = wk[(l + v * G) * KV + k]; acc[v][k] = 0.f; }
    float m_run = -INFINITY, s_run = 0.f;
    const char* base = (const char*)y + ((int64_t)f * P) * C * E::kBytes;
    se_pool_rows<E, NV>(base, C, G, l, slot, P, R, wreg, bk, m_run, s_run, acc);
    float* mine = sm + slot * (C + 2);
    float* wr = sm + R * (C + 2);       // R merge weights
    float* pooled = wr + 64;            // C
    float* slab = pooled + C;           // 1024
    if (l == 0) { mine[0] = m_run; mine[1] = s_run; }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < KV; ++k) mine[2 + (l + v * G) * KV + k] = acc[v][k];
    __syncthreads();
    float M = -INFINITY;
    for (int r = 0; r < R; ++r) M = fmaxf(M, sm[r * (C + 2)]);
    if (tid < R) { const float mr = sm[tid * (C + 2)]; wr[tid] = (mr == -INFINITY) ? 0.f : expf(mr - M); }
    __syncthreads();
    float S = 0.f;
    for (int r = 0; r < R; ++r) S += sm[r * (C + 2) + 1] * wr[r];
    const float inv_s = 1.f / S;
    for (int c = tid; c < C; c += 1024) {
        float o = 0.f;
        for (int r = 0; r < R; ++r) o += sm[r * (C + 2) + 2 + c] * wr[r];
        pooled[c] = o * inv_s;
    }
    __syncthreads();
    // hidden units [sl*hs, (sl+1)*hs): threads (unit, reduction slice)
    const int KS = 1024 / hs;
    const int kl = tid % hs, cs = tid / hs, k = sl * hs + kl;
    const int per = (C + KS - 1) / KS;
    float a0 = 0.f, a1 = 0.f;
    if (k < hidden) {
        const int c0 = cs * per, c1 = min(C, c0 + per);
        int c = c0;
        for (; c + 16 <= c1; c += 16) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = w1t[(int64_t)(c + u) * hidden + k];
#pragma unroll
            for (int u = 0; u < 16; u += 2) { a0 += w[u] * pooled[c + u]; a1 += w[u + 1] * pooled[c + u + 1]; }
        }
        for (; c < c1; ++c) a0 += w1t[(int64_t)c * hidden + k] * pooled[c];
    }
    slab[cs * hs + kl] = a0 + a1;
    __syncthreads();
    if (tid < hs && k < hidden) {
        float a = b1[k];
        for (int q = 0; q < KS; ++q) a += slab[q * hs + tid];
        hid[(int64_t)f * hidden + k] = a > 0.f ? a : 0.1f * a;
    }
}

// fc1 from pooling RECORDS (written by mv2_pw_se / mv2_conv3's fused epilogue / mv2_se_pool) instead of a pooling pass:
// grid (S, frames); every block merges the frame's nchunk records (a few KB) and produces hidden/S units.
__global__ __launch_bounds__(1024) void se_rec_fc1_kernel(const float* __restrict__ part, const float* __restrict__ w1t,
                                                          const float* __restrict__ b1, float* __restrict__ hid, int C, int nchunk,
                                                          int hidden, int hs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int f = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
    const int R = nchunk;
    float* wr = sm + R * (C + 2);       // R merge weights
    float* pooled = wr + R;             // C
    float* slab = pooled + C;           // 1024
    const float* pf = part + (int64_t)f * nchunk * (C + 2);
#pragma unroll 4
    for (int i = tid; i < R * (C + 2); i += 1024) sm[i] = pf[i];
    __syncthreads();
    float M = -INFINITY;
    for (int r = 0; r < R; ++r) M = fmaxf(M, sm[r * (C + 2)]);
    if (tid < R) { const float mr = sm[tid * (C + 2)]; wr[tid] = (mr == -INFINITY) ? 0.f : expf(mr - M); }
    __syncthreads();
    float S = 0.f;
    for (int r = 0; r < R; ++r) S += sm[r * (C + 2) + 1] * wr[r];
    const float inv_s = 1.f / S;
    for (int c = tid; c < C; c += 1024) {
        float o = 0.f;
        for (int r = 0; r < R; ++r) o += sm[r * (C + 2) + 2 + c] * wr[r];
        pooled[c] = o * inv_s;
    }
    __syncthreads();
    const int KS = 1024 / hs;
    const int kl = tid % hs, cs = tid / hs, k = sl * hs + kl;
    const int per = (C + KS - 1) / KS;
    float a0 = 0.f, a1 = 0.f;
    if (k < hidden) {
        const int c0 = cs * per, c1 = min(C, c0 + per);
        int c = c0;
        for (; c + 16 <= c1; c += 16) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = w1t[(int64_t)(c + u) * hidden + k];
#pragma unroll
            for (int u = 0; u < 16; u += 2) { a0 += w[u] * pooled[c + u]; a1 += w[u + 1] * pooled[c + u + 1]; }
        }
        for (; c < c1; ++c) a0 += w1t[(int64_t)c * hidden + k] * pooled[c];
    }
    slab[cs * hs + kl] = a0 + a1;
    __syncthreads();
    if (tid < hs && k < hidden) {
        float a = b1[k];
        for (int q = 0; q < KS; ++q) a += slab[q * hs + tid];
        hid[(int64_t)f * hidden + k] = a > 0.f ? a : 0.1f * a;
    }
}

extern "C" int mv2_se_rec_fc1(const float* part, const float* w1t, const float* b1, float* hid, int frames, int C, int nchunk,
                              int hidden, int nslice, void* stream) {
    if (!part || !w1t || !b1 || !hid || frames <= 0 || nchunk <= 0 || nslice <= 0 || C <= 0) return MV2_EINVAL;
    if (hidden <= 0 || hidden % nslice != 0) return MV2_EINVAL;
    const int hs = hidden / nslice;
    if (!is_pow2(hs) || hs > 1024) return MV2_EINVAL;
    const size_t lds = ((size_t)nchunk * (C + 2) + nchunk + C + 1024) * sizeof(float);
    if (lds > 64 * 1024) return MV2_EINVAL;
    hipLaunchKernelGGL(se_rec_fc1_kernel, dim3((unsigned)nslice, (unsigned)frames), dim3(1024), lds, (hipStream_t)stream, part, w1t, b1, hid,
                       C, nchunk, hidden, hs);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

template <typename E>
__global__ __launch_bounds__(1024) void se_fc2_apply_kernel(const void* __restrict__ y, const float* __restrict__ hidv,
                                                            const float* __restrict__ w2t, const float* __restrict__ b2,
                                                            const void* __restrict__ res, void* __restrict__ out, int P, int C,
                                                            int hidden, int cw) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KV = E::kVec;
    float* hs = sm;                 // hidden
    float* slab = hs + hidden;      // 1024
    float* g = slab + 1024;         // cw
    const int f = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < hidden; k += 1024) hs[k] = hidv[(int64_t)f * hidden + k];
    __syncthreads();
    const int GS = 1024 / cw;
    const int cl = tid % cw, ks = tid / cw, c = sl * cw + cl;
    const int per = (hidden + GS - 1) / GS;
    {
        const int k0 = ks * per, k1 = min(hidden, k0 + per);
        float a0 = 0.f, a1 = 0.f;
        int k = k0;
        for (; k + 16 <= k1; k += 16) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = w2t[(int64_t)(k + u) * C + c];
#pragma unroll
            for (int u = 0; u < 16; u += 2) { a0 += w[u] * hs[k + u]; a1 += w[u + 1] * hs[k + u + 1]; }
        }
        for (; k < k1; ++k) a0 += w2t[(int64_t)k * C + c] * hs[k];
        slab[ks * cw + cl] = a0 + a1;
    }
    __syncthreads();
    if (tid < cw) {
        float a = b2[c];
        for (int q = 0; q < GS; ++q) a += slab[q * cw + tid];
        g[tid] = 1.f / (1.f + expf(-a));
    }
    __syncthreads();
    const int cps = cw / KV;                         // 16-byte chunks of the slice per row (power of two)
    const int lg = 31 - __clz(cps);
    const int items = P << lg;
    const int64_t row0 = (int64_t)f * P;
    for (int i = tid; i < items; i += 1024) {
        const int p = i >> lg, ch = i & (cps - 1);
        const int64_t idx = ((row0 + p) * C + (int64_t)sl * cw) / KV + ch;
        const i32x4 yv = ((const i32x4*)y)[idx];
        float a[8], r[8];
        Chunk<E>::unpack(yv, a);
        if (res) { const i32x4 rv = ((const i32x4*)res)[idx]; Chunk<E>::unpack(rv, r); }
#pragma unroll
        for (int k = 0; k < KV; ++k) a[k] = a[k] * g[ch * KV + k] + (res ? r[k] : 0.f);
        ((i32x4*)out)[idx] = Chunk<E>::pack(a);
    }
}

extern "C" int mv2_se_pool_fc1(int dtype, const void* y, const float* wk, float bk, const float* w1t, const float* b1,
                               float* hid, int frames, int P, int C, int hidden, int nslice, void* stream) {
    if (!y || !wk || !w1t || !b1 || !hid || frames <= 0 || P <= 0 || nslice <= 0) return MV2_EINVAL;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if (C <= 0 || C % kvec != 0 || hidden <= 0 || hidden % nslice != 0) return MV2_EINVAL;
    const int hs = hidden / nslice;
    if (!is_pow2(hs) || hs > 1024) return MV2_EINVAL;
    const int nch = C / kvec;
    const int G = nch >= 64 ? 64 : nch;
    if (!is_pow2(G) || (nch > 64 && nch % 64 != 0)) return MV2_EINVAL;
    const int NV = nch / G;
    if ((NV != 1 && NV != 2 && NV != 4) || 1024 / G > 64) return MV2_EINVAL;
    const size_t lds = ((size_t)(1024 / G) * (C + 2) + 64 + C + 1024) * sizeof(float);
    if (lds > 64 * 1024) return MV2_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nslice, (unsigned)frames);
#define SE_PF(EE, NVV) hipLaunchKernelGGL((se_pool_fc1_kernel<EE, NVV>), grid, dim3(1024), lds, st, y, wk, bk, w1t, b1, hid, P, C, G, hidden, hs)
    if (dtype == MV2_BF16) { if (NV == 1) SE_PF(BF16, 1); else if (NV == 2) SE_PF(BF16, 2); else SE_PF(BF16, 4); }
    else { if (NV == 1) SE_PF(F32, 1); else if (NV == 2) SE_PF(F32, 2); else SE_PF(F32, 4); }
#undef SE_PF
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_se_fc2_apply(int dtype, const void* y, const float* hid, const float* w2t, const float* b2,
                                const void* residual, void* out, int frames, int P, int C, int hidden, int nslice,
                                void* stream) {
    if (!y || !hid || !w2t || !b2 || !out || frames <= 0 || P <= 0 || nslice <= 0) return MV2_EINVAL;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if (C <= 0 || C % nslice != 0 || hidden <= 0 || hidden > 4096) return MV2_EINVAL;
    const int cw = C / nslice;
    if (!is_pow2(cw) || cw > 1024 || cw < kvec) return MV2_EINVAL;
    if ((int64_t)P * (cw / kvec) >= (int64_t)0x7fffffff) return MV2_ETOOLARGE;
    const size_t lds = (size_t)(hidden + 1024 + cw) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nslice, (unsigned)frames);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((se_fc2_apply_kernel<BF16>), grid, dim3(1024), lds, st, y, hid, w2t, b2, residual, out, P, C, hidden, cw);
    else hipLaunchKernelGGL((se_fc2_apply_kernel<F32>), grid, dim3(1024), lds, st, y, hid, w2t, b2, residual, out, P, C, hidden, cw);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- SqueezeExcite: apply
template <typename E>
__global__ __launch_bounds__(256) void se_apply_kernel(const void* __restrict__ y, const float* __restrict__ gate,
                                                       const void* __restrict__ res, void* __restrict__ out, int64_t nchunks,
                                                       int P, int C) {
    const int cpr = C / E::kVec;   // chunks per row
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t row, f;
        if (nchunks < (int64_t)0x7fffffff) { const uint32_t r32 = (uint32_t)i / (uint32_t)cpr; row = r32; f = r32 / (uint32_t)P; }
        else { row = i / cpr; f = row / P; }
        const int c0 = (int)(i - row * cpr) * E::kVec;
        const i32x4 yv = ((const i32x4*)y)[i];
        float a[8], r[8];
        Chunk<E>::unpack(yv, a);
        const float* g = gate + f * C + c0;
        if (res) { const i32x4 rv = ((const i32x4*)res)[i]; Chunk<E>::unpack(rv, r); }
#pragma unroll
        for (int k = 0; k < E::kVec; ++k) a[k] = a[k] * g[k] + (res ? r[k] : 0.f);
        ((i32x4*)out)[i] = Chunk<E>::pack(a);
    }
}

extern "C" int mv2_se_apply(int dtype, const void* y, const float* gate, const void* residual, void* out, int frames,
                            int P, int C, void* stream) {
    if (!y || !gate || !out || frames <= 0 || P <= 0) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if ((dtype != MV2_BF16 && dtype != MV2_F32) || C % kvec != 0) return MV2_EINVAL;
    const int64_t nchunks = (int64_t)frames * P * (C / kvec);
    int64_t blocks = (nchunks + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV2_BF16) hipLaunchKernelGGL((se_apply_kernel<BF16>), dim3((unsigned)blocks), dim3(256), 0, st, y, gate, residual, out, nchunks, P, C);
    else hipLaunchKernelGGL((se_apply_kernel<F32>), dim3((unsigned)blocks), dim3(256), 0, st, y, gate, residual, out, nchunks, P, C);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- timing helpers
extern "C" void* mv2_event_create(void) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; return (void*)e; }
extern "C" int mv2_event_record(void* ev, void* stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
extern "C" int mv2_event_sync(void* ev) { return (int)hipEventSynchronize((hipEvent_t)ev); }
extern "C" float mv2_event_elapsed_ms(void* a, void* b) { float ms = -1.f; if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.f; return ms; }
extern "C" void mv2_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }

// --------------------------------------------------------------------------- conditioning (reference M:278-323, 715-753, 975-988, 1341-1353)
// The conditioning vector is per SAMPLE, so everything derived from it is a (B, n) matrix of a few KB.
//   mv2_cond_linear : out[b][o] = f(sum_i W[o][i] g(in[b][i]) + bias[o]), one wave per output.
//       mode 0  plain Linear (to_gamma, to_cond);  mode 1  Linear + SiLU (the cond_in stems);
//       mode 2  Conv3DMod's demodulation: g(x) = (x + 1)^2, no bias, f(s) = rsqrt(max(s, eps)) with W = sum over taps of w^2.
//   mv2_scale_channels : out[row][c] = act(x[row][c] * (scale[b(row)][c] + add)) — AdaptiveRMSNorm's gamma / Conv3DMod's
//       modulation on the input side (add = 0 / 1), the demodulation + ELU on the output side.
__global__ __launch_bounds__(256) void cond_linear_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int B, int I, int O,
                                                          int mode, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (int64_t)B * O) return;
    const int b = (int)(idx / O), o = (int)(idx - (int64_t)b * O);
    float s = 0.f;
    for (int i = lane; i < I; i += 64) {
        float x = in[(int64_t)b * I + i];
        if (mode == 2) { x += 1.f; x *= x; }
        s += w[(int64_t)o * I + i] * x;
    }
    s = wave_sum(s);
    if (lane == 0) {
        if (bias && mode != 2) s += bias[o];
        if (mode == 1) s = s / (1.f + expf(-s));
        if (mode == 2) s = rsqrtf(fmaxf(s, eps));
        out[idx] = s;
    }
}

extern "C" int mv2_cond_linear(const float* in, const float* w, const float* bias, float* out, int B, int I, int O, int mode, float eps,
                               void* stream) {
    if (!in || !w || !out || B <= 0 || I <= 0 || O <= 0 || mode < 0 || mode > 2) return MV2_EINVAL;
    const int64_t n = (int64_t)B * O;
    hipLaunchKernelGGL(cond_linear_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, w, bias, out, B, I, O, mode, eps);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

template <typename E>
__global__ __launch_bounds__(256) void scale_channels_kernel(const void* __restrict__ x, const float* __restrict__ scale, void* __restrict__ out,
                                                             int64_t nchunks, int64_t rows_per_sample, int C, float add, int act) {
    constexpr int KV = E::kVec;
    const int cpr = C / KV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / cpr;
        const int ch = (int)(i - row * cpr);
        const float* sc = scale + (row / rows_per_sample) * C + ch * KV;
        float a[8];
        Chunk<E>::unpack(((const i32x4*)x)[i], a);
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            float v = a[k] * (sc[k] + add);
            if (act == MV2_ACT_ELU) v = Act<E>::elu(v);
            a[k] = v;
        }
        ((i32x4*)out)[i] = Chunk<E>::pack(a);
    }
}

extern "C" int mv2_scale_channels(int dtype, const void* x, const float* scale, void* out, int64_t rows, int64_t rows_per_sample, int C,
                                  float add, int act, void* stream) {
    if (!x || !scale || !out || rows <= 0 || rows_per_sample <= 0 || rows % rows_per_sample != 0) return MV2_EINVAL;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    if (act != MV2_ACT_NONE && act != MV2_ACT_ELU) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if (C <= 0 || C % kvec != 0) return MV2_EINVAL;
    const int64_t nchunks = rows * (C / kvec);
    const unsigned blocks = (unsigned)std::min<int64_t>((nchunks + 255) / 256, 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV2_BF16) hipLaunchKernelGGL((scale_channels_kernel<BF16>), dim3(blocks), dim3(256), 0, st, x, scale, out, nchunks, rows_per_sample, C, add, act);
    else hipLaunchKernelGGL((scale_channels_kernel<F32>), dim3(blocks), dim3(256), 0, st, x, scale, out, nchunks, rows_per_sample, C, add, act);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- frame ranges (separate_first_frame_encoding, reference M:1553-1561, 1633-1639)
// dst[b][dst_t0 + t][:] = src ? src[b][src_t0 + t][:] : 0   for t < nT;  a frame is frame_bytes (multiple of 16) contiguous bytes.
// Packs [zero padding | first-frame conv | conv_in of the remaining frames] into one latent tensor and slices it again before conv_out.
__global__ __launch_bounds__(256) void copy_frames_kernel(const i32x4* __restrict__ src, i32x4* __restrict__ dst, int64_t chunks_per_frame,
                                                          int64_t total, int src_T, int src_t0, int dst_T, int dst_t0, int nT) {
    const int64_t per_sample = chunks_per_frame * nT;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_sample, r = i - b * per_sample;
        const i32x4 zero = {0, 0, 0, 0};
        dst[(b * dst_T + dst_t0) * chunks_per_frame + r] = src ? src[(b * src_T + src_t0) * chunks_per_frame + r] : zero;
    }
}

extern "C" int mv2_copy_frames(const void* src, int src_T, int src_t0, void* dst, int dst_T, int dst_t0, int nT, int B, int64_t frame_bytes,
                               void* stream) {
    if (!dst || B <= 0 || nT <= 0 || frame_bytes <= 0 || frame_bytes % 16 != 0) return MV2_EINVAL;
    if (dst_t0 < 0 || dst_t0 + nT > dst_T || (src && (src_t0 < 0 || src_t0 + nT > src_T))) return MV2_EINVAL;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15)) return MV2_EINVAL;
    const int64_t cpf = frame_bytes / 16, total = cpf * nT * B;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(copy_frames_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const i32x4*)src, (i32x4*)dst, cpf, total, src_T, src_t0,
                       dst_T, dst_t0, nT);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- non-constant padding (CausalConv3d pad_mode, reference M:899-927;
// Conv3DMod M:750): F.pad(x, (pw, pw, ph, ph, pt, 0), mode) on an NDHWC tensor as an explicit gather, so that the convolution that
// follows runs 'valid' on the padded image.  mode 1 = reflect (j < 0 -> -j, j >= n -> 2(n - 1) - j), 2 = replicate (clamp),
// 3 = circular (j mod n); time is padded in front only.  A row is row_bytes (multiple of 8) contiguous bytes; `tail_bytes` zero bytes
// follow the image (the conv_in gathers read 16 bytes at 8-byte alignment and may run past the last pixel).
__device__ __forceinline__ int pad_src(int j, int n, int mode) {
    if (mode == 1) { j = j < 0 ? -j : j; return j >= n ? 2 * (n - 1) - j : j; }
    if (mode == 2) return min(max(j, 0), n - 1);
    j %= n; return j < 0 ? j + n : j;
}
__global__ __launch_bounds__(256) void pad_ndhwc_kernel(const i32x2* __restrict__ in, i32x2* __restrict__ out, int B, int T, int H, int W, int w8,
                                                        int pt, int ph, int pw, int mode, int64_t body, int64_t total) {
    const int To = T + pt, Ho = H + 2 * ph, Wo = W + 2 * pw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        i32x2 v = {0, 0};
        if (i < body) {
            int64_t r = i / w8; const int c8 = (int)(i - r * w8);
            const int wo = (int)(r % Wo); r /= Wo;
            const int ho = (int)(r % Ho); r /= Ho;
            const int to = (int)(r % To); const int64_t b = r / To;
            const int ti = pad_src(to - pt, T, mode), hi = pad_src(ho - ph, H, mode), wi = pad_src(wo - pw, W, mode);
            v = in[((((b * T + ti) * H + hi) * (int64_t)W + wi)) * w8 + c8];
        }
        out[i] = v;
    }
}

extern "C" int mv2_pad_ndhwc(const void* in, void* out, int B, int T, int H, int W, int64_t row_bytes, int pt, int ph, int pw, int mode,
                             int64_t tail_bytes, void* stream) {
    if (!in || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0 || row_bytes <= 0 || row_bytes % 8 != 0 || tail_bytes < 0 || tail_bytes % 8 != 0) return MV2_EINVAL;
    if (pt < 0 || ph < 0 || pw < 0 || mode < 1 || mode > 3) return MV2_EINVAL;
    if (mode == 1 && (pt >= T || ph >= H || pw >= W)) return MV2_EINVAL;          // reflect needs padding < size (as F.pad)
    if (mode == 3 && (pt > T || ph > H || pw > W)) return MV2_EINVAL;             // circular: padding <= size
    if ((((uintptr_t)in | (uintptr_t)out) & 7)) return MV2_EINVAL;
    const int64_t w8 = row_bytes / 8;
    const int64_t body = (int64_t)B * (T + pt) * (H + 2 * ph) * (W + 2 * pw) * w8, total = body + tail_bytes / 8;
    if (w8 > 0x7fffffff) return MV2_ETOOLARGE;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(pad_ndhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const i32x2*)in, (i32x2*)out, B, T, H, W, (int)w8, pt, ph, pw,
                       mode, body, total);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// --------------------------------------------------------------------------- gate loop along time ('gateloop_time', reference M:1216-1222 ->
// third-party gateloop_transformer.SimpleGateLoopLayer; RECONSTRUCTION, parity unpinned: the package is absent from /root/reference)
// qkva rows (B, T, P) x 3C = [q | kv | a] of Linear(dim, 3 dim) on the RMS-normed tokens; per pixel and channel, along time:
//     h_t = sigmoid(a_t) * h_{t-1} + kv_t   (h_{-1} = 0),     out_t = q_t * h_t + residual_t          (Residual, M:167-174)
// One thread owns one 16-byte channel chunk of one pixel and walks its T steps; consecutive threads read consecutive chunks.
template <typename E>
__global__ __launch_bounds__(256) void gateloop_scan_kernel(const void* __restrict__ qkva, const void* __restrict__ res, void* __restrict__ out,
                                                            int64_t nseq, int T, int64_t P, int C) {
    constexpr int KV = E::kVec;
    const int cpr = C / KV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nseq * cpr; i += (int64_t)gridDim.x * 256) {
        const int64_t seq = i / cpr;                 // (b, p)
        const int ch = (int)(i - seq * cpr);
        const int64_t b = seq / P, p = seq - b * P;
        float h[8];
#pragma unroll
        for (int k = 0; k < KV; ++k) h[k] = 0.f;
        for (int t = 0; t < T; ++t) {
            const int64_t row = (b * T + t) * P + p;
            const i32x4* qr = (const i32x4*)qkva + row * (3 * cpr);
            float q[8], kv[8], a[8], r[8];
            Chunk<E>::unpack(qr[ch], q);
            Chunk<E>::unpack(qr[cpr + ch], kv);
            Chunk<E>::unpack(qr[2 * cpr + ch], a);
            Chunk<E>::unpack(((const i32x4*)res)[row * cpr + ch], r);
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const float g = 1.f / (1.f + expf(-a[k]));
                h[k] = g * h[k] + kv[k];
                r[k] += q[k] * h[k];
            }
            ((i32x4*)out)[row * cpr + ch] = Chunk<E>::pack(r);
        }
    }
}

extern "C" int mv2_gateloop_scan(int dtype, const void* qkva, const void* residual, void* out, int B, int T, int64_t P, int C, void* stream) {
    if (!qkva || !residual || !out || B <= 0 || T <= 0 || P <= 0) return MV2_EINVAL;
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if (C <= 0 || C % kvec != 0) return MV2_EINVAL;
    const int64_t nseq = (int64_t)B * P, total = nseq * (C / kvec);
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV2_BF16) hipLaunchKernelGGL((gateloop_scan_kernel<BF16>), dim3(blocks), dim3(256), 0, st, qkva, residual, out, nseq, T, P, C);
    else hipLaunchKernelGGL((gateloop_scan_kernel<F32>), dim3(blocks), dim3(256), 0, st, qkva, residual, out, nseq, T, P, C);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
