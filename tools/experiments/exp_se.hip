// EXPERIMENTS LIBRARY ONLY (build.py SOURCES_EXP; not part of libmv2_hip.so): the large-frame SqueezeExcite tail in ONE launch.
// mv2_se_gate (one 1024-thread block per frame: merge the pooling records, fc1, fc2, sigmoid; 8.5 us per launch at 20 frames, a pure latency
// chain) + mv2_se_apply (HBM stream, 11.6 us) are two of the three launches behind every ResidualUnit's 1x1x1 conv at the 128^2 .. 32^2
// levels.  Here every block of a frame repeats the gate computation (se_gate_kernel's body VERBATIM - same thread mapping, same sums, so the
// gate and therefore the output are bit-identical) into LDS and then streams its share of the frame.  The redundant part reads the frame's
// records and the two small weight matrices from L2 (~100 KB per block at C = 128).  Sized in DESIGN.md section 9; to be measured with
// tools/se_fuse_bench.cpp before anything is lowered to it.  Reference: SqueezeExcite.forward magvit2_pytorch.py:226-240, Residual :174.
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/mv2.h"

template <typename E>
__global__ __launch_bounds__(1024) void se_gate_apply_kernel(const float* __restrict__ part, const float* __restrict__ w1t,
                                                             const float* __restrict__ b1, const float* __restrict__ w2t,
                                                             const float* __restrict__ b2, const void* __restrict__ y,
                                                             const void* __restrict__ res, void* __restrict__ out, int P, int C,
                                                             int hidden, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pooled = sm;               // C
    float* hid = pooled + C;          // hidden
    float* wj = hid + hidden;         // nchunk: exp(m_j - M)
    float* red = wj + nchunk;         // 32 scratch
    float* slab = red + 32;           // 1024 partial sums
    float* gate = slab + 1024;        // C: this frame's gate (se_gate_kernel writes it to HBM)
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
            gate[tid] = 1.f / (1.f + expf(-a));
        }
    }
    __syncthreads();
    // ---- apply: this block's share of the frame's 16-byte chunks, out = gate * y + residual (se_apply_kernel's arithmetic)
    const int cpr = C / E::kVec;
    const int64_t per_frame = (int64_t)P * cpr;
    const int64_t share = (per_frame + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * share, i1 = i0 + share < per_frame ? i0 + share : per_frame;
    const int64_t base = (int64_t)f * per_frame;
    for (int64_t i = i0 + tid; i < i1; i += 1024) {
        const int c0 = (int)(i % cpr) * E::kVec;
        const i32x4 yv = ((const i32x4*)y)[base + i];
        float a[8], r[8];
        Chunk<E>::unpack(yv, a);
        if (res) { const i32x4 rv = ((const i32x4*)res)[base + i]; Chunk<E>::unpack(rv, r); }
#pragma unroll
        for (int k = 0; k < E::kVec; ++k) a[k] = a[k] * gate[c0 + k] + (res ? r[k] : 0.f);
        ((i32x4*)out)[base + i] = Chunk<E>::pack(a);
    }
}

extern "C" int mv2x_se_gate_apply(int dtype, const float* part, const float* w1t, const float* b1, const float* w2t, const float* b2,
                                  const void* y, const void* residual, void* out, int frames, int P, int C, int hidden, int nchunk,
                                  int blocks_per_frame, void* stream) {
    if (!part || !w1t || !b1 || !w2t || !b2 || !y || !out || frames <= 0 || nchunk <= 0 || P <= 0 || blocks_per_frame <= 0) return MV2_EINVAL;
    if (C <= 0 || C > 1024 || hidden <= 0 || hidden > 1024) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if ((dtype != MV2_BF16 && dtype != MV2_F32) || C % kvec != 0) return MV2_EINVAL;
    const size_t lds = (size_t)(C + hidden + nchunk + 32 + 1024 + C) * sizeof(float);
    if (lds > 64 * 1024) return MV2_EINVAL;
    dim3 grid((unsigned)blocks_per_frame, (unsigned)frames);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV2_BF16) hipLaunchKernelGGL((se_gate_apply_kernel<BF16>), grid, dim3(1024), lds, st, part, w1t, b1, w2t, b2, y, residual, out, P, C, hidden, nchunk);
    else hipLaunchKernelGGL((se_gate_apply_kernel<F32>), grid, dim3(1024), lds, st, part, w1t, b1, w2t, b2, y, residual, out, P, C, hidden, nchunk);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// ---- se_gate_kernel with its independent global loads hoisted in front of the first reduction (same sums, value for value): how much of the
// 8.5 us per launch is exposed L2 round trips?  tools/se_fuse_bench.cpp times it against mv2_se_gate and compares the gates bit for bit.
__global__ __launch_bounds__(1024) void se_gate_v2_kernel(const float* __restrict__ part, const float* __restrict__ w1t,
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
    // ---- v2: every global load whose ADDRESS does not depend on an earlier result is issued here, before the first reduction: the first
    // batch of record sums (needs only wj later), of w1t and of w2t.  The arithmetic below is se_gate_kernel's, value for value.
    const int JSp = max(1, 1024 / C), cp = tid % C, jsp = tid / C;
    float vpre[8];
    const bool have_v = jsp < JSp && jsp + 7 * JSp < nchunk;
#pragma unroll
    for (int u = 0; u < 8; ++u) vpre[u] = have_v ? pf[(int64_t)(jsp + u * JSp) * (C + 2) + 2 + cp] : 0.f;
    const int KSp = max(1, 1024 / hidden), kp = tid % hidden, csp = tid / hidden, perp = (C + KSp - 1) / KSp;
    const int c0p = csp * perp, c1p = min(C, c0p + perp);
    float w1pre[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w1pre[u] = (csp < KSp && c0p + u < c1p) ? w1t[(int64_t)(c0p + u) * hidden + kp] : 0.f;
    const int GSp = max(1, 1024 / C), c2p = tid % C, ksp = tid / C, per2p = (hidden + GSp - 1) / GSp;
    const int k0p = ksp * per2p, k1p = min(hidden, k0p + per2p);
    float w2pre[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w2pre[u] = (ksp < GSp && k0p + u < k1p) ? w2t[(int64_t)(k0p + u) * C + c2p] : 0.f;
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
            bool firstv = true;
            for (; j + 7 * JS < nchunk; j += 8 * JS) {             // eight loads in flight (the plain loop was a chain of L2 latencies)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = firstv ? vpre[u] : pf[(int64_t)(j + u * JS) * (C + 2) + 2 + c];
                firstv = false;
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
            bool first1 = true;
            for (; c + 16 <= c1; c += 16) {          // 16 independent loads in flight per thread
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = first1 ? w1pre[u] : w1t[(int64_t)(c + u) * hidden + k];
                first1 = false;
#pragma unroll
                for (int u = 0; u < 16; u += 2) { a0 += w[u] * pooled[c + u]; a1 += w[u + 1] * pooled[c + u + 1]; }
            }
            if (c < c1) {                                            // tail (< 16): still one batch of independent loads
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = first1 ? w1pre[u] : ((c + u < c1) ? w1t[(int64_t)(c + u) * hidden + k] : 0.f);
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
            bool first2 = true;
            for (; k + 16 <= k1; k += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = first2 ? w2pre[u] : w2t[(int64_t)(k + u) * C + c];
                first2 = false;
#pragma unroll
                for (int u = 0; u < 16; u += 2) { a0 += w[u] * hid[k + u]; a1 += w[u + 1] * hid[k + u + 1]; }
            }
            if (k < k1) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = first2 ? w2pre[u] : ((k + u < k1) ? w2t[(int64_t)(k + u) * C + c] : 0.f);
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


extern "C" int mv2x_se_gate_v2(const float* part, const float* w1t, const float* b1, const float* w2t, const float* b2,
                               float* gate, int frames, int C, int hidden, int nchunk, void* stream) {
    if (!part || !w1t || !b1 || !w2t || !b2 || !gate || frames <= 0 || nchunk <= 0) return MV2_EINVAL;
    if (C <= 0 || C > 1024 || hidden <= 0 || hidden > 1024) return MV2_EINVAL;
    const size_t lds = (size_t)(C + hidden + nchunk + 32 + 1024) * sizeof(float);
    if (lds > 64 * 1024) return MV2_EINVAL;
    hipLaunchKernelGGL(se_gate_v2_kernel, dim3((unsigned)frames), dim3(1024), lds, (hipStream_t)stream, part, w1t, b1, w2t, b2, gate, C, hidden, nchunk);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
