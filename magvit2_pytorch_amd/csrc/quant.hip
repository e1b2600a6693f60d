// LFQ / FSQ quantizers: one wavefront per token, per-token reductions by wave shuffles.
// Restates the eval arithmetic of vector_quantize_pytorch.LFQ / FSQ (see oracle/thirdparty.py for
// the algorithm statement; reference call sites magvit2_pytorch.py:1574, 1593, 1700, 1705).
#include <algorithm>
#include "common.h"

#define CB_MAX 32

struct QuantParams {
    const void* x; const float* w_in; const float* b_in; const float* w_out; const float* b_out;
    void* quantized; void* ids; float* prequant;
    int64_t rows; int C, cb_dim, ncb;
    float clamp, scale;
    int ste;                    // LFQ training mode: the forward value is the straight-through form z + (q - z)
    // FSQ constants (per codebook dim, length cb_dim)
    const float* shift; const float* half_l; const float* offset; const float* half_w; const int32_t* basis;
};

// Per-token state lives in LDS (one 32-float row per wave and quantity), one code dim per lane: the first
// version kept z / code / level in fully unrolled 32-entry register arrays and the compiler hoisted all weight
// loads of the unrolled loops, which spilled (fsq: 29 k spill slots).  Loops over code dims are runtime loops
// here; the float operation order per output is unchanged (results are bit-identical to the unrolled form).
struct WaveQ { float z[CB_MAX]; float code[CB_MAX]; float lvl[CB_MAX]; };

// z[i] = b_in[i] + sum_c W_in[i][c] x[c]   (wave-reduced) -> q.z[i]
template <typename E>
__device__ __forceinline__ void project_in(const QuantParams& p, int64_t row, int lane, int cbt, WaveQ& q) {
    const int nch = p.C / E::kVec;
    for (int i0 = 0; i0 < cbt; i0 += 4) {           // 4 code dims per trip: their weight loads and reductions overlap
        float zi[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ch = lane; ch < nch; ch += 64) {
            float xv[8];
            Chunk<E>::unpack(*(const i32x4*)((const char*)p.x + (row * p.C + (int64_t)ch * E::kVec) * E::kBytes), xv);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u < cbt) {
                    const float* w = p.w_in + (int64_t)(i0 + u) * p.C + ch * E::kVec;
#pragma unroll
                    for (int k = 0; k < E::kVec; k += 4) {
                        const f32x4 wv = *(const f32x4*)(w + k);
                        zi[u] += xv[k] * wv[0]; zi[u] += xv[k + 1] * wv[1]; zi[u] += xv[k + 2] * wv[2]; zi[u] += xv[k + 3] * wv[3];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) zi[u] = wave_sum(zi[u]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + u < cbt) q.z[i0 + u] = zi[u] + p.b_in[i0 + u];
        }
    }
}

// out[c] = b_out[c] + sum_i W_out[c][i] code[i].  The weight arrives TRANSPOSED, w_out_t[i][c] (cbt x C): a lane
// reads its kVec consecutive channels of one code dim with 16-byte loads (the (C, cbt) layout made every lane of
// every load touch its own cache line).
template <typename E>
__device__ __forceinline__ void project_out(const float* w_out_t, const float* b_out, void* out, int64_t row, int C, int lane,
                                            int cbt, const float* code) {
    const int nch = C / E::kVec;
    for (int ch = lane; ch < nch; ch += 64) {
        float o[8];
#pragma unroll
        for (int k = 0; k < E::kVec; k += 4) {
            const f32x4 b = *(const f32x4*)(b_out + ch * E::kVec + k);
            o[k] = b[0]; o[k + 1] = b[1]; o[k + 2] = b[2]; o[k + 3] = b[3];
        }
#pragma unroll 2
        for (int i = 0; i < cbt; ++i) {
            const float ci = code[i];                        // LDS broadcast
            const float* w = w_out_t + (int64_t)i * C + ch * E::kVec;
#pragma unroll
            for (int k = 0; k < E::kVec; k += 4) {
                const f32x4 wv = *(const f32x4*)(w + k);
                o[k] += wv[0] * ci; o[k + 1] += wv[1] * ci; o[k + 2] += wv[2] * ci; o[k + 3] += wv[3] * ci;
            }
        }
        *(i32x4*)((char*)out + (row * C + (int64_t)ch * E::kVec) * E::kBytes) = Chunk<E>::pack(o);
    }
}

template <typename E>
__global__ __launch_bounds__(256) void lfq_kernel(const QuantParams p) {
    __shared__ WaveQ sq[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= p.rows) return;                      // whole waves leave: no block-level barrier below
    WaveQ& q = sq[wave];
    const int cbt = p.cb_dim * p.ncb;
    project_in<E>(p, row, lane, cbt, q);
    __builtin_amdgcn_wave_barrier();
    if (lane < cbt) {                               // one code dim per lane
        float z = q.z[lane];
        if (p.clamp > 0.f) z = tanhf(z / p.clamp) * p.clamp;
        q.z[lane] = z;
        const float sgn = z > 0.f ? p.scale : -p.scale;
        q.code[lane] = p.ste ? z + (sgn - z) : sgn;     // training: x + (quantized - x).detach(), the same two fp32 roundings
        if (p.prequant) p.prequant[row * cbt + lane] = z;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < p.ncb) {                             // one codebook per lane: id = sum_i [z_i > 0] << (cb_dim-1-i)
        int64_t id = 0;
        for (int li = 0; li < p.cb_dim; ++li)
            if (q.z[lane * p.cb_dim + li] > 0.f) id |= (int64_t)1 << (p.cb_dim - 1 - li);
        ((int64_t*)p.ids)[row * p.ncb + lane] = id;
    }
    project_out<E>(p.w_out, p.b_out, p.quantized, row, p.C, lane, cbt, q.code);
}

template <typename E>
__global__ __launch_bounds__(256) void fsq_kernel(const QuantParams p) {
    __shared__ WaveQ sq[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    WaveQ& q = sq[wave];
    const int cbt = p.cb_dim * p.ncb;
    project_in<E>(p, row, lane, cbt, q);
    __builtin_amdgcn_wave_barrier();
    if (lane < cbt) {
        const int li = lane % p.cb_dim;
        const float bounded = tanhf(q.z[lane] + p.shift[li]) * p.half_l[li] - p.offset[li];
        const float r = rintf(bounded);                  // round half to even, as torch.round
        const float code = r / p.half_w[li];
        q.z[lane] = bounded;
        q.code[lane] = code;
        q.lvl[lane] = (code * p.half_w[li] + p.half_w[li]) * (float)p.basis[li];
        if (p.prequant) p.prequant[row * cbt + lane] = bounded;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < p.ncb) {
        float s = 0.f;
        for (int li = 0; li < p.cb_dim; ++li) s += q.lvl[lane * p.cb_dim + li];
        ((int32_t*)p.ids)[row * p.ncb + lane] = (int32_t)s;
    }
    project_out<E>(p.w_out, p.b_out, p.quantized, row, p.C, lane, cbt, q.code);
}

struct CodesParams {
    const void* ids; const float* w_out; const float* b_out; void* out;
    int64_t rows; int C, cb_dim, ncb; float scale;
    const int32_t* levels; const float* half_w; const int32_t* basis;
};

template <typename E, bool FSQ>
__global__ __launch_bounds__(256) void codes_kernel(const CodesParams p) {
    __shared__ float scode[4][CB_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    const int cbt = p.cb_dim * p.ncb;
    if (lane < cbt) {
        const int cbk = lane / p.cb_dim, li = lane % p.cb_dim;
        float code;
        if (FSQ) {
            const int32_t id = ((const int32_t*)p.ids)[row * p.ncb + cbk];
            const int lv = (id / p.basis[li]) % p.levels[li];
            code = ((float)lv - p.half_w[li]) / p.half_w[li];
        } else {
            const int64_t id = ((const int64_t*)p.ids)[row * p.ncb + cbk];
            code = ((id >> (p.cb_dim - 1 - li)) & 1) ? p.scale : -p.scale;
        }
        scode[wave][lane] = code;
    }
    __builtin_amdgcn_wave_barrier();
    project_out<E>(p.w_out, p.b_out, p.out, row, p.C, lane, cbt, scode[wave]);
}

static int check_q(int dtype, int64_t rows, int C, int cb_dim, int ncb) {
    if (dtype != MV2_BF16 && dtype != MV2_F32) return MV2_EINVAL;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if (rows <= 0 || C % kvec != 0 || cb_dim <= 0 || ncb <= 0 || cb_dim * ncb > CB_MAX || cb_dim > 62) return MV2_EINVAL;
    return MV2_OK;
}

static int launch_lfq(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out, const float* b_out, void* quantized,
                      int64_t* ids, float* prequant, int64_t rows, int C, int cb_dim, int num_codebooks, float clamp, float scale, int ste,
                      void* stream) {
    QuantParams p{x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, C, cb_dim, num_codebooks, clamp, scale, ste,
                  nullptr, nullptr, nullptr, nullptr, nullptr};
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((lfq_kernel<BF16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((lfq_kernel<F32>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_lfq(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                       void* quantized, int64_t* ids, float* prequant, int64_t rows, int C, int cb_dim, int num_codebooks,
                       float clamp, float scale, void* stream) {
    if (!x || !w_in || !b_in || !w_out || !b_out || !quantized || !ids) return MV2_EINVAL;
    int rc = check_q(dtype, rows, C, cb_dim, num_codebooks); if (rc) return rc;
    return launch_lfq(dtype, x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, C, cb_dim, num_codebooks, clamp, scale, 0, stream);
}

extern "C" int mv2_lfq_train(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                             void* quantized, int64_t* ids, float* prequant, int64_t rows, int C, int cb_dim, int num_codebooks,
                             float clamp, float scale, void* stream) {
    if (!x || !w_in || !b_in || !w_out || !b_out || !quantized || !ids || !prequant) return MV2_EINVAL;
    int rc = check_q(dtype, rows, C, cb_dim, num_codebooks); if (rc) return rc;
    return launch_lfq(dtype, x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, C, cb_dim, num_codebooks, clamp, scale, 1, stream);
}

extern "C" int mv2_fsq(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                       const float* shift, const float* half_l, const float* offset, const float* half_w, const int32_t* basis,
                       void* quantized, int32_t* ids, float* prequant, int64_t rows, int C, int cb_dim, int num_codebooks,
                       void* stream) {
    if (!x || !w_in || !b_in || !w_out || !b_out || !quantized || !ids || !shift || !half_l || !offset || !half_w || !basis) return MV2_EINVAL;
    int rc = check_q(dtype, rows, C, cb_dim, num_codebooks); if (rc) return rc;
    QuantParams p{x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, C, cb_dim, num_codebooks, 0.f, 1.f, 0,
                  shift, half_l, offset, half_w, basis};
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((fsq_kernel<BF16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((fsq_kernel<F32>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_lfq_codes(int dtype, const int64_t* ids, const float* w_out, const float* b_out, void* out, int64_t rows,
                             int C, int cb_dim, int num_codebooks, float scale, void* stream) {
    if (!ids || !w_out || !b_out || !out) return MV2_EINVAL;
    int rc = check_q(dtype, rows, C, cb_dim, num_codebooks); if (rc) return rc;
    CodesParams p{ids, w_out, b_out, out, rows, C, cb_dim, num_codebooks, scale, nullptr, nullptr, nullptr};
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((codes_kernel<BF16, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((codes_kernel<F32, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_fsq_codes(int dtype, const int32_t* ids, const float* w_out, const float* b_out, const int32_t* levels,
                             const float* half_w, const int32_t* basis, void* out, int64_t rows, int C, int cb_dim,
                             int num_codebooks, void* stream) {
    if (!ids || !w_out || !b_out || !out || !levels || !half_w || !basis) return MV2_EINVAL;
    int rc = check_q(dtype, rows, C, cb_dim, num_codebooks); if (rc) return rc;
    CodesParams p{ids, w_out, b_out, out, rows, C, cb_dim, num_codebooks, 1.f, levels, half_w, basis};
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((codes_kernel<BF16, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((codes_kernel<F32, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}


// ------------------------------------------------------------------------------- training-mode LFQ statistics and losses
// Third-party LFQ.forward, training branch (reconstruction, see oracle/thirdparty.py; reference call site magvit2_pytorch.py:1705,
// weights :1067-1072): per token and codebook  prob = softmax_j(2 * inv_temperature * z . c_j)  over all K = 2^d sign patterns c_j,
//   per-sample entropy  H(prob) with the log clamped at 1e-5, averaged over tokens and codebooks;
//   batch entropy       H(mean over tokens of prob), averaged over codebooks (the mean is all-reduced across ranks);
//   commitment          mean((z - sign(z) * scale)^2);
//   aux = (H_sample - gamma * H_batch) * w_entropy + commitment * w_commit.
// Determinism: every reduction has a fixed order (wave shuffles, then waves in order, then workgroups in order).
#define LFQ_KMAX_LOG2 12
#define LFQ_CPT 16                        // codes per thread at K = 4096

struct LfqStatsParams { const float* z; int64_t rows; int cb_dim, ncb; float inv_temp, scale; float* part; int64_t tok_per_block; };

__device__ __forceinline__ float block_sum_256(float v, float* red) {      // fixed order; red: 8 floats of LDS; ends with a barrier
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void lfq_stats_kernel(const LfqStatsParams p) {
    __shared__ float red[8];
    const int tid = threadIdx.x;
    const int d = p.cb_dim, K = 1 << d, cbt = d * p.ncb;
    const int64_t t0 = (int64_t)blockIdx.x * p.tok_per_block, t1 = min(p.rows, t0 + p.tok_per_block);
    float* out = p.part + (size_t)blockIdx.x * ((size_t)p.ncb * K + 2);
    float ent_sum = 0.f, commit_sum = 0.f;
    for (int c = 0; c < p.ncb; ++c) {
        float accp[LFQ_CPT];
#pragma unroll
        for (int u = 0; u < LFQ_CPT; ++u) accp[u] = 0.f;
        for (int64_t tok = t0; tok < t1; ++tok) {
            const float* zt = p.z + tok * cbt + c * d;
            float zz[LFQ_KMAX_LOG2];
            float mx = 0.f, cm = 0.f;
#pragma unroll
            for (int i = 0; i < LFQ_KMAX_LOG2; ++i) {
                zz[i] = i < d ? zt[i] : 0.f;
                mx += fabsf(zz[i]);
                const float q = zz[i] > 0.f ? p.scale : -p.scale;
                if (i < d) cm += (zz[i] - q) * (zz[i] - q);
            }
            const float k2 = 2.f * p.inv_temp * p.scale;
            mx *= k2;                                                      // logit of the matching code = the maximum
            float e[LFQ_CPT], s_local = 0.f;
#pragma unroll
            for (int u = 0; u < LFQ_CPT; ++u) {
                const int j = tid + 256 * u;
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < LFQ_KMAX_LOG2; ++i) dot += (i < d) ? (((j >> (d - 1 - i)) & 1) ? zz[i] : -zz[i]) : 0.f;     // MSB first
                e[u] = j < K ? expf(k2 * dot - mx) : 0.f;
                s_local += e[u];
            }
            const float S = block_sum_256(s_local, red);
            float h_local = 0.f;
#pragma unroll
            for (int u = 0; u < LFQ_CPT; ++u) {
                const float pr = e[u] / S;
                accp[u] += pr;
                h_local += -pr * logf(fmaxf(pr, 1e-5f));
            }
            const float H = block_sum_256(h_local, red);
            ent_sum += H;
            commit_sum += cm;
        }
#pragma unroll
        for (int u = 0; u < LFQ_CPT; ++u) { const int j = tid + 256 * u; if (j < K) out[(size_t)c * K + j] = accp[u]; }
    }
    if (tid == 0) { out[(size_t)p.ncb * K] = ent_sum; out[(size_t)p.ncb * K + 1] = commit_sum; }
}

// out[col] = sum over blocks (in order) of part[blk][col]
__global__ __launch_bounds__(256) void sum_blocks_kernel(const float* part, int nblk, int64_t ncols, float* out) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * ncols + col];
    out[col] = s;
}

__global__ __launch_bounds__(256) void lfq_loss_kernel(const float* sums, double n_tokens, int cb_dim, int ncb, float gamma, float w_ent,
                                                       float w_commit, float* losses) {
    __shared__ float red[8];
    const int K = 1 << cb_dim;
    const float inv_n = (float)(1.0 / n_tokens);
    float hb = 0.f;
    for (int c = 0; c < ncb; ++c) {
        float h_local = 0.f;
        for (int j = threadIdx.x; j < K; j += 256) { const float a = sums[(size_t)c * K + j] * inv_n; h_local += -a * logf(fmaxf(a, 1e-5f)); }
        hb += block_sum_256(h_local, red);
    }
    if (threadIdx.x == 0) {
        const float per_sample = sums[(size_t)ncb * K] * inv_n / (float)ncb;
        const float batch = hb / (float)ncb;
        const float commit = w_commit > 0.f ? sums[(size_t)ncb * K + 1] * inv_n / (float)(ncb * cb_dim) : 0.f;
        losses[0] = per_sample; losses[1] = batch; losses[2] = commit;
        losses[3] = (per_sample - gamma * batch) * w_ent + commit * w_commit;
    }
}

static int lfq_stats_blocks(int64_t rows) { return (int)std::min<int64_t>(rows, 512); }

extern "C" size_t mv2_lfq_stats_workspace_bytes(int64_t rows, int cb_dim, int num_codebooks) {
    if (rows <= 0 || cb_dim <= 0 || cb_dim > LFQ_KMAX_LOG2 || num_codebooks <= 0) return 0;
    return (size_t)lfq_stats_blocks(rows) * ((size_t)num_codebooks * ((size_t)1 << cb_dim) + 2) * sizeof(float);
}

extern "C" int mv2_lfq_stats(const float* prequant, int64_t rows, int cb_dim, int num_codebooks, float inv_temperature, float scale,
                             float* sums, void* workspace, size_t workspace_bytes, void* stream) {
    if (!prequant || !sums || rows <= 0 || cb_dim <= 0 || cb_dim > LFQ_KMAX_LOG2 || num_codebooks <= 0 || cb_dim * num_codebooks > CB_MAX) return MV2_EINVAL;
    if (!workspace || workspace_bytes < mv2_lfq_stats_workspace_bytes(rows, cb_dim, num_codebooks)) return MV2_EWORKSPACE;
    const int nblk = lfq_stats_blocks(rows);
    const int64_t ncols = (int64_t)num_codebooks * ((int64_t)1 << cb_dim) + 2;
    LfqStatsParams p{prequant, rows, cb_dim, num_codebooks, inv_temperature, scale, (float*)workspace, (rows + nblk - 1) / nblk};
    hipLaunchKernelGGL(lfq_stats_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    MV2_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_blocks_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, ncols, sums);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_lfq_losses(const float* sums, double n_tokens, int cb_dim, int num_codebooks, float diversity_gamma, float entropy_loss_weight,
                              float commitment_loss_weight, float* losses, void* stream) {
    if (!sums || !losses || !(n_tokens > 0) || cb_dim <= 0 || cb_dim > LFQ_KMAX_LOG2 || num_codebooks <= 0) return MV2_EINVAL;
    hipLaunchKernelGGL(lfq_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, n_tokens, cb_dim, num_codebooks, diversity_gamma,
                       entropy_loss_weight, commitment_loss_weight, losses);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// ------------------------------------------------------------------------------- sum of squared differences (F.mse_loss, M:1722)
__global__ __launch_bounds__(256) void sq_diff_kernel(const float* a, const float* b, int64_t n, int64_t per_block, float* part) {
    __shared__ float red[8];
    const int64_t i0 = (int64_t)blockIdx.x * per_block, i1 = min(n, i0 + per_block);
    float s = 0.f;
    for (int64_t i = i0 + threadIdx.x * 4; i < i1; i += 256 * 4) {
        if (i + 4 <= i1) {
            const f32x4 x = *(const f32x4*)(a + i), y = *(const f32x4*)(b + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dlt = x[j] - y[j]; s += dlt * dlt; }
        } else {
            for (int64_t k = i; k < i1; ++k) { const float dlt = a[k] - b[k]; s += dlt * dlt; }
        }
    }
    const float t = block_sum_256(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

static int sq_blocks(int64_t n) { return (int)std::min<int64_t>((n + 4095) / 4096, 1024); }

extern "C" size_t mv2_sq_diff_workspace_bytes(int64_t n) { return n > 0 ? (size_t)sq_blocks(n) * sizeof(float) : 0; }

extern "C" int mv2_sq_diff_sum(const float* a, const float* b, int64_t n, float* sum, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a || !b || !sum || n <= 0 || (((uintptr_t)a | (uintptr_t)b) & 15)) return MV2_EINVAL;
    if (!workspace || workspace_bytes < mv2_sq_diff_workspace_bytes(n)) return MV2_EWORKSPACE;
    const int nblk = sq_blocks(n);
    int64_t per_block = (n + nblk - 1) / nblk;
    per_block = (per_block + 3) / 4 * 4;                                  // 16-byte aligned chunk starts
    hipLaunchKernelGGL(sq_diff_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a, b, n, per_block, (float*)workspace);
    MV2_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, (int64_t)1, sum);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// ------------------------------------------------------------------------------- hinge discriminator loss (reference M:120-121)
// out[0] = mean(relu(1 + fake) + relu(1 - real)) over n logits; one workgroup, fixed summation order.
__global__ __launch_bounds__(256) void hinge_discr_kernel(const float* __restrict__ fake, const float* __restrict__ real, int n, float* out) {
    __shared__ float red[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += fmaxf(1.f + fake[i], 0.f) + fmaxf(1.f - real[i], 0.f);
    const float t = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = t / (float)n;
}

extern "C" int mv2_hinge_discr_loss(const float* fake, const float* real, int n, float* out, void* stream) {
    if (!fake || !real || !out || n <= 0) return MV2_EINVAL;
    hipLaunchKernelGGL(hinge_discr_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fake, real, n, out);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
