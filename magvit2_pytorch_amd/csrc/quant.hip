// LFQ / FSQ quantizers: one wavefront per token, per-token reductions by wave shuffles.
// Restates the eval arithmetic of vector_quantize_pytorch.LFQ / FSQ (see oracle/thirdparty.py for
// the algorithm statement; reference call sites magvit2_pytorch.py:1574, 1593, 1700, 1705).
#include "common.h"

#define CB_MAX 32

struct QuantParams {
    const void* x; const float* w_in; const float* b_in; const float* w_out; const float* b_out;
    void* quantized; void* ids; float* prequant;
    int64_t rows; int C, cb_dim, ncb;
    float clamp, scale;
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
        q.code[lane] = z > 0.f ? p.scale : -p.scale;
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

extern "C" int mv2_lfq(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                       void* quantized, int64_t* ids, float* prequant, int64_t rows, int C, int cb_dim, int num_codebooks,
                       float clamp, float scale, void* stream) {
    if (!x || !w_in || !b_in || !w_out || !b_out || !quantized || !ids) return MV2_EINVAL;
    int rc = check_q(dtype, rows, C, cb_dim, num_codebooks); if (rc) return rc;
    QuantParams p{x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, C, cb_dim, num_codebooks, clamp, scale,
                  nullptr, nullptr, nullptr, nullptr, nullptr};
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((lfq_kernel<BF16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((lfq_kernel<F32>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

extern "C" int mv2_fsq(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                       const float* shift, const float* half_l, const float* offset, const float* half_w, const int32_t* basis,
                       void* quantized, int32_t* ids, float* prequant, int64_t rows, int C, int cb_dim, int num_codebooks,
                       void* stream) {
    if (!x || !w_in || !b_in || !w_out || !b_out || !quantized || !ids || !shift || !half_l || !offset || !half_w || !basis) return MV2_EINVAL;
    int rc = check_q(dtype, rows, C, cb_dim, num_codebooks); if (rc) return rc;
    QuantParams p{x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, C, cb_dim, num_codebooks, 0.f, 1.f,
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
