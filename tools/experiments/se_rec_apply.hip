// tools/experiments/se_rec_apply.hip - BUILT, CORRECT, SLOWER: kept as the record of the measurement, not part of the library.
//
// Round 4: the small-frame SqueezeExcite tail (the C = 512 levels: 12 of a clip's 22 ResidualUnits) in ONE launch instead of
// mv2_se_rec_fc1 + mv2_se_fc2_apply.  It was wired through the whole stack (C ABI entry mv2_se_rec_apply + mv2_se_rec_apply_ok, ctypes
// binding, kernel emulation, lowering with MV2_SE_TWO as the switch back; commit 'mv2_se_rec_apply: the small-frame SqueezeExcite tail ...'),
// passed the lockstep tests at the real widths, and measured on MI355X in the model (profiles/r04_findings.md section 8):
//     two launches  6.4 + 5.8 us per unit,  step 2.798 ms          one launch  13.1 us per unit,  step 2.832 ms  (two interleaved runs each)
// Why: every block must compute ALL hidden units of fc1 before its fc2 (w1t = C x hidden fp32 = 256 KB at C = 512): one block streams
// that at one CU's L2 rate (~2 us), and the chain records -> merge -> fc1 -> fc2 -> gate -> apply is serial inside the block, where the
// two-launch form spreads fc1 over S1 x frames blocks.  A launch boundary costs 1.5-2 us on this chip; the serial fc1 costs more.
// (Excerpt of misc.hip at that commit; it compiles in that file's context: common.h, is_pow2.)

// ONE launch for the SqueezeExcite tail of a small-frame level (the C = 512 levels: 12 of the 22 ResidualUnits of a clip): what
// se_rec_fc1 + se_fc2_apply did in two.  grid (S2, frames); a block owns C / S2 channels of its frame: it merges the frame's pooling
// records (a few KB), computes ALL hidden units of fc1 itself (w1t is C x hidden fp32 = 256 KB at C = 512: as float4 columns, 16 loads
// per thread in ONE round trip; the S2 blocks of a frame repeat it - 41 MB of L2 reads per launch instead of a second launch and a
// round trip of the hidden vector through HBM), then fc2 + sigmoid for its channels and out = gate * y + residual over the frame.
// The two-launch form's floor was two dependent ~6 us launches; this one is the longer of the two chains plus one L2 round trip.
template <typename E>
__global__ __launch_bounds__(1024) void se_rec_apply_kernel(const float* __restrict__ part, const float* __restrict__ w1t,
                                                            const float* __restrict__ b1, const float* __restrict__ w2t,
                                                            const float* __restrict__ b2, const void* __restrict__ y,
                                                            const void* __restrict__ res, void* __restrict__ out, int P, int C,
                                                            int nchunk, int hidden, int cw) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KV = E::kVec;
    const int f = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
    const int R = nchunk, hq = hidden >> 2, CG = 1024 / hq;
    float* wr = sm + ((R * (C + 2) + 3) & ~3);          // R merge weights (every region starts 16-byte aligned: the fc1 slab is written as float4)
    float* pooled = wr + ((R + 3) & ~3);                // C
    float* hs = pooled + C;                             // hidden
    float* g = hs + hidden;                             // cw
    float* slab = g + cw;                               // max(CG * hidden, 1024)
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
    // fc1, all hidden units: thread = (channel group cg, hidden quad q)
    {
        const int q = tid % hq, cg = tid / hq;
        const int per = (C + CG - 1) / CG;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (cg < CG) {
            const int c0 = cg * per, c1 = min(C, c0 + per);
            int c = c0;
            for (; c + 16 <= c1; c += 16) {
                f32x4 w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = *(const f32x4*)(w1t + (int64_t)(c + u) * hidden + 4 * q);
#pragma unroll
                for (int u = 0; u < 16; ++u) { const float pc = pooled[c + u]; a[0] += w[u][0] * pc; a[1] += w[u][1] * pc; a[2] += w[u][2] * pc; a[3] += w[u][3] * pc; }
            }
            for (; c < c1; ++c) { const f32x4 w = *(const f32x4*)(w1t + (int64_t)c * hidden + 4 * q); const float pc = pooled[c]; a[0] += w[0] * pc; a[1] += w[1] * pc; a[2] += w[2] * pc; a[3] += w[3] * pc; }
            *(f32x4*)(slab + cg * hidden + 4 * q) = a;
        }
    }
    __syncthreads();
    if (tid < hidden) {
        float a = b1[tid];
        for (int cg = 0; cg < CG; ++cg) a += slab[cg * hidden + tid];
        hs[tid] = a > 0.f ? a : 0.1f * a;
    }
    __syncthreads();
    // fc2 + sigmoid for this block's channels (se_fc2_apply's code)
    const int GS = 1024 / cw;
    const int cl = tid % cw, ks = tid / cw, c = sl * cw + cl;
    {
        const int per = (hidden + GS - 1) / GS;
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

extern "C" int mv2_se_rec_apply_ok(int dtype, int P, int C, int hidden, int nchunk, int nslice) {
    if (dtype != MV2_BF16 && dtype != MV2_F32) return 0;
    const int kvec = dtype == MV2_BF16 ? 8 : 4;
    if (P <= 0 || C <= 0 || hidden <= 0 || nchunk <= 0 || nslice <= 0 || C % nslice != 0 || hidden % 4 != 0) return 0;
    const int cw = C / nslice, hq = hidden / 4;
    if (!is_pow2(cw) || cw > 1024 || cw < kvec || !is_pow2(hq) || hq > 1024 || hidden > 1024) return 0;
    if ((int64_t)P * (cw / kvec) >= (int64_t)0x7fffffff) return 0;
    const int CG = 1024 / hq;
    const size_t lds = ((size_t)((nchunk * (C + 2) + 3) & ~3) + ((nchunk + 3) & ~3) + C + hidden + cw + std::max(CG * hidden, 1024)) * sizeof(float);
    return (C % 4 == 0 && lds <= 64 * 1024) ? 1 : 0;
}

extern "C" int mv2_se_rec_apply(int dtype, const float* part, const float* w1t, const float* b1, const float* w2t, const float* b2,
                                const void* y, const void* residual, void* out, int frames, int P, int C, int hidden, int nchunk,
                                int nslice, void* stream) {
    if (!part || !w1t || !b1 || !w2t || !b2 || !y || !out || frames <= 0) return MV2_EINVAL;
    if (!mv2_se_rec_apply_ok(dtype, P, C, hidden, nchunk, nslice)) return MV2_EINVAL;
    const int cw = C / nslice, CG = 1024 / (hidden / 4);
    const size_t lds = ((size_t)((nchunk * (C + 2) + 3) & ~3) + ((nchunk + 3) & ~3) + C + hidden + cw + std::max(CG * hidden, 1024)) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nslice, (unsigned)frames);
    if (dtype == MV2_BF16) hipLaunchKernelGGL((se_rec_apply_kernel<BF16>), grid, dim3(1024), lds, st, part, w1t, b1, w2t, b2, y, residual, out, P, C, nchunk, hidden, cw);
    else hipLaunchKernelGGL((se_rec_apply_kernel<F32>), grid, dim3(1024), lds, st, part, w1t, b1, w2t, b2, y, residual, out, P, C, nchunk, hidden, cw);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

