// ResidualUnit tail at C = 128 / 256 / 512 (bf16): the 1x1x1 conv + bias + ELU (reference magvit2_pytorch.py:937-938) with the
// SqueezeExcite pooling partials (reference SqueezeExcite.forward M:226-231) taken from the output tile while it is still in LDS.
// Replaces two launches of the generic path (mv2_conv k = 1 and mv2_se_pool / the pooling half of mv2_se_pool_fc1); at these
// widths the generic GEMM is a latency chain (K = C is 8-32 MFMA steps) and the pooling pass re-reads the tensor it just wrote.
//
// One workgroup = BP (32 or 64) consecutive rows x ALL C output channels, so the per-position logit w_k . y needs no exchange
// between workgroups.  Four waves split the output channels (C/4 each); the input tile sits in LDS as the B operand
// ([row][C] bf16, rows padded by 16 bytes), the weights stream global -> VGPR in the fragment-major layout of mv2_conv3
// (pack.py::to_frag_major of the 1x1x1 kernel): no LDS traffic for weights, four k-steps of fragments in flight.
// The output tile replaces the input tile in LDS; rows leave as whole 16-byte chunks; logits, softmax weights and the
// exp-weighted channel sums are read back from that bf16 tile — exactly what mv2_se_pool would read from HBM.
// One record {max logit, sum exp, exp-weighted sums[C]} per workgroup: the record list of mv2_se_gate with nchunk = P / BP.
//
// Round 5: the number of waves is a template parameter.  With four waves at C = 512 a wave streams 128 KB of weights through a ring of
// 4-8 k-steps: 4-8 dependent L2 round trips, and the launch lasted 12-14 us whatever the frame count (a latency chain, not a stream).
// With C / 32 waves (16 at C = 512, 8 at C = 256) a wave owns ONE 32-channel tile and its ring holds 16 k-steps: one or two round
// trips for the whole GEMM; the row tile is read from LDS by every wave (16 x 32 KB at C = 512: ~0.5 us of the LDS pipe).  The OUTPUT
// rows are bit-identical between the two wave counts (every output element accumulates its k-steps in the same order); the SqueezeExcite
// records are not: their exp-weighted channel sums run over NG = NT * 2 / C row groups (1 with four waves at C = 512, 4 with sixteen), so
// the fp32 summation order - and with it the last bits of the gate and of everything downstream - follows the wave count, which the
// launch picks from rows and batch (`wide` below: the same clip can take different orders at batch 1 and batch 4; run-to-run results of
// one shape are bitwise reproducible).  profiles/r05_findings.md has the timings.
#include <hip/hip_runtime.h>
#include "common.h"
#include "conv_common.h"
#include "../../include/mv2.h"

struct PwSeParams {
    const void* in; const void* wp; const float* bias; const float* wk; float bk;
    void* out; float* part;
    int P, nchunk;              // rows per frame, records per frame
};

template <int CT, int PT, int NW> struct PwSeLds {
    static constexpr int C = CT * 128, BP = PT * 32, PITCH = C * 2 + 16, NT = NW * 64;
    static constexpr int NG = NT * 2 / C;                // row groups of the record sums (channel pairs x groups = threads)
    static constexpr int kTile = 0, kWk = BP * PITCH, kBias = kWk + C * 4, kLogit = kBias + C * 4, kEw = kLogit + BP * 4,
                         kGrp = kEw + BP * 4, kBytes = kGrp + NG * C * 4;
};

template <int CT, int PT, int NW>
__global__ __launch_bounds__(NW * 64) void pw_se_kernel(const PwSeParams p) {
    typedef PwSeLds<CT, PT, NW> L;
    constexpr int C = L::C, BP = L::BP, PITCH = L::PITCH, NT = L::NT;
    constexpr int TPW = C / 32 / NW;                     // 32-channel output tiles per wave
    static_assert(TPW >= 1 && TPW * NW * 32 == C, "waves split the output channels");
    // k-steps of 16 channels; k-steps of weight fragments in flight per wave (L2 latency): one tile per wave -> 16 (64 registers)
    constexpr int KS = C / 16, D = TPW == 1 ? (KS < 16 ? KS : 16) : ((TPW * PT >= 8) ? 4 : 8);
    constexpr int CPR = C / 8, NLD = BP * CPR / NT;      // 16-byte chunks per row; chunks per thread
    static_assert(NLD >= 1 && NLD * NT == BP * CPR, "whole chunks per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * BP;

    // ---- weights: first D k-steps on their way before anything else
    const char* wbase = (const char*)p.wp + (size_t)(wave * TPW) * 4096 + lane * 16;
    auto wfrag = [&](int ks, int ct) {                   // [step = ks / 4][C / 32 co tiles][k-slice = ks % 4][64 lanes][16 B]
        return *(const i32x4*)(wbase + ((size_t)(ks >> 2) * (C / 32) + ct) * 4096 + (ks & 3) * 1024);
    };
    i32x4 wf[D][TPW];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int ct = 0; ct < TPW; ++ct) wf[d][ct] = wfrag(d, ct);

    // ---- input tile -> LDS (coalesced 16-byte chunks), constants
    {
        i32x4 v[NLD];
        const char* src = (const char*)p.in + row0 * C * 2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) v[i] = *(const i32x4*)(src + (size_t)(tid + NT * i) * 16);
        for (int c = tid; c < C; c += NT) {
            ((float*)(smem + L::kWk))[c] = p.wk[c];
            ((float*)(smem + L::kBias))[c] = p.bias[c];
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + NT * i, row = q / CPR, ch = q % CPR;
            *(i32x4*)(smem + row * PITCH + ch * 16) = v[i];
        }
    }
    __syncthreads();

    f32x16 acc[TPW][PT];
#pragma unroll
    for (int ct = 0; ct < TPW; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        i32x4 b[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) b[pt] = *(const i32x4*)(smem + (pt * 32 + l31) * PITCH + ks * 32 + hi * 16);
#pragma unroll
        for (int ct = 0; ct < TPW; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) Mma<BF16>::run(wf[ks % D][ct], b[pt], acc[ct][pt]);
        if (ks + D < KS) {
#pragma unroll
            for (int ct = 0; ct < TPW; ++ct) wf[ks % D][ct] = wfrag(ks + D, ct);
        }
    }
    __syncthreads();                                     // every wave has read the input tile: it becomes the output tile

    // ---- bias + ELU -> bf16 tile.  acc[ct][pt][4 g + j] = channel (wave TPW + ct) 32 + 8 g + 4 hi + j of row pt 32 + l31
#pragma unroll
    for (int ct = 0; ct < TPW; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = (wave * TPW + ct) * 32 + 8 * g + 4 * hi;
            const f32x4 bv = *(const f32x4*)(smem + L::kBias + col * 4);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = Act<BF16>::elu(acc[ct][pt][4 * g + j] + bv[j]);
                lds_put4<BF16>(smem + (pt * 32 + l31) * PITCH + col * 2, v);
            }
        }
    __syncthreads();

    // ---- rows out
    {
        char* dst = (char*)p.out + row0 * C * 2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + NT * i, row = q / CPR, ch = q % CPR;
            *(i32x4*)(dst + (size_t)q * 16) = *(const i32x4*)(smem + row * PITCH + ch * 16);
        }
    }

    // ---- SqueezeExcite record of these BP rows
    constexpr int TPP = NT / BP;                         // threads per row for the logit (<= 32: one wave holds whole rows)
    static_assert(TPP <= 32 && CPR % TPP == 0, "logit threads");
    {
        const int row = tid / TPP, q = tid % TPP;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < CPR / TPP; ++i) {
            const int ch = q + TPP * i;
            float xv[8];
            Chunk<BF16>::unpack(*(const i32x4*)(smem + row * PITCH + ch * 16), xv);
            const f32x4 w0 = *(const f32x4*)(smem + L::kWk + ch * 32), w1 = *(const f32x4*)(smem + L::kWk + ch * 32 + 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) dot += xv[k] * w0[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) dot += xv[4 + k] * w1[k];
        }
#pragma unroll
        for (int o = 1; o < TPP; o <<= 1) dot += __shfl_xor(dot, o, 64);
        if (q == 0) ((float*)(smem + L::kLogit))[row] = dot + p.bk;
    }
    __syncthreads();
    float M = -INFINITY;
    {
        const float* lg = (const float*)(smem + L::kLogit);
#pragma unroll
        for (int r = 0; r < BP; r += 4) {
            const f32x4 v = *(const f32x4*)(lg + r);
            M = fmaxf(M, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        }
        if (tid < BP) ((float*)(smem + L::kEw))[tid] = expf(lg[tid] - M);
    }
    __syncthreads();
    const int f = (int)(row0 / p.P), j = (int)((row0 - (int64_t)f * p.P) / BP);
    float* dst = p.part + ((int64_t)f * p.nchunk + j) * (C + 2);
    const float* ew = (const float*)(smem + L::kEw);
    if (wave == 0) {
        const float s = wave_sum(lane < BP ? ew[lane] : 0.f);
        if (lane == 0) { dst[0] = M; dst[1] = s; }
    }
    {
        constexpr int NG = L::NG;                        // row groups when there are fewer channel pairs than threads
        const int cp = tid % (C / 2), g = tid / (C / 2);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
        for (int r = g; r < BP; r += NG) {
            const unsigned w = *(const unsigned*)(smem + r * PITCH + cp * 4);
            const float e = ew[r];
            a0 += e * __uint_as_float(w << 16);
            a1 += e * __uint_as_float(w & 0xffff0000u);
        }
        if constexpr (NG == 1) {
            *(f32x2*)(dst + 2 + 2 * cp) = f32x2{a0, a1};
        } else {
            float* grp = (float*)(smem + L::kGrp);
            *(f32x2*)(grp + g * C + 2 * cp) = f32x2{a0, a1};
            __syncthreads();
            if (tid < C) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < NG; ++q) a += grp[q * C + tid];
                dst[2 + tid] = a;
            }
        }
    }
}

static int pw_se_bp(int dtype, int64_t rows, int P, int C) {
    if (dtype != MV2_BF16 || rows <= 0 || P <= 0 || rows % P != 0) return 0;
    if (C != 128 && C != 256 && C != 512) return 0;
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_PW_SE_BP")) { const int v = atoi(e); if ((v == 32 || v == 64) && P % v == 0) return v; }
#endif
    if (P % 64 == 0 && rows / 64 >= 512) return 64;      // enough 64-row workgroups to fill the chip twice
    return P % 32 == 0 ? 32 : 0;
}

// rows per SqueezeExcite record (= rows per workgroup); 0 = not eligible (the caller uses mv2_conv + mv2_se_pool)
extern "C" int mv2_pw_se_rows(int dtype, int64_t rows, int P, int C) { return pw_se_bp(dtype, rows, P, C); }

extern "C" int mv2_pw_se(int dtype, const void* in, const void* wp, const float* bias, const float* se_wk, float se_bk, void* out,
                         float* se_part, int64_t rows, int P, int C, void* stream) {
    if (!in || !wp || !bias || !se_wk || !out || !se_part) return MV2_EINVAL;
    if ((((uintptr_t)in | (uintptr_t)wp | (uintptr_t)out) & 15) || ((uintptr_t)se_part & 7)) return MV2_EINVAL;
    const int bp = pw_se_bp(dtype, rows, P, C);
    if (!bp) return MV2_EINVAL;
    PwSeParams p{in, wp, bias, se_wk, se_bk, out, se_part, P, P / bp};
    const dim3 grid((unsigned)(rows / bp));
    hipStream_t st = (hipStream_t)stream;
#define PW_SE(CTv, PTv, NWv)                                                                                              \
    do {                                                                                                                  \
        const int lds = PwSeLds<CTv, PTv, NWv>::kBytes;                                                                   \
        if (hipFuncSetAttribute((const void*)pw_se_kernel<CTv, PTv, NWv>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) \
            return MV2_ENODEVICE;                                                                                              \
        hipLaunchKernelGGL((pw_se_kernel<CTv, PTv, NWv>), grid, dim3(NWv * 64), lds, st, p);                              \
    } while (0)
    // waves per workgroup: one 32-channel tile per wave where that is faster (see the head of the file); experiments: MV2_PW_SE_NW=4 | 0 forces a form
    // measured (profiles/r05_pwse_waves.log, us, four waves -> C / 32 waves): C = 512 at 5120 / 2560 / 1280 rows 12.3 -> 10.5 / 11.4 -> 9.7 / 11.1 ->
    // 9.4; at 20480 / 10240 rows 28.7 -> 30.9 / 18.5 -> 20.3 (one 16-wave workgroup per CU against four 4-wave ones: the same wave count,
    // fewer tiles in flight); C = 256 on 64-row tiles (81920 rows) 31.6 -> 29.2, on 32-row tiles (20480 rows) 10.9 -> 12.4
    bool wide = (C == 512 && rows <= 8192) || (C == 256 && bp == 64);
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_PW_SE_NW")) wide = atoi(e) != 4;
#endif
    if (bp == 64) {
        if (C == 128) PW_SE(1, 2, 4);
        else if (C == 256) { if (wide) PW_SE(2, 2, 8); else PW_SE(2, 2, 4); }
        else { if (wide) PW_SE(4, 2, 16); else PW_SE(4, 2, 4); }
    } else {
        if (C == 128) PW_SE(1, 1, 4);
        else if (C == 256) { if (wide) PW_SE(2, 1, 8); else PW_SE(2, 1, 4); }
        else { if (wide) PW_SE(4, 1, 16); else PW_SE(4, 1, 4); }
    }
#undef PW_SE
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
