// Shared pieces of the implicit-GEMM convolution kernels: launch parameters, the fused epilogue and
// the MFMA step (bf16 32x32x16 / exact-fp32 32x32x2).
#pragma once
#include <type_traits>
#include "common.h"

// compile-time int tag: lets a lambda body select a register set / LDS buffer statically
template <int V> using IC = std::integral_constant<int, V>;

struct ConvParams {
    mv2_conv_desc d;
    const void* in;
    const void* wp;
    const mv2_seg* segs;
    const float* bias;
    const float* rowscale;
    const void* residual;
    void* out;
    float* ws;
    uint32_t in_bytes;
    uint32_t w_bytes;
    int64_t M;
    int32_t lg_seg_chunks;
    int32_t n_co_tiles, n_pos_tiles;
    int32_t steps_per_split, splitk;
    int32_t ws_cols;
    int64_t ws_rows;
    // tile grid of the 3x3 kernel (conv3.hip)
    int32_t tiles_h, tiles_w, nchunk, chunks_per_split;
    int32_t aligned;  // conv3.hip: worker ranges snapped to tile starts (0 for every other kernel)
    // fused second stage (mv2_conv3 with wp2): 1x1x1 conv over the block's own output channels
    const void* wp2;
    const float* bias2;
    int32_t act2;
    // ... and, optionally, the SqueezeExcite pooling partials of the finished tile (mv2_conv3 se_part)
    const float* se_wk;
    float se_bk;
    float* se_part;
    // conv3.hip: stream-K ticket words (one per output tile, zero between launches) and the length of the group list
    uint32_t* counters;
    uint32_t gtot;        // units of the work list (groups + delta virtual units per tile)
    uint32_t delta;
    uint32_t ws_bytes;
    // mv2_conv_rms: the per-row scale 1 / |gathered row| (RMSNorm, reference M:258-276) is computed inside the GEMM launch from the rows
    // it stages anyway; `rowscale` is null
    int32_t rms_fused;
    // conv3.hip: per mille of extra work for the OLDER workgroup of a CU's pair (0 = equal ranges)
    int32_t skew;
};


// ------------------------------------------------------------------------------- epilogue
template <typename E>
__device__ __forceinline__ void decode_pos(const mv2_conv_desc& d, int64_t m, int& b, int& t, int& h, int& w) {
    // 32-bit on purpose: the host rejects M >= 2^31, and 64-bit integer division is a ~100-instruction
    // emulation that dominated the prologue / shuffle epilogue of the small launches
    uint32_t r = (uint32_t)m;
    const uint32_t Wo = (uint32_t)d.Wo, Ho = (uint32_t)d.Ho, To = (uint32_t)d.To;
    uint32_t q = r / Wo; w = (int)(r - q * Wo); r = q;
    q = r / Ho; h = (int)(r - q * Ho); r = q;
    q = r / To; t = (int)(r - q * To); b = (int)q;
}

// v[4]: raw accumulators of 4 consecutive packed columns n..n+3 of position row m
template <typename E>
__device__ __forceinline__ void epi_store4(const ConvParams& p, int64_t m, int n, float* v) {
    const mv2_conv_desc& d = p.d;
    if (m >= p.M || n >= d.co) return;
    if (p.rowscale) { float s = p.rowscale[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= s; }
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n + j < d.co) v[j] += p.bias[n + j]; }
    if (d.act == MV2_ACT_ELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = Act<E>::elu(v[j]);
    } else if (d.act == MV2_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = Act<E>::silu(v[j]);
    } else if (d.act == MV2_ACT_LRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.1f * v[j];
    }
    if (d.store == MV2_STORE_ROWS) {
        int64_t o = m * d.out_c + n;
        if (p.residual) { float r[4]; ld4<E>(p.residual, o, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += r[j]; }
        st4<E>(p.out, o, v);
    } else if (d.store == MV2_STORE_SHUFFLE) {
        int b, t, h, w; decode_pos<E>(d, m, b, t, h, w);
        const int cq = d.out_c;
        const int q = n / cq, c = n - q * cq;
        const int qw = q % d.rw, qh = (q / d.rw) % d.rh, qt = q / (d.rw * d.rh);
        const int64_t orow = (((int64_t)b * (d.To * d.rt) + (t * d.rt + qt)) * (d.Ho * d.rh) + (h * d.rh + qh)) * (d.Wo * d.rw) + (w * d.rw + qw);
        st4<E>(p.out, orow * cq + c, v);
    } else {   // MV2_STORE_NCTHW_F32
        int b, t, h, w; decode_pos<E>(d, m, b, t, h, w);
        const int to = t - d.crop_t;
        if (to < 0) return;
        const int64_t plane = (int64_t)d.Ho * d.Wo;
        const int Tout = d.To - d.crop_t;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < d.co) ((float*)p.out)[(((int64_t)b * d.co + (n + j)) * Tout + to) * plane + (int64_t)h * d.Wo + w] = v[j];
    }
}

// GEGLU: value columns nx..nx+3, gate columns nx+32.., output channels j0..j0+3
template <typename E>
__device__ __forceinline__ void epi_geglu4(const ConvParams& p, int64_t m, int nx, int j0, float* x, float* g) {
    const mv2_conv_desc& d = p.d;
    if (m >= p.M || nx >= d.co) return;
    if (p.rowscale) { float s = p.rowscale[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] *= s; g[j] *= s; } }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float xv = x[j], gv = g[j];
        if (p.bias) { xv += p.bias[nx + j]; gv += p.bias[nx + 32 + j]; }
        o[j] = Act<E>::gelu(gv) * xv;
    }
    st4<E>(p.out, m * d.out_c + j0, o);
}

// ------------------------------------------------------------------------------- LDS-staged epilogue
// The MFMA accumulator layout gives each lane 4 consecutive channels of ONE position, 32 positions per
// half-wave: a direct store touches 32 rows per instruction with 8 B each and is store-issue bound
// (measured: ~47 % of the 3x3 kernel at C = 64).  Instead every wave finishes its values in registers
// (rowscale, bias, activation, residual), drops them into a [position][channel] LDS tile, and the
// block then streams the tile out with 16-byte-per-lane stores over whole rows.
//   row_of(local_pos) -> global row m (>= p.M when the position is outside the tensor)
template <typename E, int BCO, int BPOS>
struct StageCfg {
    static constexpr int kPitch = BCO * E::kBytes + 16;            // bytes per staged row (pad: bank spread)
    static constexpr int kPitchWide = BCO * 4 + 16;                // ... when the tile is staged in fp32 (residual add before the one rounding)
    static constexpr int kTile = BPOS * kPitchWide;
    static constexpr int kRowTab = kTile;                          // destination-row table (int64 per staged row)
    static constexpr int kRowScale = kRowTab + BPOS * 8;           // per-row scale (fp32 per staged row)
    static constexpr int kBias = kRowScale + BPOS * 4;             // bias of the block's BCO columns (fp32)
    static constexpr int kBytes = kBias + BCO * 4;
};

template <typename E> __device__ __forceinline__ void lds_put4(char* dst, const float* v);
template <> __device__ __forceinline__ void lds_put4<BF16>(char* dst, const float* v) {
    i32x2 w;
    w[0] = (int)((unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16));
    w[1] = (int)((unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16));
    *(i32x2*)dst = w;
}
template <> __device__ __forceinline__ void lds_put4<F32>(char* dst, const float* v) {
    f32x4 w = {v[0], v[1], v[2], v[3]};
    *(f32x4*)dst = w;
}

__device__ __forceinline__ bool stage_ok(const ConvParams& p) {
    const int es = p.d.dtype == MV2_BF16 ? 2 : 4;
    return p.d.store != MV2_STORE_NCTHW_F32 && ((p.d.out_c * es) % 16) == 0 && (p.d.co % (16 / es)) == 0;
}

// owner: false for waves that hold no accumulators; they skip phase 1 and
// only help with the row stores of phase 2
template <typename E, int BCO, int BPOS, int WCO, int WPOS, int NT = 256, typename RowFn>
__device__ __forceinline__ void staged_epilogue(const ConvParams& p, char* lds, f32x16 (&acc)[WCO][WPOS], int n0, int wco,
                                                int wpos, int tid, RowFn row_of, bool owner = true, bool rs_in_lds = false,
                                                bool have_pf = false, float rs_pf = 1.f, float bias_pf = 0.f) {
    const mv2_conv_desc& d = p.d;
    typedef StageCfg<E, BCO, BPOS> SC;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const bool geglu = d.act == MV2_ACT_GEGLU;
    // with a residual the tile is staged in fp32: out = round(act(..) + residual), ONE rounding (staging in bf16 and adding
    // afterwards rounds twice: measured +39 % reconstruction error on configs[1])
    const bool wide = p.residual != nullptr && E::kBytes == 2;
    const int PITCH = wide ? SC::kPitchWide : SC::kPitch;
    int64_t* rowtab = (int64_t*)(lds + SC::kRowTab);
    float* rs_lds = (float*)(lds + SC::kRowScale);
    float* bias_lds = (float*)(lds + SC::kBias);
    // ---- phase 0: everything the per-value math needs from global memory goes to LDS with ONE round of loads (a bias /
    // rowscale / residual load per value made phase 1 a chain of dependent memory latencies: 7-24k clocks per block measured):
    // destination row of every staged row (-1: outside the tensor; decoded once per row so that phase 2 has no division),
    // its rowscale, and the bias of the block's columns.  The residual is added in phase 2, in whole 16-byte chunks.
    for (int lp = tid; lp < BPOS; lp += NT) {
        const int64_t m = row_of(lp);
        const bool valid = m < p.M;
        int64_t dr = -1;
        if (valid) {
            dr = m;
            if (d.store == MV2_STORE_SHUFFLE) {
                int b_, t_, h_, w_; decode_pos<E>(d, m, b_, t_, h_, w_);
                dr = (((int64_t)b_ * (d.To * d.rt) + t_ * d.rt) * (d.Ho * d.rh) + h_ * d.rh) * (int64_t)(d.Wo * d.rw) + w_ * d.rw;
            }
        }
        rowtab[lp] = dr;
        // rs_in_lds: the caller has put the scales there.  have_pf: thread lp holds row lp's scale / column lp's bias, requested in the
        // kernel PROLOGUE (BPOS, BCO <= NT): no dependent global round trip at the head of the epilogue.
        if (!rs_in_lds) rs_lds[lp] = have_pf ? rs_pf : ((valid && p.rowscale) ? p.rowscale[m] : 1.f);
    }
    if (have_pf) { if (tid < BCO) bias_lds[tid] = bias_pf; }
    else for (int q = tid; q < BCO; q += NT) bias_lds[q] = (p.bias && n0 + q < d.co_pad) ? p.bias[n0 + q] : 0.f;
    __syncthreads();
    // ---- phase 1: finish values per lane, write the tile
#pragma unroll
    for (int b = 0; b < WPOS; ++b) {
        if (!owner) break;
        const int lp = wpos * 32 * WPOS + b * 32 + l31;
        const float rs = rs_lds[lp];
        char* rowp = lds + lp * PITCH;
        if (geglu) {
            if constexpr (WCO == 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int off = 8 * g + 4 * hi;
                    const int cx = wco * 64 + off;               // block-local packed value column; gate = +32
                    const f32x4 bx = *(const f32x4*)(bias_lds + cx), bg = *(const f32x4*)(bias_lds + cx + 32);
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xv = acc[0][b][4 * g + j] * rs + bx[j], gv = acc[1][b][4 * g + j] * rs + bg[j];
                        o[j] = Act<E>::gelu(gv) * xv;
                    }
                    lds_put4<E>(rowp + (wco * 32 + off) * E::kBytes, o);
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < WCO; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = wco * 32 * WCO + a * 32 + 8 * g + 4 * hi;
                    const f32x4 bv = *(const f32x4*)(bias_lds + col);
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = acc[a][b][4 * g + j] * rs + bv[j];
                        if (d.act == MV2_ACT_ELU) x = Act<E>::elu(x);
                        else if (d.act == MV2_ACT_SILU) x = Act<E>::silu(x);
                        else if (d.act == MV2_ACT_LRELU) x = x > 0.f ? x : 0.1f * x;
                        v[j] = x;
                    }
                    if (wide) lds_put4<F32>(rowp + col * 4, v);
                    else lds_put4<E>(rowp + col * E::kBytes, v);
                }
        }
    }
    __syncthreads();
    // ---- phase 2: coalesced 16-byte stores of whole rows (+ the residual, read the same way).  The chunk count per row is a
    // power of two that divides the thread count, so a thread keeps ONE channel chunk and walks rows: everything that depends
    // on the channel (incl. the depth-to-space sub-position) is computed once, outside the loop.
    constexpr int CPR_FULL = BCO * E::kBytes / 16;                 // 16-byte chunks per staged row (power of two)
    static_assert((CPR_FULL & (CPR_FULL - 1)) == 0 && NT % CPR_FULL == 0, "chunk-per-thread mapping");
    const int cpr = geglu ? CPR_FULL / 2 : CPR_FULL;
    const int nbase = geglu ? n0 / 2 : n0;
    const int nvalid = geglu ? d.out_c : d.co;                    // valid output columns
    const int ch = tid & (cpr - 1);
    const int lp0 = geglu ? tid / (CPR_FULL / 2) : tid / CPR_FULL;
    const int lstep = geglu ? NT / (CPR_FULL / 2) : NT / CPR_FULL;
    const int n = nbase + ch * E::kVec;
    if (n >= nvalid) return;
    int64_t coloff = n, rowmul = d.out_c;
    if (d.store == MV2_STORE_SHUFFLE) {
        const uint32_t cq = (uint32_t)d.out_c;
        const uint32_t qq = (uint32_t)n / cq, c = (uint32_t)n - qq * cq;
        const uint32_t qw = qq % (uint32_t)d.rw, qh = (qq / (uint32_t)d.rw) % (uint32_t)d.rh, qt = qq / (uint32_t)(d.rw * d.rh);
        coloff = (((int64_t)qt * (d.Ho * d.rh) + qh) * (int64_t)(d.Wo * d.rw) + qw) * cq + c;
        rowmul = cq;
    }
    const bool res = p.residual != nullptr;                       // MV2_STORE_ROWS only (checked by the host)
    for (int lp = lp0; lp < BPOS; lp += lstep) {
        const int64_t dr = rowtab[lp];
        if (dr < 0) continue;
        const int64_t off = (dr * rowmul + coloff) * E::kBytes;
        i32x4 val;
        if (res) {
            const i32x4 rv = *(const i32x4*)((const char*)p.residual + off);
            float a_[E::kVec], r_[E::kVec];
            bool got = false;
            if constexpr (E::kVec == 8) {
                if (wide) {           // 8 fp32 values = two 16-byte chunks of the wide tile
                    const f32x4 lo = *(const f32x4*)(lds + lp * PITCH + ch * 32), hi4 = *(const f32x4*)(lds + lp * PITCH + ch * 32 + 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a_[j] = lo[j]; a_[4 + j] = hi4[j]; }
                    got = true;
                }
            }
            if (!got) Chunk<E>::unpack(*(const i32x4*)(lds + lp * PITCH + ch * 16), a_);
            Chunk<E>::unpack(rv, r_);
#pragma unroll
            for (int j = 0; j < E::kVec; ++j) a_[j] += r_[j];
            val = Chunk<E>::pack(a_);
        } else {
            val = *(const i32x4*)(lds + lp * PITCH + ch * 16);
        }
        *(i32x4*)((char*)p.out + off) = val;
    }
}

// ------------------------------------------------------------------------------- MFMA step
template <typename E> struct Mma;
template <> struct Mma<BF16> {
    static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<F32> {
    // 16 B = 4 fp32 per lane -> 4 x v_mfma_f32_32x32x2_f32 (exact fp32 fma chain); the k order is
    // a fixed permutation shared by both operands (lane half hi owns k = 4*hi + j of each 8-block).
    static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& acc) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a[j]), __int_as_float(b[j]), acc, 0, 0, 0);
    }
};

