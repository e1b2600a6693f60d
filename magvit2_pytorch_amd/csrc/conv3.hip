// mv2_conv3 — the ResidualUnit kt x 3 x 3 causal convolution (reference magvit2_pytorch.py:937, CausalConv3d
// :892-928; ~81 % of the path's FLOPs), optionally fused with the unit's 1x1x1 conv + ELU (:938-940) and the
// SqueezeExcite pooling partials (:226-231).
//
// Decomposition (gfx950: 256 CUs, 4 SIMDs per CU, 512 registers per lane per SIMD):
//   * TWO 4-wave workgroups per CU (256 registers per wave).  A workgroup owns a 64-channel x 8x16-position output tile;
//     its four waves split K: wave w multiplies k-slice w (16 of the 64 channels of every 128-byte chunk) of EVERY
//     tap and keeps the whole tile in its own 128 accumulator registers (2 x 4 MFMA tiles of 32x32).  A wave
//     therefore issues 8 MFMAs per 6 operand fragments, and no two waves share a weight byte:
//   * weights never touch LDS: they are packed on the host in MFMA-fragment order
//     ([step][co/32][k-slice][half][row][16 B], pack.py frag_major) and each wave streams its own 1 KB fragments
//     global -> VGPR through a ring of slots, three taps ahead;
//   * only the 10 x 18 input halo of one (frame, chunk) lives in LDS (double buffered, 144-byte padded rows, odd tile
//     rows rotated: conflict-free); the nine spatial taps are nine immediate offsets into it.  One barrier per NINE taps;
//   * the four partial tiles are reduced through LDS once per tile (fixed order -> deterministic); afterwards every wave
//     owns all 64 channels of its 32 positions and runs its epilogue without a block barrier;
//   * groups (chunk, time tap) whose input frame lies in the causal zero padding are skipped, not multiplied;
//   * stream-K: worker w walks the contiguous range [w G/W, (w+1) G/W) of the tile-major list of all G cost-weighted
//     units.  A tile cut by a range boundary is finished in-kernel: every part drops its fp32 partial into a slab, takes a
//     ticket, and the LAST arriver sums all parts in part order (bitwise deterministic) and runs the epilogue
//     (write-through sc1 stores / loads, cdna_hip_programming.md §5; no spinning, so no residency requirement).  Where a
//     worker owns >= 4 tiles the range ends are snapped to tile starts instead and nothing is cut.  There is no finish
//     kernel and no second pass over the output.
// What limits it (measured, profiles/r02_k3_findings.md): one wave per SIMD issues an MFMA every ~49 cycles here whatever its
// loads do; the MV2_C3_* macros (conv3_knobs.h) are the experiment variants of tools/build_c3_variants.sh (constants in the product build).
// Zero padding in space comes from out-of-range buffer offsets (hardware returns 0).
#include <algorithm>
#include <cstdlib>
#include "conv_common.h"
#include "conv3p.h"

namespace {

// ---- position index inside a tile -> (row, col).  With 16-wide tiles the two tile rows of a 32-lane MFMA operand
// sit HWD = 18 halo rows apart; rotating the odd row by 14 columns makes the 16 lanes of every ds_read_b128
// service group hit 16 distinct rows mod 16, i.e. conflict free in the swizzled image (plain order is 2-way on 2
// of 16 slots: the 24 % SQ_LDS_BANK_CONFLICT of the round-1 kernel).
template <int TW>
__device__ __forceinline__ void pos_of(int pp, int& ph, int& pw) {
    ph = pp / TW;
    const int x = pp % TW;
    pw = (TW == 16 && (ph & 1)) ? ((x + 14) & 15) : x;
}

struct TileInfo {
    int co_tile, b, t, thi, twi, v;     // v = number of valid time taps of this output frame (causal front)
    uint32_t G0, ng;                    // virtual index of the tile's first unit, REAL groups in the tile
    uint32_t L;                         // linear tile id (ticket index)
};

// frames-before-t in units of valid time taps: c(t) = sum_{j<t} min(j, pt) + 1
__device__ __forceinline__ uint32_t cum_taps(int t, int pt) {
    return t <= pt ? (uint32_t)(t * (t + 1) / 2) : (uint32_t)(pt * (pt + 1) / 2 + (t - pt) * (pt + 1));
}

// The work list is tile-major.  A tile of ng groups occupies ng + delta consecutive units of it: delta "virtual" units
// for the per-tile fixed cost (cross-wave reduce, finishing the tile, epilogue) followed by its ng groups of 9 taps.
// Equal shares of UNITS then give a worker that owns many tile starts fewer groups (measured imbalance with delta = 0:
// slowest workgroup 1.35 x the average at C = 512).  Decodes unit x into its tile; g = position inside the tile (may
// be < delta).
__device__ __forceinline__ TileInfo decode_unit(const ConvParams& p, uint32_t x, uint32_t& g_virtual) {
    const mv2_conv_desc& d = p.d;
    const int pt = d.kt - 1;
    const uint32_t delta = p.delta;
    const uint32_t u = (uint32_t)(p.tiles_h * p.tiles_w);
    auto F = [&](int t) { return u * ((uint32_t)p.nchunk * cum_taps(t, pt) + delta * (uint32_t)t); };     // units before frame t
    const uint32_t X1 = F(d.Ti);
    TileInfo ti;
    const uint32_t cb = x / X1, r = x - cb * X1;
    ti.co_tile = (int)(cb / (uint32_t)d.B);
    ti.b = (int)(cb - (uint32_t)ti.co_tile * (uint32_t)d.B);
    int t = 0;
    if (r < F(pt)) { while (F(t + 1) <= r) ++t; }
    else t = pt + (int)((r - F(pt)) / (u * ((uint32_t)p.nchunk * (uint32_t)(pt + 1) + delta)));
    ti.t = t;
    ti.v = min(t, pt) + 1;
    const uint32_t r2 = r - F(t);
    ti.ng = (uint32_t)(p.nchunk * ti.v);
    const uint32_t vs = ti.ng + delta;
    const uint32_t sp = r2 / vs;
    g_virtual = r2 - sp * vs;
    ti.thi = (int)(sp / (uint32_t)p.tiles_w);
    ti.twi = (int)(sp - (uint32_t)ti.thi * (uint32_t)p.tiles_w);
    ti.G0 = x - g_virtual;
    ti.L = (((uint32_t)ti.co_tile * (uint32_t)d.B + (uint32_t)ti.b) * (uint32_t)d.Ti + (uint32_t)t) * u + sp;
    return ti;
}

// Range starts.  skew = 0: W workers split G units equally, start(w) = floor(w G / W).
// skew > 0 (two workgroups per CU, W a multiple of 16): the two workgroups of a CU are (blockIdx, blockIdx + W / 2); the issue arbiter
// favours the OLDER one (priority, then age), which therefore runs at its solo rate and finishes early, while the younger one does a
// large part of its taps alone afterwards at the same solo rate (measured: 87k against 121k clocks for equal ranges at C = 512, the
// launch lasts as long as the younger).  Ranges proportional to the two rates make the pair finish together: the older workgroup of
// every pair takes (1000 + skew) / 2000 of the pair's units.  In worker order (after the XCD remap) the older workgroups are the
// first half of each XCD's segment of W / 8 workers.
__device__ __forceinline__ uint32_t start_of(uint32_t w, uint32_t W, uint32_t G, uint32_t skew) {
    if (skew == 0) return (uint32_t)((uint64_t)w * G / W);
    const uint32_t q = W >> 3, h = q >> 1;
    const uint32_t seg = w / q, j = w - seg * q;
    const uint64_t cw = (uint64_t)seg * q * 1000u + (j <= h ? (uint64_t)j * (1000u + skew) : (uint64_t)h * (1000u + skew) + (uint64_t)(j - h) * (1000u - skew));
    return (uint32_t)(cw * G / ((uint64_t)W * 1000u));
}
// worker that owns linear unit x: the w with start(w) <= x < start(w + 1)
__device__ __forceinline__ uint32_t worker_of(uint32_t x, uint32_t W, uint32_t G, uint32_t skew) {
    if (skew == 0) return (uint32_t)((((uint64_t)x + 1) * W - 1) / G);
    uint32_t lo = 0, hi = W - 1;                      // start(lo) <= x always; find the last w with start(w) <= x
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (start_of(mid, W, G, skew) <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---- geometry (one tile class: 64 output channels x 8x16 positions, four waves, two workgroups per CU)
constexpr int BCO = 64, TH = 8, TW = 16, BPOS = TH * TW;
constexpr int WCO = BCO / 32, WPOS = BPOS / 32;                       // per-wave MFMA tiles (2 x 4): the whole block tile
constexpr int HWD = TW + 2, HR = (TH + 2) * HWD;                      // halo: 10 x 18 rows
constexpr int RPP = 32, NH = (HR + RPP - 1) / RPP;                    // halo rows staged per pass / passes
constexpr int ROWP = 144;                                             // LDS pitch of a 128-byte halo row (see pbase)
constexpr int HALO_BYTES = ((HR * ROWP + 255) / 256) * 256;
constexpr int RED_BYTES = 4 * 4 * 4096;                               // cross-wave reduce round: [src wave][4 tiles] x 4 KB
constexpr int SLAB_BYTES = BCO * BPOS * 4;
// scheduling / ablation knobs (constants in the product build): conv3_knobs.h
#include "conv3_knobs.h"
constexpr int PSETS = MV2_C3_PSETS;
static_assert(PSETS == 2 || PSETS == 3, "position fragment sets");
template <typename E> struct TapMma { static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& acc) { Mma<E>::run(a, b, acc); } };
#if MV2_C3_AGPR == 1
template <> struct TapMma<BF16> {
    static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& acc) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
};
#endif
constexpr int RING = MV2_C3_RING;
static_assert(18 % RING == 0 && RING >= 2, "ring slots must divide the 18 fragments of a group");
constexpr int HWT = MV2_C3_HWT;                                       // tap at which the prefetched halo goes into the idle image (HSPLIT = 0); <= 7: tap 8 opens with the barrier

template <typename E, bool FUSED>
struct C3Lds {
    static constexpr int PITCH = BCO * E::kBytes + 16;                // row of a wave-private epilogue tile
    static constexpr int PRIV = 32 * PITCH + 128;                     // 32 positions + 32 floats of pooling scratch
    static constexpr int kMain = 2 * HALO_BYTES > RED_BYTES ? 2 * HALO_BYTES : RED_BYTES;
    static_assert(4 * PRIV <= kMain, "wave-private epilogue tiles live in the reduce region");
    static constexpr int kBias = kMain;                               // 64 fp32 bias values of the segment's channel tile
    static constexpr int kConst2 = kBias + 256;                       // fused stage: bias2[64], se_wk[64] (fp32), staged once
    static constexpr int kW2 = kConst2 + (FUSED ? 512 : 0);           // fused stage: 64 x 64 pointwise weights, staged once per workgroup
    static constexpr int kFlag = kW2 + (FUSED ? BCO * 128 : 0);
    static constexpr int kLds = kFlag + 16;
    static constexpr int kHtab = 2 * HALO_BYTES;                      // HTAB: [NH][256] row offsets behind the halo images (reduce region: dead during taps)
    static_assert(!MV2_C3_HTAB || kHtab + NH * 256 * 4 <= kMain, "halo offset table fits behind the halo images");
    // next_seg() writes the NEXT segment's table before the current segment's epilogue runs in the wave-private tiles at smem + wave * PRIV:
    // nothing between next_seg() and the next tap loop may touch [kHtab, kHtab + NH * 1024)
    static_assert(!MV2_C3_HTAB || 4 * PRIV <= kHtab, "wave-private epilogue tiles stay below the halo offset table");
    static_assert(kLds <= 80 * 1024, "two workgroups per CU");
};

}  // namespace

#ifdef MV2_EXPERIMENTS
// phase clocks (s_memtime) per workgroup: [0] prologue [1] tap loop [2] cross-wave reduce [3] partial/ticket [4] epilogue
// [5] segments [6] groups [7] whole kernel; written to g_conv3_prof when tools/k3_bench asks for it
__device__ unsigned long long* g_conv3_prof = nullptr;
#define C3_PROF_DECL unsigned long long pf_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long qf_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_ = __builtin_amdgcn_s_memtime(); const unsigned long long pt0_ = pt_;
#define C3_SUB(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); qf_[k] += n_ - pt_; pf_[4] += n_ - pt_; pt_ = n_; }
#define C3_PROF(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_[k] += n_ - pt_; pt_ = n_; }
#define C3_PROF_COUNT(k, v) pf_[k] += (v);
#define C3_PROF_END if (g_conv3_prof && threadIdx.x == 0) { pf_[7] = __builtin_amdgcn_s_memtime() - pt0_; qf_[7] = pt0_; pf_[5] |= ((unsigned long long)((__builtin_amdgcn_s_getreg(63508) & 0xf) << 8 | ((__builtin_amdgcn_s_getreg(63492) >> 8) & 0xff))) << 32; for (int i_ = 0; i_ < 8; ++i_) { g_conv3_prof[(size_t)blockIdx.x * 16 + i_] = pf_[i_]; g_conv3_prof[(size_t)blockIdx.x * 16 + 8 + i_] = qf_[i_]; } }
#else
#define C3_PROF_DECL
#define C3_PROF(k)
#define C3_PROF_COUNT(k, v)
#define C3_SUB(k)
#define C3_PROF_END
#endif

template <typename E, bool FUSED>
__global__ __launch_bounds__(256, 2) void conv3_kernel(const ConvParams p) {
    typedef C3Lds<E, FUSED> L;
    static_assert(!FUSED || std::is_same<E, BF16>::value, "fused second stage: bf16 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const mv2_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = tid & 7, r0 = tid >> 3;
    const int hi = lane >> 5, l31 = lane & 31;
    const int T = d.Ti, H = d.Hi, W_ = d.Wi;
    const int kt = d.kt;

    // ---- this block's range of groups (XCD-aware: the blocks of one XCD walk one contiguous eighth of the list,
    // i.e. few co tiles' weights: they stay in that XCD's L2)
    const uint32_t NW = gridDim.x, G = p.gtot;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t q = NW >> 3, r = NW & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    // Worker ranges.  Stream-K proper cuts tiles at arbitrary units (equal work, but every cut costs a slab round trip, a ticket
    // and a second cross-wave reduce).  Where a worker owns >= 4 tiles (p.aligned, decided by the host: C = 64 at batch 1, every
    // level up to C = 256 at batch 4) the range ends are snapped to the nearest tile start instead: no tile is cut, so no slab
    // traffic, no ticket, no partial reduce, at the price of up to half a tile of imbalance per range end.  Measured (k3_bench,
    // profiles/r02_k3_align.log): batch 4 C = 64 / 128 / 256 350 / 291 / 276 -> 319 / 273 / 247 us; batch 1 C = 64 94 -> 83 us.
    auto range_start = [&](uint32_t w) -> uint32_t {
        const uint32_t x = start_of(w, NW, G, (uint32_t)p.skew);
        if (!p.aligned || x >= G) return x;
        uint32_t gv;
        const TileInfo t = decode_unit(p, x, gv);
        const uint32_t vs = t.ng + p.delta;
        return 2 * gv < vs ? t.G0 : t.G0 + vs;
    };
    const uint32_t x0 = __builtin_amdgcn_readfirstlane(range_start(bid)), x1 = __builtin_amdgcn_readfirstlane(range_start(bid + 1));

    const uint32_t row_bytes = (uint32_t)d.Ci * E::kBytes;
    const uint32_t frame_bytes = (uint32_t)((int64_t)H * W_ * d.Ci * E::kBytes);
    const uint32_t w_step = (uint32_t)d.co_pad * 128u;
    const uint32_t tapstride = (uint32_t)p.nchunk * w_step;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.w_bytes, 0x00020000);
    // bias: zero-length descriptor when absent -> every load returns 0
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : (const float*)p.wp), 0,
                                                                          p.bias ? (uint32_t)d.co_pad * 4u : 0u, 0x00020000);
    // Halo image: row r of the 10 x 18 halo at r * 144 bytes (128 data + 16 pad).  A 32-lane MFMA operand reads 16 bytes
    // per lane at (row, this wave's chunk): the ds_read_b128 service groups of 16 lanes hit 16 distinct 16-byte slots of
    // the 256-byte bank row iff their rows are distinct mod 16 (9 = 144/16 is odd) -> conflict free with pos_of()'s
    // rotation, and the nine taps are nine IMMEDIATE offsets from one per-lane address (no per-tap VALU).
    const uint32_t lds_wr = (uint32_t)r0 * ROWP + (uint32_t)c * 16u;
    uint32_t pbase[WPOS];
#pragma unroll
    for (int bb = 0; bb < WPOS; ++bb) {
        int ph, pw; pos_of<TW>(bb * 32 + l31, ph, pw);
        pbase[bb] = (uint32_t)(ph * HWD + pw) * ROWP + (uint32_t)(wave * 2 + hi) * 16u;
    }
    if constexpr (FUSED) {      // pointwise weights -> LDS once (swizzled rows); visible after the first segment's barrier
        for (int q = tid; q < BCO * 8; q += 256) {
            const int row = q >> 3, ch = q & 7;
            *(i32x4*)(smem + L::kW2 + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = *(const i32x4*)((const char*)p.wp2 + (size_t)row * 128 + ch * 16);
        }
        if (tid < 16) *(f32x4*)(smem + L::kConst2 + tid * 16) = p.bias2 ? *(const f32x4*)(p.bias2 + tid * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        else if (tid < 32) *(f32x4*)(smem + L::kConst2 + tid * 16) = p.se_wk ? *(const f32x4*)(p.se_wk + (tid - 16) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- per-segment state: scalars of the tile + the incremental (chunk, time tap) walk over its groups
    struct Seg {
        TileInfo ti; uint32_t x, xe; int gb, ge;
        uint32_t hmask, hw_base, w_off0;
        int cur_chunk, cur_it, nxt_chunk, nxt_it;
        uint32_t cur_wb, nxt_wb;
    };
    i32x4 hreg[NH];
#if MV2_C3_HPRO
    i32x4 hpro[NH];      // the segment's first halo travels in its own registers (live only outside the tap loop)
#endif
    i32x4 wring[RING], pset[PSETS][WPOS];
    i32x4 bias_v;                                       // lanes 0..15 of the block: the segment's 64 bias values (part of the prefetch stream)

    auto advance = [&](const Seg& S, int& chunk, int& it) {
        const bool wrap = it + 1 == kt;
        it = wrap ? kt - S.ti.v : it + 1;
        chunk += wrap ? 1 : 0;
    };
    auto wbase_of = [&](int chunk, int it) { return (uint32_t)(it * 9 * p.nchunk + chunk) * w_step; };
    auto make_seg = [&](uint32_t x) {
        Seg S;
        uint32_t gv;
        S.ti = decode_unit(p, x, gv);
        S.x = x; S.xe = min(x1, S.ti.G0 + S.ti.ng + p.delta);
        S.gb = (int)max(gv, p.delta) - (int)p.delta;                      // real groups [gb, ge) of the tile; ge == gb: only virtual units
        S.ge = (int)max(S.xe - S.ti.G0, p.delta) - (int)p.delta;
        const int h0 = S.ti.thi * TH, w0 = S.ti.twi * TW;
        S.hmask = 0;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int hr = r0 + RPP * i;
            if (hr < HR) {
                const int hy = hr / HWD, hx = hr - hy * HWD;
                const int hin = h0 - 1 + hy, win = w0 - 1 + hx;
                if ((unsigned)hin < (unsigned)H && (unsigned)win < (unsigned)W_) S.hmask |= 1u << i;
            }
        }
        S.hw_base = (uint32_t)(((int64_t)(h0 - 1) * W_ + (w0 - 1)) * (int64_t)row_bytes);
        // weights: this wave's fragment of co-group a lives at  step * w_step + ((n0/32 + a) * 4 + wave) * 1024 + lane * 16
        S.w_off0 = (uint32_t)(((S.ti.co_tile * (BCO / 32)) * 4 + wave) * 1024 + lane * 16);
        // (chunk, time tap) of group gi of the tile: chunk-major; `cur` = group gi, `nxt` = gi + 1, clamped to the last group of
        // the segment (prefetches past the end re-read valid, cached addresses and are never consumed)
        const int v = S.ti.v;
        S.cur_chunk = v == 1 ? S.gb : (v == 2 ? (S.gb >> 1) : S.gb / 3);
        S.cur_it = (kt - v) + (S.gb - S.cur_chunk * v);
        S.nxt_chunk = S.cur_chunk; S.nxt_it = S.cur_it;
        if (S.gb + 1 < S.ge) advance(S, S.nxt_chunk, S.nxt_it);
        S.cur_wb = wbase_of(S.cur_chunk, S.cur_it); S.nxt_wb = wbase_of(S.nxt_chunk, S.nxt_it);
        return S;
    };
    // every load is unconditional in instruction count (spatial padding -> out-of-range offset -> zeros, no traffic)
    auto issue_halo = [&](const Seg& S, int chunk, int it, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        const int t_in = S.ti.t + it - d.pt;                        // >= 0 by construction of the group list
        const uint32_t base = (uint32_t)(((int64_t)S.ti.b * T + t_in) * (int64_t)frame_bytes) + (uint32_t)chunk * 128u + (uint32_t)c * 16u + S.hw_base;
        int r0_ = r0;
        asm volatile("" : "+v"(r0_));          // recompute the six row offsets per group: cheaper than six live registers (VGPR budget)
#if MV2_C3_HPRO
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));        // ... and nothing lane-derived is carried through the loop in a (spillable) register
        r0_ = tid_o >> 3;
        const uint32_t base_h = base - (uint32_t)c * 16u + (uint32_t)(tid_o & 7) * 16u;
#define C3_HBASE base_h
#else
#define C3_HBASE base
#endif
#pragma unroll
        for (int i = LO; i < HI; ++i) {
            const int hr = r0_ + RPP * i;
            const int hy = hr / HWD, hx = hr - hy * HWD;
            const bool ok = (S.hmask >> i) & 1u;
            const uint32_t vo = ok ? C3_HBASE + (uint32_t)(hy * W_ + hx) * row_bytes : p.in_bytes;
            hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)vo, 0, 0);
        }
    };
#if MV2_C3_HTAB
    // thread-private table of the segment's halo row offsets (see MV2_C3_HTAB above); rewritten by next_seg() for every segment
    auto write_htab = [&](const Seg& S) {
        const int h0 = S.ti.thi * TH, w0 = S.ti.twi * TW;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));        // opaque: otherwise the lane-only parts are hoisted to kernel entry and spilled around the tap loop
        const int r0_ = tid_o >> 3, c_ = tid_o & 7;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int hr = r0_ + RPP * i;
            const int hy = hr / HWD, hx = hr - hy * HWD;
            const bool ok = hr < HR && (unsigned)(h0 - 1 + hy) < (unsigned)H && (unsigned)(w0 - 1 + hx) < (unsigned)W_;
            // wrapping arithmetic: hw_base may be "negative" for edge tiles, the sum of a valid row is a true offset inside the frame
            const uint32_t e = ok ? S.hw_base + (uint32_t)(hy * W_ + hx) * row_bytes + (uint32_t)c_ * 16u : 0xffffffffu;
            *(uint32_t*)(smem + L::kHtab + (i * 256 + tid_o) * 4) = e;
        }
    };
    uint32_t htv[NH];
    auto read_htab = [&](auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));        // the table address is re-derived: nothing lane-derived is carried through the loop
#pragma unroll
        for (int i = LO; i < HI; ++i) htv[i] = *(const uint32_t*)(smem + L::kHtab + (i * 256 + tid_o) * 4);
    };
    auto issue_halo_tab = [&](const Seg& S, int chunk, int it, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        const int t_in = S.ti.t + it - d.pt;
        const uint32_t gbase = (uint32_t)(((int64_t)S.ti.b * T + t_in) * (int64_t)frame_bytes) + (uint32_t)chunk * 128u;      // uniform
#pragma unroll
        for (int i = LO; i < HI; ++i)
            hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)__builtin_elementwise_add_sat(htv[i], gbase), 0, 0);
    };
#endif
    auto write_halo = [&](char* img, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#if MV2_C3_HPRO
        int tid_w = tid;
        asm volatile("" : "+v"(tid_w));
        const int r0w = tid_w >> 3;
        const uint32_t wr = (uint32_t)r0w * ROWP + (uint32_t)(tid_w & 7) * 16u;
#pragma unroll
        for (int i = LO; i < HI; ++i)
            if (r0w + RPP * i < HR) *(i32x4*)(img + wr + i * RPP * ROWP) = hreg[i];
#else
#pragma unroll
        for (int i = LO; i < HI; ++i)
            if ((MV2_C3_HTAB && (i + 1) * RPP <= HR) || r0 + RPP * i < HR) *(i32x4*)(img + lds_wr + i * RPP * ROWP) = hreg[i];     // only the last pass is partial
#endif
    };
    auto issue_w1 = [&](const Seg& S, uint32_t so, int a, i32x4& dst) {
        dst = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)S.w_off0, (int)(so + (uint32_t)(a * 4096)), 0);
    };
#if MV2_C3_HPRO
    auto issue_halo_first = [&](const Seg& S, int chunk, int it) {
        const int t_in = S.ti.t + it - d.pt;
        const uint32_t base = (uint32_t)(((int64_t)S.ti.b * T + t_in) * (int64_t)frame_bytes) + (uint32_t)chunk * 128u + (uint32_t)c * 16u + S.hw_base;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int hr = r0 + RPP * i;
            const int hy = hr / HWD, hx = hr - hy * HWD;
            const bool ok = (S.hmask >> i) & 1u;
            const uint32_t vo = ok ? base + (uint32_t)(hy * W_ + hx) * row_bytes : p.in_bytes;
            hpro[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)vo, 0, 0);
        }
    };
    auto write_halo_first = [&](char* img) {
#pragma unroll
        for (int i = 0; i < NH; ++i)
            if (r0 + RPP * i < HR) *(i32x4*)(img + lds_wr + i * RPP * ROWP) = hpro[i];
    };
#endif
    auto issue_first = [&](const Seg& S) {     // halo of the segment's first group + the first RING weight fragments + its bias values
#if MV2_C3_HPRO
        issue_halo_first(S, S.cur_chunk, S.cur_it);
#elif MV2_C3_HTAB
        read_htab(IC<0>{}, IC<NH>{});
        issue_halo_tab(S, S.cur_chunk, S.cur_it, IC<0>{}, IC<NH>{});
#else
        issue_halo(S, S.cur_chunk, S.cur_it, IC<0>{}, IC<NH>{});
#endif
        bias_v = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)(tid < 16 ? (uint32_t)(S.ti.co_tile * BCO + tid * 4) * 4u : 0xfffffff0u), 0, 0);
#pragma unroll
        for (int f = 0; f < RING; ++f) issue_w1(S, S.cur_wb + (uint32_t)(f / WCO) * tapstride, f % WCO, wring[f]);
    };
    auto read_p = [&](uint32_t img, int tp, auto set_c) {     // img: LDS byte offset of the halo image
        constexpr int SET = decltype(set_c)::value;
        const int toff = ((tp / 3) * HWD + (tp % 3)) * ROWP;
#pragma unroll
        for (int bb = 0; bb < WPOS; ++bb) pset[SET][bb] = *(const i32x4*)(smem + (pbase[bb] + img) + toff);
    };

#if MV2_C3_AGPR == 2
    { float z_ = 0.f; asm volatile("; this kernel uses the accumulator file" : "+a"(z_)); }      // makes the compiler select the AGPR form of its own MFMAs
#endif
    const int prio_par = MV2_C3_PRIO ? (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0) : 0;     // the CU's pair is (w, w + NW / 2)
    C3_PROF_DECL
    // first segment with real groups (a range may begin or end inside a tile's virtual units)
    auto next_seg = [&](uint32_t x, Seg& O) {
        while (x < x1) {
            O = make_seg(x);
#if MV2_C3_HTAB
            if (O.ge > O.gb) { write_htab(O); return true; }
#else
            if (O.ge > O.gb) return true;
#endif
            x = O.xe;
        }
        return false;
    };
    Seg S;
    if (!next_seg(x0, S)) { C3_PROF_END return; }
#if MV2_C3_LATEPF != 2
    issue_first(S);
#endif
    for (;;) {
#if MV2_C3_LATEPF == 2
        issue_first(S);      // experiment: no prefetch across the segment boundary (nothing loop-carried in registers between segments)
#endif
        C3_PROF(4)
        f32x16 acc[WCO][WPOS];
#pragma unroll
        for (int a = 0; a < WCO; ++a)
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;
#if MV2_C3_HPRO
        write_halo_first(smem);
#else
        write_halo(smem, IC<0>{}, IC<NH>{});
#endif
        __syncthreads();
        read_p(0u, 0, IC<0>{});
        C3_PROF(0)

        // one tap.  Fragment f = 2 TP + a of the group lives in ring slot f % RING, requested RING fragments ago; pset[TP % PSETS] holds
        // the position fragments, read one tap ago.  The reload of a slot is issued right behind the last MFMA that reads it, so
        // loads, LDS reads and MFMAs interleave instead of forming phases.
        auto tap = [&](uint32_t img, uint32_t other, auto tp_c) {
            constexpr int TP = decltype(tp_c)::value;
            constexpr int SET = TP % PSETS, NSET = (TP + 1) % PSETS;
            if constexpr (MV2_C3_PRIO == 2) { if ((TP & 1) ^ prio_par) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
            if constexpr (MV2_C3_HSPLIT) {
                if constexpr (TP == 0) issue_halo(S, S.nxt_chunk, S.nxt_it, IC<0>{}, IC<NH / 2>{});
                if constexpr (TP == 4) { write_halo(smem + other, IC<0>{}, IC<NH / 2>{}); issue_halo(S, S.nxt_chunk, S.nxt_it, IC<NH / 2>{}, IC<NH>{}); }
                if constexpr (TP == 8) write_halo(smem + other, IC<NH / 2>{}, IC<NH>{});
            } else if constexpr (MV2_C3_HTAB == 1) {
#if MV2_C3_HTAB
                if constexpr (TP == 0) read_htab(IC<0>{}, IC<NH>{});
#endif
            } else if constexpr (MV2_C3_HTAB == 2) {
#if MV2_C3_HTAB
                if constexpr (TP < 3) read_htab(IC<2 * TP>{}, IC<2 * TP + 2>{});
#endif
            } else {
                // ABL 5 / 6 / 7 (experiments, TIMING ONLY): 5 = the next group's halo is never requested (the stale registers are still
                // written), 6 = nor written to LDS, 7 = and the per-group barrier is dropped, 8 = 4 + 7 (bare MFMA stream): which part sets the solo rate
                if constexpr (TP == 0 && MV2_C3_ABL < 5) issue_halo(S, S.nxt_chunk, S.nxt_it, IC<0>{}, IC<NH>{});
            }
            if constexpr (TP == 8 && MV2_C3_ABL != 7 && MV2_C3_ABL != 8) __syncthreads();       // the next group's image is complete; this one has no reader left
            if constexpr (MV2_C3_ABL == 3 || MV2_C3_ABL == 4 || MV2_C3_ABL == 8) {
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) { pset[NSET][bb] = pset[SET][bb]; asm volatile("" : "+v"(pset[NSET][bb])); }
            } else if constexpr (TP < 8) read_p(img, TP + 1, IC<NSET>{});
            else read_p(other, 0, IC<NSET>{});
            __builtin_amdgcn_sched_barrier(0);            // keep the fragment reads of the NEXT tap at the top (the scheduler sinks them to their use)
            auto frag = [&](auto a_c) {
                constexpr int A = decltype(a_c)::value, F = TP * WCO + A, SLOT = F % RING, G = F + RING;
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) TapMma<E>::run(wring[SLOT], pset[SET][bb], acc[A][bb]);
                const uint32_t so = G < 9 * WCO ? S.cur_wb + (uint32_t)(G / WCO) * tapstride : S.nxt_wb + (uint32_t)((G - 9 * WCO) / WCO) * tapstride;
                if constexpr (!(MV2_C3_ABL == 2 || MV2_C3_ABL == 4 || MV2_C3_ABL == 8 || (MV2_C3_ABL == 1 && A == 1))) issue_w1(S, so, G % WCO, wring[SLOT]);
                else asm volatile("" : "+v"(wring[SLOT]));       // keep the slot live and opaque
            };
            static_assert(WCO == 2, "two channel tiles per wave");
            frag(IC<0>{});
#if MV2_C3_HTAB
            // the table values have had four MFMAs of flight; the loads go out under the second batch
            if constexpr (MV2_C3_HTAB == 1 && !MV2_C3_HSPLIT && TP == 0) issue_halo_tab(S, S.nxt_chunk, S.nxt_it, IC<0>{}, IC<NH>{});
            if constexpr (MV2_C3_HTAB == 2 && !MV2_C3_HSPLIT && TP < 3) issue_halo_tab(S, S.nxt_chunk, S.nxt_it, IC<2 * TP>{}, IC<2 * TP + 2>{});
#endif
            frag(IC<1>{});
            if constexpr (MV2_C3_HTAB == 2 && !MV2_C3_HSPLIT) {
                if constexpr (TP >= HWT && TP < HWT + 3) write_halo(smem + other, IC<2 * (TP - HWT)>{}, IC<2 * (TP - HWT) + 2>{});
            } else
            if constexpr (!MV2_C3_HSPLIT && TP == HWT && MV2_C3_ABL < 6) write_halo(smem + other, IC<0>{}, IC<NH>{});   // image of group gi-1: its last reader passed the previous barrier
            if constexpr (PSETS == 2 && TP == 8) {      // the next group's tap 0 reads set 0
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) pset[0][bb] = pset[1][bb];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int gi = S.gb; gi < S.ge; ++gi) {
            const int par = (gi - S.gb) & 1;
            if constexpr (MV2_C3_PRIO == 1) { if ((gi + prio_par) & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
            if constexpr (MV2_C3_PRIO == 3) { if ((int)((__builtin_amdgcn_s_memtime() >> 13) & 1) ^ prio_par) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
            const uint32_t img = (uint32_t)(par * HALO_BYTES), other = (uint32_t)((par ^ 1) * HALO_BYTES);
            tap(img, other, IC<0>{}); tap(img, other, IC<1>{}); tap(img, other, IC<2>{});
            tap(img, other, IC<3>{}); tap(img, other, IC<4>{}); tap(img, other, IC<5>{});
            tap(img, other, IC<6>{}); tap(img, other, IC<7>{}); tap(img, other, IC<8>{});
            S.cur_wb = S.nxt_wb;
            if (gi + 2 < S.ge) { advance(S, S.nxt_chunk, S.nxt_it); S.nxt_wb = wbase_of(S.nxt_chunk, S.nxt_it); }
        }
        if constexpr (MV2_C3_PRIO != 0) __builtin_amdgcn_s_setprio(3);     // fixed phases: short, and they gate the next tile
        if (tid < 16) *(i32x4*)(smem + L::kBias + tid * 16) = bias_v;     // epilogue reads bias from LDS: no global load behind the next prefetch
        __syncthreads();                                    // halo images are dead: LDS becomes the reduce region
        C3_PROF(1)
        C3_PROF_COUNT(5, 1)
        C3_PROF_COUNT(6, (unsigned long long)(S.ge - S.gb))

        // Lane-derived values of the code below are re-derived from an opaque copy of the lane id: otherwise loop-invariant
        // code motion hoists every address of the reduce / ticket / epilogue to kernel entry, they are spilled around the tap loop,
        // and each scratch reload waits vmcnt(0) behind the stores and prefetches in flight (measured: ~10k clocks per tile).
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int hi_o = lane_o >> 5, l31_o = lane_o & 31;
        // ---- cross-wave reduce: wave w ends up with BOTH channel tiles of position tile w (all 64 channels of its 32
        // positions), summed over the 4 k-slices in wave order.  Two rounds (round = channel tile a): every wave drops its four
        // partial tiles of the round into LDS (straight from the accumulator registers), then reads the four partials of the
        // one tile it owns.  Ownership only enters through an LDS address: no accumulator is indexed at run time.
        f32x16 acc_e[WCO];
#if MV2_C3_OWNQ
        // experiment: the partial a wave owns anyway (position tile == its wave index) stays in registers: a quarter less LDS traffic in
        // both directions; the sum keeps the k-slice order 0..3, so results are bitwise those of the product form
#pragma unroll
        for (int a = 0; a < WCO; ++a) {
            f32x16 own;
            if (wave == 0) own = acc[a][0]; else if (wave == 1) own = acc[a][1]; else if (wave == 2) own = acc[a][2]; else own = acc[a][3];
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb) {
                if (bb == wave) continue;
                char* dst = smem + (wave * 4 + bb) * 4096 + lane_o * 16;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 vv = {acc[a][bb][4 * r4], acc[a][bb][4 * r4 + 1], acc[a][bb][4 * r4 + 2], acc[a][bb][4 * r4 + 3]};
                    *(f32x4*)(dst + r4 * 1024) = vv;
                }
            }
            __syncthreads();
            f32x16 sum;
#pragma unroll
            for (int src = 0; src < 4; ++src) {
                const char* sp = smem + (src * 4 + wave) * 4096 + lane_o * 16;
                f32x16 part;
                if (src == wave) part = own;
                else {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 vv = *(const f32x4*)(sp + r4 * 1024);
                        part[4 * r4] = vv[0]; part[4 * r4 + 1] = vv[1]; part[4 * r4 + 2] = vv[2]; part[4 * r4 + 3] = vv[3];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[r] = src == 0 ? part[r] : sum[r] + part[r];
            }
            acc_e[a] = sum;
            __syncthreads();
        }
#else
#pragma unroll
        for (int a = 0; a < WCO; ++a) {
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb) {
                char* dst = smem + (wave * 4 + bb) * 4096 + lane_o * 16;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 vv = {acc[a][bb][4 * r4], acc[a][bb][4 * r4 + 1], acc[a][bb][4 * r4 + 2], acc[a][bb][4 * r4 + 3]};
                    *(f32x4*)(dst + r4 * 1024) = vv;
                }
            }
            __syncthreads();
            f32x16 sum;
#pragma unroll
            for (int src = 0; src < 4; ++src) {
                const char* sp = smem + (src * 4 + wave) * 4096 + lane_o * 16;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 vv = *(const f32x4*)(sp + r4 * 1024);
                    if (src == 0) { sum[4 * r4] = vv[0]; sum[4 * r4 + 1] = vv[1]; sum[4 * r4 + 2] = vv[2]; sum[4 * r4 + 3] = vv[3]; }
                    else { sum[4 * r4] += vv[0]; sum[4 * r4 + 1] += vv[1]; sum[4 * r4 + 2] += vv[2]; sum[4 * r4 + 3] += vv[3]; }
                }
            }
            acc_e[a] = sum;
            __syncthreads();
        }
#endif
        // ---- the next segment's first halo image and weight taps start their flight now (the accumulators are dead, the
        // registers of the loads are free until the next tap loop): they land under this segment's ticket / epilogue
        Seg N = S;
        const bool has_next = next_seg(S.xe, N);
#if MV2_C3_LATEPF
        // experiment (AGPR form, 128 arch VGPRs): the next segment's prefetch waits until the epilogue has released its registers
#else
        if (has_next) issue_first(N);
#endif
        C3_PROF(2)

        // ---- a tile cut by a range boundary: publish the partial, take a ticket, the last arriver sums every part.
        // Slab stores are write-through (sc1) and the last arriver reads with sc1 loads: no L2 write-back / invalidate
        // fence on either side (cdna_hip_programming.md §5 "in-launch split-K reduction", the write-through form).
        const TileInfo& ti = S.ti;
        const bool partial = S.gb != 0 || (uint32_t)S.ge != ti.ng;
        bool finish = true;
        if (partial) {
            constexpr int WSLAB = SLAB_BYTES / 4;     // bytes per wave inside a slab
            const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, p.ws_bytes, 0x00020000);
            const uint32_t slot = 2 * bid + (S.x == x0 ? 0u : 1u);      // S.x == x0 <=> this is the worker's first segment with real groups
            const uint32_t soff = slot * (uint32_t)SLAB_BYTES + (uint32_t)wave * WSLAB + (uint32_t)lane_o * 16u;
#pragma unroll
            for (int a = 0; a < WCO; ++a)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    i32x4 vv = {__float_as_int(acc_e[a][4 * r4]), __float_as_int(acc_e[a][4 * r4 + 1]),
                                __float_as_int(acc_e[a][4 * r4 + 2]), __float_as_int(acc_e[a][4 * r4 + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(vv, rs_ws, (int)(soff + (uint32_t)((a * 4 + r4) * 1024)), 0, 16 /* sc1 */);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains before the ticket
            __syncthreads();
            const uint32_t w_first = worker_of(ti.G0 + p.delta, NW, G, (uint32_t)p.skew), w_last = worker_of(ti.G0 + p.delta + ti.ng - 1, NW, G, (uint32_t)p.skew);
            const uint32_t nparts = w_last - w_first + 1;
            uint32_t* flag = (uint32_t*)(smem + L::kFlag);
            if (wave == 0 && lane_o == 0) {
                const uint32_t ticket = __hip_atomic_fetch_add(p.counters + ti.L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t last = ticket == nparts - 1 ? 1u : 0u;
                if (last) __hip_atomic_store(p.counters + ti.L, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                *flag = last;
            }
            __syncthreads();
            finish = *flag != 0;
            if (finish) {
                const uint32_t s_first = start_of(w_first, NW, G, (uint32_t)p.skew);
#pragma unroll
                for (int a = 0; a < WCO; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_e[a][r] = 0.f;
                auto slab_off = [&](uint32_t pi) {
                    const uint32_t wi = w_first + pi;
                    const uint32_t sl = 2 * wi + ((pi == 0 && s_first < ti.G0) ? 1u : 0u);
                    return sl * (uint32_t)SLAB_BYTES + (uint32_t)wave * WSLAB + (uint32_t)lane_o * 16u;
                };
                // parts are summed in part order (bitwise reproducible); two parts' loads are in flight together
                for (uint32_t pi = 0; pi < nparts; pi += 2) {
                    const bool two = pi + 1 < nparts;
                    const uint32_t so0 = slab_off(pi), so1 = two ? slab_off(pi + 1) : p.ws_bytes;     // odd tail: out of range -> zeros
                    i32x4 v0[WCO * 4], v1[WCO * 4];
#pragma unroll
                    for (int q = 0; q < WCO * 4; ++q) v0[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (int)(so0 + (uint32_t)(q * 1024)), 0, 16 /* sc1 */);
#pragma unroll
                    for (int q = 0; q < WCO * 4; ++q) v1[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (int)(so1 + (uint32_t)(q * 1024)), 0, 16 /* sc1 */);
#pragma unroll
                    for (int q = 0; q < WCO * 4; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc_e[q >> 2][4 * (q & 3) + j] += __int_as_float(v0[q][j]);
#pragma unroll
                    for (int q = 0; q < WCO * 4; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc_e[q >> 2][4 * (q & 3) + j] += __int_as_float(v1[q][j]);
                }
            }
            // no barrier: the flag word is rewritten only after the next segment's barriers
        }
        C3_PROF(3)

        // ---- epilogue of a finished tile: wave-private (this wave owns every channel of its 32 positions), no block barrier.
        // acc_e[a][4 g + j] = channel a*32 + 8 g + 4 hi + j of position l31 of the wave's position tile.
        if (finish) {
            char* priv = smem + wave * L::PRIV;
            const int h0 = ti.thi * TH, w0 = ti.twi * TW;
            const int64_t frame_row0 = ((int64_t)ti.b * T + ti.t) * H;
            const int n0 = ti.co_tile * BCO;
            auto put_tile = [&](f32x16 (&v16)[WCO], int bias_off, bool elu) {     // bias_off: LDS offset of 64 fp32 bias values
                f32x4 bvs[WCO * 4];                       // all eight bias reads first: one LDS latency instead of eight in a chain
#pragma unroll
                for (int q = 0; q < WCO * 4; ++q) bvs[q] = *(const f32x4*)(smem + bias_off + ((q >> 2) * 32 + 8 * (q & 3) + 4 * hi_o) * 4);
#pragma unroll
                for (int a = 0; a < WCO; ++a)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = a * 32 + 8 * g + 4 * hi_o;
                        const f32x4 bv = bvs[a * 4 + g];
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xv = v16[a][4 * g + j] + bv[j];
                            const float e = Act<E>::elu(fminf(xv, 0.f));
                            v[j] = (elu && xv < 0.f) ? e : xv;
                        }
                        lds_put4<E>(priv + l31_o * L::PITCH + col * E::kBytes, v);
                    }
            };
            put_tile(acc_e, L::kBias, d.act == MV2_ACT_ELU);
            C3_SUB(0)
            if constexpr (FUSED) {
                // second GEMM out of LDS: out[co][pos] = sum_k W2[co][k] * tile[pos][k]; K = 64 = one 128-byte step
#pragma unroll
                for (int a = 0; a < WCO; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_e[a][r] = 0.f;
                __builtin_amdgcn_wave_barrier();
                const uint32_t swz = (uint32_t)((l31_o >> 1) & 7);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const uint32_t ck = (uint32_t)(ks * 2 + hi_o);
                    const i32x4 fp = *(const i32x4*)(priv + l31_o * L::PITCH + ck * 16);
#pragma unroll
                    for (int a = 0; a < WCO; ++a) {
                        const i32x4 fw = *(const i32x4*)(smem + L::kW2 + (a * 32 + l31_o) * 128 + ((ck ^ swz) << 4));
                        Mma<E>::run(fw, fp, acc_e[a]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                C3_SUB(1)
                put_tile(acc_e, L::kConst2, p.act2 == MV2_ACT_ELU);
                C3_SUB(2)
            }
            __builtin_amdgcn_wave_barrier();
            // whole rows out: 8 lanes per row (128 B bf16 / 16 lanes at fp32), 16 bytes per lane
            constexpr int CPR = BCO * E::kBytes / 16, RPI = 64 / CPR;            // chunks per row, rows per instruction
            {
                const int ch = lane_o % CPR, rr = lane_o / CPR;
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i) {
                    const int row = rr + i * RPI;
                    int ph, pw; pos_of<TW>(wave * 32 + row, ph, pw);
                    const bool ok = (h0 + ph) < H && (w0 + pw) < W_ && (n0 + ch * E::kVec) < d.co;
                    if (ok) {
                        const int64_t m = (frame_row0 + (h0 + ph)) * W_ + (w0 + pw);
                        *(i32x4*)((char*)p.out + (m * d.out_c + n0 + ch * E::kVec) * E::kBytes) = *(const i32x4*)(priv + row * L::PITCH + ch * 16);
                    }
                }
            }
            C3_SUB(3)
            if constexpr (FUSED) {
                if (p.se_part) {
                    // SqueezeExcite pooling partial of this wave's 2 x 16 strip of positions (reference SqueezeExcite.forward
                    // M:226-231; mv2_se_pool's record {max logit, sum exp, exp-weighted channel sums}), from the bf16 tile in LDS
                    float* ev = (float*)(priv + 32 * L::PITCH);
                    const int row = lane_o >> 1, half = lane_o & 1;
                    float dot = 0.f;
#pragma unroll
                    for (int c8 = 0; c8 < 4; ++c8) {
                        float xv[8];
                        Chunk<E>::unpack(*(const i32x4*)(priv + row * L::PITCH + (half * 32 + c8 * 8) * 2), xv);
                        const f32x4 w0v = *(const f32x4*)(smem + L::kConst2 + 256 + (half * 32 + c8 * 8) * 4),
                                    w1v = *(const f32x4*)(smem + L::kConst2 + 256 + (half * 32 + c8 * 8 + 4) * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) dot += xv[k] * w0v[k];
#pragma unroll
                        for (int k = 0; k < 4; ++k) dot += xv[4 + k] * w1v[k];
                    }
                    dot += __shfl_xor(dot, 1, 64);
                    int ph, pw; pos_of<TW>(wave * 32 + row, ph, pw);
                    const bool valid = (h0 + ph) < H && (w0 + pw) < W_;
                    const float logit = valid ? dot + p.se_bk : -INFINITY;
                    const float mx = wave_max(logit);
                    const float e = (valid && half == 0) ? expf(logit - mx) : 0.f;
                    const float sw = wave_sum(e);
                    if (half == 0) ev[row] = e;
                    __builtin_amdgcn_wave_barrier();
                    float a0 = 0.f, a1 = 0.f;                     // lane = channel
#pragma unroll 4
                    for (int r = 0; r < 32; r += 2) {
                        a0 += ev[r] * bf2f(*(const unsigned short*)(priv + r * L::PITCH + lane_o * 2));
                        a1 += ev[r + 1] * bf2f(*(const unsigned short*)(priv + (r + 1) * L::PITCH + lane_o * 2));
                    }
                    const int srow = ti.thi * 4 + wave;            // strip row inside the frame (2 image rows each)
                    if (srow * 2 < H) {                            // strips below the frame do not exist in the record list
                        float* dst = p.se_part + ((((int64_t)ti.b * T + ti.t) * ((H + 1) / 2) + srow) * p.tiles_w + ti.twi) * (BCO + 2);
                        dst[2 + lane_o] = a0 + a1;
                        if (lane_o == 0) { dst[0] = mx; dst[1] = sw; }
                    }
                    C3_SUB(4)
                }
            }
        }
        if (!has_next) break;
#if MV2_C3_LATEPF == 1
        issue_first(N);
#endif
        C3_SUB(5)
        __syncthreads();                                    // every wave left its private tile: LDS becomes the halo image again
        C3_SUB(6)
        S = N;
    }
    C3_PROF(4)
    C3_PROF_END
}

// ------------------------------------------------------------------------------- host side
namespace {

struct Plan3 {
    int tiles_h, tiles_w, nchunk, n_co_tiles;
    int64_t M, n_tiles;
    uint64_t gtot, units;      // real 9-tap groups; work-list units = groups + delta per tile
    uint32_t delta;
    int workers;
    int aligned;               // worker ranges snapped to tile starts (no cut tiles)
    int skew;                  // per mille of extra work for the older workgroup of a CU's pair (start_of)
    size_t slab_bytes;
};

inline int cus_of_current_device() {      // not cached: a process may drive several devices
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

inline bool eligible3(const mv2_conv_desc& d) {
    if (d.dtype != MV2_BF16 && d.dtype != MV2_F32) return false;
    const int es = d.dtype == MV2_BF16 ? 2 : 4;
    return d.kh == 3 && d.kw == 3 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.ph == 1 && d.pw == 1 && d.kt >= 1 && d.kt <= 3 &&
           d.pt == d.kt - 1 && d.check_bounds && (d.Ci * es) % 128 == 0 && d.To == d.Ti && d.Ho == d.Hi && d.Wo == d.Wi &&
           d.nsteps == d.kt * 9 * ((d.Ci * es) / 128) && (d.act == MV2_ACT_NONE || d.act == MV2_ACT_ELU) && d.co >= 64 && d.co_pad % 128 == 0 &&
           d.co <= d.co_pad && (d.store == MV2_STORE_ROWS || (d.store == MV2_STORE_ROWS_F32 && d.dtype == MV2_BF16)) &&
           (d.co % (16 / es)) == 0 && (d.out_c % (16 / es)) == 0 && d.out_c >= d.co;
}

inline Plan3 plan3(const mv2_conv_desc& d) {
    Plan3 P;
    const int es = d.dtype == MV2_BF16 ? 2 : 4;
    P.tiles_h = (d.Ho + TH - 1) / TH;
    P.tiles_w = (d.Wo + TW - 1) / TW;
    P.nchunk = (d.Ci * es) / 128;
    P.n_co_tiles = (d.co + BCO - 1) / BCO;
    P.M = (int64_t)d.B * d.To * d.Ho * d.Wo;
    P.n_tiles = (int64_t)P.n_co_tiles * d.B * d.Ti * P.tiles_h * P.tiles_w;
    const int pt = d.kt - 1, T = d.Ti;
    const int64_t cT = T <= pt ? (int64_t)T * (T + 1) / 2 : (int64_t)pt * (pt + 1) / 2 + (int64_t)(T - pt) * (pt + 1);
    P.gtot = (uint64_t)P.n_co_tiles * d.B * P.tiles_h * P.tiles_w * P.nchunk * (uint64_t)cT;
    // per-tile fixed cost (cross-wave reduce, finish, epilogue ~ 10k clocks) in units of one 9-tap group (~ 2.4-4.8k clocks)
    P.delta = 3;
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_C3_DELTA")) P.delta = (uint32_t)atoi(e);
#endif
    P.units = P.gtot + (uint64_t)P.delta * (uint64_t)P.n_tiles;
    const int cus = cus_of_current_device();
    P.workers = (int)std::min<uint64_t>((uint64_t)cus * 2, P.gtot);      // two workgroups per CU ...
    // ... unless that leaves a worker fewer than 4 groups: such launches are chains of fixed latencies (prefetch, ticket,
    // part sums), one workgroup per CU with twice the range is faster (measured at C = 512: T = 5, 3 groups per worker:
    // 31.6 -> 28.6 us; T = 10, 6.75 groups: 47.1 -> 49.8 us)
    if (P.gtot < (uint64_t)P.workers * 4) P.workers = (int)std::min<uint64_t>((uint64_t)cus, P.gtot);
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_C3_WORKERS")) { const int v = atoi(e); if (v > 0) P.workers = (int)std::min<uint64_t>((uint64_t)v, P.gtot); }
#endif
    // tile-aligned ranges where a worker owns at least four tiles (see the kernel): half a tile of imbalance is then <= 12 %
    P.aligned = P.n_tiles >= (int64_t)4 * P.workers ? 1 : 0;
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_C3_ALIGN")) P.aligned = atoi(e) != 0;
#endif
    // unequal ranges inside a CU's pair of workgroups (see start_of): only where there IS a pair (two workgroups per CU, worker count a
    // multiple of 16 so that the XCD segments split into an older and a younger half)
    P.skew = 0;
    if (P.workers == 2 * cus && (P.workers & 15) == 0) P.skew = MV2_C3_SKEW_DEFAULT;
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_C3_SKEW")) P.skew = (P.workers == 2 * cus && (P.workers & 15) == 0) ? atoi(e) : 0;
#endif
    // two slab slots per worker (a worker's range cuts at most its first and its last tile)
    P.slab_bytes = P.workers > 1 ? (size_t)2 * P.workers * SLAB_BYTES : 0;
    return P;
}

template <typename E, bool FUSED>
int launch3(const ConvParams& p, int workers, hipStream_t st) {
    typedef C3Lds<E, FUSED> L;
    // set on every call: the attribute is per device, a cached flag would be wrong for a second device of the process
    hipError_t e = hipFuncSetAttribute((const void*)conv3_kernel<E, FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3_kernel<E, FUSED>), dim3((unsigned)workers), dim3(256), L::kLds, st, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

}  // namespace

#ifdef MV2_EXPERIMENTS
extern "C" int mv2_conv3_set_profile(void* device_buffer) {      // experiments library only (tools/k3_bench)
    unsigned long long* v = (unsigned long long*)device_buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv3_prof), &v, sizeof(v));
}
#endif

// Which kernel serves a launch: the position-tile kernel of conv3p.hip takes every bf16 shape, this file's K-split kernel the exact-fp32
// mode.  A pure function of the descriptor (the experiments library lets tools/k3_bench force either: MV2_C3_KERNEL=old|new).
static bool use_conv3p(const mv2_conv_desc& d) {
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_C3_KERNEL")) return e[0] == 'n' && conv3p_eligible(d);
#endif
    return conv3p_eligible(d);
}

extern "C" int mv2_conv3_eligible(const mv2_conv_desc* d) { return d && eligible3(*d) ? 1 : 0; }

extern "C" int mv2_conv3_fusable(const mv2_conv_desc* d) {
    return d && eligible3(*d) && d->dtype == MV2_BF16 && d->store == MV2_STORE_ROWS && d->co == BCO && d->out_c == d->co ? 1 : 0;
}

extern "C" size_t mv2_conv3_workspace_bytes(const mv2_conv_desc* d) {
    if (!d || !eligible3(*d)) return 0;
#ifdef MV2_EXPERIMENTS
    return std::max(plan3(*d).slab_bytes, conv3p_eligible(*d) ? conv3p_workspace_bytes(*d) : (size_t)0);      // the harness switches kernels between calls
#else
    return use_conv3p(*d) ? conv3p_workspace_bytes(*d) : plan3(*d).slab_bytes;
#endif
}

extern "C" size_t mv2_conv3_counter_bytes(const mv2_conv_desc* d) {
    if (!d || !eligible3(*d)) return 0;
    size_t extra = 0;
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_CONV3_EXTRA_COUNTERS")) extra = (size_t)atoi(e);      // tools/experiments/conv3p_persistent.hip keeps its queue words behind the tickets
#endif
    return ((size_t)plan3(*d).n_tiles + extra) * sizeof(uint32_t);
}

extern "C" int mv2_conv3_tiles(const mv2_conv_desc* d, int* tile_h, int* tile_w) {
    if (!d || !mv2_conv3_fusable(d)) return MV2_EINVAL;
    if (tile_h) *tile_h = 2;
    if (tile_w) *tile_w = TW;
    return ((d->Ho + 1) / 2) * ((d->Wo + TW - 1) / TW);
}

extern "C" int mv2_conv3(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp3, const float* bias,
                         const void* wp2, const float* bias2, int act2,
                         const float* se_wk, float se_bk, float* se_part, void* out, void* workspace, size_t workspace_bytes,
                         uint32_t* counters, size_t counter_bytes, void* stream) {
    if (!d || !in || !wp3 || !out) return MV2_EINVAL;
    if (!eligible3(*d)) return MV2_EINVAL;
    if (in_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    const Plan3 P = plan3(*d);
    if (P.M >= (int64_t)0x7fffffff || P.units >= 0xffffffffull || P.n_tiles >= (int64_t)0x7fffffff || P.slab_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    const size_t w_bytes = (size_t)d->nsteps * d->co_pad * 128;
    if (w_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    if (wp2) {       // fused second stage: bf16, the block owns every channel of its positions
        if (!mv2_conv3_fusable(d)) return MV2_EINVAL;
        if (act2 != MV2_ACT_NONE && act2 != MV2_ACT_ELU) return MV2_EINVAL;
    } else if (bias2 || se_part) return MV2_EINVAL;
    if (se_part && !se_wk) return MV2_EINVAL;
    if (d->store == MV2_STORE_ROWS_F32 && !conv3p_eligible(*d)) return MV2_EINVAL;      // fp32 rows out of a bf16 launch: conv3p's epilogue only
    if (use_conv3p(*d) || d->store == MV2_STORE_ROWS_F32)
        return conv3p_launch(d, in, in_bytes, wp3, bias, wp2, bias2, act2, se_wk, se_bk, se_part, out, workspace, workspace_bytes, counters, counter_bytes,
                             (hipStream_t)stream);
    if (P.slab_bytes > 0 && (!workspace || workspace_bytes < P.slab_bytes)) return MV2_EWORKSPACE;
    if (P.workers > 1 && (!counters || counter_bytes < (size_t)P.n_tiles * sizeof(uint32_t))) return MV2_EWORKSPACE;
    ConvParams p{};
    p.d = *d; p.in = in; p.wp = wp3; p.segs = nullptr; p.bias = bias; p.rowscale = nullptr; p.residual = nullptr;
    p.out = out; p.ws = (float*)workspace;
    p.wp2 = wp2; p.bias2 = bias2; p.act2 = act2;
    p.se_wk = se_wk; p.se_bk = se_bk; p.se_part = se_part;
    p.in_bytes = (uint32_t)in_bytes; p.w_bytes = (uint32_t)w_bytes;
    p.M = P.M; p.n_co_tiles = P.n_co_tiles; p.n_pos_tiles = (int)(P.n_tiles / P.n_co_tiles);
    p.splitk = 1; p.tiles_h = P.tiles_h; p.tiles_w = P.tiles_w; p.nchunk = P.nchunk; p.chunks_per_split = P.nchunk;
    p.counters = counters; p.gtot = (uint32_t)P.units; p.delta = P.delta; p.ws_bytes = (uint32_t)P.slab_bytes;
    p.aligned = P.aligned;
    p.skew = P.skew;
    hipStream_t st = (hipStream_t)stream;
    if (wp2) return launch3<BF16, true>(p, P.workers, st);
    return d->dtype == MV2_BF16 ? launch3<BF16, false>(p, P.workers, st) : launch3<F32, false>(p, P.workers, st);
}
