// mv2_conv3 — the ResidualUnit kt x 3 x 3 causal convolution (reference magvit2_pytorch.py:937, CausalConv3d
// :892-928; ~81 % of the path's FLOPs), optionally fused with the unit's 1x1x1 conv + ELU (:938-940) and the
// SqueezeExcite pooling partials (:226-231).
//
// Decomposition (gfx950: 256 CUs, 4 SIMDs per CU, 512 registers per lane at one wave per SIMD):
//   * ONE 4-wave workgroup per CU.  A block owns a BCO x (TH x TW) output tile (128co x 8x16 or 64co x 16x16);
//     the four waves split K: wave w multiplies k-slice w (16 of the 64 channels of every 128-byte chunk) of EVERY
//     tap and keeps the whole tile in its own 256 accumulator registers (4x4 or 2x8 MFMA tiles of 32x32).  A wave
//     therefore issues 16 MFMAs per 8 (or 10) operand fragments, and no two waves share a weight byte:
//   * weights never touch LDS: they are packed on the host in MFMA-fragment order
//     ([step][co/32][k-slice][lane][16 B], pack.py frag_major) and each wave streams its own 1 KB fragments
//     global -> VGPR, three taps ahead;
//   * only the (TH+2) x (TW+2) input halo of one (frame, chunk) lives in LDS (double buffered, XOR-swizzled
//     128-byte rows); the nine spatial taps are nine row offsets into it.  One barrier per NINE taps;
//   * the four partial tiles are reduced through LDS once per tile (fixed order -> deterministic);
//   * groups (chunk, time tap) whose input frame lies in the causal zero padding are skipped, not multiplied;
//   * stream-K: the grid is one block per CU; block w walks the contiguous range [w G/W, (w+1) G/W) of the
//     tile-major list of all G groups, so every CU gets the same number of 9-tap groups (no tail at batch 1).  A
//     tile cut by a range boundary is finished in-kernel: every part drops its fp32 partial into a slab, takes a
//     ticket, and the LAST arriver sums all parts in part order (bitwise deterministic) and runs the epilogue
//     (agent-scope release / acquire as cdna_hip_programming.md §6 G16 prescribes; no spinning, so no residency
//     requirement).  There is no finish kernel and no second pass over the output.
// Zero padding in space comes from out-of-range buffer offsets (hardware returns 0).
#include <algorithm>
#include "conv_common.h"
#include "conv_fused_pw.h"

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
    uint32_t G0, ng;                    // linear index of the tile's first group, groups in the tile
    uint32_t L;                         // linear tile id (ticket index)
};

// frames-before-t in units of valid time taps: c(t) = sum_{j<t} min(j, pt) + 1
__device__ __forceinline__ uint32_t cum_taps(int t, int pt) {
    return t <= pt ? (uint32_t)(t * (t + 1) / 2) : (uint32_t)(pt * (pt + 1) / 2 + (t - pt) * (pt + 1));
}

__device__ __forceinline__ TileInfo decode_group(const ConvParams& p, uint32_t x) {
    const mv2_conv_desc& d = p.d;
    const int pt = d.kt - 1;
    const uint32_t u = (uint32_t)(p.tiles_h * p.tiles_w);
    const uint32_t unit = u * (uint32_t)p.nchunk;
    const uint32_t X1 = unit * cum_taps(d.Ti, pt);
    TileInfo ti;
    const uint32_t cb = x / X1, r = x - cb * X1;
    ti.co_tile = (int)(cb / (uint32_t)d.B);
    ti.b = (int)(cb - (uint32_t)ti.co_tile * (uint32_t)d.B);
    const uint32_t y = r / unit;
    int t = 0;
    const uint32_t ramp = (uint32_t)(pt * (pt + 1) / 2);
    if (y < ramp) { while (cum_taps(t + 1, pt) <= y) ++t; }
    else t = pt + (int)((y - ramp) / (uint32_t)(pt + 1));
    ti.t = t;
    ti.v = min(t, pt) + 1;
    const uint32_t r2 = r - unit * cum_taps(t, pt);
    ti.ng = (uint32_t)(p.nchunk * ti.v);
    const uint32_t sp = r2 / ti.ng, g = r2 - sp * ti.ng;
    ti.thi = (int)(sp / (uint32_t)p.tiles_w);
    ti.twi = (int)(sp - (uint32_t)ti.thi * (uint32_t)p.tiles_w);
    ti.G0 = x - g;
    ti.L = (((uint32_t)ti.co_tile * (uint32_t)d.B + (uint32_t)ti.b) * (uint32_t)d.Ti + (uint32_t)t) * u + sp;
    return ti;
}

// worker that owns linear group x when W workers split G groups as start(w) = floor(w G / W)
__device__ __forceinline__ uint32_t worker_of(uint32_t x, uint32_t W, uint32_t G) {
    return (uint32_t)((((uint64_t)x + 1) * W - 1) / G);
}
__device__ __forceinline__ uint32_t start_of(uint32_t w, uint32_t W, uint32_t G) { return (uint32_t)((uint64_t)w * G / W); }

template <typename E, int BCO, int TH, int TW>
struct C3Cfg {
    static constexpr int BPOS = TH * TW;
    static constexpr int WCO = BCO / 32, WPOS = BPOS / 32;          // per-wave MFMA tiles: the whole block tile
    static constexpr int NTILE = WCO * WPOS;                          // 16
    static constexpr int WCO_E = WCO == 4 ? 1 : 2, WPOS_E = WCO == 4 ? 4 : 2;   // tiles a wave owns after the reduce
    static constexpr int HWD = TW + 2, HR = (TH + 2) * HWD;
    static constexpr int RPP = 32;                                    // halo rows staged per pass (256 threads / 8)
    static constexpr int NH = (HR + RPP - 1) / RPP;
    static constexpr int HALO_BYTES = ((HR * 128 + 255) / 256) * 256;
    static constexpr int RED_BYTES = 4 * 8 * 4096;                    // cross-wave reduce: [src wave][8 tiles of a round] x 4 KB
    static constexpr int SLAB_BYTES = BCO * BPOS * 4;
    static constexpr size_t lds_main = 2 * (size_t)HALO_BYTES;
    static constexpr size_t lds_epi = (size_t)StageCfg<E, BCO, BPOS>::kBytes;
    static constexpr size_t lds_pw = std::is_same<E, BF16>::value
                                         ? ((size_t)StageCfg<BF16, BCO, BPOS>::kBytes + 255) / 256 * 256 + (size_t)(BCO * 2 / 128) * BCO * 128 : 0;
    static constexpr size_t m1 = lds_main > lds_epi ? lds_main : lds_epi;
    static constexpr size_t m2 = m1 > lds_pw ? m1 : lds_pw;
    static constexpr size_t m3 = m2 > (size_t)RED_BYTES ? m2 : (size_t)RED_BYTES;
    static constexpr size_t kFlagOff = (m3 + 15) / 16 * 16;           // 16-byte scratch word after everything else
    static constexpr size_t kLds = kFlagOff + 16;
    static_assert(NTILE == 16 && (WCO == 4 || WCO == 2), "wave tile: 4x4 or 2x8 MFMA tiles");
    static_assert(kLds <= 160 * 1024, "LDS budget");
};

}  // namespace

template <typename E, int BCO, int TH, int TW>
__global__ __launch_bounds__(256, 1) void conv3_kernel(const ConvParams p) {
    typedef C3Cfg<E, BCO, TH, TW> Cfg;
    constexpr int BPOS = Cfg::BPOS, WCO = Cfg::WCO, WPOS = Cfg::WPOS, WCO_E = Cfg::WCO_E, WPOS_E = Cfg::WPOS_E;
    constexpr int HWD = Cfg::HWD, HR = Cfg::HR, RPP = Cfg::RPP, NH = Cfg::NH, HALO_BYTES = Cfg::HALO_BYTES;
    constexpr int NT = 256;
    constexpr int HWT = 3;                          // tap at which the prefetched halo is written into the idle image
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const mv2_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = tid & 7, r0 = tid >> 3;
    const int hi = lane >> 5, l31 = lane & 31;
    const int T = d.Ti, H = d.Hi, W_ = d.Wi;
    const int kt = d.kt;

    // ---- this block's range of groups (XCD-aware: the blocks of one XCD walk one contiguous eighth of the list,
    // i.e. one co tile's weights where there are several: their 3.4 MB at C = 512 stay in that XCD's L2)
    const uint32_t NW = gridDim.x, G = p.gtot;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t q = NW >> 3, r = NW & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const uint32_t x0 = start_of(bid, NW, G), x1 = start_of(bid + 1, NW, G);

    const uint32_t row_bytes = (uint32_t)d.Ci * E::kBytes;
    const uint32_t frame_bytes = (uint32_t)((int64_t)H * W_ * d.Ci * E::kBytes);
    const uint32_t w_step = (uint32_t)d.co_pad * 128u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.w_bytes, 0x00020000);
    const uint32_t lds_wr = (uint32_t)r0 * 128u + (uint32_t)((c ^ ((r0 >> 1) & 7)) << 4);
    const uint32_t ck = (uint32_t)(wave * 2 + hi);          // this lane's 16-byte chunk of every 128-byte row
    // halo row of this lane's positions for the (dh = -1, dw = -1) tap
    int prow0[WPOS];
#pragma unroll
    for (int bb = 0; bb < WPOS; ++bb) {
        int ph, pw; pos_of<TW>(bb * 32 + l31, ph, pw);
        prow0[bb] = ph * HWD + pw;
    }

    for (uint32_t x = x0; x < x1;) {
        const TileInfo ti = decode_group(p, x);
        const uint32_t xe = min(x1, ti.G0 + ti.ng);
        const int gb = (int)(x - ti.G0), ge = (int)(xe - ti.G0);
        const int v = ti.v, t = ti.t, b = ti.b;
        const int h0 = ti.thi * TH, w0 = ti.twi * TW;
        const int n0 = ti.co_tile * BCO;

        f32x16 acc[WCO][WPOS];
#pragma unroll
        for (int a = 0; a < WCO; ++a)
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;

        // ---- halo gather state of this tile
        uint32_t hmask = 0;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int hr = r0 + RPP * i;
            if (hr < HR) {
                const int hy = hr / HWD, hx = hr - hy * HWD;
                const int hin = h0 - 1 + hy, win = w0 - 1 + hx;
                if ((unsigned)hin < (unsigned)H && (unsigned)win < (unsigned)W_) hmask |= 1u << i;
            }
        }
        const uint32_t hw_base = (uint32_t)(((int64_t)(h0 - 1) * W_ + (w0 - 1)) * (int64_t)row_bytes);
        // weights: this wave's fragment of co-group a lives at  step * w_step + ((n0/32 + a) * 4 + wave) * 1024 + lane * 16
        const uint32_t w_off0 = (uint32_t)(((n0 >> 5) * 4 + wave) * 1024 + lane * 16);

        i32x4 hreg[NH];
        i32x4 wset[3][WCO], pset[3][WPOS];

        // (chunk, time tap) of the groups of this tile: chunk-major list, advanced incrementally in scalar registers
        // (no per-tap division, no branches).  `cur` = group gi, `nxt` = group gi + 1 (clamped to the last group of
        // the segment: the prefetches past the end re-read valid, already cached addresses and are never consumed).
        int cur_chunk = v == 1 ? gb : (v == 2 ? (gb >> 1) : gb / 3);
        int cur_it = (kt - v) + (gb - cur_chunk * v);
        auto advance = [&](int& chunk, int& it) {
            const bool wrap = it + 1 == kt;
            it = wrap ? kt - v : it + 1;
            chunk += wrap ? 1 : 0;
        };
        int nxt_chunk = cur_chunk, nxt_it = cur_it;
        if (gb + 1 < ge) advance(nxt_chunk, nxt_it);
        const uint32_t tapstride = (uint32_t)p.nchunk * w_step;
        auto wbase_of = [&](int chunk, int it) { return (uint32_t)(it * 9 * p.nchunk + chunk) * w_step; };
        auto issue_halo = [&](int chunk, int it) {
            const int t_in = t + it - d.pt;                        // >= 0 by construction of the group list
            const uint32_t base = (uint32_t)(((int64_t)b * T + t_in) * (int64_t)frame_bytes) + (uint32_t)chunk * 128u + (uint32_t)c * 16u + hw_base;
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const int hr = r0 + RPP * i;
                const int hy = hr / HWD, hx = hr - hy * HWD;
                const bool ok = (hmask >> i) & 1u;
                const uint32_t vo = ok ? base + (uint32_t)(hy * W_ + hx) * row_bytes : p.in_bytes;     // padding: out of range -> zeros
                hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)vo, 0, 0);
            }
        };
        auto write_halo = [&](char* img) {
#pragma unroll
            for (int i = 0; i < NH; ++i)
                if (r0 + RPP * i < HR) *(i32x4*)(img + lds_wr + i * RPP * 128) = hreg[i];
        };
        auto issue_w = [&](uint32_t wbase, int tp, auto set_c) {
            constexpr int SET = decltype(set_c)::value;
            const uint32_t so = wbase + (uint32_t)tp * tapstride;
#pragma unroll
            for (int a = 0; a < WCO; ++a) wset[SET][a] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)w_off0, (int)(so + (uint32_t)(a * 4096)), 0);
        };
        auto read_p = [&](const char* img, int tp, auto set_c) {
            constexpr int SET = decltype(set_c)::value;
            const int toff = (tp / 3) * HWD + (tp % 3);
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb) {
                int pr_ = prow0[bb];
                asm volatile("" : "+v"(pr_));       // keep the unrolled taps from hoisting all their LDS addresses (VGPR budget)
                const int pr = pr_ + toff;
                pset[SET][bb] = *(const i32x4*)(img + (uint32_t)pr * 128u + ((ck ^ (uint32_t)((pr >> 1) & 7)) << 4));
            }
        };

        // ---- prologue: halo image of the first group, weights of taps 0..2, fragments of tap 0
        uint32_t cur_wb = wbase_of(cur_chunk, cur_it), nxt_wb = wbase_of(nxt_chunk, nxt_it);
        issue_halo(cur_chunk, cur_it);
        issue_w(cur_wb, 0, IC<0>{});
        issue_w(cur_wb, 1, IC<1>{});
        issue_w(cur_wb, 2, IC<2>{});
        write_halo(smem);
        __syncthreads();
        read_p(smem, 0, IC<0>{});

        // one tap: sets TP % 3 hold W(gi, TP) (requested three taps ago) and P(gi, TP) (read one tap ago)
        auto tap = [&](const char* img, char* other, auto tp_c) {
            constexpr int TP = decltype(tp_c)::value;
            constexpr int SET = TP % 3, NSET = (TP + 1) % 3;
            if constexpr (TP == 0) issue_halo(nxt_chunk, nxt_it);
            if constexpr (TP == 8) __syncthreads();       // the next group's image is complete; this one has no reader left
            if constexpr (TP < 8) read_p(img, TP + 1, IC<NSET>{});
            else read_p(other, 0, IC<NSET>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < WCO; ++a)
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) Mma<E>::run(wset[SET][a], pset[SET][bb], acc[a][bb]);
            if constexpr (TP + 3 < 9) issue_w(cur_wb, TP + 3, IC<SET>{});
            else issue_w(nxt_wb, TP + 3 - 9, IC<SET>{});
            if constexpr (TP == HWT) write_halo(other);   // image of group gi-1: its last reader passed the previous barrier
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int gi = gb; gi < ge; ++gi) {
            const int par = (gi - gb) & 1;
            const char* img = smem + par * HALO_BYTES;
            char* other = smem + (par ^ 1) * HALO_BYTES;
            tap(img, other, IC<0>{}); tap(img, other, IC<1>{}); tap(img, other, IC<2>{});
            tap(img, other, IC<3>{}); tap(img, other, IC<4>{}); tap(img, other, IC<5>{});
            tap(img, other, IC<6>{}); tap(img, other, IC<7>{}); tap(img, other, IC<8>{});
            cur_chunk = nxt_chunk; cur_it = nxt_it; cur_wb = nxt_wb;
            if (gi + 2 < ge) { advance(nxt_chunk, nxt_it); nxt_wb = wbase_of(nxt_chunk, nxt_it); }
        }
        __syncthreads();                                    // halo images are dead: LDS becomes the reduce region

        // ---- cross-wave reduce: wave w ends up with 4 of the 16 MFMA tiles, summed over the 4 k-slices in wave order.
        // Two rounds of 8 tiles: every wave drops its 8 partial tiles of the round into LDS (straight from the
        // accumulator registers), then reads the four partials of the two tiles it owns.  Ownership only enters
        // through an LDS address, so no accumulator is indexed at run time.
        //   BCO = 128: round(a, b) = b >> 1, slot j = 2 a + (b & 1),  owner = a       -> acc_e[0][2 rnd + k]
        //   BCO =  64: round(a, b) = b & 1,  slot j = 2 (b >> 1) + a, owner = b >> 1  -> acc_e[k][rnd]
        f32x16 acc_e[WCO_E][WPOS_E];
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
            for (int a = 0; a < WCO; ++a)
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) {
                    const int tr = WCO == 4 ? (bb >> 1) : (bb & 1);
                    const int j = WCO == 4 ? (2 * a + (bb & 1)) : (2 * (bb >> 1) + a);
                    if (tr != rnd) continue;
                    char* dst = smem + (wave * 8 + j) * 4096 + lane * 16;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        f32x4 vv = {acc[a][bb][4 * r4], acc[a][bb][4 * r4 + 1], acc[a][bb][4 * r4 + 2], acc[a][bb][4 * r4 + 3]};
                        *(f32x4*)(dst + r4 * 1024) = vv;
                    }
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x16 s;
#pragma unroll
                for (int src = 0; src < 4; ++src) {
                    const char* sp = smem + (src * 8 + wave * 2 + k) * 4096 + lane * 16;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 vv = *(const f32x4*)(sp + r4 * 1024);
                        if (src == 0) { s[4 * r4] = vv[0]; s[4 * r4 + 1] = vv[1]; s[4 * r4 + 2] = vv[2]; s[4 * r4 + 3] = vv[3]; }
                        else { s[4 * r4] += vv[0]; s[4 * r4 + 1] += vv[1]; s[4 * r4 + 2] += vv[2]; s[4 * r4 + 3] += vv[3]; }
                    }
                }
                if constexpr (WCO == 4) acc_e[0][2 * rnd + k] = s; else acc_e[k][rnd] = s;
            }
            __syncthreads();
        }
        const int wco_e = WCO == 4 ? wave : 0, wpos_e = WCO == 4 ? 0 : wave;

        // ---- a tile cut by a range boundary: publish the partial, take a ticket, the last arriver sums every part
        const bool partial = gb != 0 || (uint32_t)ge != ti.ng;
        bool finish = true;
        if (partial) {
            const uint32_t slot = 2 * bid + (x == x0 ? 0u : 1u);
            char* slab = (char*)p.ws + (size_t)slot * Cfg::SLAB_BYTES + (size_t)wave * 16384 + lane * 16;
#pragma unroll
            for (int a = 0; a < WCO_E; ++a)
#pragma unroll
                for (int bb = 0; bb < WPOS_E; ++bb)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        f32x4 vv = {acc_e[a][bb][4 * r4], acc_e[a][bb][4 * r4 + 1], acc_e[a][bb][4 * r4 + 2], acc_e[a][bb][4 * r4 + 3]};
                        *(f32x4*)(slab + ((a * WPOS_E + bb) * 4 + r4) * 1024) = vv;
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const uint32_t w_first = worker_of(ti.G0, NW, G), w_last = worker_of(ti.G0 + ti.ng - 1, NW, G);
            const uint32_t nparts = w_last - w_first + 1;
            uint32_t* flag = (uint32_t*)(smem + Cfg::kFlagOff);
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint32_t ticket = __hip_atomic_fetch_add(p.counters + ti.L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t last = ticket == nparts - 1 ? 1u : 0u;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    __hip_atomic_store(p.counters + ti.L, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                }
                *flag = last;
            }
            __syncthreads();
            finish = *flag != 0;
            if (finish) {
                const uint32_t s_first = start_of(w_first, NW, G);
#pragma unroll
                for (int a = 0; a < WCO_E; ++a)
#pragma unroll
                    for (int bb = 0; bb < WPOS_E; ++bb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc_e[a][bb][r] = 0.f;
                for (uint32_t pi = 0; pi < nparts; ++pi) {
                    const uint32_t wi = w_first + pi;
                    const uint32_t sl = 2 * wi + ((pi == 0 && s_first < ti.G0) ? 1u : 0u);
                    const char* src = (const char*)p.ws + (size_t)sl * Cfg::SLAB_BYTES + (size_t)wave * 16384 + lane * 16;
#pragma unroll
                    for (int a = 0; a < WCO_E; ++a)
#pragma unroll
                        for (int bb = 0; bb < WPOS_E; ++bb)
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) {
                                const f32x4 vv = *(const f32x4*)(src + ((a * WPOS_E + bb) * 4 + r4) * 1024);
                                acc_e[a][bb][4 * r4] += vv[0]; acc_e[a][bb][4 * r4 + 1] += vv[1];
                                acc_e[a][bb][4 * r4 + 2] += vv[2]; acc_e[a][bb][4 * r4 + 3] += vv[3];
                            }
                }
            }
            __syncthreads();                                // the flag word is free again
        }

        // ---- epilogue of a finished tile
        if (finish) {
            const int64_t frame_row0 = ((int64_t)b * T + t) * H;
            auto row_of = [&](int lp) -> int64_t {
                int ph, pw; pos_of<TW>(lp, ph, pw);
                const bool ok = (h0 + ph) < H && (w0 + pw) < W_;
                return ok ? (frame_row0 + (h0 + ph)) * W_ + (w0 + pw) : p.M;
            };
            bool done = false;
            if constexpr (std::is_same<E, BF16>::value) {
                if (p.wp2) {       // fused 1x1x1 conv + activation (+ SqueezeExcite pooling partials); one co tile by construction
                    const int64_t tile_index = (((int64_t)b * T + t) * p.tiles_h + ti.thi) * p.tiles_w + ti.twi;
                    fused_pointwise_epilogue<BCO, BPOS, WCO_E, WPOS_E, NT>(p, smem, acc_e, wco_e, wpos_e, tid, tile_index, row_of);
                    done = true;
                }
            }
            if (!done) {
                if (stage_ok(p)) {
                    staged_epilogue<E, BCO, BPOS, WCO_E, WPOS_E, NT>(p, smem, acc_e, n0, wco_e, wpos_e, tid, row_of);
                } else {
                    const int nbase = n0 + wco_e * 32 * WCO_E;
#pragma unroll
                    for (int a = 0; a < WCO_E; ++a)
#pragma unroll
                        for (int bb = 0; bb < WPOS_E; ++bb) {
                            const int64_t m = row_of(wpos_e * 32 * WPOS_E + bb * 32 + l31);
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                float vv[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) vv[j] = acc_e[a][bb][4 * gq + j];
                                epi_store4<E>(p, m, nbase + a * 32 + 8 * gq + 4 * hi, vv);
                            }
                        }
                }
            }
        }
        __syncthreads();                                    // LDS is reused by the next segment's halo image
        x = xe;
    }
}

// ------------------------------------------------------------------------------- host side
namespace {

struct Plan3 {
    int bco, th, tw, tiles_h, tiles_w, nchunk, n_co_tiles;
    int64_t M, n_tiles;
    uint64_t gtot;
    int workers;
    size_t slab_bytes;
};

inline int cus_of_current_device() {
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
           d.nsteps == d.kt * 9 * ((d.Ci * es) / 128) && d.act != MV2_ACT_GEGLU && d.co >= 64 && d.co_pad % 128 == 0 &&
           d.co <= d.co_pad && d.store != MV2_STORE_NCTHW_F32 && (d.co % 4) == 0 && (d.out_c % 4) == 0;
}

inline Plan3 plan3(const mv2_conv_desc& d) {
    Plan3 P;
    const int es = d.dtype == MV2_BF16 ? 2 : 4;
    if (d.co <= 64) { P.bco = 64; P.th = 16; P.tw = 16; } else { P.bco = 128; P.th = 8; P.tw = 16; }
    P.tiles_h = (d.Ho + P.th - 1) / P.th;
    P.tiles_w = (d.Wo + P.tw - 1) / P.tw;
    P.nchunk = (d.Ci * es) / 128;
    P.n_co_tiles = (d.co + P.bco - 1) / P.bco;
    P.M = (int64_t)d.B * d.To * d.Ho * d.Wo;
    P.n_tiles = (int64_t)P.n_co_tiles * d.B * d.Ti * P.tiles_h * P.tiles_w;
    const int pt = d.kt - 1, T = d.Ti;
    const int64_t cT = T <= pt ? (int64_t)T * (T + 1) / 2 : (int64_t)pt * (pt + 1) / 2 + (int64_t)(T - pt) * (pt + 1);
    P.gtot = (uint64_t)P.n_co_tiles * d.B * P.tiles_h * P.tiles_w * P.nchunk * (uint64_t)cT;
    const int cus = cus_of_current_device();
    P.workers = (int)std::min<uint64_t>((uint64_t)cus, P.gtot);
    // two slab slots per worker (a worker's range cuts at most its first and its last tile)
    P.slab_bytes = P.workers > 1 ? (size_t)2 * P.workers * (size_t)P.bco * (P.th * P.tw) * 4 : 0;
    return P;
}

template <typename E, int BCO, int TH, int TW>
int launch3(const ConvParams& p, int workers, hipStream_t st) {
    typedef C3Cfg<E, BCO, TH, TW> Cfg;
    // set unconditionally: the attribute is per device, a cached flag would be wrong for a second device of the process
    hipError_t e = hipFuncSetAttribute((const void*)conv3_kernel<E, BCO, TH, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3_kernel<E, BCO, TH, TW>), dim3((unsigned)workers), dim3(256), Cfg::kLds, st, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

}  // namespace

extern "C" int mv2_conv3_eligible(const mv2_conv_desc* d) { return d && eligible3(*d) ? 1 : 0; }

extern "C" size_t mv2_conv3_workspace_bytes(const mv2_conv_desc* d) {
    if (!d || !eligible3(*d)) return 0;
    return plan3(*d).slab_bytes;
}

extern "C" size_t mv2_conv3_counter_bytes(const mv2_conv_desc* d) {
    if (!d || !eligible3(*d)) return 0;
    return (size_t)plan3(*d).n_tiles * sizeof(uint32_t);
}

extern "C" int mv2_conv3_tiles(const mv2_conv_desc* d, int* tile_h, int* tile_w) {
    if (!d || !eligible3(*d)) return MV2_EINVAL;
    const Plan3 P = plan3(*d);
    if (tile_h) *tile_h = P.th;
    if (tile_w) *tile_w = P.tw;
    return P.tiles_h * P.tiles_w;
}

extern "C" int mv2_conv3(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp3, const float* bias,
                         const float* rowscale, const void* residual, const void* wp2, const float* bias2, int act2,
                         const float* se_wk, float se_bk, float* se_part, void* out, void* workspace, size_t workspace_bytes,
                         uint32_t* counters, size_t counter_bytes, void* stream) {
    if (!d || !in || !wp3 || !out) return MV2_EINVAL;
    if (!eligible3(*d)) return MV2_EINVAL;
    if (in_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    const Plan3 P = plan3(*d);
    if (P.M >= (int64_t)0x7fffffff || P.gtot >= 0xffffffffull || P.n_tiles >= (int64_t)0x7fffffff) return MV2_ETOOLARGE;
    const size_t w_bytes = (size_t)d->nsteps * d->co_pad * 128;
    if (w_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    if (residual && d->store != MV2_STORE_ROWS) return MV2_EINVAL;
    if (wp2) {       // fused second stage: bf16, the block owns every channel of its positions
        if (d->dtype != MV2_BF16 || P.n_co_tiles != 1 || (d->co != 64 && d->co != 128) || d->store != MV2_STORE_ROWS || d->out_c != d->co ||
            rowscale || residual) return MV2_EINVAL;
        if (act2 != MV2_ACT_NONE && act2 != MV2_ACT_ELU && act2 != MV2_ACT_SILU) return MV2_EINVAL;
    } else if (bias2 || se_part) return MV2_EINVAL;
    if (se_part && !se_wk) return MV2_EINVAL;
    if (P.slab_bytes > 0 && (!workspace || workspace_bytes < P.slab_bytes)) return MV2_EWORKSPACE;
    if (P.workers > 1 && (!counters || counter_bytes < (size_t)P.n_tiles * sizeof(uint32_t))) return MV2_EWORKSPACE;
    ConvParams p{};
    p.d = *d; p.in = in; p.wp = wp3; p.segs = nullptr; p.bias = bias; p.rowscale = rowscale; p.residual = residual;
    p.out = out; p.ws = (float*)workspace;
    p.wp2 = wp2; p.bias2 = bias2; p.act2 = act2;
    p.se_wk = se_wk; p.se_bk = se_bk; p.se_part = se_part;
    p.in_bytes = (uint32_t)in_bytes; p.w_bytes = (uint32_t)w_bytes;
    p.M = P.M; p.n_co_tiles = P.n_co_tiles; p.n_pos_tiles = (int)(P.n_tiles / P.n_co_tiles);
    p.splitk = 1; p.tiles_h = P.tiles_h; p.tiles_w = P.tiles_w; p.nchunk = P.nchunk; p.chunks_per_split = P.nchunk;
    p.counters = counters; p.gtot = (uint32_t)P.gtot;
    hipStream_t st = (hipStream_t)stream;
    const bool bf = d->dtype == MV2_BF16;
    if (P.bco == 128) return bf ? launch3<BF16, 128, 8, 16>(p, P.workers, st) : launch3<F32, 128, 8, 16>(p, P.workers, st);
    return bf ? launch3<BF16, 64, 16, 16>(p, P.workers, st) : launch3<F32, 64, 16, 16>(p, P.workers, st);
}
