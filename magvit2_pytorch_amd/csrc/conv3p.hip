// mv2_conv3, bf16 — the position-tile kernel of the ResidualUnit kt x 3 x 3 causal convolution (reference
// magvit2_pytorch.py:937, CausalConv3d :892-928), optionally fused with the unit's 1x1x1 conv + ELU (:938-940) and the
// SqueezeExcite pooling partials (:226-231).  Round 4 replacement of conv3.hip's decomposition for bf16 (conv3.hip keeps the
// exact-fp32 mode).
//
// What was wrong with the K-split-across-four-waves form (conv3.hip; profiles/r03_findings.md section 5): every wave holds the whole
// 64 x 128 tile in 128 accumulator registers, so a tile ends with a 4-way cross-wave reduce of 128 KB through LDS, a slab / ticket
// round for tiles cut by the stream-K ranges and an LDS-staged epilogue: 43 % of a CU's launch time were those fixed phases, with
// the partner workgroup running alone at 68 % of the matrix pipe's rate.
//
// Decomposition here (gfx950: 256 CUs x 4 SIMDs, 512 registers per lane per SIMD, 160 KB LDS per CU):
//   * a workgroup = 4 waves = 2 output-channel slices (32 channels each) x 2 K halves, on a 64-channel x 8x16-position tile of one
//     output frame.  A wave keeps 32 channels x 128 positions in 64 accumulator registers (1 x 4 MFMA tiles of 32x32), so THREE
//     workgroups fit a CU (168 registers, 52 KB LDS each): twelve waves, three per SIMD, and the fixed phases of one workgroup run
//     under the tap loops of the other two;
//   * per 16-channel k-step a wave issues 4 MFMAs for 1 weight fragment + 4 position fragments.  Weights never touch LDS: the wave
//     streams ITS 1 KB fragments (fragment-major image of pack.py / mv2_pack_conv3d, the same image conv3.hip reads) global -> VGPR
//     through a ring, three taps ahead; no two waves of a workgroup load the same weight byte (0.25 KB per MFMA = half the L1 rate);
//   * the 10 x 18 halo of one (input frame, 64-channel chunk) goes global -> LDS by buffer_load ... lds (no staging registers, no
//     ds_write: 13-16 clocks per wave-instruction, profiles/r03_findings.md section 8), double buffered, one barrier per NINE taps.
//     The LDS image keeps conv3.hip's 144-byte pixel pitch (nine taps = nine immediate offsets, conflict-free under the measured
//     ds_read_b128 pass partition with the odd tile rows rotated by 14 columns): a DMA piece is 64 consecutive 16-byte slots of the
//     image, lane -> (pixel, part) = (slot / 9, slot % 9), part 8 is the padding (out-of-range source: zeros, no traffic);
//   * the tile ends with a 2-way exchange between the K halves (32 KB through LDS; afterwards each wave owns 32 channels x 64
//     positions) and a register epilogue: bias, ELU, bf16, v_permlane32_swap so that a lane holds 8 consecutive channels, 16-byte
//     stores.  No LDS staging, no block barrier;
//   * one workgroup per (tile, K part): the grid is the work list and the hardware dispatcher balances it (tiles in descending
//     cost: the causal-front frames with one or two valid time taps come last and fill the tail).  Deep levels (few tiles) split
//     a tile's group list into S parts across workgroups: parts publish fp32 slabs with write-through stores, take a ticket, the
//     last arriver sums the parts in part order (bitwise deterministic) and runs the epilogue — the protocol of conv3.hip;
//   * groups (chunk, time tap) whose input frame lies in the causal zero padding are skipped, not multiplied; spatial zero padding
//     comes from out-of-range buffer offsets.
#include <algorithm>
#include <cstdlib>
#include "conv3p.h"

namespace {

constexpr int BCO = 64, TH = 8, TW = 16, BPOS = TH * TW, WPOS = BPOS / 32;
constexpr int HWD = TW + 2, HR = (TH + 2) * HWD;                      // halo: 10 x 18 pixels
constexpr int ROWP = 144;                                             // LDS pitch of a 128-byte halo pixel
constexpr int NPIECE = (HR * ROWP + 1023) / 1024;                     // 1 KB DMA pieces per halo image (26)
constexpr int IMG = NPIECE * 1024;
constexpr int PPW = (NPIECE + 3) / 4;                                 // pieces per wave (7; waves 2, 3 repeat pieces 24, 25)
constexpr int RING = 6;                                               // weight fragments in flight per wave (6 = 3 taps; must divide the 18 fragments of a group)
static_assert(18 % RING == 0, "ring slots must divide the 18 fragments of a group");
// The ablation / scheduling knobs this kernel was tuned with (ring depth, burst vs interleaved fragment reads, priorities, the timing-only
// ablations, frames-inside-tiles block order, occupancy override) live with their measurements: tools/experiments/conv3p_r5_oneshot.hip
// (the same kernel with the knobs, built by tools/build_c3p_r5.sh), profiles/r04_findings.md sections 2-3.  The product source has one path.
constexpr int SLAB = BCO * BPOS * 4;
constexpr int XCH = 4 * 8192;                                         // exchange region: [wave][2 tiles][4][64 lanes][16 B]
static_assert(NPIECE == 26 && PPW == 7, "piece bookkeeping below");
static_assert(XCH <= 2 * IMG, "the exchange region overlays the halo images");

template <bool FUSED>
struct PLds {
    static constexpr int kY1 = XCH;                                   // fused stage: [128 positions][144 B] bf16 tile of the first stage
    static constexpr int kConst2 = kY1 + BPOS * ROWP;                 // fused stage: bias2[64], se_wk[64] (fp32)
    static constexpr int kEv = 0;                                     // fused stage: 32 floats of pooling scratch per wave (exchange region, dead by then)
    static constexpr int kFlag = 2 * IMG;
    static constexpr int kLds = kFlag + 64;
    static_assert(kConst2 + 512 <= 2 * IMG, "fused-stage scratch lives in the halo images");
    static_assert(3 * kLds <= 160 * 1024, "three workgroups per CU");
};

template <int TW_>
__device__ __forceinline__ void pos_of(int pp, int& ph, int& pw) {     // as conv3.hip: odd tile rows rotated by 14 columns (bank-conflict freedom)
    ph = pp / TW_;
    const int x = pp % TW_;
    pw = (ph & 1) ? ((x + 14) & 15) : x;
}

// Division of 32-bit unsigned numbers by a launch-time constant (Granlund-Montgomery, the round-up form: exact for every n): a
// workgroup decodes its block id with five of them; as emulated integer divisions (~40 dependent VALU instructions each, a reciprocal
// among them) they were 4k of the 6k clocks a workgroup spent before its first load (profiles/r04_findings.md).
struct FastDiv { uint32_t m, sh1, sh2; };
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t l = 0; while ((1ull << l) < d) ++l;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1; f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
// Kernel arguments, ordered by first use (the block decode reads the first 24 dwords): ConvParams (conv_common.h) is 400 bytes of which
// this kernel needs a third, and its scalar loads came in five dependent rounds.
struct P3Params {
    FastDiv per_co, S, tiles_sp, T, tiles_w;
    uint32_t grid;                        // = gridDim.x (the hidden kernel argument sits in another cache line)
    int32_t nA, tilesA, SB, tiles_per_xcd; // two-class work list (nA = 0: uniform): per XCD nA class-A blocks = tilesA tiles in `splitk` parts, then tiles in SB parts
    FastDiv dSB;
    int32_t B, Ti, Hi, Wi, kt, nchunk, splitk, tiles_w_n, tiles_h_n;
    int32_t Ci, co, co_pad, out_c, act, act2;
    uint32_t in_bytes, w_bytes, ws_bytes;
    float se_bk;
    const void* in; const void* wp; const float* bias; void* out;
    float* ws; uint32_t* counters;
    const void* wp2; const float* bias2; const float* se_wk; float* se_part;
};

template <int N_> struct WaitVmP;
#define MV2_WAITVMP(n) template <> struct WaitVmP<n> { static __device__ __forceinline__ void run() { asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); } };
MV2_WAITVMP(0) MV2_WAITVMP(6)
#undef MV2_WAITVMP

}  // namespace

#ifdef MV2_EXPERIMENTS
// phase clocks (s_memtime) per workgroup, the record layout of conv3.hip so that tools/k3_bench prints both: [0] prologue [1] tap loop
// [2] exchange [3] slab / ticket [4] epilogue [5] segments (1) [6] groups [7] whole kernel; [8 + 7] absolute start clock
__device__ unsigned long long* g_conv3p_prof = nullptr;
#define C3P_PROF_DECL unsigned long long pf_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long qf_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_ = __builtin_amdgcn_s_memtime(); const unsigned long long pt0_ = pt_;
#define C3P_PROF(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_[k] += n_ - pt_; pt_ = n_; }
#define C3P_PROF_COUNT(k, v) pf_[k] += (v);
#define C3P_SUB(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); qf_[k] += n_ - pt_; pf_[4] += n_ - pt_; pt_ = n_; }
#define C3P_SUBP(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); qf_[k] += n_ - pt_; pf_[0] += n_ - pt_; pt_ = n_; }
#define C3P_PROF_END if (g_conv3p_prof && threadIdx.x == 0 && blockIdx.x < 1024) { pf_[7] = __builtin_amdgcn_s_memtime() - pt0_; pf_[5] |= ((unsigned long long)((__builtin_amdgcn_s_getreg(63508) & 0xf) << 8 | ((__builtin_amdgcn_s_getreg(63492) >> 8) & 0xff))) << 32; for (int i_ = 0; i_ < 8; ++i_) { g_conv3p_prof[(size_t)blockIdx.x * 16 + i_] = pf_[i_]; g_conv3p_prof[(size_t)blockIdx.x * 16 + 8 + i_] = i_ == 7 ? pt0_ : qf_[i_]; } }
#else
#define C3P_PROF_DECL
#define C3P_PROF(k)
#define C3P_PROF_COUNT(k, v)
#define C3P_SUB(k)
#define C3P_SUBP(k)
#define C3P_PROF_END
#endif

// OUT32 (MV2_STORE_ROWS_F32, the split-bf16 form of the id-exact mode): the epilogue stores the fp32 sums (exact ELU, nothing rounded)
template <bool FUSED, bool OUT32 = false>
__global__ __launch_bounds__(256, 3) void conv3p_kernel(const P3Params p) {
    static_assert(!(FUSED && OUT32), "the fused second stage is a bf16 stage");
    const P3Params& pp = p;
    typedef BF16 E;
    typedef PLds<FUSED> L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = wave & 1, kh = wave >> 1;
    const int T = p.Ti, H = p.Hi, W_ = p.Wi, kt = p.kt, pt = p.kt - 1;
    C3P_PROF_DECL
    asm volatile("" :: "s"(p.in), "s"(p.wp), "s"(p.out), "s"(p.in_bytes));      // every cache line of the kernel arguments is requested in the first round of scalar loads

    // ---- block -> (channel tile, batch, frame, spatial tile, K part).  XCD-aware: the blocks of one XCD walk one contiguous eighth of
    // the list = few channel tiles, whose weights stay in that XCD's L2, and neighbouring spatial tiles (shared halo borders).
    // Two forms of the work list:
    //  uniform : every tile in S parts; block id -> (tile, part) after the XCD remap (any grid size);
    //  two-class (p.nA > 0; host: conv3p_plan): a launch whose tiles fit the chip's workgroup slots ONCE is dispatched in one wave, and
    //            the dispatcher gives CU i the blocks i, i + CUs, i + 2 CUs (seen in the residency reports): with 2.5 tiles per CU half
    //            the CUs get three tiles and the launch lasts as long as they do.  Here the first tiles are cut into SA parts, the last
    //            ones into SB = 2 SA, so that the grid is EXACTLY the slot count and every CU receives the same number of groups (C = 256 at
    //            batch 1: 512 whole tiles + 128 tiles in halves = 768 blocks, 2.5 tiles per CU).  Per XCD: nA class-A blocks, then the
    //            class-B blocks; raw block b sits on XCD b % 8 with local index b / 8.  A speed rule only: any placement is correct.
    uint32_t S, part, r_, q_;
    const uint32_t tiles_sp = (uint32_t)(p.tiles_h_n * p.tiles_w_n);
    const uint32_t per_co = (uint32_t)p.B * (uint32_t)T * tiles_sp;      // tiles per channel tile
    bool classB = false;
    if (p.nA > 0) {
        const uint32_t xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        uint32_t tl;
        if (local < (uint32_t)p.nA) { S = (uint32_t)p.splitk; tl = fdiv(local, pp.S); part = local - tl * S; }
        else { classB = true; S = (uint32_t)p.SB; const uint32_t l2 = local - (uint32_t)p.nA; const uint32_t t2 = fdiv(l2, pp.dSB); part = l2 - t2 * S; tl = (uint32_t)p.tilesA + t2; }
        r_ = xcd * (uint32_t)p.tiles_per_xcd + tl;                        // position in the tile order
    } else {
        uint32_t bid = blockIdx.x;
        const uint32_t NW = p.grid, q = NW >> 3, r = NW & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        S = (uint32_t)p.splitk;
        q_ = fdiv(bid, pp.S); part = bid - q_ * S; r_ = q_;
    }
    const int co_tile = (int)fdiv(r_, pp.per_co);
    r_ -= (uint32_t)co_tile * per_co;
    // frames are the outer loop (descending, so that the cheap causal-front frames come last and fill the tail), spatial tiles inside
    q_ = fdiv(r_, pp.tiles_sp); const uint32_t sp = r_ - q_ * tiles_sp; r_ = q_;
    const int b = (int)fdiv(r_, pp.T);
    const int t = T - 1 - (int)(r_ - (uint32_t)b * (uint32_t)T);
    const int thi = (int)fdiv(sp, pp.tiles_w), twi = (int)(sp - (uint32_t)thi * (uint32_t)p.tiles_w_n);
    const int v = min(t, pt) + 1;                                       // valid time taps of this output frame (causal front)
    const uint32_t ng = (uint32_t)(p.nchunk * v);                       // groups of nine taps in the tile: (chunk, time tap), chunk-major
    const uint32_t nparts = min(S, ng);
    if (part >= nparts) return;
    // nparts == S: the group list in S nearly equal ranges; nparts == ng < S: one group per part
    const FastDiv& dS = classB ? pp.dSB : pp.S;
    const int gb = nparts == S ? (int)fdiv(part * ng, dS) : (int)part, ge = nparts == S ? (int)fdiv((part + 1) * ng, dS) : (int)part + 1;
    const uint32_t Smax = p.nA > 0 ? (uint32_t)p.SB : S;               // slab slots per tile
    const uint32_t tile_lin = (((uint32_t)co_tile * (uint32_t)p.B + (uint32_t)b) * (uint32_t)T + (uint32_t)t) * tiles_sp + sp;
    const int h0 = thi * TH, w0 = twi * TW;
    const int n0 = co_tile * BCO;
    if constexpr (!FUSED) { asm volatile("" :: "s"(gb), "s"(ge), "s"(h0), "s"(w0), "s"(tile_lin)); C3P_SUBP(0) }      // experiments: clocks until the block is decoded

    const uint32_t row_bytes = (uint32_t)p.Ci * 2u;
    const uint32_t frame_bytes = (uint32_t)((int64_t)H * W_ * p.Ci * 2);
    const uint32_t w_step = (uint32_t)p.co_pad * 128u;
    const uint32_t tapstride = (uint32_t)p.nchunk * w_step;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : (const float*)p.wp), 0,
                                                                          p.bias ? (uint32_t)p.co_pad * 4u : 0u, 0x00020000);

    uint32_t tab[PPW], pb[WPOS];          // halo source table / position-fragment bases: filled in the prologue, behind the first weight loads
    // weights: fragment (step, 32-channel group cg, k-slice ks) of the fragment-major image at step * w_step + (cg * 4 + ks) * 1024
    const uint32_t w_voff = (uint32_t)lane * 16u;
    const uint32_t w_frag0 = (uint32_t)(((co_tile * 2 + cs) * 4 + 2 * kh) * 1024);

    // group g of the tile -> (chunk, time tap): chunk-major.  The walk is incremental and branch-free (scalar selects).
    auto gbase_of = [&](int ch, int it) {      // byte offset of the group's (input frame, chunk)
        return (uint32_t)(((int64_t)b * T + (t + it - pt)) * (int64_t)frame_bytes) + (uint32_t)ch * 128u;
    };
    auto wbase_of = [&](int ch, int it) {      // byte offset of the group's first tap in the weight image
        return (uint32_t)(it * 9 * p.nchunk + ch) * w_step + w_frag0;
    };
    const int ch0 = v == 1 ? gb : (v == 2 ? (gb >> 1) : gb / 3), it0 = (kt - v) + (gb - ch0 * v);
    // A dead group (past the part's end) keeps its DMA instructions (a load inside a branch makes the wait-count pass drain the queue)
    // but reads through a zero-length descriptor: zeros land in an image nobody reads, no memory access.
    const __amdgpu_buffer_rsrc_t rs_dead = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
    auto dma_piece = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t so, int img, auto i_c) {
        constexpr int I = decltype(i_c)::value;
        int pi = wave + 4 * I;
        if (pi >= NPIECE) pi -= 2;
        __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(smem + img * IMG + pi * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)tab[I], (int)so, 0, 0);
    };
    i32x4 wring[RING], pset[2][WPOS];
    auto issue_w = [&](uint32_t so, i32x4& dst) { dst = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)w_voff, (int)so, 0); };
    auto read_p = [&](const uint32_t (&base)[WPOS], int toff, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int bb = 0; bb < WPOS; ++bb) pset[SET][bb] = *(const i32x4*)(smem + base[bb] + toff);
    };

    // ---- prologue: the first RING weight fragments (they need nothing lane-derived) and the first halo image are in flight together
    // (c1, i1) = the group after the current one, clamped to the part's last group (prefetches past the end re-read valid, cached
    // weight addresses and are never consumed)
    int c1 = ch0, i1 = it0;
    auto advance1 = [&]() { const bool wrap = i1 + 1 == kt; i1 = wrap ? kt - v : i1 + 1; c1 += wrap ? 1 : 0; };
    if (gb + 1 < ge) advance1();
    uint32_t cur_wb = wbase_of(ch0, it0), nxt_wb = wbase_of(c1, i1);
#pragma unroll
    for (int f = 0; f < RING; ++f) issue_w(cur_wb + (uint32_t)(f >> 1) * tapstride + (uint32_t)(f & 1) * 1024u, wring[f]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!FUSED) { C3P_SUBP(1) }                                                                                 // ... until the weight ring is requested
    // ---- halo source table: DMA piece pi fills image bytes [pi * 1024, pi * 1024 + 1024) = slots pi * 64 + lane (16 B each);
    // slot -> (pixel, part) = (slot / 9, slot % 9); part 8 is the 16-byte padding of the 144-byte pitch.  The entry is the byte offset
    // inside the (frame, chunk) the group reads (the uniform part travels in soffset, which the range check does not see); pixels in
    // the spatial zero padding, the pad part and slots past the image read out of range: zeros, no memory access.
    // Each entry's DMA piece is issued as soon as the entry exists: the first pieces are on their way while the later entries and the
    // fragment bases are computed.
    const uint32_t so0 = gbase_of(ch0, it0);
    auto tab_entry = [&](auto i_c) {
        constexpr int I = decltype(i_c)::value;
        int pi = wave + 4 * I;
        if (pi >= NPIECE) pi -= 2;
        const int slot = pi * 64 + lane;
        const int px = slot / 9, prt = slot - px * 9;
        const int hy = px / HWD, hx = px - hy * HWD;
        const int hin = h0 - 1 + hy, win = w0 - 1 + hx;
        const bool ok = px < HR && prt < 8 && (unsigned)hin < (unsigned)H && (unsigned)win < (unsigned)W_;
        tab[I] = ok ? (uint32_t)(hin * W_ + win) * row_bytes + (uint32_t)prt * 16u : p.in_bytes;
        dma_piece(rs_in, so0, 0, i_c);
        __builtin_amdgcn_sched_barrier(0);
    };
    tab_entry(IC<0>{}); tab_entry(IC<1>{}); tab_entry(IC<2>{}); tab_entry(IC<3>{}); tab_entry(IC<4>{}); tab_entry(IC<5>{}); tab_entry(IC<6>{});
    // ---- position fragments: MFMA B operand of position tile bb = 16 bytes per lane at (pixel of position 32 bb + l31, k-chunk
    // 2 (2 kh + s) + hi); the nine taps and the two k-steps s of this wave's K half are immediate offsets.
#pragma unroll
    for (int bb = 0; bb < WPOS; ++bb) {
        int ph, pw; pos_of<TW>(bb * 32 + l31, ph, pw);
        pb[bb] = (uint32_t)(ph * HWD + pw) * ROWP + (uint32_t)(kh * 64 + hi * 16);
    }
    if constexpr (!FUSED) { asm volatile("" :: "v"(pb[0]), "v"(pb[3])); C3P_SUBP(2) }       // ... until the tables are computed and the first image is requested
    C3P_SUBP(5)
    f32x16 acc[WPOS];
#pragma unroll
    for (int bb = 0; bb < WPOS; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[bb][r] = 0.f;
    WaitVmP<0>::run();                                // this wave's pieces of the first image have landed ...
    __builtin_amdgcn_s_barrier();                     // ... and everybody's
    read_p(pb, 0, IC<0>{});
    C3P_SUBP(6)
    C3P_PROF(0)

    for (int gi = gb; gi < ge; ++gi) {
        const int par = (gi - gb) & 1;
        uint32_t pbc[WPOS], pbn[WPOS];
#pragma unroll
        for (int bb = 0; bb < WPOS; ++bb) { pbc[bb] = pb[bb] + (uint32_t)(par * IMG); pbn[bb] = pb[bb] + (uint32_t)((par ^ 1) * IMG); }
        // the next group's halo: one DMA piece behind each of the first seven k-steps, into the image the previous group released
        const bool live1 = gi + 1 < ge;
        const __amdgpu_buffer_rsrc_t rs1 = live1 ? rs_in : rs_dead;
        const uint32_t so1 = gbase_of(c1, i1);
        // the weight prefetch past the part's last group is never consumed: it goes through a zero-length descriptor (no memory access, the
        // ring keeps its instruction count), so the drain behind the loop does not wait for six L2 round trips nobody needs
        const __amdgpu_buffer_rsrc_t rs_wn = live1 ? rs_w : __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0, 0x00020000);
        // one k-step: fragment F = 2 tap + s of the group lives in ring slot F % RING (requested RING fragments ago), pset[s] holds
        // its position fragments (read one k-step ago).  The reload of the slot goes out right behind its MFMAs.
        auto kstep = [&](auto f_c) {
            constexpr int F = decltype(f_c)::value, S_ = F & 1, SLOT = F % RING, G = F + RING;
            constexpr int NT = (F + 1) >> 1, NS = (F + 1) & 1, NOFF = ((NT / 3) * HWD + (NT % 3)) * ROWP + NS * 32;
            // one fragment read of the NEXT k-step in each MFMA gap (a single wave then issues an MFMA every ~34 clocks instead of
            // one burst of four reads per four MFMAs: measured 333 clocks per tap solo with the burst form, 256 is the pipe's floor).
            // k-step 16 keeps the burst form so that this group's last reads are long done at the boundary's lgkmcnt(0).
            constexpr bool ILV = F != 16 && F != 17;
            if constexpr (F == 17) {
                // group boundary: the next image is complete (this wave's pieces: every load older than the last RING has landed;
                // everybody's: the barrier), and nobody reads this group's image any more (its last fragments are in registers)
                WaitVmP<RING>::run();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else if constexpr (!ILV) {
                read_p(pbc, NOFF, IC<NS>{});
            }
            __builtin_amdgcn_sched_barrier(0);        // keep the fragment reads of the NEXT k-step at the top (the scheduler sinks them to their use)
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb) {
                Mma<E>::run(wring[SLOT], pset[S_][bb], acc[bb]);
                if constexpr (ILV) { pset[NS][bb] = *(const i32x4*)(smem + pbc[bb] + NOFF); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (F == 17) { pset[0][bb] = *(const i32x4*)(smem + pbn[bb]); __builtin_amdgcn_sched_barrier(0); }
            }
            const uint32_t so = G < 18 ? cur_wb + (uint32_t)(G >> 1) * tapstride + (uint32_t)(G & 1) * 1024u
                                       : nxt_wb + (uint32_t)((G - 18) >> 1) * tapstride + (uint32_t)(G & 1) * 1024u;
            if constexpr (G < 18) issue_w(so, wring[SLOT]);
            else wring[SLOT] = __builtin_amdgcn_raw_buffer_load_b128(rs_wn, (int)w_voff, (int)so, 0);
            if constexpr (F < PPW) dma_piece(rs1, so1, par ^ 1, IC<F>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        kstep(IC<0>{}); kstep(IC<1>{}); kstep(IC<2>{}); kstep(IC<3>{}); kstep(IC<4>{}); kstep(IC<5>{});
        kstep(IC<6>{}); kstep(IC<7>{}); kstep(IC<8>{}); kstep(IC<9>{}); kstep(IC<10>{}); kstep(IC<11>{});
        kstep(IC<12>{}); kstep(IC<13>{}); kstep(IC<14>{}); kstep(IC<15>{}); kstep(IC<16>{}); kstep(IC<17>{});
        cur_wb = nxt_wb;
        if (gi + 2 < ge) advance1();
        nxt_wb = wbase_of(c1, i1);
    }
    WaitVmP<0>::run();                                // every DMA write of this wave (and its trailing weight prefetch) has landed ...
    __syncthreads();                                  // ... everybody's: LDS becomes the exchange region
    // bias of this wave's 16 channels per lane (channel 32 cs + 8 g + 4 hi + j): requested now, lands under the exchange
    f32x4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        bv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)((uint32_t)(n0 + cs * 32 + 8 * g + 4 * hi) * 4u), 0, 0));
    C3P_PROF(1)
    C3P_PROF_COUNT(5, 1)
    C3P_PROF_COUNT(6, (unsigned long long)(ge - gb))

    // fused second stage: bias2 / se_wk for the LDS constants are requested now (their round trip hides under the exchange).  The W2
    // fragments are NOT (32 registers across the exchange spill: measured 85 -> 88 us per fused launch)
    f32x4 c2v = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FUSED) {
        if (tid < 16) { if (p.bias2) c2v = *(const f32x4*)(p.bias2 + tid * 4); }
        else if (tid < 32) { if (p.se_wk) c2v = *(const f32x4*)(p.se_wk + (tid - 16) * 4); }
    }
    // ---- 2-way exchange between the K halves: wave (cs, kh) keeps position tiles 2 kh, 2 kh + 1 and hands the other two to its partner
    // (cs, 1 - kh).  a + b == b + a bitwise, so both partners form the same sum.
    f32x16 own[2];
    {
        char* mine = smem + wave * 8192 + lane * 16;
        const char* theirs = smem + (wave ^ 2) * 8192 + lane * 16;
        auto put = [&](const f32x16& a0, const f32x16& a1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16& a = j == 0 ? a0 : a1;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 vv = {a[4 * r4], a[4 * r4 + 1], a[4 * r4 + 2], a[4 * r4 + 3]};
                    *(f32x4*)(mine + (j * 4 + r4) * 1024) = vv;
                }
            }
        };
        if (kh == 0) { put(acc[2], acc[3]); own[0] = acc[0]; own[1] = acc[1]; }
        else { put(acc[0], acc[1]); own[0] = acc[2]; own[1] = acc[3]; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 vv = *(const f32x4*)(theirs + (j * 4 + r4) * 1024);
#pragma unroll
                for (int q = 0; q < 4; ++q) own[j][4 * r4 + q] += vv[q];
            }
    }
    C3P_PROF(2)

    // ---- a tile split across workgroups: publish the partial, take a ticket, the last arriver sums every part in part order.
    // Slab stores are write-through (sc1) and the last arriver reads with sc1 loads: no L2 write-back / invalidate fence on either
    // side (cdna_hip_programming.md section 5 "in-launch split-K reduction", the write-through form; as conv3.hip).
    if (nparts > 1) {
        const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, p.ws_bytes, 0x00020000);
        const uint32_t inslab = (uint32_t)((cs * 4 + 2 * kh) * 4096 + lane * 16);        // [cs][position tile][4][64 lanes][16 B]
        const uint32_t soff = (tile_lin * Smax + part) * (uint32_t)SLAB + inslab;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const i32x4 vv = {__float_as_int(own[j][4 * r4]), __float_as_int(own[j][4 * r4 + 1]),
                                  __float_as_int(own[j][4 * r4 + 2]), __float_as_int(own[j][4 * r4 + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(vv, rs_ws, (int)(soff + (uint32_t)((j * 4 + r4) * 1024)), 0, 16 /* sc1 */);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains before the ticket
        __syncthreads();
        uint32_t* flag = (uint32_t*)(smem + L::kFlag);
        if (tid == 0) {
            const uint32_t ticket = __hip_atomic_fetch_add(p.counters + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t last = ticket == nparts - 1 ? 1u : 0u;
            if (last) __hip_atomic_store(p.counters + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) { C3P_PROF(3) C3P_PROF_END return; }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) own[j][r] = 0.f;
        for (uint32_t q = 0; q < nparts; q += 2) {                // two parts' loads in flight together; odd tail: out of range -> zeros
            const uint32_t so0 = (tile_lin * Smax + q) * (uint32_t)SLAB + inslab;
            const uint32_t so1 = q + 1 < nparts ? so0 + (uint32_t)SLAB : p.ws_bytes;
            i32x4 v0[8], v1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v0[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (int)(so0 + (uint32_t)(k * 1024)), 0, 16 /* sc1 */);
#pragma unroll
            for (int k = 0; k < 8; ++k) v1[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (int)(so1 + (uint32_t)(k * 1024)), 0, 16 /* sc1 */);
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) own[k >> 2][4 * (k & 3) + jj] += __int_as_float(v0[k][jj]);
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) own[k >> 2][4 * (k & 3) + jj] += __int_as_float(v1[k][jj]);
        }
    }
    C3P_PROF(3)

    // ---- epilogue.  own[j][4 g + q] = channel n0 + 32 cs + 8 g + 4 hi + q of position 32 (2 kh + j) + l31 of the tile.
    const int64_t frame_row0 = ((int64_t)b * T + t) * H;
    const bool elu1 = p.act == MV2_ACT_ELU;
    if constexpr (OUT32) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int ph, pw; pos_of<TW>((2 * kh + j) * 32 + l31, ph, pw);
            const bool pos_ok = (h0 + ph) < H && (w0 + pw) < W_;
            const int64_t m = (frame_row0 + (h0 + ph)) * W_ + (w0 + pw);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xv = own[j][4 * g + q] + bv[g][q];
                    x[q] = elu1 ? Act<F32>::elu(xv) : xv;
                }
                const int n = n0 + cs * 32 + 8 * g + 4 * hi;
                if (pos_ok && n < p.co) *(f32x4*)((char*)p.out + (m * p.out_c + n) * 4) = x;
            }
        }
    } else if constexpr (!FUSED) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int ph, pw; pos_of<TW>((2 * kh + j) * 32 + l31, ph, pw);
            const bool pos_ok = (h0 + ph) < H && (w0 + pw) < W_;
            const int64_t m = (frame_row0 + (h0 + ph)) * W_ + (w0 + pw);
            uint32_t pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xv = own[j][4 * g + q] + bv[g][q];
                    const float e = Act<E>::elu(fminf(xv, 0.f));
                    x[q] = (elu1 && xv < 0.f) ? e : xv;
                }
                pk[g][0] = (uint32_t)f2bf(x[0]) | ((uint32_t)f2bf(x[1]) << 16);
                pk[g][1] = (uint32_t)f2bf(x[2]) | ((uint32_t)f2bf(x[3]) << 16);
            }
            // lanes l and l + 32 hold channels 8 g + {0..3} and 8 g + {4..7}: v_permlane32_swap of (g even, g odd) pairs gives lane
            // half 0 channels 16 gp + 0..7 and lane half 1 channels 16 gp + 8..15 -> one 16-byte store each (cdna_hip_programming.md T21)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
                const int n = n0 + cs * 32 + gp * 16 + hi * 8;
                if (pos_ok && n < p.co) {
                    const i32x4 vv = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                    *(i32x4*)((char*)p.out + (m * p.out_c + n) * 2) = vv;
                }
            }
        }
    } else {
        // ---- fused second stage (C = 64: the tile holds every channel of its positions).  Stage 1 -> bf16 tile in LDS, then wave w2 runs
        // out[co][pos] = sum_k W2[co][k] * tile[pos][k] for position tile w2 (K = 64 = four k-steps), bias2 + ELU, stores whole rows out of
        // a wave-private LDS tile and writes the SqueezeExcite pooling record of its 2 x 16 strip (conv3.hip's epilogue, unchanged values).
        constexpr int PITCH = ROWP;
        // W2 fragments straight from the plain packed image [row 64][128 B]: lane (row 32 a + l31, chunk 2 ks + hi)
        i32x4 w2f[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) w2f[a][ks] = *(const i32x4*)((const char*)p.wp2 + (size_t)(a * 32 + l31) * 128 + (size_t)(2 * ks + hi) * 16);
        if (tid < 32) *(f32x4*)(smem + L::kConst2 + tid * 16) = c2v;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            char* rowp = smem + L::kY1 + ((2 * kh + j) * 32 + l31) * PITCH;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xv = own[j][4 * g + q] + bv[g][q];
                    const float e = Act<E>::elu(fminf(xv, 0.f));
                    x[q] = (elu1 && xv < 0.f) ? e : xv;
                }
                lds_put4<E>(rowp + (cs * 32 + 8 * g + 4 * hi) * 2, x);
            }
        }
        __syncthreads();
        C3P_SUB(0)
        f32x16 acc2[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[a][r] = 0.f;
        char* priv = smem + L::kY1 + wave * 32 * PITCH;              // this wave's 32 rows of the stage-1 tile; they become its output tile
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const i32x4 fp = *(const i32x4*)(priv + l31 * PITCH + (2 * ks + hi) * 16);
#pragma unroll
            for (int a = 0; a < 2; ++a) Mma<E>::run(w2f[a][ks], fp, acc2[a]);
        }
        __builtin_amdgcn_wave_barrier();
        C3P_SUB(1)
        const bool elu2 = p.act2 == MV2_ACT_ELU;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = a * 32 + 8 * g + 4 * hi;
                const f32x4 b2 = *(const f32x4*)(smem + L::kConst2 + col * 4);
                float x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xv = acc2[a][4 * g + q] + b2[q];
                    const float e = Act<E>::elu(fminf(xv, 0.f));
                    x[q] = (elu2 && xv < 0.f) ? e : xv;
                }
                lds_put4<E>(priv + l31 * PITCH + col * 2, x);
            }
        __builtin_amdgcn_wave_barrier();
        C3P_SUB(2)
        {   // whole rows out: 8 lanes per 128-byte row, 16 bytes per lane
            const int ch = lane & 7, rr = lane >> 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rr + i * 8;
                int ph, pw; pos_of<TW>(wave * 32 + row, ph, pw);
                if ((h0 + ph) < H && (w0 + pw) < W_ && (n0 + ch * 8) < p.co) {
                    const int64_t m = (frame_row0 + (h0 + ph)) * W_ + (w0 + pw);
                    *(i32x4*)((char*)p.out + (m * p.out_c + n0 + ch * 8) * 2) = *(const i32x4*)(priv + row * PITCH + ch * 16);
                }
            }
        }
        C3P_SUB(3)
        if (p.se_part) {
            // SqueezeExcite pooling partial of this wave's 2 x 16 strip (reference SqueezeExcite.forward M:226-231; mv2_se_pool's record
            // {max logit, sum exp, exp-weighted channel sums}), from the bf16 tile in LDS
            float* ev = (float*)(smem + L::kEv + wave * 128);
            const int row = lane >> 1, half = lane & 1;
            float dot = 0.f;
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                float xv[8];
                Chunk<E>::unpack(*(const i32x4*)(priv + row * PITCH + (half * 32 + c8 * 8) * 2), xv);
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
#pragma unroll
            for (int r = 0; r < 32; r += 2) {
                a0 += ev[r] * bf2f(*(const unsigned short*)(priv + r * PITCH + lane * 2));
                a1 += ev[r + 1] * bf2f(*(const unsigned short*)(priv + (r + 1) * PITCH + lane * 2));
            }
            const int srow = thi * 4 + wave;               // strip row inside the frame (2 image rows each)
            if (srow * 2 < H) {                            // strips below the frame do not exist in the record list
                float* dst = p.se_part + ((((int64_t)b * T + t) * ((H + 1) / 2) + srow) * p.tiles_w_n + twi) * (BCO + 2);
                dst[2 + lane] = a0 + a1;
                if (lane == 0) { dst[0] = mx; dst[1] = sw; }
            }
            C3P_SUB(4)
        }
    }
    C3P_PROF(4)
    C3P_PROF_END
}

// ------------------------------------------------------------------------------- host side
namespace {

inline int cus_of_current_device_p() {      // not cached: a process may drive several devices
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

}  // namespace

namespace {
struct Plan3P {
    int tiles_h, tiles_w, nchunk, n_co_tiles, splitk;
    int nA, tilesA, SB, tiles_per_xcd;      // two-class work list (nA = 0: uniform), see the kernel's block decode
    int64_t M, n_tiles, grid;
    size_t slab_bytes;
};
Plan3P conv3p_plan(const mv2_conv_desc& d);
}  // namespace

// bf16 shapes this kernel takes (a subset of conv3.hip's eligible3; checked there first)
bool conv3p_eligible(const mv2_conv_desc& d) {
    return d.dtype == MV2_BF16 && d.Ci % 64 == 0 && d.kh == 3 && d.kw == 3 && d.kt >= 1 && d.kt <= 3 && d.co >= 64;
}

namespace {
Plan3P conv3p_plan(const mv2_conv_desc& d) {
    Plan3P P;
    P.tiles_h = (d.Ho + TH - 1) / TH;
    P.tiles_w = (d.Wo + TW - 1) / TW;
    P.nchunk = d.Ci / 64;
    P.n_co_tiles = (d.co + BCO - 1) / BCO;
    P.M = (int64_t)d.B * d.To * d.Ho * d.Wo;
    P.n_tiles = (int64_t)P.n_co_tiles * d.B * d.Ti * P.tiles_h * P.tiles_w;
    // K parts per tile.  A launch with at least as many tiles as workgroup slots (three per CU) is balanced by the dispatcher and keeps
    // whole tiles.  Below that (the deep levels at batch 1: 80-320 tiles of 24 groups for 768 slots) every tile's group list is cut into
    // parts of about eight groups: measured on MI355X (profiles/r04_k3_sweep_S.log, us per launch at S = 1 / 2 / 3 / 4 / 6) C = 512:
    // T = 20 77 / 69 / 68 / 68 / 75, T = 10 41 / 40 / 37 / 39 / 41, T = 5 36 / 25 / 21 / 22 / 24; C = 256 (12 groups, 640 tiles) 65 / 68 / 76:
    // a part pays ~10k clocks of fixed phases (prologue, exchange, slab + ticket) whatever its length, so fewer, longer parts win even
    // where the grid then has fewer workgroups than the chip has slots.
    const int64_t slots = (int64_t)3 * cus_of_current_device_p();
    const int64_t smax = std::min<int64_t>(8, (int64_t)P.nchunk * d.kt);
    int64_t s = P.n_tiles >= slots ? 1 : ((int64_t)P.nchunk * d.kt) / 8;
    s = std::max<int64_t>(1, std::min<int64_t>(s, smax));
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_C3P_S")) { const int v = atoi(e); if (v > 0) s = std::min<int64_t>(v, smax); }
#endif
    P.splitk = (int)s;
    P.nA = P.tilesA = P.SB = P.tiles_per_xcd = 0;
    P.grid = P.n_tiles * P.splitk;
    // Two-class list: the launch fits the slots once (slots / 8 <= tiles < slots: the measured regime below includes C = 512 at T = 20 with
    // 320 tiles for 768 slots; the per-split checks inside the loop - at least six groups per part, whole slot layers per class - are what
    // keeps T = 10 / T = 5 on the uniform rule) and the slot count is a multiple of 8.  The first
    // tiles in S0 parts, the last nB tiles in 2 S0 parts, with S0 (tiles + nB) == slots: every CU receives the same number of groups
    // instead of 3 against 2 tiles (C = 256 at batch 1: S0 = 1, 512 + 128 tiles; C = 512, T = 20: S0 = 2, 256 + 64 tiles).  Only where the
    // shorter parts keep at least six groups (a part pays ~10k clocks of fixed phases) and every count divides over the 8 XCDs; measured
    // against the uniform rule in profiles/r04_k3_two_class.log: C = 256 63.6 -> 60.7 us, C = 512 T = 20 65.1 -> 62.1 us.
    {
        const int64_t G = (int64_t)P.nchunk * d.kt;
        int two = 1;
#ifdef MV2_EXPERIMENTS
        if (const char* e = getenv("MV2_C3P_TWOCLASS")) two = atoi(e);
        if (getenv("MV2_C3P_S")) two = 0;
#endif
        if (two && slots % 8 == 0 && P.n_tiles < slots && 2 * P.n_tiles >= slots / 4) {
            for (int64_t s0 = 1; s0 <= 4; ++s0) {
                if (slots % s0) continue;
                const int64_t nB = slots / s0 - P.n_tiles, nAt = P.n_tiles - nB;
                if (nB < 0 || nAt < 0 || nB > P.n_tiles) continue;
                if (G / (2 * s0) < 6 || nAt % 8 || nB % 8 || P.n_tiles % 8) continue;
                const int64_t cus = slots / 3;
                if ((nAt * s0) % cus || (nB * 2 * s0) % cus) continue;      // whole slot layers per class: every CU gets the same mix
                if ((P.n_tiles * P.splitk) % cus == 0) break;               // the uniform split is balanced already
                if (nB == 0 && s0 == P.splitk) break;      // the uniform rule already is this split
                P.splitk = (int)s0; P.SB = (int)(2 * s0);
                P.tilesA = (int)(nAt / 8); P.nA = (int)(nAt / 8 * s0); P.tiles_per_xcd = (int)(P.n_tiles / 8);
                P.grid = slots;
                if (P.nA == 0) { P.nA = 0; P.splitk = P.SB; P.SB = 0; P.grid = P.n_tiles * P.splitk; }      // every tile in class B: a uniform split
                break;
            }
        }
    }
    const int smax_used = P.nA > 0 ? P.SB : P.splitk;
    P.slab_bytes = smax_used > 1 ? (size_t)P.n_tiles * smax_used * SLAB : 0;
    return P;
}
}  // namespace

size_t conv3p_workspace_bytes(const mv2_conv_desc& d) { return conv3p_plan(d).slab_bytes; }

template <bool FUSED, bool OUT32 = false>
static int launch3p(const P3Params& p, unsigned grid, hipStream_t st) {
    typedef PLds<FUSED> L;
    hipError_t e = hipFuncSetAttribute((const void*)conv3p_kernel<FUSED, OUT32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::kLds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3p_kernel<FUSED, OUT32>), dim3(grid), dim3(256), (int)L::kLds, st, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// called by mv2_conv3 (conv3.hip) after its argument checks
int conv3p_launch(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp3, const float* bias,
                  const void* wp2, const float* bias2, int act2, const float* se_wk, float se_bk, float* se_part, void* out,
                  void* workspace, size_t workspace_bytes, uint32_t* counters, size_t counter_bytes, hipStream_t st) {
    const Plan3P P = conv3p_plan(*d);
    const int64_t grid = P.grid;
    if (P.M >= (int64_t)0x7fffffff || grid >= (int64_t)0x7fffffff || P.slab_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    const size_t w_bytes = (size_t)d->nsteps * d->co_pad * 128;
    if (w_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    if (P.slab_bytes > 0 && (!workspace || workspace_bytes < P.slab_bytes)) return MV2_EWORKSPACE;
    if ((P.splitk > 1 || P.nA > 0) && (!counters || counter_bytes < (size_t)P.n_tiles * sizeof(uint32_t))) return MV2_EWORKSPACE;
    const uint32_t tiles_sp = (uint32_t)(P.tiles_h * P.tiles_w);
    P3Params pp{};
    pp.per_co = make_fastdiv((uint32_t)d->B * (uint32_t)d->Ti * tiles_sp);
    pp.nA = P.nA; pp.tilesA = P.tilesA; pp.SB = P.SB; pp.tiles_per_xcd = P.tiles_per_xcd; pp.dSB = make_fastdiv((uint32_t)std::max(P.SB, 1));
    pp.S = make_fastdiv((uint32_t)P.splitk); pp.tiles_sp = make_fastdiv(tiles_sp); pp.T = make_fastdiv((uint32_t)d->Ti); pp.tiles_w = make_fastdiv((uint32_t)P.tiles_w);
    pp.grid = (uint32_t)grid;
    pp.B = d->B; pp.Ti = d->Ti; pp.Hi = d->Hi; pp.Wi = d->Wi; pp.kt = d->kt; pp.nchunk = P.nchunk; pp.splitk = P.splitk; pp.tiles_w_n = P.tiles_w; pp.tiles_h_n = P.tiles_h;
    pp.Ci = d->Ci; pp.co = d->co; pp.co_pad = d->co_pad; pp.out_c = d->out_c; pp.act = d->act; pp.act2 = act2;
    pp.in_bytes = (uint32_t)in_bytes; pp.w_bytes = (uint32_t)w_bytes; pp.ws_bytes = (uint32_t)P.slab_bytes; pp.se_bk = se_bk;
    pp.in = in; pp.wp = wp3; pp.bias = bias; pp.out = out; pp.ws = (float*)workspace; pp.counters = counters;
    pp.wp2 = wp2; pp.bias2 = bias2; pp.se_wk = se_wk; pp.se_part = se_part;
    if (d->store == MV2_STORE_ROWS_F32) return wp2 ? MV2_EINVAL : launch3p<false, true>(pp, (unsigned)grid, st);
    return wp2 ? launch3p<true>(pp, (unsigned)grid, st) : launch3p<false>(pp, (unsigned)grid, st);
}

#ifdef MV2_EXPERIMENTS
extern "C" int mv2_conv3p_set_profile(void* device_buffer) {      // experiments library only (tools/k3_bench)
    unsigned long long* v = (unsigned long long*)device_buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv3p_prof), &v, sizeof(v));
}
#endif
