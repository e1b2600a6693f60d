// mv2_conv: implicit-GEMM convolution on MFMA with fused epilogues (see include/mv2.h).
//
// Formulation (transposed so the epilogue stores 4 consecutive channels per lane):
//     D[co][pos] = sum_k Wp[k][co] * A[pos][k]
// MFMA "A" operand = packed weight rows (co), "B" operand = gathered activation rows (pos).
// One K step = 128 bytes per row (64 bf16 / 32 fp32) = 8 chunks of 16 B.  Both operands are
// staged global -> VGPR -> LDS (double buffered, one barrier per step) in an XOR-swizzled
// [row][8 chunk] image: physical chunk = chunk ^ ((row >> 1) & 7), which makes the
// ds_read_b128 fragment reads of a 32-row MFMA operand conflict-free (MI355X_MICROARCH §LDS:
// ds_read_b128 is served in 16-lane groups over 64 banks).
// Activation gathers go through a raw buffer descriptor: a tap that falls into the causal /
// "same" zero padding is redirected to offset = num_records, for which the hardware returns 0,
// so the halo costs no VALU selects and no branches.
#include <cstdio>
#include <cstdlib>
#include "conv_common.h"

#ifdef MV2_EXPERIMENTS
// phase clocks per workgroup (tools/k3_bench): [0] prologue (row decode, first loads) [1] K loop [2] epilogue [7] whole kernel
__device__ unsigned long long* g_conv_prof = nullptr;
extern "C" int mv2_conv_set_profile(void* device_buffer) {
    unsigned long long* v = (unsigned long long*)device_buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof), &v, sizeof(v));
}
#define CV_PROF_DECL unsigned long long pf_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_ = __builtin_amdgcn_s_memtime(); const unsigned long long pt0_ = pt_;
#define CV_PROF(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_[k] += n_ - pt_; pt_ = n_; }
#define CV_PROF_END if (g_conv_prof && threadIdx.x == 0 && blockIdx.y == 0) { pf_[7] = __builtin_amdgcn_s_memtime() - pt0_; for (int i_ = 0; i_ < 8; ++i_) g_conv_prof[(size_t)blockIdx.x * 8 + i_] = pf_[i_]; }
#else
#define CV_PROF_DECL
#define CV_PROF(k)
#define CV_PROF_END
#endif

// ------------------------------------------------------------------------------- main kernel
template <typename E, int BCO, int BPOS, int WCO, int WPOS, int NSET>
__global__ __launch_bounds__(256, 2) void conv_kernel(const ConvParams p) {
    static_assert(NSET == 2 || NSET == 4, "register sets of the load pipeline (even: LDS buffer = set parity)");
    constexpr int NWPOS = BPOS / (32 * WPOS);
    static_assert((BCO / (32 * WCO)) * NWPOS == 4, "4 waves per block");
    constexpr int RW = BCO / 32;    // weight rows staged per thread
    constexpr int RP = BPOS / 32;   // position rows staged per thread
    constexpr int STAGE = (BCO + BPOS) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const mv2_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid & 7, r0 = tid >> 3;
    const int hi = lane >> 5, l31 = lane & 31;
    CV_PROF_DECL
#if defined(MV2_EXPERIMENTS) && defined(MV2_CONV_AGPR)
    // experiment: a dummy use of the accumulator file switches the compiler's MFMAs to the AGPR form (C / D in a[...]); under the
    // 256-register budget it otherwise selects the VGPR form (see conv3.hip MV2_C3_AGPR and tools/mfma_cadence.cpp)
    { float z_ = 0.f; asm volatile("; this kernel uses the accumulator file" : "+a"(z_)); }
#endif

    // ---- block -> (co tile, pos tile); XCD-aware: blocks of one XCD walk neighbouring tiles
    int bid = blockIdx.x;
    {
        const int nblk = p.n_co_tiles * p.n_pos_tiles;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int co_tile = bid % p.n_co_tiles, pos_tile = bid / p.n_co_tiles;
    const int n0 = co_tile * BCO;
    const int64_t m0 = (int64_t)pos_tile * BPOS;
    const int z = blockIdx.y;
    const int s_begin = z * p.steps_per_split;
    const int s_end = min(d.nsteps, s_begin + p.steps_per_split);

    // ---- gather state of the block's BPOS position rows: {byte offset of the tap origin (mod 2^32), tap-validity
    // mask}.  Each row is decoded ONCE (one thread per row, 32-bit divisions, closed-form masks) and shared through
    // LDS: the eight lanes of a row used to repeat three 64-bit divisions for each of their RP rows, which was a
    // ~1200-instruction prologue -- most of the fixed ~15 us floor of the small launches.
    uint32_t rowbase[RP], rowmask[RP];
    // 1x1x1, stride 1, no padding: input row == output row, nothing to decode and nothing to mask.  The table + two barriers below are
    // ~2 us of a launch whose whole K loop is 4-22 steps (the feed-forward / projection / up-sampling GEMMs: 30 launches per clip).
    const bool linear_rows = d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0 &&
                             d.To == d.Ti && d.Ho == d.Hi && d.Wo == d.Wi;
    if (linear_rows) {
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int64_t m = m0 + r0 + 32 * i;
            rowbase[i] = (uint32_t)m * (uint32_t)d.Ci * (uint32_t)E::kBytes;
            rowmask[i] = m < p.M ? 0x00ffffffu : 0u;
        }
    } else {
        uint32_t* tab = (uint32_t*)smem;        // [BPOS][2]; LDS is idle before the K loop
        for (int r = tid; r < BPOS; r += 256) {
            const int64_t m = m0 + r;
            uint32_t base = 0, mk = 0;
            if (m < p.M) {
                int b, to, ho, wo; decode_pos<E>(d, m, b, to, ho, wo);
                const int t_in = to * d.st - d.pt, h_in = ho * d.sh - d.ph, w_in = wo * d.sw - d.pw;
                // wrapping 32-bit arithmetic: base + tap offset is a true offset < in_bytes < 2^32 whenever the tap is valid
                base = ((((uint32_t)b * (uint32_t)d.Ti + (uint32_t)t_in) * (uint32_t)d.Hi + (uint32_t)h_in) * (uint32_t)d.Wi +
                        (uint32_t)w_in) * (uint32_t)d.Ci * (uint32_t)E::kBytes;
                mk = 0x00ffffffu;
                if (d.check_bounds) {
                    auto span = [](int in0, int k, int n) -> uint32_t {    // bits j in [0, k) with 0 <= in0 + j < n
                        const int lo = max(0, -in0), hi_ = min(k, n - in0);
                        return hi_ > lo ? (((1u << hi_) - 1u) & ~((1u << lo) - 1u)) : 0u;
                    };
                    mk = span(t_in, d.kt, d.Ti) | (span(h_in, d.kh, d.Hi) << 8) | (span(w_in, d.kw, d.Wi) << 16);
                }
            }
            tab[2 * r] = base; tab[2 * r + 1] = mk;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RP; ++i) { rowbase[i] = tab[2 * (r0 + 32 * i)]; rowmask[i] = tab[2 * (r0 + 32 * i) + 1]; }
        __syncthreads();
    }
    // epilogue operands that only depend on the tile are requested now: their round trip hides under the K loop
    static_assert(BPOS <= 256 && BCO <= 256, "one prefetched epilogue operand per thread");
    const float bias_pf = (p.bias && tid < BCO && n0 + tid < d.co_pad) ? p.bias[n0 + tid] : 0.f;
    const float rs_pf = (p.rowscale && tid < BPOS && m0 + tid < p.M) ? p.rowscale[m0 + tid] : 1.f;
    const int lg = p.lg_seg_chunks;
    const int seg_sub = c >> lg;
    const uint32_t seg_inner = (uint32_t)(c & ((1 << lg) - 1)) * 16u;
    const int segs_per_step = 8 >> lg;
    const uint32_t w_off0 = (uint32_t)(n0 + r0) * 128u + (uint32_t)c * 16u;
    const uint32_t w_step = (uint32_t)d.co_pad * 128u;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.w_bytes, 0x00020000);

    // LDS write offsets (swizzled chunk is thread-constant: ((r0 + 32 i) >> 1) & 7 == (r0 >> 1) & 7)
    const uint32_t lds_wr = (uint32_t)r0 * 128u + (uint32_t)((c ^ ((r0 >> 1) & 7)) << 4);
    // LDS fragment read offsets
    const int wco = wave / NWPOS, wpos = wave % NWPOS;
    const uint32_t swz = (uint32_t)((l31 >> 1) & 7);
    const uint32_t rd_w = (uint32_t)(wco * 32 * WCO + l31) * 128u;
    const uint32_t rd_p = (uint32_t)(BCO + wpos * 32 * WPOS + l31) * 128u;

    f32x16 acc[WCO][WPOS];
#pragma unroll
    for (int a = 0; a < WCO; ++a)
#pragma unroll
        for (int b = 0; b < WPOS; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // NSET register sets, two LDS buffers (both selected statically): while step s computes out of LDS buffer s % 2, step s + 1 (in
    // flight since NSET steps) is written to the other buffer and the loads of step s + NSET are issued into the set step s has just
    // left, so every gather has NSET steps of flight time before it is waited for.  NSET = 4 was measured on the small tiles (r03h:
    // every load of a <= 4-step launch in flight before the first wait) and is NOT faster (qkv 13.8 -> 15.8 us, out 8.9 -> 9.6,
    // kv 13.2 -> 14.7: the launches are not bound by the depth of this pipeline); every instantiation uses 2.
    i32x4 gw[NSET][RW], gp[NSET][RP];
    float rss[RP];                  // rms_fused: this thread's share (one 16-byte chunk per step) of the sum of squares of its RP rows
#pragma unroll
    for (int i = 0; i < RP; ++i) rss[i] = 0.f;

    // NB every load below is issued UNCONDITIONALLY (steps past the end read the out-of-range offset: zeros,
    // no traffic).  A load inside a branch makes the wait-count pass fall back to vmcnt(0) at the next LDS
    // write, which silently removes all flight time from the prefetch.
    auto load_seg = [&](int s) { return p.segs[min(s, d.nsteps - 1) * segs_per_step + seg_sub]; };
    auto issue_loads = [&](int s, const mv2_seg sg, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        const bool live = s < s_end;
        const uint32_t so = (uint32_t)sg.elem_offset * (uint32_t)E::kBytes + seg_inner;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const bool ok = live && (rowmask[i] & sg.sel) == sg.sel;
            const uint32_t vo = ok ? rowbase[i] + so : p.in_bytes;
            gp[PAR][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)vo, 0, 0);
        }
        const uint32_t wo_ = live ? w_off0 + (uint32_t)s * w_step : p.w_bytes;
#pragma unroll
        for (int i = 0; i < RW; ++i)
            gw[PAR][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wo_ + (uint32_t)i * 32u * 128u), 0, 0);
    };
    auto write_lds = [&](auto par_c) {                 // register set PAR -> LDS buffer PAR % 2
        constexpr int PAR = decltype(par_c)::value;
        char* base = smem + (PAR & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < RW; ++i) *(i32x4*)(base + lds_wr + i * 32 * 128) = gw[PAR][i];
#pragma unroll
        for (int i = 0; i < RP; ++i) *(i32x4*)(base + BCO * 128 + lds_wr + i * 32 * 128) = gp[PAR][i];
        if (p.rms_fused) {          // every staged chunk passes here exactly once (the dummy behind the last step is zeros)
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                float f[8];
                Chunk<E>::unpack(gp[PAR][i], f);
#pragma unroll
                for (int j = 0; j < E::kVec; ++j) rss[i] += f[j] * f[j];
            }
        }
    };

    if (s_begin < s_end) {
        // parity is relative to s_begin: step s_begin uses set / buffer 0
        // vmcnt retires IN ORDER: a descriptor fetched after a batch of gathers cannot be consumed before that
        // batch has landed.  So every descriptor is fetched BEFORE the gathers issued in the same step.
        mv2_seg sgp[NSET];
#pragma unroll
        for (int k = 0; k < NSET; ++k) sgp[k] = load_seg(s_begin + k);
        mv2_seg sg = load_seg(s_begin + NSET);
        issue_loads(s_begin, sgp[0], IC<0>{});
        issue_loads(s_begin + 1, sgp[1], IC<1>{});
        if constexpr (NSET == 4) { issue_loads(s_begin + 2, sgp[2], IC<2>{}); issue_loads(s_begin + 3, sgp[3], IC<3>{}); }
        write_lds(IC<0>{});
        __syncthreads();
        CV_PROF(0)
        auto step = [&](int s, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
            const mv2_seg nxt = load_seg(s + NSET + 1);
            issue_loads(s + NSET, sg, IC<PAR>{});          // set PAR held step s, which is in LDS since the previous step
            sg = nxt;
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetch at the top of the step (the scheduler would sink it)
            const char* base = smem + (PAR & 1) * STAGE;
            // fragment reads software-pipelined over two register sets
            i32x4 fw[2][WCO], fp[2][WPOS];
            auto read_frags = [&](int ks, auto set_c) {
                constexpr int SET = decltype(set_c)::value;
                const uint32_t ck = (uint32_t)(((ks * 2 + hi) ^ swz) << 4);
#pragma unroll
                for (int a = 0; a < WCO; ++a) fw[SET][a] = *(const i32x4*)(base + rd_w + a * 32 * 128 + ck);
#pragma unroll
                for (int b = 0; b < WPOS; ++b) fp[SET][b] = *(const i32x4*)(base + rd_p + b * 32 * 128 + ck);
            };
            auto mma_frags = [&](auto set_c) {
                constexpr int SET = decltype(set_c)::value;
#pragma unroll
                for (int a = 0; a < WCO; ++a)
#pragma unroll
                    for (int b = 0; b < WPOS; ++b) Mma<E>::run(fw[SET][a], fp[SET][b], acc[a][b]);
            };
            read_frags(0, IC<0>{});
            read_frags(1, IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            mma_frags(IC<0>{});
            read_frags(2, IC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            mma_frags(IC<1>{});
            read_frags(3, IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            mma_frags(IC<0>{});
            mma_frags(IC<1>{});
            write_lds(IC<(PAR + 1) % NSET>{});       // unconditional: after the last step it stores the zero dummy
            __syncthreads();
        };
        int s = s_begin;
        if constexpr (NSET == 2) {
            for (; s + 1 < s_end; s += 2) {
                step(s, IC<0>{});
                step(s + 1, IC<1>{});
            }
            if (s < s_end) step(s, IC<0>{});
        } else {
            for (; s + 3 < s_end; s += 4) {
                step(s, IC<0>{});
                step(s + 1, IC<1>{});
                step(s + 2, IC<2>{});
                step(s + 3, IC<3>{});
            }
            if (s < s_end) step(s, IC<0>{});
            if (s + 1 < s_end) step(s + 1, IC<1>{});
            if (s + 2 < s_end) step(s + 2, IC<2>{});
        }
    }
    CV_PROF(1)

    // ---- epilogue: acc[a][b][r] is D[co = 8*(r>>2) + 4*hi + (r&3)][pos = l31] of tile (a, b)
    const int nbase = n0 + wco * 32 * WCO;
    const int64_t mbase = m0 + wpos * 32 * WPOS;
    if (p.splitk > 1) {
#pragma unroll
        for (int a = 0; a < WCO; ++a)
#pragma unroll
            for (int b = 0; b < WPOS; ++b) {
                const int64_t m = mbase + b * 32 + l31;
                float* row = p.ws + ((int64_t)z * p.ws_rows + m) * p.ws_cols + nbase + a * 32 + 4 * hi;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                    *(f32x4*)(row + 8 * g) = v;
                }
            }
        return;
    }
    if (stage_ok(p)) {   // LDS-staged, fully coalesced stores (the K loop ended on a barrier: LDS is free)
        if (p.rms_fused) {   // the eight lanes of a row hold its eight chunk columns: fold, then one lane per row publishes 1 / |row|
            float* rs_lds = (float*)(smem + StageCfg<E, BCO, BPOS>::kRowScale);
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                float ss = rss[i];
                ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
                if (c == 0) rs_lds[r0 + 32 * i] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
            }
        }
        staged_epilogue<E, BCO, BPOS, WCO, WPOS>(p, smem, acc, n0, wco, wpos, tid, [&](int lp) { return m0 + lp; }, true, p.rms_fused != 0, true, rs_pf, bias_pf);
        CV_PROF(2)
        CV_PROF_END
        return;
    }
    if (d.act == MV2_ACT_GEGLU) {
        if constexpr (WCO == 2) {
#pragma unroll
            for (int b = 0; b < WPOS; ++b) {
                const int64_t m = mbase + b * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float x[4], gt[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { x[j] = acc[0][b][4 * g + j]; gt[j] = acc[1][b][4 * g + j]; }
                    const int off = 8 * g + 4 * hi;
                    epi_geglu4<E>(p, m, nbase + off, (nbase >> 1) + off, x, gt);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < WCO; ++a)
#pragma unroll
        for (int b = 0; b < WPOS; ++b) {
            const int64_t m = mbase + b * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[a][b][4 * g + j];
                epi_store4<E>(p, m, nbase + a * 32 + 8 * g + 4 * hi, v);
            }
        }
}

// ------------------------------------------------------------------------------- pointwise kernel: LDS-DMA staging
// 1x1x1, stride-1, un-padded convolutions over whole 128-byte K steps (the feed-forward / projection / up-sampling GEMMs of the path:
// reference magvit2_pytorch.py:298-339 FeedForward, :380-470 the attention projections, :760-800 the up-sampling convs) are plain
// GEMMs: input row == output row, nothing to gather and nothing to mask.  For them the VGPR round trip of conv_kernel (16 B loads ->
// registers -> ds_write -> barrier) is pure overhead: profiles/r03_generic_default_phase_clocks.log has its K loop at 27 % of the MFMA
// rate with the LDS-write path and the L1s both half busy.  Here both operands go global -> LDS DIRECTLY (buffer_load ... lds): a
// wave-instruction fills 8 rows x 128 bytes of a stage (lane l -> row l / 8, 16-byte slot l % 8) and the XOR swizzle of the stage
// image (the one conv_kernel writes) is applied on the SOURCE side: slot s of row r receives global chunk s ^ ((r >> 1) & 7).
// No staging registers, no ds_write; the waits are hand-placed (a __syncthreads() would wait for vmcnt(0) and drain the pipeline),
// one bare s_barrier per step.  Same packed weights, same MFMA order, same epilogue: results are bit-identical to conv_kernel's.
// Measured on MI355X (profiles/r03_pw_dma.log, prototype tools/experiments/pw_dma_bench.cpp): ff1 C=512 32.8 -> 22.7 us,
// ff2 21.7 -> 16.5, ff1 C=256 41.4 -> 31.6, ff2 22.1 -> 14.6, qkv 13.8 -> 9.5, out 8.9 -> 6.0, kv 13.2 -> 8.2, up 128->256 46 -> 19.
// (prototype figures, trivial epilogue; behind the real epilogue and inside the step graph the gain is smaller, see pw_dma_ok below).
template <int N_> struct WaitVm;
#define MV2_WAITVM(n) template <> struct WaitVm<n> { static __device__ __forceinline__ void run() { asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); } };
MV2_WAITVM(0) MV2_WAITVM(8)
#undef MV2_WAITVM
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// workgroups per CU the dispatch rule counts on (LDS: 2 stages x (BCO + BPOS) x 128 B): 128 x 128 -> 2, 128 x 64 / 64 x 128 -> 3, 64 x 64 -> 5;
// the register budget follows (nothing tied the two together)
template <int BCO, int BPOS> struct PwDmaOcc { static constexpr int kWgs = BCO + BPOS >= 256 ? 2 : (BCO + BPOS >= 192 ? 3 : 5); };

//
// GATHER (round 6): the same pipeline for convolutions WITH taps, strides and zero padding whose K steps are whole 128-byte pieces of one
// input row each (Ci % 64 == 0: the space / time down-sampling convolutions, reference magvit2_pytorch.py:770-807).  The only difference is
// where a position row's 128 bytes come from: its tap origin (decoded once per row, as conv_kernel does) + the step's segment offset, and a
// tap that falls into the padding reads the out-of-range offset, for which the DMA writes zeros.  Same stage image, same MFMA order, same
// epilogue: bit-identical to conv_kernel's un-split result.
template <int BCO, int BPOS, int WCO, int WPOS, int NS, bool RMS, bool GATHER = false>
__global__ __launch_bounds__(256, (PwDmaOcc<BCO, BPOS>::kWgs)) void conv_pw_dma_kernel(const ConvParams p) {
    typedef BF16 E;
    static_assert(!(RMS && GATHER), "the fused row norm reads whole rows: pointwise launches only");
    constexpr int NWPOS = BPOS / (32 * WPOS);
    static_assert((BCO / (32 * WCO)) * NWPOS == 4, "4 waves per block");
    constexpr int STAGE = (BCO + BPOS) * 128;
    constexpr int LPW = BCO / 32;                                // DMA loads per thread and step that fetch weight rows ...
    constexpr int LPS = (BCO + BPOS) / 32;                       // ... and in total (128 x 128: 4 weight + 4 activation)
    static_assert(NS == 2, "two stages: the vmcnt immediates below and the launcher (launch_pw_dma) know no other depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const mv2_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / NWPOS, wpos = wave % NWPOS;
    int bid = blockIdx.x;
    {   // XCD-aware: blocks of one XCD walk neighbouring tiles (as conv_kernel)
        const int nblk = p.n_co_tiles * p.n_pos_tiles;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int co_tile = bid % p.n_co_tiles, pos_tile = bid / p.n_co_tiles;
    const int n0 = co_tile * BCO;
    const int64_t m0 = (int64_t)pos_tile * BPOS;
    const int nsteps = d.nsteps;
    // epilogue operands that only depend on the tile: requested now, their round trip hides under the K loop
    const float bias_pf = (p.bias && tid < BCO && n0 + tid < d.co_pad) ? p.bias[n0 + tid] : 0.f;
    const float rs_pf = (p.rowscale && tid < BPOS && m0 + tid < p.M) ? p.rowscale[m0 + tid] : 1.f;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.w_bytes, 0x00020000);
    // per-thread source offsets: load i covers rows 8 (wave + 4 i) + lane / 8 of the stage image [weight rows BCO | position rows BPOS].
    // Rows past M read the out-of-range offset (no memory access); what lands in their LDS rows only reaches output rows that are never stored.
    uint32_t voff[LPS];
    uint32_t rowmask[LPS];                                       // GATHER: tap-validity bits of the load's position row (conv_kernel's)
    if constexpr (GATHER) {
        // one thread per position row decodes it (32-bit divisions, closed-form masks) into LDS, idle before the first DMA
        uint32_t* tab = (uint32_t*)smem;        // [BPOS][2]
        for (int r = tid; r < BPOS; r += 256) {
            const int64_t m = m0 + r;
            uint32_t base = 0, mk = 0;
            if (m < p.M) {
                int b, to, ho, wo; decode_pos<E>(d, m, b, to, ho, wo);
                const int t_in = to * d.st - d.pt, h_in = ho * d.sh - d.ph, w_in = wo * d.sw - d.pw;
                // wrapping 32-bit arithmetic: base + tap offset is a true offset < in_bytes < 2^32 whenever the tap is valid
                base = ((((uint32_t)b * (uint32_t)d.Ti + (uint32_t)t_in) * (uint32_t)d.Hi + (uint32_t)h_in) * (uint32_t)d.Wi +
                        (uint32_t)w_in) * (uint32_t)d.Ci * 2u;
                mk = 0x00ffffffu;
                if (d.check_bounds) {
                    auto span = [](int in0, int k, int n) -> uint32_t {    // bits j in [0, k) with 0 <= in0 + j < n
                        const int lo = max(0, -in0), hi_ = min(k, n - in0);
                        return hi_ > lo ? (((1u << hi_) - 1u) & ~((1u << lo) - 1u)) : 0u;
                    };
                    mk = span(t_in, d.kt, d.Ti) | (span(h_in, d.kh, d.Hi) << 8) | (span(w_in, d.kw, d.Wi) << 16);
                }
            }
            tab[2 * r] = base; tab[2 * r + 1] = mk;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int row = 8 * (wave + 4 * i) + (lane >> 3);
            const uint32_t chunk = (uint32_t)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
            if (i < LPW) { voff[i] = (uint32_t)(n0 + row) * 128u + chunk; rowmask[i] = 0u; }
            else { voff[i] = tab[2 * (row - BCO)] + chunk; rowmask[i] = tab[2 * (row - BCO) + 1]; }
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            const int row = 8 * (wave + 4 * i) + (lane >> 3);
            const uint32_t chunk = (uint32_t)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
            rowmask[i] = 0u;
            if (i < LPW) voff[i] = (uint32_t)(n0 + row) * 128u + chunk;
            else {
                const int64_t m = m0 + row - BCO;
                voff[i] = m < p.M ? (uint32_t)m * (uint32_t)d.Ci * 2u + chunk : p.in_bytes;
            }
        }
    }
    const uint32_t w_step = (uint32_t)d.co_pad * 128u;
    // GATHER: the K program (one segment per step: byte offset of the tap + channel piece, tap-selection bits) is read one step ahead of
    // its use through the scalar cache
    auto load_seg = [&](int s) { return p.segs[min(s, nsteps - 1)]; };
    mv2_seg sg_next = {0, 0};
    if constexpr (GATHER) sg_next = load_seg(0);
    auto issue = [&](int s, int stage) {          // every load unconditional: steps past the end read out of range (the check does not see soffset)
        const bool live = s < nsteps;
        const uint32_t so_w = live ? (uint32_t)s * w_step : 0u, so_x = live ? (uint32_t)s * 128u : 0u;
        const mv2_seg sg = sg_next;
        if constexpr (GATHER) sg_next = load_seg(s + 1);
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(smem + stage * STAGE + 8 * (wave + 4 * i) * 128);
            if (i < LPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, (int)(live ? voff[i] : p.w_bytes), (int)so_w, 0, 0);
            else if constexpr (GATHER) {
                // the tap offset goes into the VECTOR offset: a row at the padding border has a wrapped (huge) tap origin that only becomes a
                // true offset once the tap is added, and the range check sees the vector offset alone
                const bool ok = live && (rowmask[i] & sg.sel) == sg.sel;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, dst, 16, (int)(ok ? voff[i] + (uint32_t)sg.elem_offset * 2u : p.in_bytes), 0, 0, 0);
            } else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, dst, 16, (int)(live ? voff[i] : p.in_bytes), (int)so_x, 0, 0);
        }
    };
    f32x16 acc[WCO][WPOS];
#pragma unroll
    for (int a = 0; a < WCO; ++a)
#pragma unroll
        for (int b = 0; b < WPOS; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float rss[WPOS];                                             // RMS: sum of squares of this lane's chunks of rows b * 32 + l31 (waves wco == 0)
#pragma unroll
    for (int b = 0; b < WPOS; ++b) rss[b] = 0.f;
    const uint32_t swz = (uint32_t)((l31 >> 1) & 7);
    const uint32_t rd_w = (uint32_t)(wco * 32 * WCO + l31) * 128u, rd_p = (uint32_t)(BCO + wpos * 32 * WPOS + l31) * 128u;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s);                // prologue: NS - 1 steps in flight
    int stage = 0;
    for (int s = 0; s < nsteps; ++s) {
        WaitVm<(NS - 2) * LPS>::run();                           // this thread's loads of step s have landed ...
        __builtin_amdgcn_s_barrier();                            // ... everybody's, and everybody has read the stage that is refilled next
        const int nxt = stage == 0 ? NS - 1 : stage - 1;         // (s + NS - 1) % NS
        issue(s + NS - 1, nxt);
        const char* base = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t ck = (uint32_t)(((ks * 2 + hi) ^ swz) << 4);
            i32x4 fw[WCO], fp[WPOS];
#pragma unroll
            for (int a = 0; a < WCO; ++a) fw[a] = *(const i32x4*)(base + rd_w + a * 32 * 128 + ck);
#pragma unroll
            for (int b = 0; b < WPOS; ++b) fp[b] = *(const i32x4*)(base + rd_p + b * 32 * 128 + ck);
#pragma unroll
            for (int a = 0; a < WCO; ++a)
#pragma unroll
                for (int b = 0; b < WPOS; ++b) Mma<E>::run(fw[a], fp[b], acc[a][b]);
            if constexpr (RMS) {
                if (wco == 0) {                                  // wave-uniform: the two waves of a position half read the same rows
#pragma unroll
                    for (int b = 0; b < WPOS; ++b) {
                        // NB bit_cast of the WHOLE fragment, then element pairs: __builtin_bit_cast(bf16x2, fp[b][q]) on a vector element
                        // compiles to element 0 for every q (hipcc 7.2, seen in the ISA)
                        const bf16x8 h = __builtin_bit_cast(bf16x8, fp[b]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bf16x2 v = {h[2 * q], h[2 * q + 1]};
                            rss[b] = __builtin_amdgcn_fdot2_f32_bf16(v, v, rss[b], false);
                        }
                    }
                }
            }
        }
        stage = stage == NS - 1 ? 0 : stage + 1;
    }
    WaitVm<0>::run();
    __syncthreads();                                             // every DMA write has landed, every fragment read is done: LDS is free
    if constexpr (RMS) {    // lanes l31 and l31 + 32 hold the two halves of a row's chunks: fold, one lane per row publishes 1 / |row|
        float* rs_lds = (float*)(smem + StageCfg<E, BCO, BPOS>::kRowScale);
        if (wco == 0) {
#pragma unroll
            for (int b = 0; b < WPOS; ++b) {
                float ss = rss[b];
                ss += __shfl_xor(ss, 32, 64);
                if (hi == 0) rs_lds[wpos * 32 * WPOS + b * 32 + l31] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
            }
        }
    }
    staged_epilogue<E, BCO, BPOS, WCO, WPOS>(p, smem, acc, n0, wco, wpos, tid, [&](int lp) { return m0 + lp; }, true, RMS, true, rs_pf, bias_pf);
}

// split-K finish: sum the fp32 slabs and run the same epilogue
template <typename E>
__global__ __launch_bounds__(256) void conv_finish_kernel(const ConvParams p) {
    const mv2_conv_desc& d = p.d;
    const bool geglu = d.act == MV2_ACT_GEGLU;
    const int groups = geglu ? (d.co / 8) : ((d.co + 3) / 4);   // 4-wide column groups per row
    const int64_t total = p.M * groups;
    const bool small = total < (int64_t)0x7fffffff;   // 32-bit index math whenever it fits (always, in practice)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = small ? (int64_t)((uint32_t)idx / (uint32_t)groups) : idx / groups;
        const int gi = (int)(idx - m * groups);
        if (!geglu) {
            const int n = gi * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < p.splitk; ++z) {
                const f32x4 t = *(const f32x4*)(p.ws + ((int64_t)z * p.ws_rows + m) * p.ws_cols + n);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            }
            epi_store4<E>(p, m, n, v);
        } else {
            const int blk = gi / 8, i4 = (gi % 8) * 4;
            const int nx = blk * 64 + i4;
            float x[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < p.splitk; ++z) {
                const float* row = p.ws + ((int64_t)z * p.ws_rows + m) * p.ws_cols;
                const f32x4 tx = *(const f32x4*)(row + nx), tg = *(const f32x4*)(row + nx + 32);
#pragma unroll
                for (int j = 0; j < 4; ++j) { x[j] += tx[j]; g[j] += tg[j]; }
            }
            epi_geglu4<E>(p, m, nx, blk * 32 + i4, x, g);
        }
    }
}

// ------------------------------------------------------------------------------- host side
namespace {

struct TileCfg { int bco, bpos; };

inline int device_cus();

inline TileCfg pick_tile(const mv2_conv_desc& d) {
    // Launches whose 128x128 tiling gives fewer than 2 blocks per CU (the deep levels at batch 1: M = 1280..20480
    // rows) are chains of latencies, not MFMA work.  They use 64x64 tiles (64x128 for GEGLU, which needs value and
    // gate in one wave): 4x the blocks, a quarter of the per-block epilogue, several blocks per CU to overlap the
    // chains.  Measured at C = 512: 1x1x1 conv 19 -> 9-12 us, to_out 13.5 -> 6.8 us; C = 256: 20.5 -> 16 us.
    // (experiments build: MV2_SMALL_TILE=0 disables, =1 restricts the rule to launches with fewer than 1 block per CU)
#ifdef MV2_EXPERIMENTS
    static const int small_policy = getenv("MV2_SMALL_TILE") ? atoi(getenv("MV2_SMALL_TILE")) : 2;
#else
    constexpr int small_policy = 2;
#endif
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_TILE")) {        // tools/k3_bench tile sweep: "bco,bpos" (read per call)
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && !(d.act == MV2_ACT_GEGLU && a == 32)) return {a, b};
    }
#endif
    const int64_t M = (int64_t)d.B * d.To * d.Ho * d.Wo;
    const int64_t blocks128 = ((d.co + 127) / 128) * ((M + 127) / 128);
    const bool small = small_policy > 0 && d.co >= 64 && blocks128 < (int64_t)small_policy * device_cus();      // product: fewer than 2 tiles per CU
    if (d.act == MV2_ACT_GEGLU) {
        if (d.co <= 64) return {64, 256};
        return small ? TileCfg{64, 128} : TileCfg{128, 128};
    }
    if (small) return {64, 64};
    if (d.co <= 32) return {32, 256};
    if (d.co <= 64) return {64, 256};
    return {128, 128};
}

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// conv_pw_dma_kernel instead of conv_kernel: bf16 1x1x1 stride-1 un-padded convolutions over whole 128-byte K steps of contiguous input
// rows, stored through the staged epilogue.  The launch never splits K.  Which tile (measured per launch and in the step graph on MI355X,
// profiles/r04_pw_tile_sweep.log / r04_pw_rule_ab.log; "tiles" below = 128 x 128 tiles of the launch):
//   GEGLU (value and gate of a packed column pair in one wave):  >= 256 tiles -> 128 x 128 (ff1 C = 256: 39.2 us against 40.0-40.4 with the
//         smaller tiles, at batch 4 157 against 168); below: conv_kernel, as before;
//   other epilogues: >= 200 tiles -> 128 x 64 (48 KB of LDS: three workgroups per CU instead of two, half the epilogue chain per tile:
//         kv 22.5 -> 17.2 us, up-sampling 256 -> 4 x 128: 28.8 -> 21.7, ff2 C = 256: 21.0 -> 18.0, out 13.6 -> 11.1; at batch 4 up-sampling
//         128 -> 4 x 64: 130 -> 113, ff2 C = 512: 49.6 -> 43.1), 64 x 128 where there are at most 64 output columns;
//         32 .. 199 tiles -> 64 x 64 (32 KB: five workgroups per CU; the launches the 64 x 64 tiles of conv_kernel used to take: ff2 C = 512
//         21.5 -> 15.8 us, out 9.0 -> 7.4, time up-sampling 13.1 -> 10.7 / 10.7 -> 8.4, and the T = 5 ff2 that conv_kernel split over K:
//         15.6 -> 10.3); below 32 tiles: conv_kernel (split-K where K is long).
// Whole step at B = 1 on one box, two interleaved runs each: 2.689 ms with the round-3 rule (>= 256 tiles -> 128 x 128), 2.635 with
// ">= 256 -> 128 x 64, >= 96 -> 64 x 64"; B = 4: 7.31 -> 7.23 ms.  Results do not depend on the tile (every output element accumulates its
// K steps in the same order): bit-identical to conv_kernel's except against its split-K form.
// A pure function of the descriptor; oracle/ndhwc_ops.py::CpuOps._pw_dma restates it.
inline bool pw_dma_rule(const mv2_conv_desc& d, int64_t tiles128, TileCfg* t) {
    if (d.act == MV2_ACT_GEGLU) {
        if (tiles128 < 256) return false;
        *t = {128, 128};
        return true;
    }
    if (tiles128 >= 200) { *t = d.co <= 64 ? TileCfg{64, 128} : TileCfg{128, 64}; return true; }
    if (tiles128 >= 32) { *t = {64, 64}; return true; }
    return false;
}

inline bool pointwise_rows(const mv2_conv_desc& d) {
    return d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0 &&
           d.To == d.Ti && d.Ho == d.Hi && d.Wo == d.Wi;
}

// conv_pw_dma_kernel<..., GATHER> instead of conv_kernel: bf16 convolutions WITH taps / strides / zero padding whose K steps are whole
// 128-byte pieces of one input row (Ci % 64 == 0), stored through the staged epilogue, never split over K.  On the path: the space and
// time down-sampling convolutions (reference magvit2_pytorch.py:770-807).  Tiles as for the pointwise launches, 128 x 128 from 1024 tiles
// on; below 32 tiles of 128 x 128 conv_kernel keeps the launch (split-K where K is long).  mv2_conv only: mv2_conv_rms stays on
// conv_kernel / the pointwise form.  Measured on MI355X (tools/k3_bench gather, profiles/r06_gather_dma.log), conv_kernel -> this kernel,
// us per launch at B = 1: space 64 -> 128 34.6 -> 27.2, 128 -> 256 28.0 -> 23.2, 256 -> 512 30.0 -> 26.9, time 512 -> 512 at T = 20
// 16.5 -> 13.2, at T = 10 (conv_kernel: split-K + finish launch) 15.9 -> 11.2; at B = 4: 99.5 -> 87.3, 78.5 -> 74.6, 82.1 -> 73.8,
// 34.2 -> 28.1, 22.3 -> 19.8.  Outputs bit-identical to conv_kernel's un-split form.
inline bool gather_dma_pick(const mv2_conv_desc& d, TileCfg* tile) {
    if (d.dtype != MV2_BF16 || d.splitk > 1 || pointwise_rows(d) || d.act == MV2_ACT_GEGLU) return false;
    if (d.seg_bytes != 128 || d.Ci % 64 != 0) return false;
    if (d.store == MV2_STORE_NCTHW_F32 || (d.out_c * 2) % 16 != 0 || d.co % 8 != 0) return false;                   // stage_ok
    const int64_t M = (int64_t)d.B * d.To * d.Ho * d.Wo;
    const int64_t tiles128 = ((d.co + 127) / 128) * ((M + 127) / 128);
    TileCfg t{0, 0};
#ifdef MV2_EXPERIMENTS
    if (getenv("MV2_TILE")) return false;
    if (const char* e = getenv("MV2_GATHER_DMA")) {        // 0 = off, N = tile threshold; tile from MV2_GATHER_DMA_TILE ("bco,bpos")
        const int thr = atoi(e);
        if (thr <= 0 || tiles128 < thr) return false;
        if (const char* tl = getenv("MV2_GATHER_DMA_TILE")) {
            if (sscanf(tl, "%d,%d", &t.bco, &t.bpos) == 2) { if (tile) *tile = t; return true; }
        }
    }
#endif
    if (tiles128 >= 1024 && d.co > 64) t = {128, 128};
    else if (tiles128 >= 200) t = d.co <= 64 ? TileCfg{64, 128} : TileCfg{128, 64};
    else if (tiles128 >= 32) t = {64, 64};
    else return false;
    if (tile) *tile = t;
    return true;
}

inline bool pw_dma_pick(const mv2_conv_desc& d, TileCfg* tile) {
#ifdef MV2_EXPERIMENTS
    if (getenv("MV2_TILE")) return false;                                        // tools/k3_bench tile sweep of conv_kernel
    struct EnvRule { int min_blocks, bco, bpos; } env_rules[8];                  // first entry (descending thresholds) the launch's tile count reaches
    int nrules = -1;
    if (const char* e = getenv("MV2_PW_DMA_RULE")) {       // "1024:128x128,256:128x64,96:64x64"; "off" = never
        nrules = 0;
        while (*e && nrules < 8) {
            int mb = 0, a = 0, b2 = 0, n = 0;
            if (sscanf(e, "%d:%dx%d%n", &mb, &a, &b2, &n) != 3) break;
            env_rules[nrules++] = {mb, a, b2};
            e += n; if (*e == ',') ++e;
        }
    }
    if (const char* e = getenv("MV2_PW_DMA")) {            // 0 = off, N = tile threshold of a single rule; tile from MV2_PW_DMA_TILE ("bco,bpos")
        env_rules[0] = {atoi(e), 128, 128};
        if (env_rules[0].min_blocks <= 0) return false;
        if (const char* t = getenv("MV2_PW_DMA_TILE")) sscanf(t, "%d,%d", &env_rules[0].bco, &env_rules[0].bpos);
        nrules = 1;
    }
#endif
    if (d.dtype != MV2_BF16 || d.splitk > 1) return false;
    if (!pointwise_rows(d)) return false;
    if (d.seg_bytes != 128 || d.Ci % 64 != 0 || d.nsteps != d.Ci / 64) return false;
    if (d.store == MV2_STORE_NCTHW_F32 || (d.out_c * 2) % 16 != 0 || d.co % 8 != 0) return false;                   // stage_ok
    const int64_t M = (int64_t)d.B * d.To * d.Ho * d.Wo;
    const int64_t tiles128 = ((d.co + 127) / 128) * ((M + 127) / 128);
    TileCfg t{0, 0};
#ifdef MV2_EXPERIMENTS
    if (nrules >= 0) {
        for (int i = 0; i < nrules; ++i)
            if (tiles128 >= env_rules[i].min_blocks) {
                t = {env_rules[i].bco, env_rules[i].bpos};
                if (d.act == MV2_ACT_GEGLU && t.bco == 64 && t.bpos == 64) t.bpos = 128;      // value and gate live in one wave: 64 packed columns per wave
                if (tile) *tile = t;
                return true;
            }
        return false;
    }
#endif
    if (!pw_dma_rule(d, tiles128, &t)) return false;
    if (tile) *tile = t;
    return true;
}
inline bool pw_dma_ok(const mv2_conv_desc& d) { return pw_dma_pick(d, nullptr); }

struct Launch {
    TileCfg t;
    bool dma;             // conv_pw_dma_kernel
    bool gather;          // ... its GATHER form (taps / strides / padding)
    int n_co_tiles, n_pos_tiles, splitk, steps_per_split;
    int64_t M, ws_rows;
    int ws_cols;
    // unused by the generic kernel (conv3.hip plans its own tiles); kept zero
    bool halo;
    int th, tw, tiles_h, tiles_w, nchunk, chunks_per_split;
};

inline int device_cus() {      // of the CURRENT device; not cached: a process may drive several devices
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

inline int validate(const mv2_conv_desc& d) {
    if (d.dtype != MV2_BF16 && d.dtype != MV2_F32) return MV2_EINVAL;
    if (d.seg_bytes != 16 && d.seg_bytes != 32 && d.seg_bytes != 64 && d.seg_bytes != 128) return MV2_EINVAL;
    if (d.co_pad % 128 != 0 || d.co <= 0 || d.co > d.co_pad || d.nsteps <= 0) return MV2_EINVAL;
    if (d.kt > 8 || d.kh > 8 || d.kw > 8 || d.kt < 1 || d.kh < 1 || d.kw < 1) return MV2_EINVAL;
    if (d.store != MV2_STORE_NCTHW_F32 && d.act != MV2_ACT_GEGLU && (d.co % 4) != 0) return MV2_EINVAL;
    if (d.act == MV2_ACT_GEGLU && ((d.co % 64) != 0 || d.store != MV2_STORE_ROWS || d.out_c != d.co / 2)) return MV2_EINVAL;
    if (d.store == MV2_STORE_SHUFFLE && (d.out_c % 4 != 0 || d.out_c * d.rt * d.rh * d.rw != d.co)) return MV2_EINVAL;
    if (d.store == MV2_STORE_ROWS && d.act != MV2_ACT_GEGLU && d.out_c < d.co) return MV2_EINVAL;
    if ((d.out_c % 4) != 0 && d.store != MV2_STORE_NCTHW_F32) return MV2_EINVAL;
    return MV2_OK;
}

inline Launch plan_launch(const mv2_conv_desc& d, bool allow_gather = true) {
    Launch L;
    L.halo = false; L.th = L.tw = L.tiles_h = L.tiles_w = L.nchunk = L.chunks_per_split = 0;
    L.M = (int64_t)d.B * d.To * d.Ho * d.Wo;
    L.dma = pw_dma_pick(d, &L.t);
    L.gather = !L.dma && allow_gather && gather_dma_pick(d, &L.t);
    L.dma = L.dma || L.gather;
    if (!L.dma) L.t = pick_tile(d);
    L.n_co_tiles = (d.co + L.t.bco - 1) / L.t.bco;
    L.n_pos_tiles = (int)((L.M + L.t.bpos - 1) / L.t.bpos);
    int sk = L.dma ? 1 : d.splitk;
    if (sk <= 0) {
        sk = 1;
        const int64_t blocks = (int64_t)L.n_co_tiles * L.n_pos_tiles;
        const int cus = device_cus();
        if (blocks < cus && d.nsteps >= 16) {
            sk = (int)((2 * cus + blocks - 1) / blocks);
            if (sk > d.nsteps / 8) sk = d.nsteps / 8;
            if (sk > 16) sk = 16;
            if (sk < 1) sk = 1;
        }
    }
    if (sk > d.nsteps) sk = d.nsteps;
    L.steps_per_split = (d.nsteps + sk - 1) / sk;
    L.splitk = (d.nsteps + L.steps_per_split - 1) / L.steps_per_split;
    L.ws_rows = (int64_t)L.n_pos_tiles * L.t.bpos;
    L.ws_cols = L.n_co_tiles * L.t.bco;
    return L;
}

template <typename E, int BCO, int BPOS, int WCO, int WPOS, int NSET>
int launch_cfg(const ConvParams& p, const Launch& L, hipStream_t st) {
    constexpr size_t lds_main = 2 * (size_t)(BCO + BPOS) * 128;
    constexpr size_t lds_epi = (size_t)StageCfg<E, BCO, BPOS>::kBytes;
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    if (lds > 64 * 1024) {   // per-device attribute: set on every call rather than cached in a process-wide flag
        hipError_t e = hipFuncSetAttribute((const void*)conv_kernel<E, BCO, BPOS, WCO, WPOS, NSET>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)(L.n_co_tiles * L.n_pos_tiles), (unsigned)L.splitk);
    hipLaunchKernelGGL((conv_kernel<E, BCO, BPOS, WCO, WPOS, NSET>), grid, dim3(256), lds, st, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

template <int BCO, int BPOS, int WCO, int WPOS, bool RMS, bool GATHER>
int launch_pw_dma(const ConvParams& p, const Launch& L, hipStream_t st) {
    constexpr int NS = 2;
    constexpr size_t lds_main = (size_t)NS * (BCO + BPOS) * 128, lds_epi = (size_t)StageCfg<BF16, BCO, BPOS>::kBytes;
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    hipError_t e = hipFuncSetAttribute((const void*)conv_pw_dma_kernel<BCO, BPOS, WCO, WPOS, NS, RMS, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv_pw_dma_kernel<BCO, BPOS, WCO, WPOS, NS, RMS, GATHER>), dim3((unsigned)(L.n_co_tiles * L.n_pos_tiles)), dim3(256), lds, st, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

template <bool RMS, bool GATHER = false>
int launch_pw_dma_tile(const ConvParams& p, const Launch& L, hipStream_t st) {
    // two stages: 128 x 128 = 64 KB of LDS = two workgroups per CU (measured better than three stages, 96 KB, one workgroup per CU, on every
    // launch with more than one tile per CU: profiles/r03_pw_dma.log: ff1 22.7 vs 29.9 us, ff2 at B = 4 47.3 vs 50.8); 128 x 64 and
    // 64 x 128 = 48 KB = three, 64 x 64 = 32 KB = five
    if (L.t.bco == 128 && L.t.bpos == 128) return launch_pw_dma<128, 128, 2, 2, RMS, GATHER>(p, L, st);
    if (L.t.bco == 128 && L.t.bpos == 64) return launch_pw_dma<128, 64, 2, 1, RMS, GATHER>(p, L, st);
    if (L.t.bco == 64 && L.t.bpos == 128) return launch_pw_dma<64, 128, 2, 1, RMS, GATHER>(p, L, st);
    if (L.t.bco == 64 && L.t.bpos == 64 && p.d.act != MV2_ACT_GEGLU) return launch_pw_dma<64, 64, 1, 1, RMS, GATHER>(p, L, st);
    return MV2_EINVAL;
}

template <typename E>
int launch_dtype(const ConvParams& p, const Launch& L, hipStream_t st) {
    int rc;
    if (L.dma) {
        if constexpr (std::is_same<E, BF16>::value) {
            if (L.gather) return p.rms_fused ? MV2_EINVAL : launch_pw_dma_tile<false, true>(p, L, st);
            return p.rms_fused ? launch_pw_dma_tile<true>(p, L, st) : launch_pw_dma_tile<false>(p, L, st);
        } else {
            return MV2_EINVAL;
        }
    }
    if (L.t.bco == 64 && L.t.bpos == 64) rc = launch_cfg<E, 64, 64, 1, 1, 2>(p, L, st);
    else if (L.t.bco == 64 && L.t.bpos == 128) rc = launch_cfg<E, 64, 128, 2, 1, 2>(p, L, st);
    else if (L.t.bco == 128) rc = launch_cfg<E, 128, 128, 2, 2, 2>(p, L, st);
    else if (L.t.bco == 64) rc = launch_cfg<E, 64, 256, 2, 2, 2>(p, L, st);
    else rc = launch_cfg<E, 32, 256, 1, 2, 2>(p, L, st);
    if (rc != MV2_OK) return rc;
    if (L.splitk > 1) {
        const int groups = p.d.act == MV2_ACT_GEGLU ? p.d.co / 8 : (p.d.co + 3) / 4;
        int64_t total = p.M * groups;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL((conv_finish_kernel<E>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        MV2_LAUNCH_CHECK();
    }
    return MV2_OK;
}

}  // namespace

extern "C" size_t mv2_conv_workspace_bytes(const mv2_conv_desc* d) {
    if (!d || validate(*d) != MV2_OK) return 0;
    const Launch L = plan_launch(*d);
    if (L.splitk <= 1) return 0;
    return (size_t)L.splitk * (size_t)L.ws_rows * (size_t)L.ws_cols * sizeof(float);
}

static int conv_impl(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp, const mv2_seg* segs,
                     const float* bias, const float* rowscale, const void* residual, void* out,
                     void* workspace, size_t workspace_bytes, void* stream, int rms_fused);

extern "C" int mv2_conv_pointwise_dma(const mv2_conv_desc* d) { return d && validate(*d) == MV2_OK && pw_dma_ok(*d) ? 1 : 0; }
extern "C" int mv2_conv_gather_dma(const mv2_conv_desc* d) { return d && validate(*d) == MV2_OK && !pw_dma_ok(*d) && gather_dma_pick(*d, nullptr) ? 1 : 0; }

extern "C" int mv2_conv(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp, const mv2_seg* segs,
                        const float* bias, const float* rowscale, const void* residual, void* out,
                        void* workspace, size_t workspace_bytes, void* stream) {
    return conv_impl(d, in, in_bytes, wp, segs, bias, rowscale, residual, out, workspace, workspace_bytes, stream, 0);
}

// 1 when mv2_conv_rms can run this descriptor: the launch keeps the whole reduction extent in one workgroup (no split-K) and stores
// through the staged epilogue (rows / depth-to-space stores with 16-byte chunks)
extern "C" int mv2_conv_rms_ok(const mv2_conv_desc* d) {
    if (!d || validate(*d) != MV2_OK) return 0;
    if (d->store == MV2_STORE_NCTHW_F32) return 0;
    const int es = d->dtype == MV2_BF16 ? 2 : 4;
    if (((d->out_c * es) % 16) != 0 || (d->co % (16 / es)) != 0) return 0;
    return plan_launch(*d, false).splitk == 1 ? 1 : 0;
}

extern "C" int mv2_conv_rms(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp, const mv2_seg* segs,
                            const float* bias, const void* residual, void* out, void* stream) {
    if (!mv2_conv_rms_ok(d)) return MV2_EINVAL;
    return conv_impl(d, in, in_bytes, wp, segs, bias, nullptr, residual, out, nullptr, 0, stream, 1);
}

static int conv_impl(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp, const mv2_seg* segs,
                     const float* bias, const float* rowscale, const void* residual, void* out,
                     void* workspace, size_t workspace_bytes, void* stream, int rms_fused) {
    if (!d || !in || !wp || !segs || !out) return MV2_EINVAL;
    int rc = validate(*d);
    if (rc != MV2_OK) return rc;
    if (in_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    if ((int64_t)d->B * d->To * d->Ho * d->Wo >= (int64_t)0x7fffffff) return MV2_ETOOLARGE;   // device row math is 32-bit
    const size_t w_bytes = (size_t)d->nsteps * d->co_pad * 128;
    if (w_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    if (residual && (d->store != MV2_STORE_ROWS || d->act == MV2_ACT_GEGLU)) return MV2_EINVAL;
    const Launch L = plan_launch(*d, rms_fused == 0);
    if (d->act == MV2_ACT_GEGLU && L.t.bco == 32) return MV2_EINVAL;
    ConvParams p;
    p.d = *d; p.in = in; p.wp = wp; p.segs = segs; p.bias = bias; p.rowscale = rowscale; p.residual = residual;
    p.out = out; p.ws = (float*)workspace;
    p.wp2 = nullptr; p.bias2 = nullptr; p.act2 = MV2_ACT_NONE;
    p.se_wk = nullptr; p.se_bk = 0.f; p.se_part = nullptr;
    p.counters = nullptr; p.gtot = 0;
    p.rms_fused = rms_fused;
    p.in_bytes = (uint32_t)in_bytes; p.w_bytes = (uint32_t)w_bytes;
    p.M = L.M; p.lg_seg_chunks = ilog2(d->seg_bytes / 16);
    p.n_co_tiles = L.n_co_tiles; p.n_pos_tiles = L.n_pos_tiles;
    p.steps_per_split = L.steps_per_split; p.splitk = L.splitk;
    p.ws_cols = L.ws_cols; p.ws_rows = L.ws_rows;
    p.tiles_h = L.tiles_h; p.tiles_w = L.tiles_w; p.nchunk = L.nchunk; p.chunks_per_split = L.chunks_per_split;
    p.aligned = 0;
    if (L.splitk > 1) {
        const size_t need = (size_t)L.splitk * (size_t)L.ws_rows * (size_t)L.ws_cols * sizeof(float);
        if (!workspace || workspace_bytes < need) return MV2_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    return d->dtype == MV2_BF16 ? launch_dtype<BF16>(p, L, st) : launch_dtype<F32>(p, L, st);
}

extern "C" int mv2_abi_version(void) { return MV2_ABI_VERSION; }
extern "C" int mv2_device_cus(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MV2_ENODEVICE;
    return device_cus();
}
