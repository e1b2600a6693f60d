// Halo-tiled implicit-GEMM kernel for the stride-1 kt x 3 x 3 causal convolutions (the ResidualUnit
// conv: ~81 % of the path's FLOPs, reference magvit2_pytorch.py:937 / CausalConv3d :892-928).
//
// The generic kernel (conv.hip) gathers the activation tile once per tap: every input row is pulled
// through the vector-memory path 27 times.  Here a block owns a TH x TW rectangle of one output frame and
// stages the (TH+2) x (TW+2) input halo of one 128-byte channel chunk of one input frame into LDS ONCE;
// the nine spatial taps of that frame are then nine row offsets into the same LDS image.  Global -> LDS
// activation traffic drops from 27x to kt * (TH+2)(TW+2)/(TH*TW) ~ 4x; only the per-tap weight tile
// (BCO x 128 B, L2-resident) streams.
//
// Schedule (all compile-time: a group = the 9 taps of one (chunk, frame); tap parity selects the weight
// register set / LDS buffer):
//   * weights: W(s+2) is issued into registers at the top of tap s, W(s+1) is written to the other LDS
//     buffer after tap s's MFMAs -> two taps of flight;
//   * halo: the next group's halo is issued at tap 0 of the current group and written at tap HWT into the
//     OTHER halo image (DBH = double-buffered halo) -> ~6 taps of flight, no extra barrier; with a single
//     image (DBH = false, tiles too big for two) it is written behind an extra barrier after tap 8;
//   * every load is unconditional (past-the-end -> out-of-range offset, zeros, no traffic) so the
//     compiler's wait counts stay exact (see conv.hip);
//   * MFMA fragments are software-pipelined over two register sets.
// Zero padding (causal time front, "same" space) comes from out-of-range buffer offsets returning 0.
// Epilogue / split-K are shared with conv.hip.
#include <cstdlib>
#include "conv_common.h"


#include "conv_fused_pw.h"

// WD: weight prefetch depth in taps (register sets): the weight tile of tap s + WD is requested at the top of tap s
// FD: fragment register sets per tap (2: reads of k-slice ks+2 issue after the MFMAs of ks; 4: all 16 reads first)
// EW (with WD == 3): the LDS write of the next tap's weight tile is issued at the TOP of the tap (it has been in flight
// for two taps) instead of at the end, so the write latency runs under the MFMAs and the barrier only waits for skew
// ABL: TIMING-ONLY ablations of one per-tap component (results are wrong; selected by MV2_HALO_ABL, never by default):
//   1 no MFMAs (fragments still read)   2 no LDS fragment reads (MFMAs on constant registers)
//   3 no LDS writes of weights / halo   4 no per-tap barrier   5 no global weight loads
template <typename E, int BCO, int TH, int TW, int WCO, bool DBH, int NWV, bool GLW, int WD, int FD, bool EW, int ABL>
__global__ __launch_bounds__(64 * NWV, 2) void conv_halo_kernel(const ConvParams p) {
    constexpr int BPOS = TH * TW;
    constexpr int NT = 64 * NWV;                   // threads: 4 waves (2 blocks / CU) or 8 waves (1 block / CU)
    constexpr int RPP = NT / 8;                    // rows staged per pass (8 lanes per 128-byte row)
    constexpr int NWCO = BCO / (32 * WCO);
    constexpr int NWPOS = NWV / NWCO;
    constexpr int WPOS = BPOS / (32 * NWPOS);
    static_assert(NWCO * NWPOS == NWV && WPOS >= 1 && NWPOS * WPOS * 32 == BPOS, "wave tiling");
    constexpr int HWD = TW + 2;                    // halo row pitch (positions)
    constexpr int HR = (TH + 2) * HWD;             // halo rows
    constexpr int NH = (HR + RPP - 1) / RPP;       // halo rows staged per thread
    constexpr int RW = BCO / RPP;
    static_assert(RW >= 1, "weight rows per thread");
    constexpr int HALO_BYTES = ((HR * 128 + 255) / 256) * 256;
    constexpr int WBUF = BCO * 128;
    constexpr int HWT = 6;                         // tap at which the prefetched halo is written (DBH)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo0 = smem;
    char* const wlds = smem + (DBH ? 2 : 1) * HALO_BYTES;

    const mv2_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid & 7, r0 = tid >> 3;
    const int hi = lane >> 5, l31 = lane & 31;
    const int T = d.Ti, H = d.Hi, W = d.Wi;

    int bid = blockIdx.x;
    {
        const int nblk = p.n_co_tiles * p.n_pos_tiles;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int co_tile = bid % p.n_co_tiles;
    int pt_ = bid / p.n_co_tiles;
    const int twi = pt_ % p.tiles_w; pt_ /= p.tiles_w;
    const int thi = pt_ % p.tiles_h; pt_ /= p.tiles_h;
    const int t = pt_ % T; const int b = pt_ / T;
    const int h0 = thi * TH, w0 = twi * TW;
    const int n0 = co_tile * BCO;
    const int z = blockIdx.y;
    const int c_begin = z * p.chunks_per_split;
    const int c_end = min(p.nchunk, c_begin + p.chunks_per_split);
    const int kt = d.kt;
    const bool prio = (p.ablate & 1) != 0;          // runtime A/B switches (env MV2_ABLATE): 1 setprio, 2 stagger
    if ((p.ablate & 2) && (((blockIdx.x >> 3) / 32) & 1)) {   // de-phase the second block of each CU by ~half a tap
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const int G = (c_end - c_begin) * kt;           // halo groups: (chunk, it)
    const int S = G * 9;                            // taps

    // ---- halo gather state: validity bits are held, offsets are recomputed where used (VGPR budget)
    uint32_t hmask = 0;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int hr = r0 + RPP * i;
        if (hr < HR) {
            const int hy = hr / HWD, hx = hr - hy * HWD;
            const int hin = h0 - 1 + hy, win = w0 - 1 + hx;
            if ((unsigned)hin < (unsigned)H && (unsigned)win < (unsigned)W) hmask |= 1u << i;
        }
    }
    const uint32_t row_bytes = (uint32_t)d.Ci * E::kBytes;
    const uint32_t hw_base = (uint32_t)(((int64_t)(h0 - 1) * W + (w0 - 1)) * (int64_t)row_bytes);
    const uint32_t frame_bytes = (uint32_t)((int64_t)H * W * d.Ci * E::kBytes);
    const uint32_t w_off0 = (uint32_t)(n0 + r0) * 128u + (uint32_t)c * 16u;
    const uint32_t w_step = (uint32_t)d.co_pad * 128u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.w_bytes, 0x00020000);

    const uint32_t lds_wr = (uint32_t)r0 * 128u + (uint32_t)((c ^ ((r0 >> 1) & 7)) << 4);
    const int wco = wave / NWPOS, wpos = wave % NWPOS;
    const uint32_t swz = (uint32_t)((l31 >> 1) & 7);
    const uint32_t rd_w = (uint32_t)(wco * 32 * WCO + l31) * 128u;
    // this lane's positions inside the tile -> halo row of their (dh = -1, dw = -1) tap
    int prow0[WPOS];
#pragma unroll
    for (int bb = 0; bb < WPOS; ++bb) {
        const int pp = wpos * 32 * WPOS + bb * 32 + l31;
        prow0[bb] = (pp / TW) * HWD + (pp % TW);
    }

    f32x16 acc[WCO][WPOS];
#pragma unroll
    for (int a = 0; a < WCO; ++a)
#pragma unroll
        for (int bb = 0; bb < WPOS; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;

    i32x4 hreg[NH];
    static_assert(WD == 2 || WD == 3, "weight register sets");
    i32x4 wreg[WD][GLW ? 1 : RW];
    // GLW: the weight tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no
    // ds_write_b128 at ~79 B/clk).  The LDS image of a DMA is lane-linear (wave base + lane * 16), so the
    // chunk swizzle is applied to the SOURCE address: lane (row = tid >> 3, physical chunk = tid & 7) fetches
    // logical chunk (tid & 7) ^ ((row >> 1) & 7).  __syncthreads() drains the DMA (vmcnt(0)) every tap.
    const int wv_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto dma_w = [&](int s, int buf) {
        const int sc = s < S ? s : S - 1;              // past the end: re-fetch the last tile into the idle buffer
        const int g = sc / 9, tp = sc - g * 9;
        const int chunk = c_begin + g / kt, it = g % kt;
        const char* src = (const char*)p.wp + (size_t)((it * 9 + tp) * p.nchunk + chunk) * w_step + (size_t)(n0 + r0) * 128u
                          + (size_t)((c ^ ((r0 >> 1) & 7)) << 4);
#pragma unroll
        for (int i = 0; i < RW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * RPP * 128),
                                             (__attribute__((address_space(3))) void*)(wlds + buf * WBUF + (i * RPP + wv_u * 8) * 128), 16, 0, 0);
    };

    // unconditional in instruction count: a group past the end (or a frame before t = 0) reads out of range
    auto issue_halo = [&](int g) {
        const int chunk = c_begin + g / kt, it = g % kt;
        const int t_in = t + it - d.pt;
        const bool tok = g < G && t_in >= 0 && t_in < T;
        const uint32_t base = (uint32_t)(((int64_t)b * T + t_in) * (int64_t)frame_bytes) + (uint32_t)chunk * 128u + (uint32_t)c * 16u + hw_base;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int hr = r0 + RPP * i;
            const int hy = hr / HWD, hx = hr - hy * HWD;
            const bool ok = tok && ((hmask >> i) & 1u);
            const uint32_t vo = ok ? base + (uint32_t)(hy * W + hx) * row_bytes : p.in_bytes;
            hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)vo, 0, 0);
        }
    };
    auto write_halo = [&](char* img) {
#pragma unroll
        for (int i = 0; i < NH; ++i)
            if (r0 + RPP * i < HR) *(i32x4*)(img + lds_wr + i * RPP * 128) = hreg[i];
    };
    auto issue_w = [&](int s, auto set_c) {     // weights of tap s into register set SET
        constexpr int SET = decltype(set_c)::value;
        const int g = s / 9, tp = s - g * 9;
        const int chunk = c_begin + g / kt, it = g % kt;
        const uint32_t wo_ = (s < S) ? w_off0 + (uint32_t)((it * 9 + tp) * p.nchunk + chunk) * w_step : p.w_bytes;
        if constexpr (ABL == 5) {
#pragma unroll
            for (int i = 0; i < RW; ++i) wreg[SET][i] = i32x4{(int)wo_, 0, 0, 0};
            return;
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) wreg[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wo_ + (uint32_t)(i * RPP * 128)), 0, 0);
    };

    if (S > 0) {
        // prologue: halo of group 0, weights of taps 0 and 1
        issue_halo(0);
        if constexpr (GLW) {
            dma_w(0, 0);
        } else {
            issue_w(0, IC<0>{});
            issue_w(1, IC<1>{});
            if constexpr (WD == 3) issue_w(2, IC<2>{});
        }
        write_halo(halo0);
        if constexpr (!GLW) {
#pragma unroll
            for (int i = 0; i < RW; ++i) *(i32x4*)(wlds + lds_wr + i * RPP * 128) = wreg[0][i];
        }
        __syncthreads();

        // FD == 5: cross-tap software pipeline.  Sets 0,1 hold k-slices 0,1 of the current tap, sets 2,3 hold
        // k-slices 2,3; the MFMAs of slices 2,3 are issued at the top of the NEXT tap (after the barrier), under
        // that tap's first fragment reads, so every MFMA batch runs under a read batch.
        i32x4 xw[FD == 5 ? 4 : 1][WCO], xp[FD == 5 ? 4 : 1][WPOS];
        if constexpr (FD == 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int a = 0; a < WCO; ++a) xw[q][a] = i32x4{0, 0, 0, 0};
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) xp[q][bb] = i32x4{0, 0, 0, 0};
            }
        }
        // one tap of group g: TP = tap inside the group, PAR = (g*9 + TP) & 1, GP = g & 1 — all compile-time
        auto tap = [&](int g, auto tp_c, auto par_c, auto gp_c) {
            constexpr int TP = decltype(tp_c)::value;
            constexpr int PAR = decltype(par_c)::value;
            constexpr int GP = decltype(gp_c)::value;
            const int s = g * 9 + TP;
            if constexpr (GLW) dma_w(s + 1, PAR ^ 1);   // buffer PAR^1 was last read in tap s-1 (barrier passed)
            else if constexpr (WD == 2) issue_w(s + 2, IC<PAR>{});   // wreg[PAR] is free: W(s) went to LDS during tap s-1
            else issue_w(s + 3, IC<TP % 3>{});      // three sets: set = s % 3 = TP % 3 (9 taps per group), also free
            if (TP == 0) issue_halo(g + 1);
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetches at the top of the tap
            static_assert(!EW || (WD == 3 && !GLW && FD == 2), "early weight write: third register set, FD == 2 schedule");
            auto early_write = [&]() {
                // buffer PAR^1 was last read in tap s-1 (all reads retired before its barrier): free to overwrite.
                // Placed after the first MFMA batch so that a late W(s+1) stalls under running MFMAs, not before them.
                if constexpr (EW) {
#pragma unroll
                    for (int i = 0; i < RW; ++i) *(i32x4*)(wlds + (PAR ^ 1) * WBUF + lds_wr + i * RPP * 128) = wreg[(TP + 1) % 3][i];
                    if (DBH && TP == HWT) write_halo(halo0 + (GP ^ 1) * HALO_BYTES);
                }
            };
            const char* wb = wlds + PAR * WBUF;
            const char* himg = halo0 + (DBH ? GP * HALO_BYTES : 0);
            constexpr int toff = (TP / 3) * HWD + (TP % 3);
            uint32_t pbase[WPOS], psw[WPOS];
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb) {
                // opaque copy: without it loop-invariant code motion hoists the LDS addresses of all 18 unrolled
                // taps out of the group loop and the kernel spills at 256 VGPRs
                int pr_ = prow0[bb];
                asm volatile("" : "+v"(pr_));
                const int pr = pr_ + toff;
                pbase[bb] = (uint32_t)pr * 128u;
                psw[bb] = (uint32_t)((pr >> 1) & 7);
            }
            i32x4 fw_[FD == 5 ? 1 : FD][WCO], fp_[FD == 5 ? 1 : FD][WPOS];
            auto& fw = *[&]() { if constexpr (FD == 5) return &xw; else return &fw_; }();
            auto& fp = *[&]() { if constexpr (FD == 5) return &xp; else return &fp_; }();
            auto read_frags = [&](int ks, auto set_c) {
                constexpr int SET = decltype(set_c)::value;
                const uint32_t ck = (uint32_t)(ks * 2 + hi);
                if constexpr (ABL == 2) {       // constants instead of LDS reads
#pragma unroll
                    for (int a = 0; a < WCO; ++a) fw[SET][a] = i32x4{(int)ck, 0, 0, 0};
#pragma unroll
                    for (int bb = 0; bb < WPOS; ++bb) fp[SET][bb] = i32x4{(int)pbase[bb], 0, 0, 0};
                    return;
                }
#pragma unroll
                for (int a = 0; a < WCO; ++a) fw[SET][a] = *(const i32x4*)(wb + rd_w + a * 32 * 128 + ((ck ^ swz) << 4));
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) fp[SET][bb] = *(const i32x4*)(himg + pbase[bb] + ((ck ^ psw[bb]) << 4));
            };
            auto mma_frags = [&](auto set_c) {
                constexpr int SET = decltype(set_c)::value;
                if constexpr (ABL == 1) {       // keep the fragment reads alive, skip the matrix work
#pragma unroll
                    for (int a = 0; a < WCO; ++a) asm volatile("" :: "v"(fw[SET][a]));
#pragma unroll
                    for (int bb = 0; bb < WPOS; ++bb) asm volatile("" :: "v"(fp[SET][bb]));
                    return;
                }
#pragma unroll
                for (int a = 0; a < WCO; ++a)
#pragma unroll
                    for (int bb = 0; bb < WPOS; ++bb) Mma<E>::run(fw[SET][a], fp[SET][bb], acc[a][bb]);
            };
            if constexpr (FD == 5) {
                read_frags(0, IC<0>{}); read_frags(1, IC<1>{});
                __builtin_amdgcn_sched_barrier(0);
                mma_frags(IC<2>{}); mma_frags(IC<3>{});        // k-slices 2,3 of the PREVIOUS tap (zeros before tap 0)
                __builtin_amdgcn_sched_barrier(0);
                read_frags(2, IC<2>{}); read_frags(3, IC<3>{});
                __builtin_amdgcn_sched_barrier(0);
                mma_frags(IC<0>{}); mma_frags(IC<1>{});
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (FD == 4) {
                read_frags(0, IC<0>{}); read_frags(1, IC<1>{}); read_frags(2, IC<2>{}); read_frags(3, IC<3>{});
                __builtin_amdgcn_sched_barrier(0);
                mma_frags(IC<0>{}); mma_frags(IC<1>{}); mma_frags(IC<2>{}); mma_frags(IC<3>{});
            } else {
                read_frags(0, IC<0>{});
                read_frags(1, IC<1>{});
                __builtin_amdgcn_sched_barrier(0);
                // the two blocks of a CU run identical timelines; raising this wave's priority while it feeds the
                // matrix pipe lets the partner wave fall behind into its LDS / load phase (role split, guide T5)
                if (prio) __builtin_amdgcn_s_setprio(1);
                mma_frags(IC<0>{});
                read_frags(2, IC<0>{});
                if constexpr (EW) early_write();
                __builtin_amdgcn_sched_barrier(0);
                mma_frags(IC<1>{});
                read_frags(3, IC<1>{});
                __builtin_amdgcn_sched_barrier(0);
                mma_frags(IC<0>{});
                mma_frags(IC<1>{});
                if (prio) __builtin_amdgcn_s_setprio(0);
            }
            // W(s+1) -> the other weight buffer (after the last tap: the zero dummy into the idle buffer)
            if constexpr (!GLW && !EW) {
                if constexpr (ABL == 3) {
#pragma unroll
                    for (int i = 0; i < RW; ++i) asm volatile("" :: "v"(wreg[WD == 2 ? (PAR ^ 1) : ((TP + 1) % 3)][i]));
                } else {
#pragma unroll
                    for (int i = 0; i < RW; ++i) *(i32x4*)(wlds + (PAR ^ 1) * WBUF + lds_wr + i * RPP * 128) = wreg[WD == 2 ? (PAR ^ 1) : ((TP + 1) % 3)][i];
                }
            }
            if (DBH) {
                if (!EW && ABL != 3 && TP == HWT) write_halo(halo0 + (GP ^ 1) * HALO_BYTES);   // image of group g-1: idle since 6+ barriers
            } else if (TP == 8) {
                __syncthreads();                  // every wave is done with the single halo image
                write_halo(halo0);
            }
            if constexpr (ABL != 4) __syncthreads();
        };
        auto group = [&](int g, auto gp_c) {
            constexpr int GP = decltype(gp_c)::value;
            tap(g, IC<0>{}, IC<(GP + 0) & 1>{}, gp_c); tap(g, IC<1>{}, IC<(GP + 1) & 1>{}, gp_c);
            tap(g, IC<2>{}, IC<(GP + 2) & 1>{}, gp_c); tap(g, IC<3>{}, IC<(GP + 3) & 1>{}, gp_c);
            tap(g, IC<4>{}, IC<(GP + 4) & 1>{}, gp_c); tap(g, IC<5>{}, IC<(GP + 5) & 1>{}, gp_c);
            tap(g, IC<6>{}, IC<(GP + 6) & 1>{}, gp_c); tap(g, IC<7>{}, IC<(GP + 7) & 1>{}, gp_c);
            tap(g, IC<8>{}, IC<(GP + 8) & 1>{}, gp_c);
        };
        int g = 0;
        for (; g + 1 < G; g += 2) { group(g, IC<0>{}); group(g + 1, IC<1>{}); }
        if (g < G) group(g, IC<0>{});
        if constexpr (FD == 5) {       // drain: k-slices 2,3 of the last tap
#pragma unroll
            for (int q = 2; q < 4; ++q)
#pragma unroll
                for (int a = 0; a < WCO; ++a)
#pragma unroll
                    for (int bb = 0; bb < WPOS; ++bb) Mma<E>::run(xw[q][a], xp[q][bb], acc[a][bb]);
        }
    }

    // ---- epilogue (shared with conv.hip)
    const int nbase = n0 + wco * 32 * WCO;
    int64_t mrow[WPOS];
#pragma unroll
    for (int bb = 0; bb < WPOS; ++bb) {
        const int pp = wpos * 32 * WPOS + bb * 32 + l31;
        const int ph = pp / TW, pw = pp % TW;
        const bool ok = (h0 + ph) < H && (w0 + pw) < W;
        mrow[bb] = ok ? ((((int64_t)b * T + t) * H + (h0 + ph)) * W + (w0 + pw)) : p.M;
    }
    if constexpr (std::is_same<E, BF16>::value) {
        if (p.wp2) {       // mv2_conv_pw: fused pointwise second stage (host guarantees one co tile, no split-K)
            const int64_t frame_row0 = ((int64_t)b * T + t) * H;
            const int64_t tile_index = (((int64_t)b * T + t) * p.tiles_h + thi) * p.tiles_w + twi;
            fused_pointwise_epilogue<BCO, BPOS, WCO, WPOS, NT>(p, smem, acc, wco, wpos, tid, tile_index, [&](int lp) {
                const int ph = lp / TW, pw = lp - ph * TW;
                const bool ok = (h0 + ph) < H && (w0 + pw) < W;
                return ok ? (frame_row0 + (h0 + ph)) * W + (w0 + pw) : p.M;
            });
            return;
        }
    }
    if (p.splitk > 1) {
#pragma unroll
        for (int a = 0; a < WCO; ++a)
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb) {
                if (mrow[bb] >= p.M) continue;
                float* row = p.ws + ((int64_t)z * p.ws_rows + mrow[bb]) * p.ws_cols + nbase + a * 32 + 4 * hi;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 v = {acc[a][bb][4 * gq], acc[a][bb][4 * gq + 1], acc[a][bb][4 * gq + 2], acc[a][bb][4 * gq + 3]};
                    *(f32x4*)(row + 8 * gq) = v;
                }
            }
        return;
    }
    if (stage_ok(p)) {   // LDS-staged, fully coalesced stores (the tap loop ended on a barrier: LDS is free)
        const int64_t frame_row0 = ((int64_t)b * T + t) * H;
        staged_epilogue<E, BCO, BPOS, WCO, WPOS, NT>(p, smem, acc, n0, wco, wpos, tid, [&](int lp) {
            const int ph = lp / TW, pw = lp - ph * TW;
            const bool ok = (h0 + ph) < H && (w0 + pw) < W;
            return ok ? (frame_row0 + (h0 + ph)) * W + (w0 + pw) : p.M;
        });
        return;
    }
#pragma unroll
    for (int a = 0; a < WCO; ++a)
#pragma unroll
        for (int bb = 0; bb < WPOS; ++bb) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[a][bb][4 * gq + j];
                epi_store4<E>(p, mrow[bb], nbase + a * 32 + 8 * gq + 4 * hi, v);
            }
        }
}

template <typename E, int BCO, int TH, int TW, int WCO, bool DBH, int NWV, bool GLW = false, int WD = 2, int FD = 2, bool EW = false, int ABL = 0>
static int launch_one(const ConvParams& p, hipStream_t st) {
    constexpr int HR = (TH + 2) * (TW + 2);
    constexpr size_t halo_bytes = (size_t)((HR * 128 + 255) / 256) * 256;
    constexpr size_t lds_main = (DBH ? 2 : 1) * halo_bytes + 2 * (size_t)BCO * 128;
    constexpr size_t lds_epi = (size_t)StageCfg<E, BCO, TH * TW>::kBytes;
    // fused pointwise stage (bf16, one co tile): stage-1 tile + C x C weights
    constexpr size_t lds_pw = std::is_same<E, BF16>::value && BCO <= 128
                                  ? ((size_t)StageCfg<BF16, BCO, TH * TW>::kBytes + 255) / 256 * 256 + (size_t)(BCO * 2 / 128) * BCO * 128 : 0;
    constexpr size_t lds_a = lds_main > lds_epi ? lds_main : lds_epi;
    constexpr size_t lds = lds_a > lds_pw ? lds_a : lds_pw;
    static_assert(lds <= (NWV == 8 ? 160 : 80) * 1024, "LDS budget: two 4-wave blocks or one 8-wave block per CU");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo_kernel<E, BCO, TH, TW, WCO, DBH, NWV, GLW, WD, FD, EW, ABL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    dim3 grid((unsigned)(p.n_co_tiles * p.n_pos_tiles), (unsigned)p.splitk);
    hipLaunchKernelGGL((conv_halo_kernel<E, BCO, TH, TW, WCO, DBH, NWV, GLW, WD, FD, EW, ABL>), grid, dim3(64 * NWV), lds, st, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}

// tile menu: (BCO, TH x TW):  (64, 16x16) one image | (128, 8x16) two images | (64, 8x16) two images, 32co x 64pos waves
int mv2_launch_halo(const ConvParams& p, int bco, int th, int tw, hipStream_t st) {
    const bool bf = p.d.dtype == MV2_BF16;
    // MV2_HALO_RS=1: role-specialised 8-wave kernel (conv_halo_rs.hip) where it applies; NOT yet validated on a GPU
    if (getenv("MV2_HALO_RS") != nullptr && mv2_halo_rs_eligible(p, bco, th, tw)) return mv2_launch_halo_rs(p, st);
    static const bool glw = getenv("MV2_HALO_GLDS") != nullptr;   // weight tile by LDS-DMA (A/B switch)
    if (bco == 128 && th == 8 && tw == 16 && glw)
        return bf ? launch_one<BF16, 128, 8, 16, 2, true, 4, true>(p, st) : launch_one<F32, 128, 8, 16, 2, true, 4, true>(p, st);
    if (bco == 64 && th == 16 && tw == 16 && glw)
        return bf ? launch_one<BF16, 64, 16, 16, 2, false, 4, true>(p, st) : launch_one<F32, 64, 16, 16, 2, false, 4, true>(p, st);
    // MV2_HALO_ABL=1..5: timing-only component ablations of the 128co x 8x16 bf16 kernel (results are WRONG; see the kernel header)
    const int abl = getenv("MV2_HALO_ABL") ? atoi(getenv("MV2_HALO_ABL")) : 0;      // re-read per launch: tools/halo_ablation.py switches it in-process
    if (bf && abl && bco == 128 && th == 8 && tw == 16) {
        switch (abl) {
            case 1: return launch_one<BF16, 128, 8, 16, 2, true, 4, false, 2, 2, false, 1>(p, st);
            case 2: return launch_one<BF16, 128, 8, 16, 2, true, 4, false, 2, 2, false, 2>(p, st);
            case 3: return launch_one<BF16, 128, 8, 16, 2, true, 4, false, 2, 2, false, 3>(p, st);
            case 4: return launch_one<BF16, 128, 8, 16, 2, true, 4, false, 2, 2, false, 4>(p, st);
            case 5: return launch_one<BF16, 128, 8, 16, 2, true, 4, false, 2, 2, false, 5>(p, st);
            default: break;
        }
    }
    // MV2_HALO_EW=1: early LDS write of the next weight tile (three register sets); NOT YET MEASURED OR VALIDATED ON A GPU
    // (written after the round-1 GPU budget was spent; next round: lockstep tests with the switch on, then A/B)
    const bool ew = getenv("MV2_HALO_EW") != nullptr;
    if (bf && ew && bco == 64 && th == 16 && tw == 16) return launch_one<BF16, 64, 16, 16, 2, false, 4, false, 3, 2, true>(p, st);
    if (bf && ew && bco == 128 && th == 8 && tw == 16) return launch_one<BF16, 128, 8, 16, 2, true, 4, false, 3, 2, true>(p, st);
    // MV2_HALO_FDEPTH=4: all fragment reads of a tap before its MFMAs; =5: cross-tap pipeline (A/B switches, bf16 only)
    static const int fdepth = getenv("MV2_HALO_FDEPTH") ? atoi(getenv("MV2_HALO_FDEPTH")) : 2;
    if (bf && fdepth == 5 && bco == 64 && th == 16 && tw == 16) return launch_one<BF16, 64, 16, 16, 2, false, 4, false, 2, 5>(p, st);
    if (bf && fdepth == 5 && bco == 128 && th == 8 && tw == 16) return launch_one<BF16, 128, 8, 16, 2, true, 4, false, 2, 5>(p, st);
    if (bf && fdepth == 4 && bco == 64 && th == 16 && tw == 16) return launch_one<BF16, 64, 16, 16, 2, false, 4, false, 2, 4>(p, st);
    if (bf && fdepth == 4 && bco == 128 && th == 8 && tw == 16) return launch_one<BF16, 128, 8, 16, 2, true, 4, false, 2, 4>(p, st);
    // MV2_HALO_WDEPTH=3: three weight register sets (prefetch three taps ahead; measured: no gain over two)
    static const int wdepth = getenv("MV2_HALO_WDEPTH") ? atoi(getenv("MV2_HALO_WDEPTH")) : 2;
    if (bco == 64 && th == 16 && tw == 16 && wdepth == 3)
        return bf ? launch_one<BF16, 64, 16, 16, 2, false, 4, false, 3>(p, st) : launch_one<F32, 64, 16, 16, 2, false, 4, false, 3>(p, st);
    if (bco == 128 && th == 8 && tw == 16 && wdepth == 3)
        return bf ? launch_one<BF16, 128, 8, 16, 2, true, 4, false, 3>(p, st) : launch_one<F32, 128, 8, 16, 2, true, 4, false, 3>(p, st);
    if (bco == 128 && th == 8 && tw == 16)
        return bf ? launch_one<BF16, 128, 8, 16, 2, true, 4>(p, st) : launch_one<F32, 128, 8, 16, 2, true, 4>(p, st);
    if (bco == 64 && th == 16 && tw == 16)
        return bf ? launch_one<BF16, 64, 16, 16, 2, false, 4>(p, st) : launch_one<F32, 64, 16, 16, 2, false, 4>(p, st);
    if (bco == 64 && th == 8 && tw == 16)
        return bf ? launch_one<BF16, 64, 8, 16, 1, true, 4>(p, st) : launch_one<F32, 64, 8, 16, 1, true, 4>(p, st);
    if (bco == 128 && th == 16 && tw == 16)     // 8 waves, one block per CU: weight tile staged once per 256 positions
        return bf ? launch_one<BF16, 128, 16, 16, 2, true, 8>(p, st) : launch_one<F32, 128, 16, 16, 2, true, 8>(p, st);
    return MV2_EINVAL;
}
