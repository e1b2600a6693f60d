// conv_halo_rs_kernel — role-specialised variant of the halo-tiled 3x3 causal convolution (conv_halo.hip).
//
// STATUS: written after the round-1 GPU budget was spent; compiled and schedule-checked (ISA), NOT yet run on a
// GPU.  Selected only by MV2_HALO_RS=1; the default path never reaches this file.  First thing to do with it:
// the lockstep tests with the switch on, then tools/halo_ablation.py.
//
// Why: the component ablation of the default kernel (profiles/r01_halo_ablation.*) shows that its
// per-tap components ADD UP — MFMAs 21 %, LDS fragment reads 18 %, LDS weight/halo writes 21 %, barrier 9 %,
// weight loads 8 % — because every wave does every job in lock-step and the partner block's waves do the same
// jobs at the same time.  Here one 8-wave block per CU splits the jobs:
//   waves 0-3  CONSUMERS: fragment reads + MFMAs only (wave tile 64co x 64pos, 128co x 8x16 block tile),
//              cross-tap pipelined: the MFMAs of k-slices 2,3 of tap s-1 run under the fragment reads of tap s;
//   waves 4-7  PRODUCERS: weight-tile loads three taps ahead (registers are free here), halo loads a group
//              ahead, and all LDS writes (weights of tap s+1 into the idle buffer, halo image of group g+1).
// One barrier per tap keeps the double-buffer protocol of the default kernel, so its hazard analysis carries over:
// buffer PAR^1 / image GP^1 were last read before the previous barrier.  A SIMD hosts one consumer and one
// producer wave: the producer's LDS writes and address arithmetic issue in the shadow of the consumer's MFMAs.
// bf16, 128 output channels per block, stride-1 "same"/causal 3x3 only; epilogues shared with conv.hip
// (split-K slab or LDS-staged rows).  Reference: ResidualUnit CausalConv3d, magvit2_pytorch.py:937.
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"

namespace {

constexpr int RS_BCO = 128, RS_TH = 8, RS_TW = 16, RS_BPOS = RS_TH * RS_TW;
constexpr int RS_WCO = 2, RS_WPOS = 2;                 // consumer wave tile: 2 x 2 MFMA tiles of 32 x 32
constexpr int RS_HWD = RS_TW + 2, RS_HR = (RS_TH + 2) * RS_HWD;
constexpr int RS_RPP = 32;                             // rows staged per producer pass (256 producer threads / 8)
constexpr int RS_NH = (RS_HR + RS_RPP - 1) / RS_RPP;   // halo rows per producer thread
constexpr int RS_RW = RS_BCO / RS_RPP;                 // weight rows per producer thread
constexpr int RS_HALO_BYTES = ((RS_HR * 128 + 255) / 256) * 256;
constexpr int RS_WBUF = RS_BCO * 128;
constexpr int RS_HWT = 6;                              // tap at which the prefetched halo image is written

}  // namespace

__global__ __launch_bounds__(512, 1) void conv_halo_rs_kernel(const ConvParams p) {
    typedef BF16 E;
    constexpr int TW = RS_TW, HWD = RS_HWD, HR = RS_HR, RPP = RS_RPP, NH = RS_NH, RW = RS_RW;
    constexpr int WCO = RS_WCO, WPOS = RS_WPOS, BCO = RS_BCO, BPOS = RS_BPOS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo0 = smem;
    char* const wlds = smem + 2 * RS_HALO_BYTES;

    const mv2_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
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
    const int h0 = thi * RS_TH, w0 = twi * RS_TW;
    const int n0 = co_tile * BCO;
    const int z = blockIdx.y;
    const int c_begin = z * p.chunks_per_split;
    const int c_end = min(p.nchunk, c_begin + p.chunks_per_split);
    const int kt = d.kt;
    const int G = (c_end - c_begin) * kt;           // halo groups: (chunk, it)
    const int S = G * 9;                            // taps

    // ---- producer state (threads 256..511 -> loader index 0..255: 8 lanes per 128-byte row, 32 rows per pass)
    const int ltid = tid & 255;
    const int c = ltid & 7, r0 = ltid >> 3;
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
    const uint32_t row_bytes = (uint32_t)d.Ci * 2u;
    const uint32_t hw_base = (uint32_t)(((int64_t)(h0 - 1) * W + (w0 - 1)) * (int64_t)row_bytes);
    const uint32_t frame_bytes = (uint32_t)((int64_t)H * W * d.Ci * 2);
    const uint32_t w_off0 = (uint32_t)(n0 + r0) * 128u + (uint32_t)c * 16u;
    const uint32_t w_step = (uint32_t)d.co_pad * 128u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.w_bytes, 0x00020000);
    const uint32_t lds_wr = (uint32_t)r0 * 128u + (uint32_t)((c ^ ((r0 >> 1) & 7)) << 4);

    // ---- consumer state (waves 0..3: 2 x 2 over (co, pos))
    const int cw = wave & 3;
    const int wco = cw >> 1, wpos = cw & 1;
    const uint32_t swz = (uint32_t)((l31 >> 1) & 7);
    const uint32_t rd_w = (uint32_t)(wco * 32 * WCO + l31) * 128u;
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
    // fragment sets: 0,1 = k-slices 0,1 of the current tap; 2,3 = k-slices 2,3 (consumed at the top of the next tap)
    i32x4 xw[4][WCO], xp[4][WPOS];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int a = 0; a < WCO; ++a) xw[q][a] = i32x4{0, 0, 0, 0};
#pragma unroll
        for (int bb = 0; bb < WPOS; ++bb) xp[q][bb] = i32x4{0, 0, 0, 0};
    }

    i32x4 hreg[NH];
    i32x4 wreg[3][RW];                              // weight tiles in flight: taps s+1 .. s+3 (set = tap % 3 = TP % 3)

    // every producer load is unconditional in instruction count (past-the-end / padding -> out-of-range offset -> zeros)
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
    auto issue_w = [&](int s, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int g = s / 9, tp = s - g * 9;
        const int chunk = c_begin + g / kt, it = g % kt;
        const uint32_t wo_ = (s < S) ? w_off0 + (uint32_t)((it * 9 + tp) * p.nchunk + chunk) * w_step : p.w_bytes;
#pragma unroll
        for (int i = 0; i < RW; ++i) wreg[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wo_ + (uint32_t)(i * RPP * 128)), 0, 0);
    };
    auto write_w = [&](int buf, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < RW; ++i) *(i32x4*)(wlds + buf * RS_WBUF + lds_wr + i * RPP * 128) = wreg[SET][i];
    };

    // The two roles run SEPARATE loops (whole waves take one side; both sides execute exactly 1 + S barriers).  A
    // common loop with an if / else per tap made the compiler merge the roles' registers through phi copies at
    // every join and drain vmcnt(0) on the producer side each tap, which would have thrown the prefetch away.
    if (S > 0 && producer) {
        // prologue: halo image of group 0 and the weight tile of tap 0 into LDS, taps 1..3 in flight
        issue_halo(0);
        issue_w(0, IC<0>{});
        issue_w(1, IC<1>{});
        issue_w(2, IC<2>{});
        write_halo(halo0);
        write_w(0, IC<0>{});
        issue_w(3, IC<0>{});                    // set 0 is free again: tap 3 -> set 3 % 3
        __syncthreads();
        // one tap of group g: TP = tap inside the group, PAR = (g*9 + TP) & 1, GP = g & 1 — all compile-time
        auto tap = [&](int g, auto tp_c, auto par_c, auto gp_c) {
            constexpr int TP = decltype(tp_c)::value;
            constexpr int PAR = decltype(par_c)::value;
            constexpr int GP = decltype(gp_c)::value;
            const int s = g * 9 + TP;
            // in flight on entry: W(s+1) in set (TP+1)%3, W(s+2) in (TP+2)%3, W(s+3) in TP%3.  Write W(s+1) into the
            // buffer the consumers left before the last barrier, then reuse its set for W(s+4).
            if (TP == 0) issue_halo(g + 1);
            write_w(PAR ^ 1, IC<(TP + 1) % 3>{});
            issue_w(s + 4, IC<(TP + 1) % 3>{});
            if (TP == RS_HWT) write_halo(halo0 + (GP ^ 1) * RS_HALO_BYTES);     // image of group g-1: idle for 6+ barriers
            __syncthreads();
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
    } else if (S > 0) {
        __syncthreads();                        // pairs with the producers' prologue barrier
        auto tap = [&](int g, auto tp_c, auto par_c, auto gp_c) {
            constexpr int TP = decltype(tp_c)::value;
            constexpr int PAR = decltype(par_c)::value;
            constexpr int GP = decltype(gp_c)::value;
            const char* wb = wlds + PAR * RS_WBUF;
            const char* himg = halo0 + GP * RS_HALO_BYTES;
            constexpr int toff = (TP / 3) * HWD + (TP % 3);
            uint32_t pbase[WPOS], psw[WPOS];
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb) {
                int pr_ = prow0[bb];
                asm volatile("" : "+v"(pr_));       // keep the 18 unrolled taps from hoisting all their LDS addresses
                const int pr = pr_ + toff;
                pbase[bb] = (uint32_t)pr * 128u;
                psw[bb] = (uint32_t)((pr >> 1) & 7);
            }
            auto read_frags = [&](int ks, auto set_c) {
                constexpr int SET = decltype(set_c)::value;
                const uint32_t ck = (uint32_t)(ks * 2 + hi);
#pragma unroll
                for (int a = 0; a < WCO; ++a) xw[SET][a] = *(const i32x4*)(wb + rd_w + a * 32 * 128 + ((ck ^ swz) << 4));
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) xp[SET][bb] = *(const i32x4*)(himg + pbase[bb] + ((ck ^ psw[bb]) << 4));
            };
            auto mma_frags = [&](auto set_c) {
                constexpr int SET = decltype(set_c)::value;
#pragma unroll
                for (int a = 0; a < WCO; ++a)
#pragma unroll
                    for (int bb = 0; bb < WPOS; ++bb) Mma<E>::run(xw[SET][a], xp[SET][bb], acc[a][bb]);
            };
            (void)g;
            read_frags(0, IC<0>{}); read_frags(1, IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            mma_frags(IC<2>{}); mma_frags(IC<3>{});        // k-slices 2,3 of the PREVIOUS tap (zeros before tap 0)
            __builtin_amdgcn_sched_barrier(0);
            read_frags(2, IC<2>{}); read_frags(3, IC<3>{});
            __builtin_amdgcn_sched_barrier(0);
            mma_frags(IC<0>{}); mma_frags(IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
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
        // drain: k-slices 2,3 of the last tap
#pragma unroll
        for (int q = 2; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < WCO; ++a)
#pragma unroll
                for (int bb = 0; bb < WPOS; ++bb) Mma<E>::run(xw[q][a], xp[q][bb], acc[a][bb]);
    }

    // ---- epilogue (shared with conv.hip): consumers own the accumulators, all 512 threads store rows
    if (p.splitk > 1) {
        if (!producer) {
            const int nbase = n0 + wco * 32 * WCO;
#pragma unroll
            for (int bb = 0; bb < WPOS; ++bb) {
                const int pp = wpos * 32 * WPOS + bb * 32 + l31;
                const int ph = pp / TW, pw = pp % TW;
                if ((h0 + ph) >= H || (w0 + pw) >= W) continue;
                const int64_t m = (((int64_t)b * T + t) * H + (h0 + ph)) * W + (w0 + pw);
#pragma unroll
                for (int a = 0; a < WCO; ++a) {
                    float* row = p.ws + ((int64_t)z * p.ws_rows + m) * p.ws_cols + nbase + a * 32 + 4 * hi;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f32x4 v = {acc[a][bb][4 * gq], acc[a][bb][4 * gq + 1], acc[a][bb][4 * gq + 2], acc[a][bb][4 * gq + 3]};
                        *(f32x4*)(row + 8 * gq) = v;
                    }
                }
            }
        }
        return;
    }
    const int64_t frame_row0 = ((int64_t)b * T + t) * H;
    staged_epilogue<E, BCO, BPOS, WCO, WPOS, 512>(p, smem, acc, n0, wco, wpos, tid, [&](int lp) {
        const int ph = lp / TW, pw = lp - ph * TW;
        const bool ok = (h0 + ph) < H && (w0 + pw) < W;
        return ok ? (frame_row0 + (h0 + ph)) * W + (w0 + pw) : p.M;
    }, !producer);
}

// host: eligibility + launch (called from mv2_launch_halo when MV2_HALO_RS is set)
bool mv2_halo_rs_eligible(const ConvParams& p, int bco, int th, int tw) {
    // bf16, the 128co x 8x16 tile, not the fused 1x1x1 stage, and an output the LDS-staged epilogue can store
    // (the kernel has no direct-store fallback): same rule as stage_ok() in conv_common.h
    const bool staged = p.d.store != MV2_STORE_NCTHW_F32 && ((p.d.out_c * 2) % 16) == 0 && (p.d.co % 8) == 0;
    return p.d.dtype == MV2_BF16 && bco == RS_BCO && th == RS_TH && tw == RS_TW && !p.wp2 && (staged || p.splitk > 1);
}

int mv2_launch_halo_rs(const ConvParams& p, hipStream_t st) {
    constexpr size_t lds_main = 2 * (size_t)RS_HALO_BYTES + 2 * (size_t)RS_WBUF;
    constexpr size_t lds_epi = (size_t)StageCfg<BF16, RS_BCO, RS_BPOS>::kBytes;
    constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    static_assert(lds <= 160 * 1024, "LDS budget: one 8-wave block per CU");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo_rs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    dim3 grid((unsigned)(p.n_co_tiles * p.n_pos_tiles), (unsigned)p.splitk);
    hipLaunchKernelGGL(conv_halo_rs_kernel, grid, dim3(512), lds, st, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
