// FeedForward as ONE launch (bf16): RMSNorm -> 1x1x1 conv (C -> 2 x inner) -> GEGLU -> 1x1x1 conv (inner -> C) -> + x
// (reference magvit2_pytorch.py:466-508 FeedForward / GEGLU, the Residual wrapper M:174, TokenShift M:250-253 in front of it
// for the time-axis layers).  Replaces the pair mv2_conv_rms(GEGLU) + mv2_conv(residual): the hidden tensor (rows x inner,
// 56 MB per launch at C = 256) never exists in memory, and the erf epilogue of one hidden chunk runs while other waves of the
// CU are in their MFMA phases.
//
// One workgroup = 64 consecutive rows x a contiguous range of hidden blocks (a block = 32 hidden units = 64 packed ff1 columns:
// 32 values + 32 gates, pack.py::geglu_layout).  NW = C / 64 waves (4 at C = 256, 8 at C = 512):
//   * the x tile (64 rows x C, token-shifted where asked) sits in LDS for the whole kernel: B operand of ff1, source of the
//     RMSNorm scale 1 / |row|, and the residual of the epilogue;
//   * per chunk of NW hidden blocks: wave w runs ff1 for block w (64 columns x 64 rows, K = C, accumulators acc1[2][2]),
//     finishes GEGLU in registers (value * rs + b, gelu(gate * rs + b)), writes its 64 rows x 32 bf16 into the chunk's panel in LDS
//     (double buffered, one bare s_barrier per chunk); then every wave adds panel x W2[chunk] into ITS 64 output channels
//     (acc2[2][2], K = 32 NW): the accumulators of the output tile stay in registers across all chunks;
//   * both weight matrices stream global -> VGPR in the fragment-major image of mv2_conv3 / mv2_pw_se (one contiguous 1 KB block per
//     MFMA A fragment, no LDS traffic for weights) through ONE register ring of D steps x 2 fragments that runs ahead across
//     the ff1 -> ff2 -> next-chunk boundaries (every step of either phase takes exactly two fragments);
//   * few-row launches split the hidden range of a tile over S workgroups: partial output tiles go to fp32 slabs with
//     write-through stores, a ticket per tile, the last arriver sums the parts in part order (the protocol of conv3p.hip).
// Arithmetic is that of the two-launch form: same MFMA k order in both GEMMs, h rounded to bf16 once, out = bf16(acc + b2 + x)
// rounded once; only the RMSNorm sum of squares is taken in another order (<= 1 ulp of the scale).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "conv_common.h"
#include "../../include/mv2.h"

// experiments build: phase clocks (s_memtime ticks; ratios between phases are what they are good for, profiles/r04_findings.md section 3) of
// wave 0 of every workgroup: [0] prologue (x tile, RMS scale) [1] ff1 loops [2] GEGLU [3] barrier waits [4] ff2 loops [5] slab + ticket
// [6] epilogue [7] whole kernel.  MV2_FF_PROF=1 makes mv2_ff synchronise and print the means.
#ifdef MV2_EXPERIMENTS
__device__ unsigned long long* g_ff_prof = nullptr;
#define FF_PROF_DECL unsigned long long pf_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_ = p.prof ? __builtin_amdgcn_s_memtime() : 0ull; const unsigned long long pt0_ = pt_;
#define FF_PROF(k) if (p.prof) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_[k] += n_ - pt_; pt_ = n_; }
#define FF_PROF_END if (p.prof && g_ff_prof && threadIdx.x == 0 && blockIdx.x < 4096) { pf_[7] = __builtin_amdgcn_s_memtime() - pt0_; for (int i_ = 0; i_ < 8; ++i_) g_ff_prof[(size_t)blockIdx.x * 8 + i_] = pf_[i_]; }
#else
#define FF_PROF_DECL
#define FF_PROF(k)
#define FF_PROF_END
#endif

namespace {

struct FfParams {
    const void* x; const void* w1; const float* b1; const void* w2; const float* b2; void* out;
    float* ws; uint32_t* counters;
    int64_t rows;
    uint32_t w1_bytes, w2_bytes, ws_bytes;
    int nblk;                 // hidden blocks of 32 (hidden_pad / 32)
    int n1tiles;              // 32-row tiles of the ff1 image (co_pad / 32)
    int S;                    // workgroups per row tile (parts of the hidden range)
    int tiles;                // row tiles of 64
    int rot;                  // chunk order rotated by the row tile (see the kernel)
    int prof;                 // experiments build: phase clocks on (MV2_FF_PROF)
    int abl;                  // experiments build: ablation bits (1: no GEGLU math, 2: no weight loads in the loop)
    int shift_hw, T;          // TokenShift: channels >= C / 2 come from the row shift_hw earlier, zero in the first frame (0: none)
};

template <int CT, int NW> struct FfCfg {
    static constexpr int C = CT * 128, NT = NW * 64;
    static constexpr int CW = C / (32 * NW);                            // 32-channel output tiles per wave: 2 (NW = C / 64) or 1 (NW = C / 32)
    // ring steps per chunk (two A fragments each): ff1 k-steps, then ff2 steps - CW = 2: one hidden k-step x the wave's two channel tiles,
    // CW = 1: two hidden k-steps x its one tile (they accumulate into two registers sets that are added behind the loop)
    static constexpr int KS1 = C / 16, KS2 = NW * CW, T = KS1 + KS2;
    static constexpr int PX = C * 2 + 16;                               // x tile row pitch (bytes): 16-byte slots of consecutive rows 1 apart mod 16
    static constexpr int HCH = NW * 32, PP = HCH * 2 + 16;              // hidden units per chunk, panel row pitch
    static constexpr int kX = 0, kPanel = 64 * PX, kRs = kPanel + 2 * 64 * PP, kFlag = kRs + 256, kB1 = kFlag + 16;
    static constexpr int SLAB = NW * CW * 8 * 1024;                     // one part's output tile in fragment order: [wave][a][r][reg4][64 lanes][16 B]
    static_assert((CW == 1 || CW == 2) && CW * 32 * NW == C, "every wave owns 32 or 64 output channels");
};

template <int CT, int NW, int D, int ABL = 0>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void ff_kernel(const FfParams p) {
    typedef FfCfg<CT, NW> L;
    constexpr int C = L::C, NT = L::NT, KS1 = L::KS1, KS2 = L::KS2, T = L::T, PX = L::PX, PP = L::PP, CW = L::CW;
    static_assert(T % D == 0 && KS1 % D == 0, "ring slots are static inside the unrolled chunk body");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    FF_PROF_DECL

    // ---- block -> (row tile, part).  XCD-aware: the blocks of one XCD walk a contiguous eighth of the list, the parts of a tile are
    // neighbours (they read the same x tile)
    int bid = blockIdx.x;
    {
        const int nb = p.tiles * p.S, q = nb >> 3, r = nb & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int tile = bid / p.S, part = bid - tile * p.S;
    const int blk_b = (int)((int64_t)part * p.nblk / p.S), blk_e = (int)((int64_t)(part + 1) * p.nblk / p.S);
    const int nchunks = (blk_e - blk_b + NW - 1) / NW;
    const int64_t row0 = (int64_t)tile * 64;

    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, p.w2_bytes, 0x00020000);
    // fragment of ring step t of chunk c (which = 0 / 1).  ff1: value / gate tile of this wave's block, k-step t; ff2: the wave's two
    // 32-channel tiles, hidden k-step t - KS1 of the chunk.  Dead steps (block outside the part, chunk past the end, hidden past the
    // padded width) read out of range: no memory access, zeros.
    const int lane16 = lane * 16;
    auto frag = [&](int c, int t, int which) -> i32x4 {
        if constexpr ((ABL & 2) != 0) { i32x4 z = {0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00}; asm volatile("" : "+v"(z)); return z; }      // experiments: no weight loads
        // the range check of a raw buffer load sees the vector offset only: a dead step's vector offset is the buffer size
        if (t < KS1) {
            const int blk = blk_b + c * NW + wave;
            const bool live = blk < blk_e && c < nchunks;
            const uint32_t so = ((uint32_t)((t >> 2) * p.n1tiles + 2 * blk + which) * 4u + (uint32_t)(t & 3)) * 1024u;
            return __builtin_amdgcn_raw_buffer_load_b128(rs_w1, live ? lane16 : (int)p.w1_bytes, live ? (int)so : 0, 0);
        }
        const int k16 = (blk_b + c * NW) * 2 + (CW == 2 ? (t - KS1) : 2 * (t - KS1) + which);
        const bool live = k16 < 2 * blk_e && c < nchunks;
        const uint32_t so = ((uint32_t)((k16 >> 2) * (C / 32) + wave * CW + (CW == 2 ? which : 0)) * 4u + (uint32_t)(k16 & 3)) * 1024u;
        return __builtin_amdgcn_raw_buffer_load_b128(rs_w2, live ? lane16 : (int)p.w2_bytes, live ? (int)so : 0, 0);
    };

    // Every workgroup streams the same weights.  Workgroups that start together and walk the hidden range in the same order request the
    // same lines from the same L2 channels at the same time; each row tile therefore starts at its own chunk and wraps around (the output
    // sums the chunks in that order: deterministic per tile, whatever the launch looks like).
    const int rot = p.rot ? (int)((uint32_t)tile % (uint32_t)nchunks) : 0;
    auto chunk_of = [&](int ci) { const int e = ci + rot; return e >= nchunks ? e - nchunks : e; };
    // ---- weights first: D steps of the first chunk on their way before anything else
    i32x4 wf[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d) { wf[d][0] = frag(chunk_of(0), d, 0); wf[d][1] = frag(chunk_of(0), d, 1); }

    // ---- x tile -> LDS (16-byte chunks, coalesced), the part's ff1 bias, RMSNorm scale
    {
        constexpr int CPR = C / 8, NLD = 64 * CPR / NT;                  // chunks per row; chunks per thread (8)
        i32x4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + NT * i, row = q / CPR, ch = q % CPR;
            int64_t m = row0 + row;
            bool ok = m < p.rows;
            if (p.shift_hw > 0 && ch >= CPR / 2) {                       // TokenShift: second half of the channels from the previous frame
                const uint32_t fr = (uint32_t)m / (uint32_t)p.shift_hw;
                ok = ok && (fr % (uint32_t)p.T) != 0u;
                m -= p.shift_hw;
            }
            v[i] = ok ? *(const i32x4*)((const char*)p.x + (m * CPR + ch) * 16) : i32x4{0, 0, 0, 0};
        }
        float* b1l = (float*)(smem + L::kB1);
        const int nb1 = (blk_e - blk_b) * 64;
        for (int i = tid; i < nb1; i += NT) b1l[i] = p.b1[blk_b * 64 + i];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + NT * i, row = q / CPR, ch = q % CPR;
            *(i32x4*)(smem + L::kX + row * PX + ch * 16) = v[i];
        }
    }
    __syncthreads();
    {
        constexpr int TPR = NT / 64, CPT = C / 8 / TPR;                  // threads per row, 16-byte chunks per thread
        const int row = tid / TPR, q = tid % TPR;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            float xv[8];
            Chunk<BF16>::unpack(*(const i32x4*)(smem + L::kX + row * PX + (q * CPT + i) * 16), xv);
#pragma unroll
            for (int k = 0; k < 8; ++k) ss += xv[k] * xv[k];
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
        if (q == 0) ((float*)(smem + L::kRs))[row] = 1.f / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize(x, dim = 1), M:276
    }
    __syncthreads();

    FF_PROF(0)
    f32x16 acc2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[a][r][e] = 0.f;
    const float rs0 = ((const float*)(smem + L::kRs))[l31], rs1 = ((const float*)(smem + L::kRs))[32 + l31];
    const char* xrd = smem + L::kX + l31 * PX + hi * 16;

    // B fragments (positions) are read one step ahead of their MFMAs: bq[t & 1] holds step t's pair
    auto ldsq = [&](const char* a) -> i32x4 {
        if constexpr ((ABL & 4) != 0) { i32x4 z = {0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00}; asm volatile("" : "+v"(z)); return z; }      // experiments: no fragment reads
        else return *(const i32x4*)a;
    };
    auto sched_pin = [&]() { if constexpr ((ABL & 8) == 0) __builtin_amdgcn_sched_barrier(0); };      // experiments (8): the scheduler's own order
    i32x4 bq[2][2 * (3 - CW)];      // CW = 1: an ff2 step takes two hidden k-steps x two row tiles
    bq[0][0] = ldsq(xrd); bq[0][1] = ldsq(xrd + 32 * PX);
    for (int ci = 0; ci < nchunks; ++ci) {
        const int c = chunk_of(ci), cn = ci + 1 < nchunks ? chunk_of(ci + 1) : nchunks;      // cn == nchunks: dead prefetch
        const bool active = blk_b + c * NW + wave < blk_e;               // wave-uniform
        char* panel = smem + L::kPanel + (ci & 1) * 64 * PP;
        // ---- ff1: this wave's hidden block, K = C
        f32x16 acc1[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[a][r][e] = 0.f;
#pragma unroll
        for (int t = 0; t < KS1; ++t) {
            if (t + 1 < KS1) { bq[(t + 1) & 1][0] = ldsq(xrd + (t + 1) * 32); bq[(t + 1) & 1][1] = ldsq(xrd + 32 * PX + (t + 1) * 32); }
            sched_pin();       // the scheduler otherwise sinks every load to its use (seen in the ISA: load, vmcnt(0), MFMA)
            Mma<BF16>::run(wf[t % D][0], bq[t & 1][0], acc1[0][0]);
            Mma<BF16>::run(wf[t % D][0], bq[t & 1][1], acc1[0][1]);
            Mma<BF16>::run(wf[t % D][1], bq[t & 1][0], acc1[1][0]);
            Mma<BF16>::run(wf[t % D][1], bq[t & 1][1], acc1[1][1]);
            wf[t % D][0] = frag(c, t + D, 0);                            // t + D < T always (KS2 >= D)
            wf[t % D][1] = frag(c, t + D, 1);
            sched_pin();
        }
        FF_PROF(1)
        // ---- GEGLU in registers -> panel.  acc1[0 | 1][r][4 g + j] = value | gate of hidden unit 8 g + 4 hi + j of the block, row 32 r + l31
        if (active) {
            const float* bl = (const float*)(smem + L::kB1) + (c * NW + wave) * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bx = *(const f32x4*)(bl + 8 * g + 4 * hi), bg = *(const f32x4*)(bl + 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float rs = r ? rs1 : rs0;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xv = acc1[0][r][4 * g + j] * rs + bx[j], gv = acc1[1][r][4 * g + j] * rs + bg[j];
                        if constexpr ((ABL & 1) != 0) o[j] = gv * xv;      // experiments: no GELU math
                        else o[j] = Act<BF16>::gelu(gv) * xv;
                    }
                    lds_put4<BF16>(panel + (r * 32 + l31) * PP + (wave * 32 + 8 * g + 4 * hi) * 2, o);
                }
            }
        } else {                                                        // a block past the part's range: its ff2 k-steps multiply zeros
            const float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 2; ++r) lds_put4<BF16>(panel + (r * 32 + l31) * PP + (wave * 32 + 8 * g + 4 * hi) * 2, z);
        }
        // bare barrier: only the LDS writes must have landed (a __syncthreads() would also drain the weight ring: vmcnt(0)).  The other
        // panel buffer was last read in chunk c - 1's ff2, which every wave has left before it arrives here.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FF_PROF(2)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        FF_PROF(3)
        // ---- ff2: acc2 (this wave's 64 output channels) += W2[:, chunk] x panel
        const char* prd = panel + l31 * PP + hi * 16;
        auto panel_frags = [&](int kk, i32x4* dst) {
            if constexpr (CW == 2) { dst[0] = ldsq(prd + kk * 32); dst[1] = ldsq(prd + 32 * PP + kk * 32); }
            else { dst[0] = ldsq(prd + 2 * kk * 32); dst[1] = ldsq(prd + 32 * PP + 2 * kk * 32);
                   dst[2] = ldsq(prd + (2 * kk + 1) * 32); dst[3] = ldsq(prd + 32 * PP + (2 * kk + 1) * 32); }
        };
        panel_frags(0, bq[0]);
#pragma unroll
        for (int kk = 0; kk < KS2; ++kk) {
            const int t = KS1 + kk;
            if (kk + 1 < KS2) panel_frags(kk + 1, bq[(kk + 1) & 1]);
            else { bq[0][0] = ldsq(xrd); bq[0][1] = ldsq(xrd + 32 * PX); }      // the next chunk's first ff1 step
            sched_pin();
            // CW = 2: fragment `which` = channel tile, both on this k-step's row fragments; CW = 1: fragment `which` = k-step, on its own pair
            Mma<BF16>::run(wf[t % D][0], bq[kk & 1][0], acc2[0][0]);
            Mma<BF16>::run(wf[t % D][0], bq[kk & 1][1], acc2[0][1]);
            Mma<BF16>::run(wf[t % D][1], bq[kk & 1][CW == 2 ? 0 : 2], acc2[1][0]);
            Mma<BF16>::run(wf[t % D][1], bq[kk & 1][CW == 2 ? 1 : 3], acc2[1][1]);
            if (t + D < T) { wf[t % D][0] = frag(c, t + D, 0); wf[t % D][1] = frag(c, t + D, 1); }
            else { wf[t % D][0] = frag(cn, t + D - T, 0); wf[t % D][1] = frag(cn, t + D - T, 1); }
            sched_pin();
        }
        FF_PROF(4)
    }

    if constexpr (CW == 1) {      // the two k-step parities of the one channel tile
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[0][r][e] += acc2[1][r][e];
    }
    // ---- a tile split across workgroups: publish the partial (write-through), take a ticket, the last arriver sums every part in part
    // order (cdna_hip_programming.md section 5 "in-launch split-K reduction"; same form as conv3p.hip)
    if (p.S > 1) {
        const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, p.ws_bytes, 0x00020000);
        const uint32_t inslab = (uint32_t)(wave * CW * 8 * 1024 + lane * 16);
        const uint32_t soff = (uint32_t)(tile * p.S + part) * (uint32_t)L::SLAB + inslab;
#pragma unroll
        for (int a = 0; a < CW; ++a)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const i32x4 vv = {__float_as_int(acc2[a][r][4 * r4]), __float_as_int(acc2[a][r][4 * r4 + 1]),
                                      __float_as_int(acc2[a][r][4 * r4 + 2]), __float_as_int(acc2[a][r][4 * r4 + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(vv, rs_ws, (int)(soff + (uint32_t)(((a * 2 + r) * 4 + r4) * 1024)), 0, 16 /* sc1 */);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains before the ticket
        __syncthreads();
        uint32_t* flag = (uint32_t*)(smem + L::kFlag);
        if (tid == 0) {
            const uint32_t ticket = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t last = ticket == (uint32_t)p.S - 1u ? 1u : 0u;
            if (last) __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) { FF_PROF(5) FF_PROF_END return; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[a][r][e] = 0.f;
        for (int q = 0; q < p.S; ++q) {
            const uint32_t so = (uint32_t)(tile * p.S + q) * (uint32_t)L::SLAB + inslab;
            i32x4 v[8 * CW];
#pragma unroll
            for (int k = 0; k < 8 * CW; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_ws, (int)(so + (uint32_t)(k * 1024)), 0, 16 /* sc1 */);
#pragma unroll
            for (int k = 0; k < 8 * CW; ++k)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc2[k >> 3][(k >> 2) & 1][4 * (k & 3) + jj] += __int_as_float(v[k][jj]);
        }
    }

    FF_PROF(5)
    // ---- epilogue: out = bf16(acc + b2 + x), rounded once.  acc2[a][r][4 g + j] = channel 32 CW wave + 32 a + 8 g + 4 hi + j of row 32 r + l31.
    // Every wave has left its last ff1 (the last chunk's barrier), so the x tile turns into the output tile in place: each 8-byte
    // group is read and rewritten by the one lane that owns it.  Token-shifted layers read the residual from memory (the tile holds
    // the shifted row).
#pragma unroll
    for (int a = 0; a < CW; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = wave * 32 * CW + a * 32 + 8 * g + 4 * hi;
            const f32x4 bv = *(const f32x4*)(p.b2 + col);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                char* cell = smem + L::kX + (r * 32 + l31) * PX + col * 2;
                float xr[4];
                if (p.shift_hw > 0) {
                    const int64_t m = row0 + r * 32 + l31;
                    if (m < p.rows) ld4<BF16>(p.x, m * C + col, xr);
                    else { xr[0] = xr[1] = xr[2] = xr[3] = 0.f; }
                } else {
                    const i32x2 w = *(const i32x2*)cell;
                    xr[0] = __uint_as_float(((unsigned)w[0]) << 16); xr[1] = __uint_as_float(((unsigned)w[0]) & 0xffff0000u);
                    xr[2] = __uint_as_float(((unsigned)w[1]) << 16); xr[3] = __uint_as_float(((unsigned)w[1]) & 0xffff0000u);
                }
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (acc2[a][r][4 * g + j] + bv[j]) + xr[j];
                lds_put4<BF16>(cell, o);
            }
        }
    __syncthreads();
    {
        constexpr int CPR = C / 8, NLD = 64 * CPR / NT;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + NT * i, row = q / CPR, ch = q % CPR;
            const int64_t m = row0 + row;
            if (m < p.rows) *(i32x4*)((char*)p.out + (m * CPR + ch) * 16) = *(const i32x4*)(smem + L::kX + row * PX + ch * 16);
        }
    }
    FF_PROF(6)
    FF_PROF_END
}

inline int ff_cus() {      // of the CURRENT device; not cached: a process may drive several devices
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    return cus;
}

struct FfPlan { int CT, NW, nblk, tiles, S; size_t lds, slab_bytes, counter_bytes; };

inline bool ff_shape_ok(int dtype, int64_t rows, int C, int hidden_pad) {
    return dtype == MV2_BF16 && rows > 0 && rows < (int64_t)0x7fffffff / 1024 && (C == 256 || C == 512) && hidden_pad >= 64 && hidden_pad % 64 == 0 &&
           hidden_pad <= 8192;
}

// Workgroups per row tile (S) and waves per workgroup.  The unit of time is one chunk iteration (every wave: ff1 of one hidden block + GEGLU +
// its share of ff2, for 64 rows); a workgroup pays about 0.9 iterations of fixed phases (x tile, RMS scale, epilogue: 11k of the 13.4k
// ticks of an iteration in the phase clocks) and, when a tile is split, another ~1.0 for slab + ticket + the last arriver's reduction.  Measured on MI355X (profiles/r05_findings.md): a workgroup
// that has a CU to itself takes as long as two that share one - a wave issues an MFMA every ~90 clocks whatever its neighbours do - so
// the launch lasts  ceil(workgroups / resident slots) x (iterations + fixed);  the smallest S that minimises that wins.  Parts keep at
// least half a chunk of blocks.
inline FfPlan ff_plan(int64_t rows, int C, int hidden_pad) {
    FfPlan P;
    P.CT = C / 128; P.NW = C / 64;
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_FF_NW")) { if (atoi(e) == 8 && C == 256) P.NW = 8; }
#endif
    P.nblk = hidden_pad / 32;
    P.tiles = (int)((rows + 63) / 64);
    // resident workgroups per CU: two of the 4-wave form where 2 x its LDS fits (kB1 + the largest bias slice this loop may pick: S = 1),
    // else one
    const size_t kb1_ = C == 512 ? (size_t)FfCfg<4, 8>::kB1 : (P.NW == 8 ? (size_t)FfCfg<2, 8>::kB1 : (size_t)FfCfg<2, 4>::kB1);
    const int64_t slots = (int64_t)ff_cus() * ((P.NW == 4 && 2 * (kb1_ + (size_t)(P.nblk + 1) * 256) <= (size_t)160 * 1024) ? 2 : 1);
    double best = 1e30; int bestS = 1;
    for (int s = 1; s <= 8; ++s) {
        if (s > 1 && 2 * (P.nblk / s) < P.NW) break;
        const int per = (P.nblk + s - 1) / s, chunks = (per + P.NW - 1) / P.NW;
        const double wg = chunks + 0.9 + (s > 1 ? 1.0 : 0.0);
        const double t = (double)(((int64_t)P.tiles * s + slots - 1) / slots) * wg;
        if (t < best - 1e-9) { best = t; bestS = s; }
    }
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_FF_S")) { const int v = atoi(e); if (v >= 1 && v <= 8 && (v == 1 || P.nblk / v >= 1)) bestS = v; }
#endif
    P.S = bestS;
    const int per_max = (P.nblk + P.S - 1) / P.S + 1;
    const int kb1 = C == 512 ? FfCfg<4, 8>::kB1 : (P.NW == 8 ? FfCfg<2, 8>::kB1 : FfCfg<2, 4>::kB1);
    P.lds = (size_t)kb1 + (size_t)per_max * 64 * 4;
    const size_t slab = C == 512 ? (size_t)FfCfg<4, 8>::SLAB : (P.NW == 8 ? (size_t)FfCfg<2, 8>::SLAB : (size_t)FfCfg<2, 4>::SLAB);
    P.slab_bytes = P.S > 1 ? (size_t)P.tiles * P.S * slab : 0;
    P.counter_bytes = (size_t)P.tiles * 4;
    return P;
}

}  // namespace

// The shape rule above bounds hidden_pad by 8192, the LDS does not: x tile + panels + per_max * 256 bytes of bias must fit the CU's 160 KB
// (C = 512: kB1 is ~134 KB, i.e. hidden_pad up to ~3600 unsplit).  mv2_ff_eligible and the sizing queries say no where the plan
// does not fit, so that a caller never commits to a launch hipFuncSetAttribute would refuse.
inline bool ff_fits(int64_t rows, int C, int hidden_pad) { return ff_plan(rows, C, hidden_pad).lds <= (size_t)160 * 1024; }

extern "C" int mv2_ff_eligible(int dtype, int64_t rows, int C, int hidden_pad) { return ff_shape_ok(dtype, rows, C, hidden_pad) && ff_fits(rows, C, hidden_pad) ? 1 : 0; }

extern "C" size_t mv2_ff_workspace_bytes(int64_t rows, int C, int hidden_pad) {
    if (!ff_shape_ok(MV2_BF16, rows, C, hidden_pad) || !ff_fits(rows, C, hidden_pad)) return 0;
    return ff_plan(rows, C, hidden_pad).slab_bytes;
}

extern "C" size_t mv2_ff_counter_bytes(int64_t rows, int C, int hidden_pad) {
    if (!ff_shape_ok(MV2_BF16, rows, C, hidden_pad) || !ff_fits(rows, C, hidden_pad)) return 0;
    return ff_plan(rows, C, hidden_pad).counter_bytes;
}

extern "C" int mv2_ff(int dtype, const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int64_t rows,
                      int C, int hidden_pad, int shift_hw, int T, void* workspace, size_t workspace_bytes, void* counters,
                      size_t counter_bytes, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !out || x == out) return MV2_EINVAL;
    if (!ff_shape_ok(dtype, rows, C, hidden_pad) || !ff_fits(rows, C, hidden_pad)) return MV2_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)out | (uintptr_t)b1 | (uintptr_t)b2) & 15)) return MV2_EINVAL;
    if (shift_hw < 0 || (shift_hw > 0 && (T <= 0 || rows % ((int64_t)shift_hw * T) != 0))) return MV2_EINVAL;
    const FfPlan P = ff_plan(rows, C, hidden_pad);
    const int co_pad1 = (2 * hidden_pad + 127) / 128 * 128;
    const uint64_t w1b = (uint64_t)(C / 64) * co_pad1 * 128, w2b = (uint64_t)(hidden_pad / 64) * C * 128;
    if (w1b >= 0xfffffff0ull || w2b >= 0xfffffff0ull || P.slab_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    if (P.S > 1 && (!workspace || workspace_bytes < P.slab_bytes || !counters || counter_bytes < P.counter_bytes)) return MV2_EWORKSPACE;
    if (P.S > 1 && (((uintptr_t)workspace & 15) || ((uintptr_t)counters & 3))) return MV2_EINVAL;
    FfParams p;
    p.x = x; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.out = out; p.ws = (float*)workspace; p.counters = (uint32_t*)counters;
    p.rows = rows; p.w1_bytes = (uint32_t)w1b; p.w2_bytes = (uint32_t)w2b; p.ws_bytes = (uint32_t)P.slab_bytes;
    p.nblk = P.nblk; p.n1tiles = co_pad1 / 32; p.S = P.S; p.tiles = P.tiles; p.shift_hw = shift_hw; p.T = T;
    p.rot = 1; p.abl = 0; p.prof = 0;
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_FF_ROT")) p.rot = atoi(e);
    if (const char* e = getenv("MV2_FF_ABL")) p.abl = atoi(e);
#endif
    const dim3 grid((unsigned)((int64_t)P.tiles * P.S));
    hipStream_t st = (hipStream_t)stream;
#ifdef MV2_EXPERIMENTS
    static unsigned long long* prof_store = nullptr;
    unsigned long long* prof_buf = nullptr;
    if (getenv("MV2_FF_PROF")) {
        if (!prof_store && hipMalloc((void**)&prof_store, 4096 * 8 * 8) != hipSuccess) return MV2_ENODEVICE;
        prof_buf = prof_store;
        p.prof = 1;
        (void)hipMemset(prof_buf, 0, 4096 * 8 * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ff_prof), &prof_buf, sizeof(prof_buf));
    }
#endif
#define FF_LAUNCH(CTv, NWv, Dv, Av)                                                                                              \
    do {                                                                                                                         \
        if (hipFuncSetAttribute((const void*)ff_kernel<CTv, NWv, Dv, Av>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds) != hipSuccess) \
            return MV2_ENODEVICE;                                                                                                \
        hipLaunchKernelGGL((ff_kernel<CTv, NWv, Dv, Av>), grid, dim3(NWv * 64), P.lds, st, p);                                   \
    } while (0)
#ifdef MV2_EXPERIMENTS
    if (p.abl) {
        if (C == 256) {
            switch (p.abl) { case 1: FF_LAUNCH(2, 4, 4, 1); break; case 2: FF_LAUNCH(2, 4, 4, 2); break; case 3: FF_LAUNCH(2, 4, 4, 3); break;
                             case 7: FF_LAUNCH(2, 4, 4, 7); break; case 8: FF_LAUNCH(2, 4, 4, 8); break; case 11: FF_LAUNCH(2, 4, 4, 11); break;
                             default: FF_LAUNCH(2, 4, 4, 15); }
        } else {
            switch (p.abl) { case 1: FF_LAUNCH(4, 8, 4, 1); break; case 2: FF_LAUNCH(4, 8, 4, 2); break; case 3: FF_LAUNCH(4, 8, 4, 3); break;
                             case 7: FF_LAUNCH(4, 8, 4, 7); break; case 8: FF_LAUNCH(4, 8, 4, 8); break; case 11: FF_LAUNCH(4, 8, 4, 11); break;
                             default: FF_LAUNCH(4, 8, 4, 15); }
        }
    } else
#endif
    if (C == 256) {
#ifdef MV2_EXPERIMENTS
        if (P.NW == 8) FF_LAUNCH(2, 8, 4, 0); else
#endif
        FF_LAUNCH(2, 4, 4, 0);
    } else FF_LAUNCH(4, 8, 4, 0);
#undef FF_LAUNCH
    MV2_LAUNCH_CHECK();
#ifdef MV2_EXPERIMENTS
    if (prof_buf) {
        if (hipStreamSynchronize(st) != hipSuccess) return MV2_ENODEVICE;
        const int nb = (int)std::min<int64_t>((int64_t)P.tiles * P.S, 4096);
        std::vector<unsigned long long> h((size_t)nb * 8);
        if (hipMemcpy(h.data(), prof_buf, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return MV2_ENODEVICE;
        double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < nb; ++b) for (int i = 0; i < 8; ++i) m[i] += (double)h[(size_t)b * 8 + i] / nb;
        fprintf(stderr, "[ff prof] C=%d rows=%lld S=%d blocks=%d ticks/wg: prologue %.0f ff1 %.0f geglu %.0f barrier %.0f ff2 %.0f slab %.0f epilogue %.0f total %.0f\n",
                C, (long long)rows, P.S, nb, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        // the quartiles of the workgroups' lifetimes: lone workgroups (one wave per SIMD) against pairs that share a CU
        std::vector<std::pair<unsigned long long, int>> tot;
        for (int b = 0; b < nb; ++b) tot.push_back({h[(size_t)b * 8 + 7], b});
        std::sort(tot.begin(), tot.end());
        for (int qi = 0; qi < 5; ++qi) {
            const int b = tot[(size_t)((nb - 1) * qi / 4)].second;
            fprintf(stderr, "[ff prof]   q%d: total %llu = prologue %llu ff1 %llu geglu %llu barrier %llu ff2 %llu slab %llu epilogue %llu\n", qi, h[(size_t)b * 8 + 7],
                    h[(size_t)b * 8], h[(size_t)b * 8 + 1], h[(size_t)b * 8 + 2], h[(size_t)b * 8 + 3], h[(size_t)b * 8 + 4], h[(size_t)b * 8 + 5], h[(size_t)b * 8 + 6]);
        }
        unsigned long long* none = nullptr;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ff_prof), &none, sizeof(none));
    }
#endif
    return MV2_OK;
}
