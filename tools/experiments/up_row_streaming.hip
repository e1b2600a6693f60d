// NOT IN THE LIBRARY (round 5 experiment, kept as the record of a measurement): correct on its first run and bit-identical to the tiled kernels,
// but slower in the step graph in all three forms that were tried - register epilogue with 16-byte scattered stores (+54 us per step at B = 1),
// LDS-staged epilogue with a k-slice ring (decoder_layers.7 19 -> 46 us: the plain-image fragment reads thrash the 32 KB L1), eight loads back to
// back per step (decoder_layers.10 20.5 -> 34.3, .12 30.5 -> 31.9).  profiles/r05_findings.md section 6.  To build it: add it to build.py's
// SOURCES and call mv2i_conv_up_ok / mv2i_conv_up from conv.hip's conv_impl.
//
// Row-streaming form of the up-sampling convolutions (reference SpatialUpsample2x / TimeUpsample2x, magvit2_pytorch.py:808-883:
// Conv 1x1x1 (C -> r C') + SiLU + Rearrange 'b (c p1 p2) h w -> b c (h p1) (w p2)' / 'b (c p) t -> b c (t p)'), bf16, behind mv2_conv.
//
// These launches move bytes, not flops (decoder_layers.12 at configs[1]: 21 MB in, 52 MB out, 5.4 GF): through the tiled GEMM kernels a
// workgroup spent two thirds of its life in the LDS-staged epilogue (profiles/r04_findings.md section 6: 11-22k of 16-30k clocks).  Here a
// workgroup owns 64 input rows and ALL r C' output columns:
//   * the 64 x K input tile goes to LDS once (K = C <= 512: <= 66 KB), one barrier;
//   * NW = columns / 64 waves (4 / 8 / 16), each with its own 64 columns: 2 x 2 MFMA tiles, K / 16 k-steps; the A fragments come straight
//     from the PLAIN packed image mv2_conv takes ([step][co_pad][128 B]: lane (row, half) reads chunk 2 ks + half of its row - the four
//     k-slices of a step use every byte of the 32 lines a wave touches), through a ring of four k-slices x two fragments (each slot re-requested 16 MFMAs ahead);
//   * epilogue: bias, SiLU, bf16 into a wave-private 64 x 64 tile in the LDS the input tile occupied, destination rows (depth-to-space /
//     depth-to-time) decoded once per input row, 8 lanes write each 128-byte piece of an output row.
// Results are those of the tiled kernels (same k order per output element, same epilogue arithmetic): bit-identical.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "conv_common.h"
#include "../../include/mv2.h"

namespace {

template <int NW>
__global__ __launch_bounds__(NW * 64) void conv_up_kernel(const ConvParams p) {
    typedef BF16 E;
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const mv2_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = d.Ci, PX = K * 2 + 16, nsteps = d.nsteps;            // K = 64 nsteps
    const int64_t m0 = (int64_t)blockIdx.x * 64;

    // ---- weights of the first step: this wave's two 32-column tiles, four k-slices each (lane (row l31, half hi): chunk 2 ks + hi of its row)
    const char* wrow[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) wrow[a] = (const char*)p.wp + (size_t)(wave * 64 + a * 32 + l31) * 128 + hi * 16;
    const size_t w_step = (size_t)d.co_pad * 128;
    // two register sets: ALL eight fragment loads of step s + 1 go out back to back while step s computes.  The four k-slices of a step read
    // the same 64 lines (32 bytes each per load): spread over a step's MFMAs (a first version) the lines left the 32 KB L1 between two uses
    // and the launch ran at 4x the L2 traffic (decoder_layers.7: 19 -> 46 us)
    i32x4 wf[2][4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int a = 0; a < 2; ++a) wf[0][ks][a] = *(const i32x4*)(wrow[a] + ks * 32);

    // ---- input tile -> LDS (coalesced 16-byte chunks; rows past M are zeros)
    {
        const int cpr = K / 8, total = 64 * cpr;
        for (int q = tid; q < total; q += NT) {
            const int row = q / cpr, ch = q - row * cpr;
            const int64_t m = m0 + row;
            const i32x4 v = m < p.M ? *(const i32x4*)((const char*)p.in + (m * cpr + ch) * 16) : i32x4{0, 0, 0, 0};
            *(i32x4*)(smem + row * PX + ch * 16) = v;
        }
    }
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][r][e] = 0.f;
    const char* xrd = smem + l31 * PX + hi * 16;
    auto step = [&](int s, auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value, NXT = CUR ^ 1;
        const size_t wnext = (size_t)(s + 1 < nsteps ? s + 1 : s) * w_step;      // the last step re-reads its own (cached) fragments: no branch around a load
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) wf[NXT][ks][a] = *(const i32x4*)(wrow[a] + wnext + ks * 32);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const i32x4 b0 = *(const i32x4*)(xrd + (s * 4 + ks) * 32), b1 = *(const i32x4*)(xrd + 32 * PX + (s * 4 + ks) * 32);
            Mma<E>::run(wf[CUR][ks][0], b0, acc[0][0]);
            Mma<E>::run(wf[CUR][ks][0], b1, acc[0][1]);
            Mma<E>::run(wf[CUR][ks][1], b0, acc[1][0]);
            Mma<E>::run(wf[CUR][ks][1], b1, acc[1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(s, IC<0>{});
        if (s + 1 < nsteps) step(s + 1, IC<1>{});
    }

    // ---- epilogue.  acc[a][r][4 g + j] = packed column 64 wave + 32 a + 8 g + 4 hi + j of row 32 r + l31.  A first version stored from
    // registers (16 bytes per lane after v_permlane32_swap, as conv3p.hip does): correct and 50 us slower per step - a wave instruction
    // then carries 64 partial writes into 64 different 128-byte lines.  So: bias + SiLU -> a wave-private [64 rows][64 columns] bf16 tile
    // (144-byte pitch) in the LDS the input tile occupied, destination rows decoded once per row, and 8 lanes write each 128-byte piece.
    constexpr int TP = 144;                                     // staged row pitch
    __syncthreads();                                            // every wave has read the input tile
    int64_t* rowtab = (int64_t*)(smem + NW * 64 * TP);
    if (tid < 64) {
        const int64_t m = m0 + tid;
        int64_t dr = -1;
        if (m < p.M) {
            int b_, t_, h_, w_;
            decode_pos<E>(d, m, b_, t_, h_, w_);
            // destination row of sub-position (0, 0, 0); sub-position q adds ((qt Ho' + qh) Wo' + qw) rows (Ho' = Ho rh, Wo' = Wo rw)
            dr = (((int64_t)b_ * (d.To * d.rt) + t_ * d.rt) * (d.Ho * d.rh) + h_ * d.rh) * (int64_t)(d.Wo * d.rw) + w_ * d.rw;
        }
        rowtab[tid] = dr;
    }
    char* priv = smem + wave * (64 * TP);
    const bool silu = d.act == MV2_ACT_SILU;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int lc = a * 32 + 8 * g + 4 * hi;             // column inside the wave's 64
            const f32x4 bv = p.bias ? *(const f32x4*)(p.bias + wave * 64 + lc) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[a][r][4 * g + j] + bv[j];
                    x[j] = silu ? Act<E>::silu(v) : v;
                }
                lds_put4<E>(priv + (r * 32 + l31) * TP + lc * 2, x);
            }
        }
    __syncthreads();                                            // the tiles and the row table are complete
    {
        const int cq = d.out_c;
        const int ch = lane & 7, rr = lane >> 3;                // 8 lanes per 128-byte piece, 8 rows per instruction
        const int n = wave * 64 + ch * 8;                       // first of this lane's 8 packed columns: one sub-position (out_c % 8 == 0)
        const int q = n / cq, c = n - q * cq;
        const int qw = q % d.rw, qh = (q / d.rw) % d.rh, qt = q / (d.rw * d.rh);
        const int64_t qoff = ((int64_t)qt * (d.Ho * d.rh) + qh) * (int64_t)(d.Wo * d.rw) + qw;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rr + 8 * i;
            const int64_t dr = rowtab[row];
            if (dr >= 0) *(i32x4*)((char*)p.out + ((dr + qoff) * cq + c) * 2) = *(const i32x4*)(priv + row * TP + ch * 16);
        }
    }
}

}  // namespace

// Launches mv2_conv hands over (conv.hip checks nothing else for them): bf16 1x1x1 stride-1 un-padded convolution over whole 128-byte
// steps with K <= 512, depth-to-space / depth-to-time store, SiLU or no activation, no row scale, no residual, 256 / 512 / 1024 packed
// columns, 8-channel groups that do not straddle a sub-position, and enough rows to fill the chip.  A pure function of the descriptor.
bool mv2i_conv_up_ok(const mv2_conv_desc& d) {
#ifdef MV2_EXPERIMENTS
    if (const char* e = getenv("MV2_CONV_UP")) { if (atoi(e) == 0) return false; }
#endif
    if (d.dtype != MV2_BF16 || d.splitk > 1) return false;
    if (!(d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0 &&
          d.To == d.Ti && d.Ho == d.Hi && d.Wo == d.Wi)) return false;
    if (d.seg_bytes != 128 || d.Ci % 64 != 0 || d.nsteps != d.Ci / 64 || d.Ci > 512) return false;
    if (d.store != MV2_STORE_SHUFFLE || (d.act != MV2_ACT_SILU && d.act != MV2_ACT_NONE)) return false;
    if (d.co != d.co_pad || (d.co != 256 && d.co != 512) || d.out_c % 8 != 0) return false;
    const int64_t M = (int64_t)d.B * d.To * d.Ho * d.Wo;
    return M >= 4096;
}

int mv2i_conv_up(const ConvParams& p, void* stream) {
    const mv2_conv_desc& d = p.d;
    const int nw = d.co / 64;
    const size_t lds = std::max<size_t>((size_t)64 * (d.Ci * 2 + 16), (size_t)nw * 64 * 144 + 64 * 8);
    const dim3 grid((unsigned)((p.M + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
#define UP_LAUNCH(NWv)                                                                                                                  \
    do {                                                                                                                                \
        if (hipFuncSetAttribute((const void*)conv_up_kernel<NWv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
            return MV2_ENODEVICE;                                                                                                       \
        hipLaunchKernelGGL((conv_up_kernel<NWv>), grid, dim3(NWv * 64), lds, st, p);                                                    \
    } while (0)
    if (d.co == 256) UP_LAUNCH(4); else if (d.co == 512) UP_LAUNCH(8); else UP_LAUNCH(16);
#undef UP_LAUNCH
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
