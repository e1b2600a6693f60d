// mv2_conv_in — the encoder's input convolution (reference magvit2_pytorch.py:1109, 1557: CausalConv3d(channels -> init_dim, 7 x 7 x 7))
// over the pre-padded channel-last clip mv2_video_to_ndhwc writes: 4-channel bf16 pixels (8 bytes), 64 output channels, bf16 rows out.
//
// Why its own kernel: through the generic gather kernel every input row travels from L2 to the CU once per (time, height) tap — 1280
// workgroups x 25 K steps x 40 KB = 1.28 GB through the L1s for a 3.7 MB clip, 107 us per clip on MI355X (profiles/r03): the launch is
// bound by the gathers, not by its 43 GF.  The same defect cost the output convolution 2 x (convout.hip).  Here a workgroup owns an
// 8 x 32 tile of ONE output frame: the 7 input frames of its 14 x 40 halo (31 KB) go to LDS once and every tap is an immediate offset.
//
// GEMM shape: K = 7 (it) x 7 (ih) x 8 pixels (iw 0..6 + one zero-weight pad) x 4 channels = 49 x 32; one k16 step = 4 pixels x 4
// channels of one (it, ih) row, so the B operand of v_mfma_f32_32x32x16_bf16 (32 positions of a tile row; lane = (position, pixel pair))
// is 16 contiguous bytes of the halo row: two pixels.  A operand = weights: the 28 KB slice of one `it` sits in LDS ([step][co tile][lane]
// fragments, written by the host packer), the next slice travels global -> registers under the MFMAs.  Wave w owns tile rows 2w, 2w + 1
// and all 64 channels: 2 x 2 accumulator tiles, 4 MFMAs per 2 + 2 LDS fragment reads.
#include <hip/hip_runtime.h>
#include "common.h"
#include "conv_common.h"
#include "../../include/mv2.h"

namespace {

constexpr int CI_TH = 8, CI_TW = 32;                       // output tile
constexpr int CI_K = 7;                                    // kh = kw = 7; kt <= 7
constexpr int CI_HH = CI_TH + CI_K - 1;                    // 14 halo rows
constexpr int CI_HWP = CI_TW + 8;                          // 40 halo pixels per row: 32 + 6, + 2 so that the zero-weight pad pixel exists
constexpr int CI_ROWB = CI_HWP * 8;                        // 320 bytes per halo row
constexpr int CI_FRAME = CI_HH * CI_ROWB;                  // 4480 bytes per halo frame
constexpr int CI_HALO = 7 * CI_FRAME;                      // 31360
constexpr int CI_STEPS = CI_K * 2;                         // k16 steps per `it` slice
constexpr int CI_SLICE = CI_STEPS * 2 * 1024;              // 28672 bytes: [step][co tile][64 lanes][16 B]
constexpr int CI_W = ((CI_HALO + 255) / 256) * 256;        // LDS offset of the weight slice
constexpr int CI_PITCH = 128 + 16;                         // staged output row (64 bf16 + pad)
constexpr int CI_LDS = CI_W + CI_SLICE;
static_assert(4 * 64 * CI_PITCH <= CI_LDS, "the four wave-private output tiles live in the (dead) halo + weight regions");

struct ConvInParams {
    const void* in; const void* wfrag; const float* bias; void* out;
    uint32_t in_bytes;
    int B, To, Ho, Wo, Tp, Hp, Wp, kt, tiles_h, tiles_w;
};

__global__ __launch_bounds__(256, 2) void conv_in_kernel(const ConvInParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- tile (XCD-aware: contiguous eighths of the (b, t, tile row, tile col) list per XCD: neighbours share halo rows in one L2)
    uint32_t bid = blockIdx.x;
    {
        const uint32_t NW = gridDim.x, q = NW >> 3, r = NW & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int twi = (int)(bid % (uint32_t)p.tiles_w); uint32_t rr = bid / (uint32_t)p.tiles_w;
    const int thi = (int)(rr % (uint32_t)p.tiles_h); rr /= (uint32_t)p.tiles_h;
    const int t = (int)(rr % (uint32_t)p.To), b = (int)(rr / (uint32_t)p.To);
    const int h0 = thi * CI_TH, w0 = twi * CI_TW;

    // ---- loads: halo (kt frames x 14 rows x 20 pixel pairs) and the first weight slice, all issued before the first wait
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    constexpr int CPF = CI_HH * (CI_HWP / 2);              // 280 16-byte chunks (pixel pairs) per frame
    constexpr int NLD = (7 * CPF + 255) / 256;             // 8
    const int total = p.kt * CPF;
    i32x4 hv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        uint32_t off = p.in_bytes;                          // past the clip (ragged right / bottom tiles, tail) -> zeros
        if (idx < total) {
            const int fr = idx / CPF, rem = idx - fr * CPF;
            const int hy = rem / (CI_HWP / 2), hp2 = rem - hy * (CI_HWP / 2);
            const int hin = h0 + hy, win = w0 + 2 * hp2;    // the input is pre-padded: output (h, w) reads input rows h .. h + 6
            if (hin < p.Hp && win < p.Wp)                   // Wp is even (host check): a pair never straddles the row end
                off = (uint32_t)(((((int64_t)b * p.Tp + (t + fr)) * p.Hp + hin) * p.Wp + win) * 8);
        }
        hv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0);
    }
    constexpr int WLD = CI_SLICE / 16 / 256;                // 7 chunks of a weight slice per thread
    i32x4 wv[WLD];
    auto load_slice = [&](int it) {
#pragma unroll
        for (int i = 0; i < WLD; ++i) wv[i] = *(const i32x4*)((const char*)p.wfrag + (size_t)it * CI_SLICE + (size_t)(tid + 256 * i) * 16);
    };
    auto write_slice = [&]() {
#pragma unroll
        for (int i = 0; i < WLD; ++i) *(i32x4*)(smem + CI_W + (tid + 256 * i) * 16) = wv[i];
    };
    load_slice(0);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        if (idx < total) {
            const int fr = idx / CPF, rem = idx - fr * CPF;
            const int hy = rem / (CI_HWP / 2), hp2 = rem - hy * (CI_HWP / 2);
            *(i32x4*)(smem + fr * CI_FRAME + hy * CI_ROWB + hp2 * 16) = hv[i];
        }
    }
    write_slice();
    __syncthreads();

    // ---- K loop.  B fragment of tile row r, tap row (it, ih), step s: lane (x = l31, pixel pair hi) reads pixels x + 4 s + 2 hi, + 1
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][q][r] = 0.f;
    const uint32_t pb = (uint32_t)(wave * 2) * CI_ROWB + (uint32_t)(l31 + 2 * hi) * 8u;
    const uint32_t wb = (uint32_t)CI_W + (uint32_t)lane * 16u;
    for (int it = 0; it < p.kt; ++it) {
        if (it + 1 < p.kt) load_slice(it + 1);              // lands under this slice's MFMAs
        const uint32_t fb = pb + (uint32_t)it * CI_FRAME;
#pragma unroll
        for (int st = 0; st < CI_STEPS; ++st) {
            const int ih = st >> 1, s = st & 1;
            const i32x4 a0 = *(const i32x4*)(smem + wb + (st * 2 + 0) * 1024), a1 = *(const i32x4*)(smem + wb + (st * 2 + 1) * 1024);
            i32x4 bq[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const char* src = smem + fb + (uint32_t)((q + ih) * CI_ROWB + s * 32);
                const i32x2 lo = *(const i32x2*)src, hi2 = *(const i32x2*)(src + 8);
                bq[q] = i32x4{lo[0], lo[1], hi2[0], hi2[1]};
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) { Mma<BF16>::run(a0, bq[q], acc[0][q]); Mma<BF16>::run(a1, bq[q], acc[1][q]); }
        }
        __syncthreads();                                    // every wave is done with this slice
        if (it + 1 < p.kt) { write_slice(); __syncthreads(); }
    }

    // ---- epilogue: + bias -> bf16, a wave-private [64 positions][64 channels] tile in the (dead) halo / weight regions, whole 128-byte rows out.
    // acc[a][q][4 g + j] = channel a * 32 + 8 g + 4 hi + j of position l31 of tile row 2 wave + q
    char* priv = smem + wave * (64 * CI_PITCH);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = a * 32 + 8 * g + 4 * hi;
            const f32x4 bv = p.bias ? *(const f32x4*)(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[a][q][4 * g + j] + bv[j];
                lds_put4<BF16>(priv + (q * 32 + l31) * CI_PITCH + col * 2, v);
            }
        }
    __builtin_amdgcn_wave_barrier();
    {
        const int ch = lane & 7, rr2 = lane >> 3;           // 8 lanes per 128-byte row, 8 rows per instruction
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rr2 + 8 * i, q = row >> 5, x = row & 31;
            const int h = h0 + wave * 2 + q, w = w0 + x;
            if (h < p.Ho && w < p.Wo) {
                const int64_t m = (((int64_t)b * p.To + t) * p.Ho + h) * p.Wo + w;
                *(i32x4*)((char*)p.out + m * 128 + ch * 16) = *(const i32x4*)(priv + row * CI_PITCH + ch * 16);
            }
        }
    }
}

inline bool conv_in_ok(const mv2_conv_desc& d) {
    // the descriptor mv2_conv takes for conv_in over the pre-padded clip (pack.py::pack_conv_in: one tap per (it, ih), kw x 4 channels
    // contiguous): kh = 7 rows of kw = 1 "wide" taps, Ci = 4, no bounds checks, input extents = output extents + 6
    return d.dtype == MV2_BF16 && d.Ci == 4 && d.co == 64 && d.out_c == 64 && d.kh == CI_K && d.kw == 1 && d.kt >= 1 && d.kt <= 7 && d.st == 1 &&
           d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0 && d.check_bounds == 0 && d.Ti == d.To + d.kt - 1 && d.Hi == d.Ho + CI_K - 1 &&
           d.Wi == d.Wo + CI_K - 1 && (d.Wi & 1) == 0 && d.store == MV2_STORE_ROWS && d.act == MV2_ACT_NONE && d.B > 0 && d.To > 0;
}

}  // namespace

extern "C" int mv2_conv_in_eligible(const mv2_conv_desc* d) { return d && conv_in_ok(*d) ? 1 : 0; }

// bytes of the weight image mv2_conv_in reads: [kt slices][14 steps][2 channel tiles][64 lanes][8 bf16]
extern "C" size_t mv2_conv_in_weight_bytes(int kt) { return kt >= 1 && kt <= 7 ? (size_t)kt * CI_SLICE : 0; }

extern "C" int mv2_conv_in(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wfrag, const float* bias, void* out,
                           void* stream) {
    if (!d || !in || !wfrag || !out) return MV2_EINVAL;
    if (!conv_in_ok(*d)) return MV2_EINVAL;
    if (in_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    if ((((uintptr_t)in | (uintptr_t)wfrag | (uintptr_t)out) & 15) || (bias && ((uintptr_t)bias & 15))) return MV2_EINVAL;
    ConvInParams p;
    p.in = in; p.wfrag = wfrag; p.bias = bias; p.out = out; p.in_bytes = (uint32_t)in_bytes;
    p.B = d->B; p.To = d->To; p.Ho = d->Ho; p.Wo = d->Wo; p.Tp = d->Ti; p.Hp = d->Hi; p.Wp = d->Wi; p.kt = d->kt;
    p.tiles_h = (d->Ho + CI_TH - 1) / CI_TH; p.tiles_w = (d->Wo + CI_TW - 1) / CI_TW;
    const int64_t blocks = (int64_t)d->B * d->To * p.tiles_h * p.tiles_w;
    if (blocks <= 0 || blocks >= (int64_t)0x7fffffff) return MV2_ETOOLARGE;
    if (hipFuncSetAttribute((const void*)conv_in_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CI_LDS) != hipSuccess) return MV2_ENODEVICE;
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)blocks), dim3(256), CI_LDS, (hipStream_t)stream, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
