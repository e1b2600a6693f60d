// mv2_conv_out — the decoder's output convolution (reference magvit2_pytorch.py:1127, 1642: CausalConv3d(dim -> channels, 3 x 3 x 3),
// zero padding): 64 input channels, <= 4 output channels, fp32 NCTHW store with the leading time-padding frames cropped.
//
// Why its own kernel: with N = 3 this is not MFMA work (3.4 GF per clip) but an HBM stream — 42 MB in, 3.3 MB out — and through the
// generic gather kernel every input row is fetched once per TAP: 1280 workgroups x 27 steps x 36 KB = 1.24 GB through the L1s, 80 us
// per clip on MI355X (9 % of the HBM roofline, profiles/r02).  Here a workgroup owns an 8 x 16 tile of ONE output frame: the kt input
// frames of its 10 x 18 halo go to LDS once (4.2 x the tile bytes instead of 27 x), all taps are immediate offsets into that image.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16 with A = weights (rows = output channel, 3-4 of 16 used), B = 16 positions of one tile row;
// a k-step is 32 of the 64 channels of one tap.  The four waves split the k-steps of a tile (K = kt * 9 * 2 = 54 steps; every wave keeps its
// <= 14 weight fragments in registers for the whole launch: no LDS or L1 traffic for weights), and meet once per tile through LDS.
// Issue order per wave per k-step: 8 ds_read_b128 (the 8 tile rows) + 8 MFMAs.
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/mv2.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int CO_TH = 8, CO_TW = 16;                      // output tile
constexpr int CO_HW = CO_TW + 2, CO_HH = CO_TH + 2;       // halo 10 x 18
constexpr int CO_PX = CO_HH * CO_HW;                      // 180 halo pixels per frame
constexpr int CO_PITCH = 144;                             // bytes per halo pixel: 64 bf16 + 16 pad
constexpr int CO_FRAME = ((CO_PX * CO_PITCH + 255) / 256) * 256;
constexpr int CO_MAXKS = 14;                              // k-steps per wave: ceil(54 / 4)

struct ConvOutParams {
    const void* in; const void* wfrag; const float* bias; float* out;
    uint32_t in_bytes;
    int B, T, H, W, co, kt, crop_t, tiles_h, tiles_w;
};

__global__ __launch_bounds__(256, 2) void conv_out_kernel(const ConvOutParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nks = p.kt * 18;                                                   // k-steps: tap-major, two channel halves per tap
    // ---- this wave's weight fragments (k-steps wave, wave + 4, ...): lane l holds A[co = l % 16][8 (l / 16) .. + 8) of the step
    i32x4 wf[CO_MAXKS];
#pragma unroll
    for (int i = 0; i < CO_MAXKS; ++i) {
        const int ks = wave + 4 * i;
        wf[i] = ks < nks ? *(const i32x4*)((const char*)p.wfrag + ((size_t)ks * 64 + lane) * 16) : i32x4{0, 0, 0, 0};
    }
    // ---- tile: blockIdx -> (b, output frame, tile row, tile col); XCD-aware: the eight XCDs take contiguous eighths of the list, so the
    // workgroups that share halo frames (neighbouring tiles, consecutive frames) meet in one L2
    const int Tout = p.T - p.crop_t;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t NW = gridDim.x, q = NW >> 3, r = NW & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int twi = (int)(bid % (uint32_t)p.tiles_w); uint32_t rr = bid / (uint32_t)p.tiles_w;
    const int thi = (int)(rr % (uint32_t)p.tiles_h); rr /= (uint32_t)p.tiles_h;
    const int to = (int)(rr % (uint32_t)Tout), b = (int)(rr / (uint32_t)Tout);
    const int t = to + p.crop_t, h0 = thi * CO_TH, w0 = twi * CO_TW;

    // ---- halo: kt frames x 180 pixels x 8 chunks of 16 bytes, every load issued before the first wait
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    constexpr int NCH = CO_PX * 8;                                               // chunks per frame
    constexpr int NLD = (3 * NCH + 255) / 256;                                   // 17 loads per thread at kt = 3
    const int total = p.kt * NCH;
    i32x4 hv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        uint32_t off = p.in_bytes;                                               // out of range -> zeros (padding, causal front, tail)
        if (idx < total) {
            const int fr = idx / NCH, rem = idx - fr * NCH;
            const int px = rem >> 3, ch = rem & 7;
            const int hy = px / CO_HW, hx = px - hy * CO_HW;
            const int t_in = t + fr - (p.kt - 1), hin = h0 - 1 + hy, win = w0 - 1 + hx;
            if (t_in >= 0 && (unsigned)hin < (unsigned)p.H && (unsigned)win < (unsigned)p.W)
                off = (uint32_t)(((((int64_t)b * p.T + t_in) * p.H + hin) * p.W + win) * 128 + ch * 16);
        }
        hv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        if (idx < total) {
            const int fr = idx / NCH, rem = idx - fr * NCH;
            *(i32x4*)(smem + fr * CO_FRAME + (rem >> 3) * CO_PITCH + (rem & 7) * 16) = hv[i];
        }
    }
    __syncthreads();

    // ---- taps.  B operand of tile row r, tap (it, ih, iw), half hf: lane l reads 16 bytes of pixel (r + ih, l % 16 + iw) at channel
    // hf * 32 + 8 (l / 16): one per-lane base + compile-time offsets
    f32x4_t acc[CO_TH];
#pragma unroll
    for (int r = 0; r < CO_TH; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t lbase = (uint32_t)(lane & 15) * CO_PITCH + (uint32_t)(lane >> 4) * 16u;
#pragma unroll
    for (int i = 0; i < CO_MAXKS; ++i) {
        const int ks = wave + 4 * i;                                             // wave-uniform
        if (ks < nks) {
            const int tap = ks >> 1, hf = ks & 1;
            const int it = tap / 9, ihw = tap - it * 9, ih = ihw / 3, iw = ihw - ih * 3;
            const uint32_t o = lbase + (uint32_t)(it * CO_FRAME + (ih * CO_HW + iw) * CO_PITCH + hf * 64);
            i32x4 bf[CO_TH];
#pragma unroll
            for (int r = 0; r < CO_TH; ++r) bf[r] = *(const i32x4*)(smem + o + r * CO_HW * CO_PITCH);
#pragma unroll
            for (int r = 0; r < CO_TH; ++r)
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i]), __builtin_bit_cast(bf16x8, bf[r]), acc[r], 0, 0, 0);
        }
    }
    __syncthreads();                                                             // the halo is dead: LDS becomes the reduce buffer

    // ---- cross-wave sum (fixed order) + bias, store.  D[co = 4 (l / 16) + i][pos = l % 16]: lanes 0..15 hold co 0..3 of their position
    float* red = (float*)smem;                                                   // [wave][row][16 positions][4]
    if (lane < 16) {
#pragma unroll
        for (int r = 0; r < CO_TH; ++r) *(f32x4_t*)(red + ((wave * CO_TH + r) * 16 + lane) * 4) = acc[r];
    }
    __syncthreads();
    if (tid < CO_TH * CO_TW) {
        const int r = tid >> 4, x = tid & 15;
        f32x4_t s = *(const f32x4_t*)(red + (r * 16 + x) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) { const f32x4_t v = *(const f32x4_t*)(red + ((w * CO_TH + r) * 16 + x) * 4); s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
        const int h = h0 + r, w_ = w0 + x;
        if (h < p.H && w_ < p.W) {
            const int64_t plane = (int64_t)p.H * p.W;
            float* dst = p.out + (((int64_t)b * p.co) * Tout + to) * plane + (int64_t)h * p.W + w_;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < p.co) dst[(int64_t)c * Tout * plane] = s[c] + (p.bias ? p.bias[c] : 0.f);
        }
    }
}

inline bool conv_out_ok(const mv2_conv_desc& d) {
    return d.dtype == MV2_BF16 && d.Ci == 64 && d.co >= 1 && d.co <= 4 && d.kh == 3 && d.kw == 3 && d.kt >= 1 && d.kt <= 3 && d.st == 1 &&
           d.sh == 1 && d.sw == 1 && d.pt == d.kt - 1 && d.ph == 1 && d.pw == 1 && d.check_bounds == 1 && d.To == d.Ti && d.Ho == d.Hi &&
           d.Wo == d.Wi && d.store == MV2_STORE_NCTHW_F32 && d.act == MV2_ACT_NONE && d.crop_t >= 0 && d.crop_t < d.To && d.B > 0;
}

}  // namespace

extern "C" int mv2_conv_out_eligible(const mv2_conv_desc* d) { return d && conv_out_ok(*d) ? 1 : 0; }

// bytes of the weight image mv2_conv_out reads: [kt * 18 k-steps][64 lanes][8 bf16]
extern "C" size_t mv2_conv_out_weight_bytes(int kt) { return kt >= 1 && kt <= 3 ? (size_t)kt * 18 * 64 * 16 : 0; }

extern "C" int mv2_conv_out(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wfrag, const float* bias, void* out,
                            void* stream) {
    if (!d || !in || !wfrag || !out) return MV2_EINVAL;
    if (!conv_out_ok(*d)) return MV2_EINVAL;
    if (in_bytes >= 0xfffffff0ull) return MV2_ETOOLARGE;
    if ((((uintptr_t)in | (uintptr_t)wfrag) & 15) || ((uintptr_t)out & 3)) return MV2_EINVAL;
    ConvOutParams p;
    p.in = in; p.wfrag = wfrag; p.bias = bias; p.out = (float*)out; p.in_bytes = (uint32_t)in_bytes;
    p.B = d->B; p.T = d->Ti; p.H = d->Hi; p.W = d->Wi; p.co = d->co; p.kt = d->kt; p.crop_t = d->crop_t;
    p.tiles_h = (d->Hi + CO_TH - 1) / CO_TH; p.tiles_w = (d->Wi + CO_TW - 1) / CO_TW;
    const int64_t blocks = (int64_t)d->B * (d->Ti - d->crop_t) * p.tiles_h * p.tiles_w;
    if (blocks <= 0 || blocks >= (int64_t)0x7fffffff) return MV2_ETOOLARGE;
    const int lds = 3 * CO_FRAME;
    if (hipFuncSetAttribute((const void*)conv_out_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MV2_ENODEVICE;
    hipLaunchKernelGGL(conv_out_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    MV2_LAUNCH_CHECK();
    return MV2_OK;
}
