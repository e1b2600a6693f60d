// tools/experiments/ds2_conv_bench.cpp — PROTOTYPE, not part of the library: the stride-2 3 x 3 convolutions of compress_space
// (reference magvit2_pytorch.py SpatialDownsample2x: Conv2d(dim, dim_out, 3, stride = 2, padding = 1) per frame) as a HALO-STAGED kernel,
// the recipe that gave conv_in 1.75 x and conv_out 2.9 x.  Through the generic gather kernel (mv2_conv) the three launches of a clip take
// 36.9 / 29.3 / 30.6 us at B = 1 (profiles/r03_per_launch_events_b1.json: encoder_layers.1 / .3 / .6) for 12 GF each = 330-410 TF/s.
//
//   out[f][ho][wo][co] = sum_{ih, iw, ci} x[f][2 ho - 1 + ih][2 wo - 1 + iw][ci] w[co][ih][iw][ci]        (zero padding)
//
// Workgroup = TH x TW output positions (64) of one frame x 128 output channels, 4 waves = 2 (64 channels) x 2 (32 positions).
// Per 64-channel chunk of the input the (2 TH + 1) x (2 TW + 1) halo goes to LDS once, split into an EVEN-column and an ODD-column plane
// so that the 16 / 32 positions of a B fragment are consecutive entries of a plane row (pitch 144: the conflict-free pattern of conv3 under
// the measured ds_read_b128 pass partition, profiles/r03_findings.md section 8); the nine taps are immediate offsets.  Weights stream
// global -> VGPR in fragment-major order (one tap ahead), as in conv3 / pw_se.  The next chunk's halo is requested before the taps of the
// current one.  Epilogue: bf16 tile through LDS, 16-byte row stores (no bias / activation: this is a K-loop prototype).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/experiments/ds2_conv_bench.cpp -o tools/ds2_conv_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct P {
    const void* x; const void* wfrag; void* out;
    uint32_t xbytes;
    int F, Hi, Wi, Ci, Co, Ho, Wo, tiles_h, tiles_w, nchunk;     // F = B * T frames; nchunk = Ci / 64
};

__device__ __forceinline__ unsigned short f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); }

constexpr int PITCH = 144;

template <int TH, int TW> struct Geo {
    static constexpr int HR = 2 * TH + 1, HC = 2 * TW + 1;       // halo rows / columns
    static constexpr int NE = TW + 1, NO = TW;                    // entries per row of the even / odd column plane
    // TW = 16: a B fragment spans two tile rows (halo rows two apart); 24 entries per plane row keep the two half-fragments on complementary
    // bank groups (brute force against the measured pass partition); TW = 32: one row per fragment, any row length works
    static constexpr int RE = TW == 16 ? 24 : NE, RO = TW == 16 ? 24 : NO;
    static constexpr int PLANE_E = HR * RE * PITCH, PLANE_O = HR * RO * PITCH;
    static constexpr int SLAB = ((PLANE_E + PLANE_O + 255) / 256) * 256;
    static constexpr int NPX = HR * HC;
    static constexpr int NLD = (NPX * 8 + 255) / 256;
    static constexpr int EPI = 64 * (128 * 2 + 16);
    static constexpr int LDS = SLAB > EPI ? SLAB : EPI;
};

// weight image: [chunk][tap 9][co tile of 32 (Co / 32)][k-slice 4][64 lanes][16 B]: lane l of (tap, ct, ks) holds w[co = 32 ct + l % 32][tap][ci = 64 chunk + 16 ks + 8 (l / 32) .. + 8)
// ABL (timing only, wrong results): 1 = no weight loads (fragments stay what tap 0 loaded), 2 = no halo loads after the first slab
template <int TH, int TW, int ABL = 0, int WD = 1>
__global__ __launch_bounds__(256, 2) void ds2_kernel(const P p) {
    typedef Geo<TH, TW> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;                     // co half (64 channels), position half (32 positions)
    // block -> (co tile of 128, frame, tile row, tile col); co tile innermost: the workgroups that share a halo are neighbours
    const int nct = p.Co / 128;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t NW = gridDim.x, q = NW >> 3, r = NW & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int ct128 = (int)(bid % (uint32_t)nct); uint32_t rr = bid / (uint32_t)nct;
    const int twi = (int)(rr % (uint32_t)p.tiles_w); rr /= (uint32_t)p.tiles_w;
    const int thi = (int)(rr % (uint32_t)p.tiles_h); const int f = (int)(rr / (uint32_t)p.tiles_h);
    const int ho0 = thi * TH, wo0 = twi * TW;

    // ---- halo staging: thread's chunks (pixel, 16-byte chunk) of a 64-channel slab; source offset without the channel-chunk term
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    uint32_t s_off[G::NLD], l_off[G::NLD];
#pragma unroll
    for (int i = 0; i < G::NLD; ++i) {
        const int idx = tid + 256 * i, px = idx >> 3, ch = idx & 7;
        s_off[i] = p.xbytes; l_off[i] = 0xffffffffu;
        if (px < G::NPX) {
            const int hr = px / G::HC, hc = px - hr * G::HC;
            const int hin = 2 * ho0 - 1 + hr, win = 2 * wo0 - 1 + hc;
            if ((unsigned)hin < (unsigned)p.Hi && (unsigned)win < (unsigned)p.Wi)
                s_off[i] = (uint32_t)((((int64_t)f * p.Hi + hin) * p.Wi + win) * p.Ci * 2 + ch * 16);
            l_off[i] = (hc & 1) ? (uint32_t)(G::PLANE_E + (hr * G::RO + (hc >> 1)) * PITCH + ch * 16) : (uint32_t)((hr * G::RE + (hc >> 1)) * PITCH + ch * 16);
        }
    }
    auto slab_loads = [&](int chunk, i32x4 (&v)[G::NLD]) {
        const bool live = chunk < p.nchunk;
#pragma unroll
        for (int i = 0; i < G::NLD; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(live ? s_off[i] : p.xbytes), live ? chunk * 128 : 0, 0);
    };
    auto slab_store = [&](const i32x4 (&v)[G::NLD]) {
#pragma unroll
        for (int i = 0; i < G::NLD; ++i)
            if (l_off[i] != 0xffffffffu) *(i32x4*)(smem + l_off[i]) = v[i];
    };

    // ---- B fragment base of this lane: position = 32 wp + l31 -> (row r, column x) of the tile
    const int pos = 32 * wp + l31, r = pos / TW, x = pos - r * TW;
    const uint32_t b_even = (uint32_t)((2 * r * G::RE + x) * PITCH + hi * 16);                 // taps iw = 0 (entry x), iw = 2 (entry x + 1)
    const uint32_t b_odd = (uint32_t)(G::PLANE_E + (2 * r * G::RO + x) * PITCH + hi * 16);     // tap iw = 1

    // ---- weights: wave's two co tiles (64 channels), one tap = 4 k-slices x 2 co tiles = 8 fragments
    const int ct0 = ct128 * 4 + wc * 2;
    const char* wbase = (const char*)p.wfrag + (size_t)lane * 16;
    auto wfrag = [&](int chunk, int tap, int ks, int a) {
        const size_t blk = (((size_t)chunk * 9 + tap) * (p.Co / 32) + ct0 + a) * 4 + ks;
        return *(const i32x4*)(wbase + blk * 1024);
    };

    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;

    i32x4 hv[G::NLD];
    slab_loads(0, hv);
    i32x4 wf[WD + 1][4][2];                                        // [set][k-slice][co tile]; WD = taps of prefetch distance (WD = 2: three sets, 9 taps per slab keep the set of tap t at t % 3)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int a = 0; a < 2; ++a) { wf[0][ks][a] = wfrag(0, 0, ks, a); if constexpr (WD == 2) wf[1][ks][a] = wfrag(0, 1, ks, a); }
    slab_store(hv);
    __syncthreads();

    // nine taps per slab: the register set of a slab's first tap alternates from slab to slab (PAR), so the slab loop is unrolled by two
    auto slab = [&](int chunk, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        if (ABL != 2) slab_loads(chunk + 1, hv);                   // next slab under this slab's taps (past the end: out of range, zeros, no traffic)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ih = tap / 3, iw = tap - ih * 3;
            // the weights WD taps ahead (the next slab's first taps behind tap 8 / 7; behind the last slab: a harmless re-read of slab 0)
            const int nt_ = tap + WD, ntap = nt_ >= 9 ? nt_ - 9 : nt_, nchunk_ = nt_ >= 9 ? (chunk + 1 < p.nchunk ? chunk + 1 : 0) : chunk;
            constexpr int dummy_ = 0; (void)dummy_;
            const int wset = WD == 2 ? (tap + 2) % 3 : (tap + 1 + PAR) & 1, rset = WD == 2 ? tap % 3 : (tap + PAR) & 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int a = 0; a < 2; ++a) wf[wset][ks][a] = ABL == 1 ? wf[rset][ks][a] : wfrag(nchunk_, ntap, ks, a);
            const uint32_t o = (iw == 1 ? b_odd + (uint32_t)(ih * G::RO * PITCH) : b_even + (uint32_t)((ih * G::RE + (iw >> 1)) * PITCH));
            i32x4 bf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bf[ks] = *(const i32x4*)(smem + o + ks * 32);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[rset][ks][a]), __builtin_bit_cast(bf16x8, bf[ks]), acc[a], 0, 0, 0);
        }
        __syncthreads();                                           // every wave has read the slab
        if (chunk + 1 < p.nchunk) {
            slab_store(hv);
            __syncthreads();
        }
    };
    for (int chunk = 0; chunk < p.nchunk; chunk += 2) {
        slab(chunk, std::integral_constant<int, 0>{});
        if (chunk + 1 < p.nchunk) slab(chunk + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: acc[a][4 g + j] = out[pos = 32 wp + l31][co = 64 wc + 32 a + 8 g + 4 hi + j]
    constexpr int EP = 128 * 2 + 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 64 * wc + 32 * a + 8 * g + 4 * hi;
            i32x2 v;
            v[0] = (int)((unsigned)f2bf(acc[a][4 * g]) | ((unsigned)f2bf(acc[a][4 * g + 1]) << 16));
            v[1] = (int)((unsigned)f2bf(acc[a][4 * g + 2]) | ((unsigned)f2bf(acc[a][4 * g + 3]) << 16));
            *(i32x2*)(smem + pos * EP + col * 2) = v;
        }
    __syncthreads();
    for (int q = tid; q < 64 * 16; q += 256) {
        const int ps = q >> 4, chn = q & 15, rr_ = ps / TW, xx = ps - rr_ * TW;
        const int ho = ho0 + rr_, wo = wo0 + xx;
        if (ho < p.Ho && wo < p.Wo)
            *(i32x4*)((char*)p.out + ((((int64_t)f * p.Ho + ho) * p.Wo + wo) * p.Co + ct128 * 128 + chn * 8) * 2) = *(const i32x4*)(smem + ps * EP + chn * 16);
    }
}

// ---- variant W: 128 positions (4 x 32) per workgroup, the halo staged in HALF-slabs of 32 channels (64-byte pixels, pitch 80: conflict-free
// for 32-lane rows under the measured pass partition), so that the slab still fits twice per CU: half the weight bytes per FLOP of the
// 64-position form.  Waves 2 (64 channels) x 2 (64 positions = two tile rows); acc[2][2].
// weight image: [half-slab][tap 9][co tile of 32][k-slice 2][64 lanes][16 B]: lane l holds w[co = 32 ct + l % 32][tap][ci = 32 hs + 16 ks + 8 (l / 32) .. + 8)
struct GeoW {
    static constexpr int TH = 4, TW = 32, HR = 9, HC = 65, RE = 33, RO = 32, PW = 80;
    static constexpr int PLANE_E = HR * RE * PW, PLANE_O = HR * RO * PW;
    static constexpr int SLAB = ((PLANE_E + PLANE_O + 255) / 256) * 256;
    static constexpr int NPX = HR * HC, NLD = (NPX * 4 + 255) / 256;
    static constexpr int EPI = 128 * (128 * 2 + 16);
    static constexpr int LDS = SLAB > EPI ? SLAB : EPI;
};

__global__ __launch_bounds__(256, 2) void ds2w_kernel(const P p) {
    typedef GeoW G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;
    const int nct = p.Co / 128, nhs = p.Ci / 32;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t NW = gridDim.x, q = NW >> 3, r = NW & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int ct128 = (int)(bid % (uint32_t)nct); uint32_t rr = bid / (uint32_t)nct;
    const int twi = (int)(rr % (uint32_t)p.tiles_w); rr /= (uint32_t)p.tiles_w;
    const int thi = (int)(rr % (uint32_t)p.tiles_h); const int f = (int)(rr / (uint32_t)p.tiles_h);
    const int ho0 = thi * G::TH, wo0 = twi * G::TW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    uint32_t s_off[G::NLD], l_off[G::NLD];
#pragma unroll
    for (int i = 0; i < G::NLD; ++i) {
        const int idx = tid + 256 * i, px = idx >> 2, ch = idx & 3;
        s_off[i] = p.xbytes; l_off[i] = 0xffffffffu;
        if (px < G::NPX) {
            const int hr = px / G::HC, hc = px - hr * G::HC;
            const int hin = 2 * ho0 - 1 + hr, win = 2 * wo0 - 1 + hc;
            if ((unsigned)hin < (unsigned)p.Hi && (unsigned)win < (unsigned)p.Wi)
                s_off[i] = (uint32_t)((((int64_t)f * p.Hi + hin) * p.Wi + win) * p.Ci * 2 + ch * 16);
            l_off[i] = (hc & 1) ? (uint32_t)(G::PLANE_E + (hr * G::RO + (hc >> 1)) * G::PW + ch * 16) : (uint32_t)((hr * G::RE + (hc >> 1)) * G::PW + ch * 16);
        }
    }
    auto slab_loads = [&](int hs, i32x4 (&v)[G::NLD]) {
        const bool live = hs < nhs;
#pragma unroll
        for (int i = 0; i < G::NLD; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(live ? s_off[i] : p.xbytes), live ? hs * 64 : 0, 0);
    };
    auto slab_store = [&](const i32x4 (&v)[G::NLD]) {
#pragma unroll
        for (int i = 0; i < G::NLD; ++i)
            if (l_off[i] != 0xffffffffu) *(i32x4*)(smem + l_off[i]) = v[i];
    };
    // B fragments: tile b of this wave = tile row 2 wp + b, column l31
    const uint32_t b_even = (uint32_t)((2 * (2 * wp) * G::RE + l31) * G::PW + hi * 16);
    const uint32_t b_odd = (uint32_t)(G::PLANE_E + (2 * (2 * wp) * G::RO + l31) * G::PW + hi * 16);
    const int ct0 = ct128 * 4 + wc * 2;
    const char* wbase = (const char*)p.wfrag + (size_t)lane * 16;
    auto wfrag = [&](int hs, int tap, int ks, int a) {
        const size_t blk = (((size_t)hs * 9 + tap) * (p.Co / 32) + ct0 + a) * 2 + ks;
        return *(const i32x4*)(wbase + blk * 1024);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    i32x4 hv[G::NLD];
    slab_loads(0, hv);
    i32x4 wf[3][2][2];                                             // three sets: two taps ahead; 9 taps per half-slab keep tap t in set t % 3
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int a = 0; a < 2; ++a) { wf[0][ks][a] = wfrag(0, 0, ks, a); wf[1][ks][a] = wfrag(0, 1, ks, a); }
    slab_store(hv);
    __syncthreads();
    for (int hs = 0; hs < nhs; ++hs) {
        slab_loads(hs + 1, hv);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ih = tap / 3, iw = tap - ih * 3;
            const int nt_ = tap + 2, ntap = nt_ >= 9 ? nt_ - 9 : nt_, nhs_ = nt_ >= 9 ? (hs + 1 < nhs ? hs + 1 : 0) : hs;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < 2; ++a) wf[(tap + 2) % 3][ks][a] = wfrag(nhs_, ntap, ks, a);
            i32x4 bf[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const uint32_t o = (iw == 1 ? b_odd + (uint32_t)((2 * b + ih) * G::RO * G::PW) : b_even + (uint32_t)(((2 * b + ih) * G::RE + (iw >> 1)) * G::PW));
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bf[ks][b] = *(const i32x4*)(smem + o + ks * 32);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[tap % 3][ks][a]), __builtin_bit_cast(bf16x8, bf[ks][b]), acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (hs + 1 < nhs) {
            slab_store(hv);
            __syncthreads();
        }
    }
    constexpr int EP = 128 * 2 + 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 64 * wc + 32 * a + 8 * g + 4 * hi, pos = (2 * wp + b) * 32 + l31;
                i32x2 v;
                v[0] = (int)((unsigned)f2bf(acc[a][b][4 * g]) | ((unsigned)f2bf(acc[a][b][4 * g + 1]) << 16));
                v[1] = (int)((unsigned)f2bf(acc[a][b][4 * g + 2]) | ((unsigned)f2bf(acc[a][b][4 * g + 3]) << 16));
                *(i32x2*)(smem + pos * EP + col * 2) = v;
            }
    __syncthreads();
    for (int q = tid; q < 128 * 16; q += 256) {
        const int ps = q >> 4, chn = q & 15, rr_ = ps >> 5, xx = ps & 31;
        const int ho = ho0 + rr_, wo = wo0 + xx;
        if (ho < p.Ho && wo < p.Wo)
            *(i32x4*)((char*)p.out + ((((int64_t)f * p.Ho + ho) * p.Wo + wo) * p.Co + ct128 * 128 + chn * 8) * 2) = *(const i32x4*)(smem + ps * EP + chn * 16);
    }
}

static uint32_t rng = 12345u;
static inline float frand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f; }
static inline uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float h_bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int TH, int TW, int ABL = 0, int WD = 1>
static void run(const char* name, int F, int Hi, int Ci, int Co, int reps) {
    typedef Geo<TH, TW> G;
    const int Wi = Hi, Ho = Hi / 2, Wo = Wi / 2, nchunk = Ci / 64;
    std::vector<uint16_t> x((size_t)F * Hi * Wi * Ci), w((size_t)Co * 9 * Ci), out((size_t)F * Ho * Wo * Co);
    for (auto& v : x) v = h_f2bf(frand());
    for (auto& v : w) v = h_f2bf(frand() * 0.05f);               // w[co][tap][ci]
    std::vector<uint16_t> wfr((size_t)nchunk * 9 * (Co / 32) * 4 * 64 * 8);
    for (int c = 0; c < nchunk; ++c)
        for (int tap = 0; tap < 9; ++tap)
            for (int ct = 0; ct < Co / 32; ++ct)
                for (int ks = 0; ks < 4; ++ks)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j)
                            wfr[((((((size_t)c * 9 + tap) * (Co / 32) + ct) * 4 + ks) * 64 + l) * 8) + j] =
                                w[((size_t)(32 * ct + l % 32) * 9 + tap) * Ci + 64 * c + 16 * ks + 8 * (l / 32) + j];
    void *dx, *dw, *dout;
    CK(hipMalloc(&dx, x.size() * 2)); CK(hipMalloc(&dw, wfr.size() * 2)); CK(hipMalloc(&dout, out.size() * 2));
    CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, wfr.data(), wfr.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, out.size() * 2));
    P p{dx, dw, dout, (uint32_t)(x.size() * 2), F, Hi, Wi, Ci, Co, Ho, Wo, (Ho + TH - 1) / TH, (Wo + TW - 1) / TW, nchunk};
    const int blocks = F * p.tiles_h * p.tiles_w * (Co / 128);
    CK(hipFuncSetAttribute((const void*)ds2_kernel<TH, TW, ABL, WD>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    hipLaunchKernelGGL((ds2_kernel<TH, TW, ABL, WD>), dim3(blocks), dim3(256), G::LDS, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0; double maxd = 0;
    for (int t = 0; t < 4096; ++t) {
        rng = rng * 1664525u + 1013904223u; const int f = (rng >> 4) % F;
        rng = rng * 1664525u + 1013904223u; const int ho = t < 64 ? (t & 1 ? Ho - 1 : 0) : (rng >> 4) % Ho;
        rng = rng * 1664525u + 1013904223u; const int wo = t < 64 ? (t & 2 ? Wo - 1 : 0) : (rng >> 4) % Wo;
        rng = rng * 1664525u + 1013904223u; const int co = (rng >> 4) % Co;
        double ref = 0;
        for (int ih = 0; ih < 3; ++ih)
            for (int iw = 0; iw < 3; ++iw) {
                const int hin = 2 * ho - 1 + ih, win = 2 * wo - 1 + iw;
                if (hin < 0 || hin >= Hi || win < 0 || win >= Wi) continue;
                for (int ci = 0; ci < Ci; ++ci)
                    ref += (double)h_bf2f(x[(((size_t)f * Hi + hin) * Wi + win) * Ci + ci]) * h_bf2f(w[((size_t)co * 9 + ih * 3 + iw) * Ci + ci]);
            }
        const double d = fabs(ref - h_bf2f(out[(((size_t)f * Ho + ho) * Wo + wo) * Co + co]));
        if (d > maxd) maxd = d;
        if (d > 0.03 + 0.01 * fabs(ref)) ++bad;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((ds2_kernel<TH, TW, ABL, WD>), dim3(blocks), dim3(256), G::LDS, 0, p);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ds2_kernel<TH, TW, ABL, WD>), dim3(blocks), dim3(256), G::LDS, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    printf("%-34s%s tile %dx%-2d F=%-3d %3dx%-3d Ci=%-3d Co=%-3d  %7.1f us  %6.0f TF/s  max|d| %.3e bad %d  (%d blocks, %d KB LDS)\n", name, ABL == 1 ? " [no weight loads]" : (ABL == 2 ? " [no halo reloads]" : (WD == 2 ? " [weights 2 taps ahead]" : "")), TH, TW, F, Hi, Wi, Ci, Co, us,
           2.0 * F * Ho * Wo * Co * 9.0 * Ci / us / 1e6, maxd, bad, blocks, G::LDS / 1024);
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dout));
}

static void run_wide(const char* name, int F, int Hi, int Ci, int Co, int reps) {
    typedef GeoW G;
    const int Wi = Hi, Ho = Hi / 2, Wo = Wi / 2, nhs = Ci / 32;
    std::vector<uint16_t> x((size_t)F * Hi * Wi * Ci), w((size_t)Co * 9 * Ci), out((size_t)F * Ho * Wo * Co);
    for (auto& v : x) v = h_f2bf(frand());
    for (auto& v : w) v = h_f2bf(frand() * 0.05f);
    std::vector<uint16_t> wfr((size_t)nhs * 9 * (Co / 32) * 2 * 64 * 8);
    for (int c = 0; c < nhs; ++c)
        for (int tap = 0; tap < 9; ++tap)
            for (int ct = 0; ct < Co / 32; ++ct)
                for (int ks = 0; ks < 2; ++ks)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j)
                            wfr[((((((size_t)c * 9 + tap) * (Co / 32) + ct) * 2 + ks) * 64 + l) * 8) + j] =
                                w[((size_t)(32 * ct + l % 32) * 9 + tap) * Ci + 32 * c + 16 * ks + 8 * (l / 32) + j];
    void *dx, *dw, *dout;
    CK(hipMalloc(&dx, x.size() * 2)); CK(hipMalloc(&dw, wfr.size() * 2)); CK(hipMalloc(&dout, out.size() * 2));
    CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, wfr.data(), wfr.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, out.size() * 2));
    P p{dx, dw, dout, (uint32_t)(x.size() * 2), F, Hi, Wi, Ci, Co, Ho, Wo, (Ho + G::TH - 1) / G::TH, (Wo + G::TW - 1) / G::TW, nhs};
    const int blocks = F * p.tiles_h * p.tiles_w * (Co / 128);
    CK(hipFuncSetAttribute((const void*)ds2w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    hipLaunchKernelGGL(ds2w_kernel, dim3(blocks), dim3(256), G::LDS, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0; double maxd = 0;
    for (int t = 0; t < 4096; ++t) {
        rng = rng * 1664525u + 1013904223u; const int f = (rng >> 4) % F;
        rng = rng * 1664525u + 1013904223u; const int ho = t < 64 ? (t & 1 ? Ho - 1 : 0) : (rng >> 4) % Ho;
        rng = rng * 1664525u + 1013904223u; const int wo = t < 64 ? (t & 2 ? Wo - 1 : 0) : (rng >> 4) % Wo;
        rng = rng * 1664525u + 1013904223u; const int co = (rng >> 4) % Co;
        double ref = 0;
        for (int ih = 0; ih < 3; ++ih)
            for (int iw = 0; iw < 3; ++iw) {
                const int hin = 2 * ho - 1 + ih, win = 2 * wo - 1 + iw;
                if (hin < 0 || hin >= Hi || win < 0 || win >= Wi) continue;
                for (int ci = 0; ci < Ci; ++ci)
                    ref += (double)h_bf2f(x[(((size_t)f * Hi + hin) * Wi + win) * Ci + ci]) * h_bf2f(w[((size_t)co * 9 + ih * 3 + iw) * Ci + ci]);
            }
        const double d = fabs(ref - h_bf2f(out[(((size_t)f * Ho + ho) * Wo + wo) * Co + co]));
        if (d > maxd) maxd = d;
        if (d > 0.03 + 0.01 * fabs(ref)) ++bad;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(ds2w_kernel, dim3(blocks), dim3(256), G::LDS, 0, p);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(ds2w_kernel, dim3(blocks), dim3(256), G::LDS, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    printf("%-34s [128 positions, half-slabs] tile 4x32 F=%-3d %3dx%-3d Ci=%-3d Co=%-3d  %7.1f us  %6.0f TF/s  max|d| %.3e bad %d  (%d blocks, %d KB LDS)\n", name, F, Hi, Wi, Ci, Co, us,
           2.0 * F * Ho * Wo * Co * 9.0 * Ci / us / 1e6, maxd, bad, blocks, G::LDS / 1024);
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dout));
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "wide")) {
        run_wide("enc.1 64->128 128^2 (36.9 us)", 20, 128, 64, 128, 30);
        run_wide("enc.3 128->256 64^2 (29.3 us)", 20, 64, 128, 256, 30);
        run_wide("enc.1 B=4", 80, 128, 64, 128, 10);
        run_wide("enc.3 B=4", 80, 64, 128, 256, 10);
        return 0;
    }
    // the compress_space launches of BASELINE configs[1] at B = 1 (20 frames) and B = 4 (80 frames); mv2_conv's times in the names
    run<2, 32>("enc.1 64->128 128^2 (36.9 us)", 20, 128, 64, 128, 30);
    run<4, 16>("enc.1 64->128 128^2 (36.9 us)", 20, 128, 64, 128, 30);
    run<2, 32>("enc.3 128->256 64^2 (29.3 us)", 20, 64, 128, 256, 30);
    run<4, 16>("enc.3 128->256 64^2 (29.3 us)", 20, 64, 128, 256, 30);
    run<4, 16>("enc.6 256->512 32^2 (30.6 us)", 20, 32, 256, 512, 30);
    run<2, 32>("enc.1 B=4", 80, 128, 64, 128, 10);
    run<4, 16>("enc.3 B=4", 80, 64, 128, 256, 10);
    run<4, 16>("enc.6 B=4", 80, 32, 256, 512, 10);
    run<4, 16, 0, 2>("enc.1", 20, 128, 64, 128, 30);
    run<4, 16, 0, 2>("enc.3", 20, 64, 128, 256, 30);
    run<4, 16, 0, 2>("enc.6", 20, 32, 256, 512, 30);
    run<4, 16, 0, 2>("enc.3 B=4", 80, 64, 128, 256, 10);
    // timing-only ablations: what bounds the prototype?
    run<4, 16, 1>("enc.1", 20, 128, 64, 128, 30);
    run<4, 16, 1>("enc.3", 20, 64, 128, 256, 30);
    run<4, 16, 1>("enc.6", 20, 32, 256, 512, 30);
    run<4, 16, 2>("enc.3", 20, 64, 128, 256, 30);
    run<4, 16, 2>("enc.6", 20, 32, 256, 512, 30);
    run<4, 16, 1>("enc.3 B=4", 80, 64, 128, 256, 10);
    return 0;
}
