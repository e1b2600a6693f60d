// Host-side weight packing behind the C ABI (include/mv2.h "weight images"): the byte-exact layouts mv2_conv / mv2_conv3 /
// mv2_pw_se read, produced from a dense fp32 convolution kernel in the reference's state_dict order (Co, Ci, kt, kh, kw)
// (reference nn.Conv3d weights, e.g. magvit2_pytorch.py:894-922).  Plain C++ on host pointers: a non-Python host (the FFI
// a maintainer would bind) needs no other code to feed the kernels.  magvit2_pytorch_amd/pack.py builds the
// same bytes (tests/test_host_logic.py::test_c_packer_matches_python_packer compares them).
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/mv2.h"

namespace {
inline uint16_t f2bf_rne(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
struct Geo { int es, seg_bytes, per_step, segs_per_tap, nsteps, co_pad, k_elems, ci_pad; };
bool geo(int dtype, int co, int ci, int kt, int kh, int kw, Geo& g) {
    if ((dtype != MV2_BF16 && dtype != MV2_F32) || co <= 0 || ci <= 0 || kt <= 0 || kh <= 0 || kw <= 0) return false;
    g.es = dtype == MV2_BF16 ? 2 : 4;
    const int kvec = 16 / g.es;
    g.ci_pad = (ci + kvec - 1) / kvec * kvec;
    const int tap_bytes = g.ci_pad * g.es;
    g.seg_bytes = gcd_i(128, tap_bytes);
    g.per_step = 128 / g.seg_bytes;
    g.segs_per_tap = tap_bytes / g.seg_bytes;
    const int64_t nsegs = (int64_t)kt * kh * kw * g.segs_per_tap;
    g.nsteps = (int)((nsegs + g.per_step - 1) / g.per_step);
    g.co_pad = (co + 127) / 128 * 128;
    g.k_elems = 128 / g.es;
    return true;
}
}  // namespace

extern "C" size_t mv2_pack_conv3d_bytes(int dtype, int co, int ci, int kt, int kh, int kw) {
    Geo g;
    if (!geo(dtype, co, ci, kt, kh, kw, g)) return 0;
    return (size_t)g.nsteps * g.co_pad * 128;
}

// K order: taps (it, ih, iw) row-major, inside a tap the input channels (zero-padded to whole 16-byte chunks), the tail of the
// last 128-byte step zero.  Plain image: [step][co_pad rows][128 bytes].  Fragment-major (mv2_conv3 / mv2_pw_se; needs whole
// 128-byte channel chunks, i.e. seg_bytes == 128): every group of 32 rows becomes [k-slice 4][half 2][row 32][16 bytes], holding
// 16-byte chunk (2 k-slice + half) of the row.
extern "C" int mv2_pack_conv3d(int dtype, const float* w, int co, int ci, int kt, int kh, int kw, int frag_major, void* out, size_t out_bytes) {
    Geo g;
    if (!w || !out || !geo(dtype, co, ci, kt, kh, kw, g)) return MV2_EINVAL;
    if (frag_major && g.seg_bytes != 128) return MV2_EINVAL;
    const size_t need = (size_t)g.nsteps * g.co_pad * 128;
    if (out_bytes < need) return MV2_EWORKSPACE;
    std::memset(out, 0, need);
    const int ntaps = kt * kh * kw;
    const int64_t wstride_ci = (int64_t)ntaps;                      // (Co, Ci, kt, kh, kw): channel stride = taps
    for (int n = 0; n < co; ++n)
        for (int tap = 0; tap < ntaps; ++tap)
            for (int c = 0; c < ci; ++c) {
                const int64_t k = (int64_t)tap * g.ci_pad + c;     // global K index (elements)
                const int step = (int)(k / g.k_elems), kin = (int)(k % g.k_elems);
                size_t byte = ((size_t)step * g.co_pad + n) * 128 + (size_t)kin * g.es;
                if (frag_major) {
                    const int chunk = (kin * g.es) / 16, within = (kin * g.es) % 16;
                    const int grp = n / 32, row = n % 32, ks = chunk >> 1, half = chunk & 1;
                    byte = (size_t)step * g.co_pad * 128 + (size_t)grp * 4096 + (size_t)ks * 1024 + (size_t)half * 512 + (size_t)row * 16 + within;
                }
                const float v = w[((int64_t)n * ci + c) * wstride_ci + tap];
                if (dtype == MV2_BF16) { const uint16_t b = f2bf_rne(v); std::memcpy((char*)out + byte, &b, 2); }
                else std::memcpy((char*)out + byte, &v, 4);
            }
    return MV2_OK;
}
