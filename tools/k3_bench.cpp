// tools/k3_bench.cpp — standalone A/B + correctness harness for mv2_conv3 (conv3.hip) against
//   (1) a plain CPU reference (small shapes, fp64 accumulate) and
//   (2) the generic kernel (mv2_conv; and mv2_conv_pw when the library still exports the round-1 fused kernel) on the real level shapes,
// plus run-to-run determinism, ticket-word hygiene and per-launch timing (HIP events, back-to-back launches).
// Build (no GPU needed):  hipcc -O2 -std=c++17 tools/k3_bench.cpp -o tools/k3_bench -ldl
// Run on the GPU box:     tools/k3_bench [path to libmv2_hip_exp.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../include/mv2.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef int (*conv_fn)(const mv2_conv_desc*, const void*, size_t, const void*, const mv2_seg*, const float*, const float*, const void*, void*, void*, size_t, void*);
typedef size_t (*ws_fn)(const mv2_conv_desc*);
typedef int (*elig_fn)(const mv2_conv_desc*);
typedef int (*conv3_fn)(const mv2_conv_desc*, const void*, size_t, const void*, const float*, const void*, const float*, int,
                        const float*, float, float*, void*, void*, size_t, uint32_t*, size_t, void*);
typedef int (*convpw_fn)(const mv2_conv_desc*, const void*, size_t, const void*, const mv2_seg*, const float*, const void*, const float*, int, const float*,
                         float, float*, void*, void*);
typedef int (*tiles_fn)(const mv2_conv_desc*, int*, int*);

typedef int (*setprof_fn)(void*);
static setprof_fn p_setprof, p_setprof_old, p_setprof_new;
static int set_both_profiles(void* b) { int r = 0; if (p_setprof_old) r |= p_setprof_old(b); if (p_setprof_new) r |= p_setprof_new(b); return r; }   // conv3.hip (K-split) and conv3p.hip (position tile) write the same record layout
static conv_fn p_conv; static ws_fn p_conv_ws, p_c3_ws, p_c3_cnt; static elig_fn p_c3_fusable; static conv3_fn p_conv3; static convpw_fn p_convpw; static tiles_fn p_c3_tiles;

static uint32_t rng_state = 12345u;
// K3_ZERO=1: all-zero operands (power-management probe: the same instruction stream on a quiet data path; rule 25 of the guide);
// K3_ZERO=x: only the activations are zero, K3_ZERO=w: only the weights (which operand path the data-dependent cost sits on)
static const char* g_zero_env = getenv("K3_ZERO");
static const bool g_zero_all = g_zero_env && g_zero_env[0] == '1', g_zero_x = g_zero_all || (g_zero_env && g_zero_env[0] == 'x'), g_zero_w = g_zero_all || (g_zero_env && g_zero_env[0] == 'w');
static bool g_zero_fill = false;
static inline float frand() { rng_state = rng_state * 1664525u + 1013904223u; return g_zero_fill ? 0.f : ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Shape { int B, T, H, W, C, kt; };

struct Case {
    Shape s; int es;          // element size 2 / 4
    mv2_conv_desc d;
    std::vector<uint8_t> in, wp, wp3, wp2;        // host images
    std::vector<float> bias, bias2, se_wk;
    std::vector<mv2_seg> segs;
    void *d_in, *d_wp, *d_wp3, *d_wp2, *d_bias, *d_bias2, *d_sewk, *d_segs;
    size_t in_bytes, M;
};

static void put_elem(std::vector<uint8_t>& v, size_t i, float f, int es) {
    if (es == 2) { uint16_t h = f2bf(f); memcpy(&v[i * 2], &h, 2); } else memcpy(&v[i * 4], &f, 4);
}
static float get_elem(const std::vector<uint8_t>& v, size_t i, int es) {
    if (es == 2) { uint16_t h; memcpy(&h, &v[i * 2], 2); return bf2f(h); }
    float f; memcpy(&f, &v[i * 4], 4); return f;
}

static void* to_dev(const void* h, size_t n) { void* d; CK(hipMalloc(&d, n ? n : 16)); if (n) CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); return d; }

static Case make_case(Shape s, int es) {
    Case c; c.s = s; c.es = es;
    const int C = s.C, kel = 128 / es, nchunk = C * es / 128, co_pad = (C + 127) / 128 * 128, nsteps = s.kt * 9 * nchunk;
    mv2_conv_desc& d = c.d; memset(&d, 0, sizeof(d));
    d.dtype = es == 2 ? MV2_BF16 : MV2_F32; d.B = s.B; d.Ti = d.To = s.T; d.Hi = d.Ho = s.H; d.Wi = d.Wo = s.W; d.Ci = C; d.co = C; d.co_pad = co_pad;
    d.st = d.sh = d.sw = 1; d.pt = s.kt - 1; d.ph = d.pw = 1; d.kt = s.kt; d.kh = d.kw = 3; d.nsteps = nsteps; d.seg_bytes = 128; d.check_bounds = 1;
    d.act = MV2_ACT_ELU; d.store = MV2_STORE_ROWS; d.rt = d.rh = d.rw = 1; d.out_c = C; d.splitk = 0;
    c.M = (size_t)s.B * s.T * s.H * s.W;
    c.in_bytes = c.M * C * es;
    c.in.resize(c.in_bytes);
    g_zero_fill = g_zero_x;
    for (size_t i = 0; i < c.M * C; ++i) put_elem(c.in, i, frand(), es);
    g_zero_fill = g_zero_w;
    // plain packed weights [step][co_pad][kel], scaled so outputs are O(1)
    const float wscale = 1.0f / sqrtf((float)(27 * C) / 3.0f);
    c.wp.assign((size_t)nsteps * co_pad * 128, 0);
    for (int st = 0; st < nsteps; ++st)
        for (int n = 0; n < C; ++n)
            for (int k = 0; k < kel; ++k) put_elem(c.wp, ((size_t)st * co_pad + n) * kel + k, frand() * wscale * 1.7f, es);
    // fragment-major: [step][co_pad/32][ks 4][half 2][row 32][16 B]  <-  16-byte chunk (2 ks + half) of row r
    c.wp3.assign(c.wp.size(), 0);
    for (int st = 0; st < nsteps; ++st)
        for (int g = 0; g < co_pad / 32; ++g)
            for (int ks = 0; ks < 4; ++ks)
                for (int hf = 0; hf < 2; ++hf)
                    for (int r = 0; r < 32; ++r) {
                        const size_t src = ((size_t)st * co_pad + g * 32 + r) * 128 + (size_t)(2 * ks + hf) * 16;
                        const size_t dst = ((((size_t)st * (co_pad / 32) + g) * 4 + ks) * 2 + hf) * 512 + (size_t)r * 16;
                        memcpy(&c.wp3[dst], &c.wp[src], 16);
                    }
    g_zero_fill = false;
    c.bias.resize(co_pad); for (auto& b : c.bias) b = frand() * 0.1f;
    // pointwise second stage (bf16 only): [C/64][co_pad][64]
    const int ns2 = C * es / 128;
    c.wp2.assign((size_t)ns2 * co_pad * 128, 0);
    const float w2s = 1.0f / sqrtf((float)C / 3.0f);
    for (int st = 0; st < ns2; ++st)
        for (int n = 0; n < C; ++n)
            for (int k = 0; k < kel; ++k) put_elem(c.wp2, ((size_t)st * co_pad + n) * kel + k, frand() * w2s, es);
    c.bias2.resize(co_pad); for (auto& b : c.bias2) b = frand() * 0.1f;
    c.se_wk.resize(C); for (auto& b : c.se_wk) b = frand() * 0.2f;
    // K program of the generic / halo kernel: step = tap * nchunk + chunk
    c.segs.resize(nsteps);
    for (int it = 0; it < s.kt; ++it) for (int ih = 0; ih < 3; ++ih) for (int iw = 0; iw < 3; ++iw) for (int ch = 0; ch < nchunk; ++ch) {
        mv2_seg sg; sg.elem_offset = ((it * s.H + ih) * s.W + iw) * C + ch * kel; sg.sel = (1u << it) | (1u << (8 + ih)) | (1u << (16 + iw));
        c.segs[((it * 3 + ih) * 3 + iw) * nchunk + ch] = sg;
    }
    c.d_in = to_dev(c.in.data(), c.in.size()); c.d_wp = to_dev(c.wp.data(), c.wp.size()); c.d_wp3 = to_dev(c.wp3.data(), c.wp3.size());
    c.d_wp2 = to_dev(c.wp2.data(), c.wp2.size()); c.d_bias = to_dev(c.bias.data(), c.bias.size() * 4); c.d_bias2 = to_dev(c.bias2.data(), c.bias2.size() * 4);
    c.d_sewk = to_dev(c.se_wk.data(), c.se_wk.size() * 4); c.d_segs = to_dev(c.segs.data(), c.segs.size() * sizeof(mv2_seg));
    return c;
}
static void free_case(Case& c) { for (void* p : {c.d_in, c.d_wp, c.d_wp3, c.d_wp2, c.d_bias, c.d_bias2, c.d_sewk, c.d_segs}) hipFree(p); }

static float elu(float v) { return v > 0 ? v : expm1f(v); }

// plain CPU reference of conv + bias + ELU (fp64 accumulate), output rounded to the storage dtype
static std::vector<float> cpu_ref(const Case& c) {
    const Shape& s = c.s; const int C = s.C, es = c.es, kel = 128 / es, nchunk = C * es / 128, co_pad = c.d.co_pad;
    std::vector<float> inf(c.M * C); for (size_t i = 0; i < inf.size(); ++i) inf[i] = get_elem(c.in, i, es);
    std::vector<float> wf(c.wp.size() / es); for (size_t i = 0; i < wf.size(); ++i) wf[i] = get_elem(c.wp, i, es);
    std::vector<float> out(c.M * C);
    for (int b = 0; b < s.B; ++b) for (int t = 0; t < s.T; ++t) for (int h = 0; h < s.H; ++h) for (int w = 0; w < s.W; ++w) {
        const size_t m = (((size_t)b * s.T + t) * s.H + h) * s.W + w;
        std::vector<double> acc(C, 0.0);
        for (int it = 0; it < s.kt; ++it) { const int ti = t + it - (s.kt - 1); if (ti < 0) continue;
            for (int ih = 0; ih < 3; ++ih) { const int hi = h + ih - 1; if (hi < 0 || hi >= s.H) continue;
                for (int iw = 0; iw < 3; ++iw) { const int wi = w + iw - 1; if (wi < 0 || wi >= s.W) continue;
                    const float* xr = &inf[((((size_t)b * s.T + ti) * s.H + hi) * s.W + wi) * C];
                    for (int ch = 0; ch < nchunk; ++ch) {
                        const size_t st = ((it * 3 + ih) * 3 + iw) * nchunk + ch;
                        for (int n = 0; n < C; ++n) { const float* wr = &wf[(st * co_pad + n) * kel]; double a = 0; for (int k = 0; k < kel; ++k) a += (double)wr[k] * xr[ch * kel + k]; acc[n] += a; }
                    } } } }
        for (int n = 0; n < C; ++n) out[m * C + n] = elu((float)acc[n] + c.bias[n]);
    }
    return out;
}

struct Diff { double max_abs, mean_abs; size_t nbad; };
static Diff compare(const std::vector<uint8_t>& a, const std::vector<float>& ref, int es, double atol, double rtol) {
    Diff d{0, 0, 0};
    for (size_t i = 0; i < ref.size(); ++i) { const double x = get_elem(a, i, es), r = ref[i], e = fabs(x - r); d.max_abs = e > d.max_abs ? e : d.max_abs; d.mean_abs += e;
        if (!(e <= atol + rtol * fabs(r))) ++d.nbad; }
    d.mean_abs /= ref.size() ? ref.size() : 1; return d;
}
static std::vector<float> to_float(const std::vector<uint8_t>& a, size_t n, int es) { std::vector<float> f(n); for (size_t i = 0; i < n; ++i) f[i] = get_elem(a, i, es); return f; }

struct Bufs { void *out, *ws, *ws3, *part; uint32_t* cnt; size_t ws_bytes, ws3_bytes, cnt_bytes, out_bytes, part_bytes; };

static int run_new(const Case& c, Bufs& b, bool fused, bool se, hipStream_t st) {
    return p_conv3(&c.d, c.d_in, c.in_bytes, c.d_wp3, (const float*)c.d_bias, fused ? c.d_wp2 : nullptr, fused ? (const float*)c.d_bias2 : nullptr,
                   MV2_ACT_ELU, se ? (const float*)c.d_sewk : nullptr, 0.05f, se ? (float*)b.part : nullptr, b.out, b.ws3, b.ws3_bytes, b.cnt, b.cnt_bytes, st);
}
static int run_old(const Case& c, Bufs& b, bool fused, bool se, hipStream_t st) {
    if (fused) return p_convpw(&c.d, c.d_in, c.in_bytes, c.d_wp, (const mv2_seg*)c.d_segs, (const float*)c.d_bias, c.d_wp2, (const float*)c.d_bias2, MV2_ACT_ELU,
                               se ? (const float*)c.d_sewk : nullptr, 0.05f, se ? (float*)b.part : nullptr, b.out, st);
    return p_conv(&c.d, c.d_in, c.in_bytes, c.d_wp, (const mv2_seg*)c.d_segs, (const float*)c.d_bias, nullptr, nullptr, b.out, b.ws, b.ws_bytes, st);
}

template <typename F> static float time_us(F f, int reps, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1000.f / reps;
}

static bool test_shape(Shape s, int es, bool do_cpu, bool do_old, bool do_fused, int reps, hipStream_t st) {
    Case c = make_case(s, es);
    Bufs b{}; b.out_bytes = c.M * s.C * es; CK(hipMalloc(&b.out, b.out_bytes));
    b.ws_bytes = p_conv_ws ? p_conv_ws(&c.d) : 0; CK(hipMalloc(&b.ws, b.ws_bytes ? b.ws_bytes : 16));
    b.ws3_bytes = p_c3_ws(&c.d); CK(hipMalloc(&b.ws3, b.ws3_bytes ? b.ws3_bytes : 16));
    b.cnt_bytes = p_c3_cnt(&c.d); CK(hipMalloc((void**)&b.cnt, b.cnt_bytes ? b.cnt_bytes : 16)); CK(hipMemset(b.cnt, 0, b.cnt_bytes ? b.cnt_bytes : 16));
    const bool fusable = es == 2 && p_c3_fusable && p_c3_fusable(&c.d);
    int th = 0, tw = 0; const int ntile = fusable ? p_c3_tiles(&c.d, &th, &tw) : 0;
    // old kernel: 16x16 tiles at C <= 64, 8x16 above
    const int ntile_old = ((s.H + (s.C <= 64 ? 15 : 7)) / (s.C <= 64 ? 16 : 8)) * ((s.W + 15) / 16);
    b.part_bytes = (size_t)s.B * s.T * std::max(ntile, ntile_old) * (s.C + 2) * 4; CK(hipMalloc(&b.part, b.part_bytes));
    bool ok = true;
    const size_t n = c.M * s.C;
    const double atol = es == 2 ? 2e-2 : 2e-4, rtol = es == 2 ? 2e-2 : 2e-4;
    printf("== C=%d T=%d H=%d W=%d B=%d kt=%d %s  class %s (fused tiles %dx%d, slabs %.1f MB, tickets %zu)\n", s.C, s.T, s.H, s.W, s.B, s.kt, es == 2 ? "bf16" : "fp32",
           "64co x 8x16", th, tw, b.ws3_bytes / 1048576.0, b.cnt_bytes / 4);
    // --- new kernel, plain
    CK(hipMemset(b.out, 0xff, b.out_bytes));
    int rc = run_new(c, b, false, false, st); CK(hipStreamSynchronize(st));
    if (rc != 0) { printf("   mv2_conv3 rc=%d\n", rc); return false; }
    std::vector<uint8_t> o_new(b.out_bytes); CK(hipMemcpy(o_new.data(), b.out, b.out_bytes, hipMemcpyDeviceToHost));
    // determinism + ticket hygiene
    std::vector<uint32_t> cnt(b.cnt_bytes / 4 + 1);
    CK(hipMemcpy(cnt.data(), b.cnt, b.cnt_bytes, hipMemcpyDeviceToHost));
    size_t nz = 0; for (size_t i = 0; i < b.cnt_bytes / 4; ++i) nz += cnt[i] != 0;
    if (nz) { printf("   FAIL: %zu ticket words non-zero after the launch\n", nz); ok = false; }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(b.out, 0xff, b.out_bytes));
        run_new(c, b, false, false, st); CK(hipStreamSynchronize(st));
        std::vector<uint8_t> o2(b.out_bytes); CK(hipMemcpy(o2.data(), b.out, b.out_bytes, hipMemcpyDeviceToHost));
        if (memcmp(o2.data(), o_new.data(), b.out_bytes) != 0) { printf("   FAIL: run %d differs bitwise from run 0\n", rep + 1); ok = false; }
    }
    if (do_cpu) {
        std::vector<float> ref = cpu_ref(c);
        Diff d = compare(o_new, ref, es, atol, rtol);
        printf("   new vs CPU ref : max|d| %.3e mean|d| %.3e bad %zu / %zu  %s\n", d.max_abs, d.mean_abs, d.nbad, n, d.nbad ? "FAIL" : "ok");
        ok &= d.nbad == 0;
    }
    std::vector<uint8_t> o_old;
    if (do_old && p_conv) {
        CK(hipMemset(b.out, 0xff, b.out_bytes));
        rc = run_old(c, b, false, false, st); CK(hipStreamSynchronize(st));
        if (rc != 0) printf("   old mv2_conv rc=%d\n", rc);
        else {
            o_old.resize(b.out_bytes); CK(hipMemcpy(o_old.data(), b.out, b.out_bytes, hipMemcpyDeviceToHost));
            Diff d = compare(o_new, to_float(o_old, n, es), es, atol * 0.5, rtol * 0.5);
            printf("   new vs old halo: max|d| %.3e mean|d| %.3e bad %zu / %zu  %s\n", d.max_abs, d.mean_abs, d.nbad, n, d.nbad ? "FAIL" : "ok");
            ok &= d.nbad == 0;
        }
    }
    if (do_fused && fusable) {
        CK(hipMemset(b.out, 0xff, b.out_bytes)); CK(hipMemset(b.part, 0xff, b.part_bytes));
        rc = run_new(c, b, true, true, st); CK(hipStreamSynchronize(st));
        if (rc != 0) { printf("   fused mv2_conv3 rc=%d\n", rc); ok = false; }
        std::vector<uint8_t> f_new(b.out_bytes), p_new(b.part_bytes);
        CK(hipMemcpy(f_new.data(), b.out, b.out_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(p_new.data(), b.part, b.part_bytes, hipMemcpyDeviceToHost));
        if (p_conv) {      // independent check of the fused stage: the plain launch (validated above) followed by the generic kernel's 1x1x1 conv + ELU, pooled SE vector on the host
            mv2_conv_desc d1; memset(&d1, 0, sizeof(d1));
            d1.dtype = MV2_BF16; d1.B = s.B; d1.Ti = d1.To = s.T; d1.Hi = d1.Ho = s.H; d1.Wi = d1.Wo = s.W; d1.Ci = s.C; d1.co = s.C; d1.co_pad = c.d.co_pad;
            d1.st = d1.sh = d1.sw = 1; d1.kt = d1.kh = d1.kw = 1; d1.nsteps = s.C / 64; d1.seg_bytes = 128; d1.check_bounds = 0; d1.act = MV2_ACT_ELU; d1.store = MV2_STORE_ROWS;
            d1.rt = d1.rh = d1.rw = 1; d1.out_c = s.C;
            std::vector<mv2_seg> sg1(d1.nsteps); for (int i = 0; i < d1.nsteps; ++i) { sg1[i].elem_offset = i * 64; sg1[i].sel = 1u | (1u << 8) | (1u << 16); }
            void* dsg1 = to_dev(sg1.data(), sg1.size() * sizeof(mv2_seg));
            void *y1, *y2; CK(hipMalloc(&y1, b.out_bytes)); CK(hipMalloc(&y2, b.out_bytes));
            CK(hipMemcpy(y1, o_new.data(), b.out_bytes, hipMemcpyHostToDevice));
            const size_t wsb1 = p_conv_ws(&d1); void* dws1; CK(hipMalloc(&dws1, wsb1 ? wsb1 : 16));
            rc = p_conv(&d1, y1, b.out_bytes, c.d_wp2, (const mv2_seg*)dsg1, (const float*)c.d_bias2, nullptr, nullptr, y2, dws1, wsb1, st); CK(hipStreamSynchronize(st));
            if (rc) { printf("   reference 1x1x1 mv2_conv rc=%d\n", rc); ok = false; }
            std::vector<uint8_t> f_ref(b.out_bytes); CK(hipMemcpy(f_ref.data(), y2, b.out_bytes, hipMemcpyDeviceToHost));
            hipFree(y1); hipFree(y2); hipFree(dws1); hipFree(dsg1);
            const std::vector<float> fr = to_float(f_ref, n, 2);
            Diff d = compare(f_new, fr, 2, 2e-2, 2e-2);
            // pooled SE vector per frame from the reference output: softmax over positions of (w . y + b), then the weighted channel means
            const int HW = s.H * s.W;
            std::vector<float> po((size_t)s.B * s.T * s.C);
            for (int fi = 0; fi < s.B * s.T; ++fi) {
                std::vector<double> lg(HW); double M = -1e300;
                for (int q = 0; q < HW; ++q) { double a = 0.05; const float* yr = &fr[((size_t)fi * HW + q) * s.C]; for (int ch = 0; ch < s.C; ++ch) a += (double)yr[ch] * c.se_wk[ch]; lg[q] = a; M = std::max(M, a); }
                double S = 0; std::vector<double> acc(s.C, 0.0);
                for (int q = 0; q < HW; ++q) { const double w = exp(lg[q] - M); S += w; const float* yr = &fr[((size_t)fi * HW + q) * s.C]; for (int ch = 0; ch < s.C; ++ch) acc[ch] += w * yr[ch]; }
                for (int ch = 0; ch < s.C; ++ch) po[(size_t)fi * s.C + ch] = (float)(acc[ch] / S);
            }
            const float* f = (const float*)p_new.data();
            std::vector<float> pn((size_t)s.B * s.T * s.C);
            for (int fi = 0; fi < s.B * s.T; ++fi) {
                float M = -INFINITY; for (int j = 0; j < ntile; ++j) M = fmaxf(M, f[((size_t)fi * ntile + j) * (s.C + 2)]);
                double S = 0; std::vector<double> acc(s.C, 0.0);
                for (int j = 0; j < ntile; ++j) { const float* r = f + ((size_t)fi * ntile + j) * (s.C + 2); if (!(r[0] > -INFINITY)) continue; const double w = exp((double)r[0] - M);
                    S += w * r[1]; for (int ch = 0; ch < s.C; ++ch) acc[ch] += w * r[2 + ch]; }
                for (int ch = 0; ch < s.C; ++ch) pn[(size_t)fi * s.C + ch] = (float)(acc[ch] / S);
            }
            double dmax = 0; size_t nb = 0; for (size_t i = 0; i < pn.size(); ++i) { const double e = fabs(pn[i] - po[i]); dmax = e > dmax ? e : dmax; nb += !(e <= 2e-3 + 1e-2 * fabs(po[i])); }
            printf("   fused vs plain + generic 1x1x1: out max|d| %.3e bad %zu; pooled SE vector max|d| %.3e bad %zu / %zu  %s\n", d.max_abs, d.nbad, dmax, nb, pn.size(), (d.nbad || nb) ? "FAIL" : "ok");
            ok &= d.nbad == 0 && nb == 0;
            // run-to-run: the fused launch twice more, output and records bitwise
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(b.out, 0xff, b.out_bytes)); CK(hipMemset(b.part, 0xff, b.part_bytes));
                run_new(c, b, true, true, st); CK(hipStreamSynchronize(st));
                std::vector<uint8_t> f2(b.out_bytes), p2(b.part_bytes);
                CK(hipMemcpy(f2.data(), b.out, b.out_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(p2.data(), b.part, b.part_bytes, hipMemcpyDeviceToHost));
                if (memcmp(f2.data(), f_new.data(), b.out_bytes) != 0 || memcmp(p2.data(), p_new.data(), b.part_bytes) != 0) { printf("   FAIL: fused run %d differs bitwise from run 0\n", rep + 1); ok = false; }
            }
        }
        if (p_convpw) {
            CK(hipMemset(b.out, 0xff, b.out_bytes)); CK(hipMemset(b.part, 0xff, b.part_bytes));
            rc = run_old(c, b, true, true, st); CK(hipStreamSynchronize(st));
            if (rc != 0) printf("   old mv2_conv_pw rc=%d\n", rc);
            else {
                std::vector<uint8_t> f_old(b.out_bytes), p_old(b.part_bytes);
                CK(hipMemcpy(f_old.data(), b.out, b.out_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(p_old.data(), b.part, b.part_bytes, hipMemcpyDeviceToHost));
                Diff d = compare(f_new, to_float(f_old, n, 2), 2, 2e-2, 2e-2);
                // the two kernels tile the frame differently: compare the POOLED vector per frame (records combined as mv2_se_gate does)
                auto pooled = [&](const std::vector<uint8_t>& pv, int nt) {
                    std::vector<float> out((size_t)s.B * s.T * s.C);
                    const float* f = (const float*)pv.data();
                    for (int fr = 0; fr < s.B * s.T; ++fr) {
                        float M = -INFINITY; for (int j = 0; j < nt; ++j) M = fmaxf(M, f[((size_t)fr * nt + j) * (s.C + 2)]);
                        double S = 0; std::vector<double> acc(s.C, 0.0);
                        for (int j = 0; j < nt; ++j) { const float* r = f + ((size_t)fr * nt + j) * (s.C + 2); const double w = exp((double)r[0] - M); if (!(r[0] > -INFINITY)) continue;
                            S += w * r[1]; for (int ch = 0; ch < s.C; ++ch) acc[ch] += w * r[2 + ch]; }
                        for (int ch = 0; ch < s.C; ++ch) out[(size_t)fr * s.C + ch] = (float)(acc[ch] / S);
                    }
                    return out;
                };
                std::vector<float> pn = pooled(p_new, ntile), po = pooled(p_old, ntile_old);
                double dmax = 0; size_t nb = 0; for (size_t i = 0; i < pn.size(); ++i) { const double e = fabs(pn[i] - po[i]); dmax = e > dmax ? e : dmax; nb += !(e <= 2e-3 + 1e-2 * fabs(po[i])); }
                printf("   fused new vs old: out max|d| %.3e bad %zu; pooled SE vector max|d| %.3e bad %zu / %zu  %s\n", d.max_abs, d.nbad, dmax, nb, pn.size(),
                       (d.nbad || nb) ? "FAIL" : "ok");
                ok &= d.nbad == 0 && nb == 0;
            }
        }
    }
    if (reps > 0 && p_setprof) {      // per-phase clocks of one launch (experiments library)
        unsigned long long* dprof; const int NWK = 1024; CK(hipMalloc((void**)&dprof, NWK * 128)); CK(hipMemset(dprof, 0, NWK * 128));
        for (int fz = 0; fz < ((do_fused && fusable) ? 2 : 1); ++fz) {
            CK(hipMemset(dprof, 0, NWK * 128)); p_setprof(dprof);
            run_new(c, b, fz, fz, st); CK(hipStreamSynchronize(st));
            p_setprof(nullptr);
            std::vector<unsigned long long> hp(NWK * 16); CK(hipMemcpy(hp.data(), dprof, NWK * 128, hipMemcpyDeviceToHost));
            double sum[8] = {0}, sub[8] = {0}; int nw = 0; double mx7 = 0;
            std::vector<unsigned> cuid(NWK, 0);
            for (int w = 0; w < NWK; ++w) { cuid[w] = (unsigned)(hp[w * 16 + 5] >> 32); hp[w * 16 + 5] &= 0xffffffffull; }
            for (int w = 0; w < NWK; ++w) if (hp[w * 16 + 7]) { ++nw; for (int k = 0; k < 8; ++k) { sum[k] += (double)hp[w * 16 + k]; sub[k] += (double)hp[w * 16 + 8 + k]; } if (hp[w * 16 + 7] > mx7) mx7 = (double)hp[w * 16 + 7]; }
            if (nw) printf("   prof %s (avg clocks / workgroup, %d wgs): prologue %.0f  taps %.0f (%.0f per tap)  reduce %.0f  ticket %.0f  epilogue %.0f | segments %.2f groups %.1f | kernel avg %.0f max %.0f\n",
                           fz ? "fused" : "plain", nw, sum[0] / nw, sum[1] / nw, sum[1] / (sum[6] * 9), sum[2] / nw, sum[3] / nw, sum[4] / nw, sum[5] / nw, sum[6] / nw, sum[7] / nw, mx7);
            if (nw && getenv("K3_DIST")) {      // start skew, end-time distribution and the slowest workgroups (qf[7] = absolute start clock)
                unsigned long long s0 = ~0ull, e1 = 0; std::vector<std::pair<double, int>> byend;
                for (int w = 0; w < NWK; ++w) if (hp[w * 16 + 7]) { s0 = std::min(s0, hp[w * 16 + 15]); e1 = std::max(e1, hp[w * 16 + 15] + hp[w * 16 + 7]); }
                double sk_sum = 0, sk_max = 0;
                for (int w = 0; w < NWK; ++w) if (hp[w * 16 + 7]) { const double sk = (double)(hp[w * 16 + 15] - s0); sk_sum += sk; sk_max = std::max(sk_max, sk); byend.push_back({(double)(hp[w * 16 + 15] + hp[w * 16 + 7] - s0), w}); }
                std::sort(byend.begin(), byend.end());
                printf("        span first start -> last end %.0f | start skew avg %.0f max %.0f | end times p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f\n", (double)(e1 - s0), sk_sum / nw, sk_max,
                       byend[nw / 10].first, byend[nw / 2].first, byend[nw * 9 / 10].first, byend[nw * 99 / 100].first, byend[nw - 1].first);
                for (int k = 0; k < 6; ++k) { const int w = byend[nw - 1 - k].second;
                    printf("        late wg %4d (xcd %d): start +%.0f dur %llu | segs %llu groups %llu taps %llu reduce %llu ticket %llu epi %llu\n", w, w & 7, (double)(hp[w * 16 + 15] - s0),
                           hp[w * 16 + 7], hp[w * 16 + 5], hp[w * 16 + 6], hp[w * 16 + 1], hp[w * 16 + 2], hp[w * 16 + 3], hp[w * 16 + 4]); }
                {   // residency: workgroups per CU (key = xcc << 8 | se/sh/cu bits of HW_ID), and how a CU's workgroups compare
                    std::vector<std::pair<unsigned, int>> bycu;
                    for (int w = 0; w < NWK; ++w) if (hp[w * 16 + 7]) bycu.push_back({cuid[w], w});
                    std::sort(bycu.begin(), bycu.end());
                    int hist[8] = {0}, ncu = 0; size_t i = 0;
                    double rate1 = 0, rate2 = 0; int n1 = 0, n2 = 0;
                    while (i < bycu.size()) { size_t j = i; while (j < bycu.size() && bycu[j].first == bycu[i].first) ++j; const int cnt = (int)(j - i); hist[std::min(cnt, 7)]++; ++ncu;
                        for (size_t k = i; k < j; ++k) { const int w = bycu[k].second; const double r = (double)hp[w * 16 + 1] / ((double)hp[w * 16 + 6] * 9); if (cnt == 1) { rate1 += r; ++n1; } else { rate2 += r; ++n2; } }
                        i = j; }
                    printf("        residency: %d distinct CUs; CUs hosting 1/2/3/4+ workgroups: %d / %d / %d / %d | clocks per tap on single-wg CUs %.0f (%d wgs), on shared CUs %.0f (%d wgs)\n", ncu, hist[1], hist[2], hist[3],
                           hist[4] + hist[5] + hist[6] + hist[7], n1 ? rate1 / n1 : 0., n1, n2 ? rate2 / n2 : 0., n2);
                    std::vector<std::pair<double, int>> bydur; for (int w = 0; w < NWK; ++w) if (hp[w * 16 + 7]) bydur.push_back({(double)hp[w * 16 + 7], w});
                    std::sort(bydur.begin(), bydur.end());
                    for (int k = 0; k < 5; ++k) { const int w = bydur[nw - 1 - k].second;
                        printf("        slow wg %4d cu %04x: dur %llu | segs %llu groups %llu taps %llu (%.0f/tap) reduce %llu ticket %llu epi %llu | same CU:", w, cuid[w], hp[w * 16 + 7], hp[w * 16 + 5], hp[w * 16 + 6], hp[w * 16 + 1],
                               (double)hp[w * 16 + 1] / (hp[w * 16 + 6] * 9.0), hp[w * 16 + 2], hp[w * 16 + 3], hp[w * 16 + 4]);
                        for (int v = 0; v < NWK; ++v) if (v != w && hp[v * 16 + 7] && cuid[v] == cuid[w]) printf(" wg %d dur %llu groups %llu start %+lld", v, hp[v * 16 + 7], hp[v * 16 + 6], (long long)(hp[v * 16 + 15] - hp[w * 16 + 15]));
                        printf("\n"); }
                    for (int k = 0; k < 5; ++k) { const int w = bydur[k].second;
                        printf("        fast wg %4d cu %04x: dur %llu | segs %llu groups %llu taps %llu (%.0f/tap) | same CU:", w, cuid[w], hp[w * 16 + 7], hp[w * 16 + 5], hp[w * 16 + 6], hp[w * 16 + 1], (double)hp[w * 16 + 1] / (hp[w * 16 + 6] * 9.0));
                        for (int v = 0; v < NWK; ++v) if (v != w && hp[v * 16 + 7] && cuid[v] == cuid[w]) printf(" wg %d dur %llu groups %llu start %+lld", v, hp[v * 16 + 7], hp[v * 16 + 6], (long long)(hp[v * 16 + 15] - hp[w * 16 + 15]));
                        printf("\n"); }
                }
                for (int k = 0; k < 3; ++k) { const int w = byend[k].second;
                    printf("        early wg %4d (xcd %d): start +%.0f dur %llu | segs %llu groups %llu taps %llu reduce %llu ticket %llu epi %llu\n", w, w & 7, (double)(hp[w * 16 + 15] - s0),
                           hp[w * 16 + 7], hp[w * 16 + 5], hp[w * 16 + 6], hp[w * 16 + 1], hp[w * 16 + 2], hp[w * 16 + 3], hp[w * 16 + 4]); }
            }
            if (nw) printf("        epilogue parts per SEGMENT (conv3p plain: prologue parts decode / weight ring / tables): put_tile %.0f  gemm2 %.0f  put_tile2 %.0f  row stores %.0f  SE %.0f  (to barrier | conv3p: DMA issue %.0f, barrier | conv3p: landed + barrier %.0f)\n",
                           sub[0] / sum[5], sub[1] / sum[5], sub[2] / sum[5], sub[3] / sum[5], sub[4] / sum[5], sub[5] / sum[5], sub[6] / sum[5]);
        }
        hipFree(dprof);
    }
    if (reps > 0) {
        const double gf = 2.0 * c.M * s.C * 27.0 * s.C * 1e-9;
        const float t_new = time_us([&] { run_new(c, b, false, false, st); }, reps, st);
        printf("   time new plain  : %7.1f us  (%6.0f TF/s)\n", t_new, gf / t_new * 1e3);
        if (getenv("K3_DELTA")) {      // interleaved rounds, min and median per variant (clock drifts under sustained load)
            const int dls[] = {0, 1, 2, 3, 4, 6};
            std::vector<float> ts[6];
            for (int round = 0; round < 7; ++round)
                for (int k = 0; k < 6; ++k) {
                    char buf[16]; snprintf(buf, sizeof buf, "%d", dls[k]); setenv("MV2_C3_DELTA", buf, 1);
                    ts[k].push_back(time_us([&] { run_new(c, b, false, false, st); }, 5, st));
                }
            unsetenv("MV2_C3_DELTA");
            printf("   delta sweep (min / median us):");
            for (int k = 0; k < 6; ++k) { std::sort(ts[k].begin(), ts[k].end()); printf("  d%d: %.1f / %.1f", dls[k], ts[k][0], ts[k][3]); }
            printf("\n");
        }
        if (getenv("K3_ALIGN")) {      // tile-aligned worker ranges (experiments build): correctness against the default split + timing, per delta
            std::vector<uint8_t> o_al(b.out_bytes);
            setenv("MV2_C3_ALIGN", "1", 1);
            CK(hipMemset(b.out, 0xff, b.out_bytes));
            run_new(c, b, false, false, st); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(o_al.data(), b.out, b.out_bytes, hipMemcpyDeviceToHost));
            Diff da = compare(o_al, to_float(o_new, n, es), es, atol * 0.5, rtol * 0.5);
            printf("   aligned ranges vs default split: max|d| %.3e bad %zu / %zu  %s\n", da.max_abs, da.nbad, n, da.nbad ? "FAIL" : "ok");
            ok &= da.nbad == 0;
            printf("   aligned sweep (min / median us over 5 rounds) default: ");
            std::vector<float> td;
            unsetenv("MV2_C3_ALIGN");
            for (int r = 0; r < 5; ++r) td.push_back(time_us([&] { run_new(c, b, false, false, st); }, 5, st));
            std::sort(td.begin(), td.end()); printf("%.1f / %.1f |", td[0], td[2]);
            for (int dl : {0, 1, 3}) for (int wk : {512, 256}) {
                char buf[16]; snprintf(buf, sizeof buf, "%d", dl); setenv("MV2_C3_DELTA", buf, 1);
                snprintf(buf, sizeof buf, "%d", wk); setenv("MV2_C3_WORKERS", buf, 1);
                setenv("MV2_C3_ALIGN", "1", 1);
                std::vector<float> ta;
                for (int r = 0; r < 5; ++r) ta.push_back(time_us([&] { run_new(c, b, false, false, st); }, 5, st));
                std::sort(ta.begin(), ta.end()); printf("  aligned d%d w%d: %.1f / %.1f", dl, wk, ta[0], ta[2]);
            }
            unsetenv("MV2_C3_ALIGN"); unsetenv("MV2_C3_DELTA"); unsetenv("MV2_C3_WORKERS"); printf("\n");
            if (do_fused && fusable) {
                const float tf0 = time_us([&] { run_new(c, b, true, true, st); }, reps, st);
                setenv("MV2_C3_ALIGN", "1", 1);
                const float tf1 = time_us([&] { run_new(c, b, true, true, st); }, reps, st);
                unsetenv("MV2_C3_ALIGN");
                printf("   fused: default %.1f us, aligned %.1f us\n", tf0, tf1);
            }
        }
        if (getenv("K3_SWEEP")) {
            printf("   workers sweep   :");
            for (int wk : {512, 448, 384, 320, 256, 192, 128}) {
                char buf[16]; snprintf(buf, sizeof buf, "%d", wk); setenv("MV2_C3_WORKERS", buf, 1);
                const float tw_ = time_us([&] { run_new(c, b, false, false, st); }, reps, st);
                printf("  %d: %.1f", wk, tw_);
            }
            unsetenv("MV2_C3_WORKERS"); printf("\n");
        }
        if (do_old && p_conv) { const float t_old = time_us([&] { run_old(c, b, false, false, st); }, reps, st); printf("   time old plain  : %7.1f us  (%6.0f TF/s)\n", t_old, gf / t_old * 1e3); }
        if (do_fused && fusable) {
            const float tf_new = time_us([&] { run_new(c, b, true, true, st); }, reps, st);
            printf("   time new fused  : %7.1f us\n", tf_new);
        }
        if (do_fused && es == 2 && s.C <= 128 && p_convpw) { const float tf_old = time_us([&] { run_old(c, b, true, true, st); }, reps, st); printf("   time old fused  : %7.1f us\n", tf_old); }
    }
    fflush(stdout);
    hipFree(b.out); hipFree(b.ws); hipFree(b.ws3); hipFree(b.cnt); hipFree(b.part); free_case(c);
    return ok;
}

// generic mv2_conv (conv.hip) on a 1x1x1 convolution = plain GEMM rows x Ci -> Co: timing + phase clocks
typedef int (*convrms_fn)(const mv2_conv_desc*, const void*, size_t, const void*, const mv2_seg*, const float*, const void*, void*, void*);
static convrms_fn p_conv_rms = nullptr;
static bool g_pw_quiet = false;
static float time_pointwise(const char* name, int rows_t, int rows_hw, int Ci, int Co, int act, bool residual, hipStream_t st, bool rms = false, int shuf = 1,
                            std::vector<uint16_t>* keep = nullptr, int shuf_t = 1) {
    typedef int (*setprof_fn2)(void*);
    static setprof_fn2 p_cvprof = nullptr; static bool looked = false;
    if (!looked) { looked = true; p_cvprof = (setprof_fn2)dlsym(RTLD_DEFAULT, "mv2_conv_set_profile"); }
    mv2_conv_desc d; memset(&d, 0, sizeof(d));
    const int co_pad = (Co + 127) / 128 * 128, nsteps = Ci / 64;
    d.dtype = MV2_BF16; d.B = 1; d.Ti = d.To = rows_t; d.Hi = d.Ho = rows_hw; d.Wi = d.Wo = rows_hw; d.Ci = Ci; d.co = Co; d.co_pad = co_pad;
    d.st = d.sh = d.sw = 1; d.kt = d.kh = d.kw = 1; d.nsteps = nsteps; d.seg_bytes = 128; d.check_bounds = 0; d.act = act; d.store = MV2_STORE_ROWS;
    d.rt = d.rh = d.rw = 1; d.out_c = act == MV2_ACT_GEGLU ? Co / 2 : Co;
    if (shuf > 1) { d.store = MV2_STORE_SHUFFLE; d.rh = d.rw = shuf; d.out_c = Co / (shuf * shuf); }
    if (shuf_t > 1) { d.store = MV2_STORE_SHUFFLE; d.rt = shuf_t; d.out_c = Co / shuf_t; }
    const size_t M = (size_t)rows_t * rows_hw * rows_hw;
    rng_state = 12345u;                                  // same operands on every call (ab_pointwise compares two runs)
    std::vector<uint16_t> hin(M * Ci), hw((size_t)nsteps * co_pad * 64);
    for (auto& v : hin) v = f2bf(frand()); for (auto& v : hw) v = f2bf(frand() * 0.05f);
    std::vector<mv2_seg> segs(nsteps); for (int i = 0; i < nsteps; ++i) { segs[i].elem_offset = i * 64; segs[i].sel = 1u | (1u << 8) | (1u << 16); }
    std::vector<float> bias(co_pad, 0.01f);
    void *din = to_dev(hin.data(), hin.size() * 2), *dw = to_dev(hw.data(), hw.size() * 2), *dsegs = to_dev(segs.data(), segs.size() * 8), *dbias = to_dev(bias.data(), bias.size() * 4);
    const size_t out_elems = M * (size_t)d.out_c * d.rh * d.rw * d.rt;
    void *dout, *dres; CK(hipMalloc(&dout, out_elems * 2)); CK(hipMalloc(&dres, out_elems * 2)); CK(hipMemset(dres, 0, out_elems * 2)); CK(hipMemset(dout, 0xff, out_elems * 2));
    const size_t wsb = p_conv_ws(&d); void* dws; CK(hipMalloc(&dws, wsb ? wsb : 16));
    auto run = [&] {
        if (rms) return p_conv_rms(&d, din, M * Ci * 2, dw, (const mv2_seg*)dsegs, (const float*)dbias, residual ? dres : nullptr, dout, st);
        return p_conv(&d, din, M * Ci * 2, dw, (const mv2_seg*)dsegs, (const float*)dbias, nullptr, residual ? dres : nullptr, dout, dws, wsb, st); };
    int rc = run(); CK(hipStreamSynchronize(st));
    if (rc) { printf("   %s: mv2_conv rc=%d\n", name, rc); return -1.f; }
    if (keep) { keep->resize(out_elems); CK(hipMemcpy(keep->data(), dout, out_elems * 2, hipMemcpyDeviceToHost)); }
    const float t = time_us(run, 20, st);
    const double gf = 2.0 * M * Ci * (double)Co * 1e-9;
    if (!g_pw_quiet) printf("== generic %-22s M=%zu Ci=%d Co=%d: %7.1f us (%5.0f TF/s)  split-K ws %.1f MB\n", name, M, Ci, Co, t, gf / t * 1e3, wsb / 1048576.0);
    if (p_cvprof && !g_pw_quiet) {
        unsigned long long* dprof; const int NB = 8192; CK(hipMalloc((void**)&dprof, NB * 64)); CK(hipMemset(dprof, 0, NB * 64));
        p_cvprof(dprof); run(); CK(hipStreamSynchronize(st)); p_cvprof(nullptr);
        std::vector<unsigned long long> hp(NB * 8); CK(hipMemcpy(hp.data(), dprof, NB * 64, hipMemcpyDeviceToHost));
        double sum[8] = {0}; int nb = 0; double mx = 0;
        for (int b = 0; b < NB; ++b) if (hp[b * 8 + 7]) { ++nb; for (int k = 0; k < 8; ++k) sum[k] += (double)hp[b * 8 + k]; mx = std::max(mx, (double)hp[b * 8 + 7]); }
        if (nb) printf("   prof (avg clocks / block, %d blocks): prologue %.0f  K loop %.0f (%d steps)  epilogue %.0f | block avg %.0f max %.0f\n", nb, sum[0] / nb, sum[1] / nb, nsteps,
                       sum[2] / nb, sum[7] / nb, mx);
        hipFree(dprof);
    }
    hipFree(din); hipFree(dw); hipFree(dsegs); hipFree(dbias); hipFree(dout); hipFree(dres); hipFree(dws);
    return t;
}

// conv_pw_dma_kernel against conv_kernel on the same launch (experiments library: MV2_PW_DMA=0 switches the DMA kernel off, =N sets the tile
// threshold): time of both + element-wise comparison of the two outputs (same MFMA order: bit-identical unless the row scale is fused)
static void ab_pointwise(const char* name, int rows_t, int rows_hw, int Ci, int Co, int act, bool residual, hipStream_t st, bool rms = false, int shuf = 1) {
    std::vector<uint16_t> a, b;
    setenv("MV2_PW_DMA", "0", 1);
    const float t0 = time_pointwise(name, rows_t, rows_hw, Ci, Co, act, residual, st, rms, shuf, &a);
    setenv("MV2_PW_DMA", "1", 1);
    const float t1 = time_pointwise(name, rows_t, rows_hw, Ci, Co, act, residual, st, rms, shuf, &b);
    unsetenv("MV2_PW_DMA");
    size_t diff = 0, nan = 0; double maxd = 0;
    for (size_t i = 0; i < a.size() && i < b.size(); ++i) {
        const float x = bf2f(a[i]), y = bf2f(b[i]);
        if (!(y == y)) ++nan;
        if (a[i] != b[i]) { ++diff; maxd = std::max(maxd, (double)fabsf(x - y)); }
    }
    printf("## %-26s generic %6.1f us -> dma %6.1f us (%+5.1f %%)   differing elements %zu of %zu (max |d| %.3g, nan %zu)%s\n", name, t0, t1, (t1 / t0 - 1.f) * 100.f, diff, a.size(), maxd, nan,
           a.size() != b.size() ? "  SIZE MISMATCH" : "");
}

// conv_pw_dma_kernel tile sweep (experiments library: MV2_PW_DMA_TILE): the library's rule (whatever kernel it picks), the generic kernel,
// then the LDS-DMA kernel forced on with each tile; every output compared element-wise with the generic kernel's
static void sweep_pointwise(const char* name, int rows_t, int rows_hw, int Ci, int Co, int act, bool residual, hipStream_t st, bool rms = false, int shuf = 1, int shuf_t = 1) {
    g_pw_quiet = true;
    std::vector<uint16_t> ref, b;
    unsetenv("MV2_PW_DMA"); unsetenv("MV2_PW_DMA_TILE");
    const float trule = time_pointwise(name, rows_t, rows_hw, Ci, Co, act, residual, st, rms, shuf, nullptr, shuf_t);
    setenv("MV2_PW_DMA", "0", 1);
    const float t0 = time_pointwise(name, rows_t, rows_hw, Ci, Co, act, residual, st, rms, shuf, &ref, shuf_t);
    printf("## %-28s rule %6.1f | generic %6.1f |", name, trule, t0);
    const char* tiles[] = {"128,128", "128,64", "64,128", "64,64"};
    setenv("MV2_PW_DMA", "1", 1);
    for (const char* tl : tiles) {
        setenv("MV2_PW_DMA_TILE", tl, 1);
        const float t1 = time_pointwise(name, rows_t, rows_hw, Ci, Co, act, residual, st, rms, shuf, &b, shuf_t);
        size_t diff = 0, nan = 0; double maxd = 0;
        for (size_t i = 0; i < ref.size() && i < b.size(); ++i) {
            const float x = bf2f(ref[i]), y = bf2f(b[i]);
            if (!(y == y)) ++nan;
            if (ref[i] != b[i]) { ++diff; maxd = std::max(maxd, (double)fabsf(x - y)); }
        }
        printf(" %s %6.1f", tl, t1);
        if (t1 < 0 || ref.size() != b.size() || nan || (diff && !rms) || (rms && maxd > 0.07)) printf(" [BAD diff %zu maxd %.3g nan %zu]", diff, maxd, nan);
        else if (diff) printf(" (d%zu %.2g)", diff, maxd);
        printf(" |");
    }
    printf("\n");
    unsetenv("MV2_PW_DMA"); unsetenv("MV2_PW_DMA_TILE");
    g_pw_quiet = false;
}

// generic mv2_conv on a convolution WITH taps / strides / padding (the down-sampling convolutions of the path): conv_kernel against the
// LDS-DMA gather form of conv_pw_dma_kernel (experiments library: MV2_GATHER_DMA=0 off, =N tile threshold, MV2_GATHER_DMA_TILE "bco,bpos"),
// outputs compared element-wise (same MFMA order: bit-identical unless conv_kernel split K)
struct GShape { const char* name; int B, Ti, Hi, Wi, Ci, Co, kt, kh, kw, st, sh, sw, pt, ph, pw; };
static float time_gather(const GShape& g, hipStream_t st, std::vector<uint16_t>* keep, int* splitk_ws = nullptr) {
    mv2_conv_desc d; memset(&d, 0, sizeof(d));
    const int nchunk = g.Ci / 64, taps = g.kt * g.kh * g.kw, nsteps = taps * nchunk, co_pad = (g.Co + 127) / 128 * 128;
    d.dtype = MV2_BF16; d.B = g.B; d.Ti = g.Ti; d.Hi = g.Hi; d.Wi = g.Wi; d.Ci = g.Ci; d.co = g.Co; d.co_pad = co_pad;
    d.To = (g.Ti + g.pt - g.kt) / g.st + 1;                     // causal: pads in front only
    d.Ho = (g.Hi + 2 * g.ph - g.kh) / g.sh + 1; d.Wo = (g.Wi + 2 * g.pw - g.kw) / g.sw + 1;
    d.st = g.st; d.sh = g.sh; d.sw = g.sw; d.pt = g.pt; d.ph = g.ph; d.pw = g.pw; d.kt = g.kt; d.kh = g.kh; d.kw = g.kw;
    d.nsteps = nsteps; d.seg_bytes = 128; d.check_bounds = 1; d.act = MV2_ACT_NONE; d.store = MV2_STORE_ROWS; d.rt = d.rh = d.rw = 1; d.out_c = g.Co; d.splitk = 0;
    const size_t Min = (size_t)g.B * g.Ti * g.Hi * g.Wi, M = (size_t)g.B * d.To * d.Ho * d.Wo;
    rng_state = 4242u;
    std::vector<uint16_t> hin(Min * g.Ci), hw((size_t)nsteps * co_pad * 64, 0);
    for (auto& v : hin) v = f2bf(frand());
    const float ws_ = 1.0f / sqrtf((float)(taps * g.Ci) / 3.0f);
    for (int s = 0; s < nsteps; ++s) for (int n = 0; n < g.Co; ++n) for (int k = 0; k < 64; ++k) hw[((size_t)s * co_pad + n) * 64 + k] = f2bf(frand() * ws_);
    std::vector<mv2_seg> segs(nsteps);
    for (int it = 0; it < g.kt; ++it) for (int ih = 0; ih < g.kh; ++ih) for (int iw = 0; iw < g.kw; ++iw) for (int ch = 0; ch < nchunk; ++ch) {
        mv2_seg sg; sg.elem_offset = ((it * g.Hi + ih) * g.Wi + iw) * g.Ci + ch * 64; sg.sel = (1u << it) | (1u << (8 + ih)) | (1u << (16 + iw));
        segs[((it * g.kh + ih) * g.kw + iw) * nchunk + ch] = sg;
    }
    std::vector<float> bias(co_pad); for (auto& b : bias) b = frand() * 0.1f;
    void *din = to_dev(hin.data(), hin.size() * 2), *dw = to_dev(hw.data(), hw.size() * 2), *dsegs = to_dev(segs.data(), segs.size() * 8), *dbias = to_dev(bias.data(), bias.size() * 4);
    void* dout; CK(hipMalloc(&dout, M * g.Co * 2)); CK(hipMemset(dout, 0xff, M * g.Co * 2));
    const size_t wsb = p_conv_ws(&d); void* dws; CK(hipMalloc(&dws, wsb ? wsb : 16));
    if (splitk_ws) *splitk_ws = (int)(wsb >> 20);
    auto run = [&] { return p_conv(&d, din, Min * g.Ci * 2, dw, (const mv2_seg*)dsegs, (const float*)dbias, nullptr, nullptr, dout, dws, wsb, st); };
    int rc = run(); CK(hipStreamSynchronize(st));
    float t = -1.f;
    if (rc) printf("   %s: mv2_conv rc=%d\n", g.name, rc);
    else {
        if (keep) { keep->resize(M * g.Co); CK(hipMemcpy(keep->data(), dout, M * g.Co * 2, hipMemcpyDeviceToHost)); }
        t = time_us(run, 20, st);
    }
    hipFree(din); hipFree(dw); hipFree(dsegs); hipFree(dbias); hipFree(dout); hipFree(dws);
    return t;
}
static void sweep_gather(const GShape& g, hipStream_t st) {
    std::vector<uint16_t> ref, b;
    int wsmb = 0;
    unsetenv("MV2_GATHER_DMA"); unsetenv("MV2_GATHER_DMA_TILE");
    const float trule = time_gather(g, st, nullptr);
    setenv("MV2_GATHER_DMA", "0", 1);
    const float t0 = time_gather(g, st, &ref, &wsmb);
    const int To = (g.Ti + g.pt - g.kt) / g.st + 1, Ho = (g.Hi + 2 * g.ph - g.kh) / g.sh + 1, Wo = (g.Wi + 2 * g.pw - g.kw) / g.sw + 1;
    const double gf = 2.0 * g.B * To * Ho * Wo * (double)g.Co * g.Ci * g.kt * g.kh * g.kw * 1e-9;
    printf("## %-30s %5.1f GF | rule %6.1f | conv_kernel %6.1f (%4.0f TF/s%s) |", g.name, gf, trule, t0, gf / t0 * 1e3, wsmb ? ", split-K" : "");
    const char* tiles[] = {"128,128", "128,64", "64,128", "64,64"};
    setenv("MV2_GATHER_DMA", "1", 1);
    for (const char* tl : tiles) {
        setenv("MV2_GATHER_DMA_TILE", tl, 1);
        const float t1 = time_gather(g, st, &b);
        size_t diff = 0, nan = 0; double maxd = 0;
        for (size_t i = 0; i < ref.size() && i < b.size(); ++i) {
            const float x = bf2f(ref[i]), y = bf2f(b[i]);
            if (!(y == y)) ++nan;
            if (ref[i] != b[i]) { ++diff; maxd = std::max(maxd, (double)fabsf(x - y)); }
        }
        printf(" %s %6.1f", tl, t1);
        if (t1 < 0 || ref.size() != b.size() || nan || (diff && !wsmb) || maxd > 0.05) printf(" [BAD diff %zu maxd %.3g nan %zu]", diff, maxd, nan);
        else if (diff) printf(" (d%zu %.2g)", diff, maxd);
        printf(" |");
    }
    printf("\n"); fflush(stdout);
    unsetenv("MV2_GATHER_DMA"); unsetenv("MV2_GATHER_DMA_TILE");
}

int main(int argc, char** argv) {
    std::string lib = argc > 1 ? argv[1] : "magvit2_pytorch_amd/libmv2_hip_exp.so";
    void* h = dlopen(lib.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) { printf("dlopen %s: %s\n", lib.c_str(), dlerror()); return 2; }
    p_conv = (conv_fn)dlsym(h, "mv2_conv"); p_conv_ws = (ws_fn)dlsym(h, "mv2_conv_workspace_bytes");
    p_conv3 = (conv3_fn)dlsym(h, "mv2_conv3"); p_c3_ws = (ws_fn)dlsym(h, "mv2_conv3_workspace_bytes"); p_c3_cnt = (ws_fn)dlsym(h, "mv2_conv3_counter_bytes"); p_c3_fusable = (elig_fn)dlsym(h, "mv2_conv3_fusable");
    p_c3_tiles = (tiles_fn)dlsym(h, "mv2_conv3_tiles"); p_convpw = (convpw_fn)dlsym(h, "mv2_conv_pw");
    p_setprof_old = (setprof_fn)dlsym(h, "mv2_conv3_set_profile"); p_setprof_new = (setprof_fn)dlsym(h, "mv2_conv3p_set_profile");
    if (p_setprof_old || p_setprof_new) p_setprof = set_both_profiles;
    printf("conv3 kernel under test: MV2_C3_KERNEL=%s\n", getenv("MV2_C3_KERNEL") ? getenv("MV2_C3_KERNEL") : "(library rule)");
    if (!p_conv3 || !p_c3_ws || !p_c3_cnt || !p_c3_tiles) { printf("missing mv2_conv3 symbols\n"); return 2; }
    if (argc > 2 && !strcmp(argv[2], "pwtile")) {
        hipStream_t st2; CK(hipStreamCreate(&st2));
        p_conv_rms = (convrms_fn)dlsym(h, "mv2_conv_rms");
        if (!p_conv_rms) { printf("no mv2_conv_rms\n"); return 2; }
        for (int rep = 0; rep < 2; ++rep) {
            printf("---- pass %d: every pointwise launch of BASELINE configs[1] (B = 1), us per launch\n", rep);
            sweep_pointwise("enc5 ff1 C=256 geglu+rms", 20, 32, 256, 2 * 704, MV2_ACT_GEGLU, false, st2, true);
            sweep_pointwise("enc5 ff2 C=256 +res", 20, 32, 704, 256, MV2_ACT_NONE, true, st2);
            sweep_pointwise("enc5 q C=256 rms", 20, 32, 256, 256, MV2_ACT_NONE, false, st2, true);
            sweep_pointwise("enc5 kv C=256 rms", 20, 32, 256, 512, MV2_ACT_NONE, false, st2, true);
            sweep_pointwise("enc5 out C=256 +res", 20, 32, 256, 256, MV2_ACT_NONE, true, st2);
            sweep_pointwise("enc8 ff1 C=512 T=20 geglu+rms", 20, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2, true);
            sweep_pointwise("enc8 ff2 C=512 T=20 +res", 20, 16, 1408, 512, MV2_ACT_NONE, true, st2);
            sweep_pointwise("enc8 qkv C=512 T=20 rms", 20, 16, 512, 768, MV2_ACT_NONE, false, st2, true);
            sweep_pointwise("enc8 out C=512 T=20 +res", 20, 16, 256, 512, MV2_ACT_NONE, true, st2);
            sweep_pointwise("enc13 ff2 C=512 T=5 +res", 5, 16, 1408, 512, MV2_ACT_NONE, true, st2);
            sweep_pointwise("enc13 out C=512 T=5 +res", 5, 16, 256, 512, MV2_ACT_NONE, true, st2);
            sweep_pointwise("dec12 up 128->4x64 64x64", 20, 64, 128, 256, MV2_ACT_SILU, false, st2, false, 2);
            sweep_pointwise("dec10 up 256->4x128 32x32", 20, 32, 256, 512, MV2_ACT_SILU, false, st2, false, 2);
            sweep_pointwise("dec7 up 512->4x256 16x16", 20, 16, 512, 1024, MV2_ACT_SILU, false, st2, false, 2);
            sweep_pointwise("dec4 time-up 512->2x512 T=10", 10, 16, 512, 1024, MV2_ACT_SILU, false, st2, false, 1, 2);
            sweep_pointwise("dec2 time-up 512->2x512 T=5", 5, 16, 512, 1024, MV2_ACT_SILU, false, st2, false, 1, 2);
            sweep_pointwise("ragged M=20x29x29 +res", 20, 29, 704, 256, MV2_ACT_NONE, true, st2);
        }
        printf("---- batch 4\n");
        sweep_pointwise("B4 ff1 C=256 geglu+rms", 80, 32, 256, 2 * 704, MV2_ACT_GEGLU, false, st2, true);
        sweep_pointwise("B4 ff2 C=256 +res", 80, 32, 704, 256, MV2_ACT_NONE, true, st2);
        sweep_pointwise("B4 kv C=256 rms", 80, 32, 256, 512, MV2_ACT_NONE, false, st2, true);
        sweep_pointwise("B4 ff1 C=512 T=20 geglu+rms", 80, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2, true);
        sweep_pointwise("B4 ff2 C=512 T=20 +res", 80, 16, 1408, 512, MV2_ACT_NONE, true, st2);
        sweep_pointwise("B4 qkv C=512 T=20 rms", 80, 16, 512, 768, MV2_ACT_NONE, false, st2, true);
        sweep_pointwise("B4 ff1 C=512 T=5 geglu+rms", 20, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2, true);
        sweep_pointwise("B4 dec12 up 128->4x64", 80, 64, 128, 256, MV2_ACT_SILU, false, st2, false, 2);
        sweep_pointwise("B4 dec10 up 256->4x128", 80, 32, 256, 512, MV2_ACT_SILU, false, st2, false, 2);
        sweep_pointwise("B4 dec7 up 512->4x256", 80, 16, 512, 1024, MV2_ACT_SILU, false, st2, false, 2);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "pwdma")) {
        hipStream_t st2; CK(hipStreamCreate(&st2));
        p_conv_rms = (convrms_fn)dlsym(h, "mv2_conv_rms");
        if (!p_conv_rms) { printf("no mv2_conv_rms\n"); return 2; }
        // the pointwise launches of BASELINE configs[1] (B = 1) ...
        ab_pointwise("ff1 C=256 T=20 geglu+rms", 20, 32, 256, 2 * 704, MV2_ACT_GEGLU, false, st2, true);
        ab_pointwise("ff2 C=256 T=20 +res", 20, 32, 704, 256, MV2_ACT_NONE, true, st2);
        ab_pointwise("ff1 C=512 T=20 geglu+rms", 20, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2, true);
        ab_pointwise("ff2 C=512 T=20 +res", 20, 16, 1408, 512, MV2_ACT_NONE, true, st2);
        ab_pointwise("qkv C=512 T=20 rms", 20, 16, 512, 768, MV2_ACT_NONE, false, st2, true);
        ab_pointwise("out C=512 T=20 +res", 20, 16, 256, 512, MV2_ACT_NONE, true, st2);
        ab_pointwise("q C=256 T=20 rms", 20, 32, 256, 256, MV2_ACT_NONE, false, st2, true);
        ab_pointwise("kv C=256 T=20 rms", 20, 32, 256, 512, MV2_ACT_NONE, false, st2, true);
        ab_pointwise("out C=256 T=20 +res", 20, 32, 256, 256, MV2_ACT_NONE, true, st2);
        ab_pointwise("up2x2 128->128 64x64 silu", 20, 64, 128, 512, MV2_ACT_SILU, false, st2, false, 2);
        ab_pointwise("up2x2 256->128 32x32 silu", 20, 32, 256, 512, MV2_ACT_SILU, false, st2, false, 2);
        ab_pointwise("up2x2 512->256 16x16 silu", 20, 16, 512, 1024, MV2_ACT_SILU, false, st2, false, 2);
        ab_pointwise("ragged M=20x29x29 +res", 20, 29, 704, 256, MV2_ACT_NONE, true, st2);
        // ... the deepest level (T = 5: 1280 rows, below the 128-tile threshold unless forced) ...
        setenv("MV2_PW_DMA", "1", 1);
        ab_pointwise("ff1 C=512 T=5 geglu+rms", 5, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2, true);
        ab_pointwise("ff2 C=512 T=5 +res", 5, 16, 1408, 512, MV2_ACT_NONE, true, st2);
        ab_pointwise("qkv C=512 T=5 rms", 5, 16, 512, 768, MV2_ACT_NONE, false, st2, true);
        ab_pointwise("out C=512 T=5 +res", 5, 16, 256, 512, MV2_ACT_NONE, true, st2);
        ab_pointwise("ff1 C=512 T=10 geglu+rms", 10, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2, true);
        ab_pointwise("ff2 C=512 T=10 +res", 10, 16, 1408, 512, MV2_ACT_NONE, true, st2);
        // ... and batch 4
        ab_pointwise("ff1 B=4 C=512 geglu+rms", 80, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2, true);
        ab_pointwise("ff2 B=4 C=512 +res", 80, 16, 1408, 512, MV2_ACT_NONE, true, st2);
        ab_pointwise("ff1 B=4 C=256 geglu+rms", 80, 32, 256, 2 * 704, MV2_ACT_GEGLU, false, st2, true);
        ab_pointwise("ff2 B=4 C=256 +res", 80, 32, 704, 256, MV2_ACT_NONE, true, st2);
        ab_pointwise("up2x2 128->128 B=4", 80, 64, 128, 512, MV2_ACT_SILU, false, st2, false, 2);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "gather")) {
        hipStream_t st2; CK(hipStreamCreate(&st2));
        const GShape shapes[] = {
            // the down-sampling convolutions of BASELINE configs[1] (B = 1) ...
            {"enc1 space 64->128 128x128", 1, 20, 128, 128, 64, 128, 1, 3, 3, 1, 2, 2, 0, 1, 1},
            {"enc3 space 128->256 64x64", 1, 20, 64, 64, 128, 256, 1, 3, 3, 1, 2, 2, 0, 1, 1},
            {"enc6 space 256->512 32x32", 1, 20, 32, 32, 256, 512, 1, 3, 3, 1, 2, 2, 0, 1, 1},
            {"enc9 time 512->512 T=20", 1, 20, 16, 16, 512, 512, 3, 1, 1, 2, 1, 1, 2, 0, 0},
            {"enc11 time 512->512 T=10", 1, 10, 16, 16, 512, 512, 3, 1, 1, 2, 1, 1, 2, 0, 0},
            // ... ragged and padded everywhere (odd extents: tiles that end inside a frame, every border tap) ...
            {"ragged space 64->128 29x31", 2, 3, 29, 31, 64, 128, 1, 3, 3, 1, 2, 2, 0, 1, 1},
            {"ragged k3 stride1 128 13x11", 1, 7, 13, 11, 128, 128, 3, 3, 3, 1, 1, 1, 2, 1, 1},
            // ... and batch 4
            {"B4 enc1 space 64->128", 4, 20, 128, 128, 64, 128, 1, 3, 3, 1, 2, 2, 0, 1, 1},
            {"B4 enc3 space 128->256", 4, 20, 64, 64, 128, 256, 1, 3, 3, 1, 2, 2, 0, 1, 1},
            {"B4 enc6 space 256->512", 4, 20, 32, 32, 256, 512, 1, 3, 3, 1, 2, 2, 0, 1, 1},
            {"B4 enc9 time 512->512 T=20", 4, 20, 16, 16, 512, 512, 3, 1, 1, 2, 1, 1, 2, 0, 0},
            {"B4 enc11 time 512->512 T=10", 4, 10, 16, 16, 512, 512, 3, 1, 1, 2, 1, 1, 2, 0, 0},
        };
        for (int rep = 0; rep < 2; ++rep) {
            printf("---- pass %d: us per launch; conv_kernel vs conv_pw_dma_kernel<GATHER> per tile (bco,bpos)\n", rep);
            for (const GShape& g : shapes) sweep_gather(g, st2);
        }
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "generic")) {
        hipStream_t st2; CK(hipStreamCreate(&st2));
        p_conv = (conv_fn)dlsym(h, "mv2_conv"); p_conv_ws = (ws_fn)dlsym(h, "mv2_conv_workspace_bytes");
        const char* tiles[] = {"", "64,64", "64,128", "128,128", "64,256", "32,256"};
        for (const char* tl : tiles) {
            if (*tl) setenv("MV2_TILE", tl, 1); else unsetenv("MV2_TILE");
            printf("---- tile %s\n", *tl ? tl : "(default rule)");
            time_pointwise("ff1 C=512 T=20 (geglu)", 20, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2);
            time_pointwise("ff2 C=512 T=20", 20, 16, 1408, 512, MV2_ACT_NONE, true, st2);
            time_pointwise("ff1 C=256 T=20 (geglu)", 20, 32, 256, 2 * 704, MV2_ACT_GEGLU, false, st2);
            time_pointwise("ff2 C=256 T=20", 20, 32, 704, 256, MV2_ACT_NONE, true, st2);
            time_pointwise("ff1 C=512 T=5 (geglu)", 5, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2);
            time_pointwise("ff2 C=512 T=5", 5, 16, 1408, 512, MV2_ACT_NONE, true, st2);
            time_pointwise("qkv C=512 T=20", 20, 16, 512, 768, MV2_ACT_NONE, false, st2);
            time_pointwise("out C=512 T=20", 20, 16, 256, 512, MV2_ACT_NONE, true, st2);
            time_pointwise("kv C=256 T=20", 20, 32, 256, 256, MV2_ACT_NONE, false, st2);
            time_pointwise("ff1 B=4 C=512 T=20 (geglu)", 80, 16, 512, 2 * 1408, MV2_ACT_GEGLU, false, st2);
            time_pointwise("ff2 B=4 C=512 T=20", 80, 16, 1408, 512, MV2_ACT_NONE, true, st2);
            // short-K launches whose 128 x 128 tiles are all epilogue (r04: 16-22k of 30k clocks per block)
            time_pointwise("up2x2 128->64 64x64 silu (dec.12)", 20, 64, 128, 256, MV2_ACT_SILU, false, st2, false, 2);
            time_pointwise("up2x2 256->128 32x32 silu (dec.10)", 20, 32, 256, 512, MV2_ACT_SILU, false, st2, false, 2);
            time_pointwise("up2x2 512->256 16x16 silu (dec.7)", 20, 16, 512, 1024, MV2_ACT_SILU, false, st2, false, 2);
            time_pointwise("kv C=256 T=20 (512 cols)", 20, 32, 256, 512, MV2_ACT_NONE, false, st2);
            time_pointwise("ff1 C=256 T=20 (geglu)", 20, 32, 256, 2 * 704, MV2_ACT_GEGLU, false, st2);
        }
        return 0;
    }
    const bool levels4 = argc > 2 && !strcmp(argv[2], "levels4");      // levels at batch 1, then the three large levels at batch 4 (no CPU-reference cases)
    const bool quick = argc > 2 && (!strcmp(argv[2], "quick") || !strcmp(argv[2], "levels") || levels4);
    const bool levels_only = argc > 2 && (!strcmp(argv[2], "levels") || levels4);
    hipStream_t st; CK(hipStreamCreate(&st));
    bool ok = true;
    // small shapes against the CPU reference: ragged tiles, kt = 1/2/3, batch > 1, both dtypes, every channel width
    const Shape small[] = {{1, 3, 12, 20, 64, 3}, {2, 2, 9, 17, 128, 3}, {1, 4, 16, 16, 256, 3}, {1, 2, 8, 8, 512, 3}, {1, 5, 4, 4, 512, 3},
                           {1, 1, 20, 36, 64, 1}, {1, 3, 10, 10, 128, 2}, {3, 1, 2, 2, 256, 3}};
    {
    if (argc > 2 && !strcmp(argv[2], "persist")) {      // experiments library: the persistent work list forced onto shapes of a few tiles (8 / 16 / 24 workgroup slots), against the CPU reference
        const Shape more[] = {{1, 5, 33, 50, 64, 3}, {2, 4, 20, 40, 128, 3}, {1, 6, 24, 24, 256, 3}, {1, 3, 16, 16, 512, 3}, {1, 7, 30, 30, 64, 2}, {4, 3, 16, 32, 64, 1}};
        for (const char* sl : {"8", "16", "24"}) {
            setenv("MV2_C3P_SLOTS", sl, 1);
            printf("######## MV2_C3P_SLOTS=%s\n", sl);
            for (const Shape& s : small) ok &= test_shape(s, 2, true, true, true, 0, st);
            for (const Shape& s : more) ok &= test_shape(s, 2, true, true, true, 0, st);
        }
        unsetenv("MV2_C3P_SLOTS");
        printf(ok ? "ALL OK\n" : "SOME FAILED\n");
        return ok ? 0 : 1;
    }
    if (!levels_only) {
        for (const Shape& s : small) ok &= test_shape(s, 2, true, true, true, 0, st);
        ok &= test_shape({1, 3, 12, 20, 64, 3}, 4, true, true, false, 0, st);
        ok &= test_shape({1, 2, 8, 16, 256, 3}, 4, true, true, false, 0, st);
    }
    // the level shapes of BASELINE configs[1] (B = 1) and configs[2] (B = 4): A/B against the round-1 kernel + timing
    const Shape levels[] = {{1, 20, 128, 128, 64, 3}, {1, 20, 64, 64, 128, 3}, {1, 20, 32, 32, 256, 3}, {1, 20, 16, 16, 512, 3}, {1, 10, 16, 16, 512, 3}, {1, 5, 16, 16, 512, 3}};
    for (const Shape& s : levels) ok &= test_shape(s, 2, false, true, true, quick ? 5 : 20, st);
    if (levels4) {
        const Shape b4[] = {{4, 20, 128, 128, 64, 3}, {4, 20, 64, 64, 128, 3}, {4, 20, 32, 32, 256, 3}};
        for (const Shape& s : b4) ok &= test_shape(s, 2, false, false, false, 5, st);
    }
    if (!quick) {
        const Shape b4[] = {{4, 20, 128, 128, 64, 3}, {4, 20, 64, 64, 128, 3}, {4, 20, 32, 32, 256, 3}, {4, 20, 16, 16, 512, 3}, {4, 5, 16, 16, 512, 3}};
        for (const Shape& s : b4) ok &= test_shape(s, 2, false, true, false, 10, st);
        ok &= test_shape({1, 20, 32, 32, 256, 3}, 4, false, true, false, 5, st);
    }
    }
    printf(ok ? "ALL OK\n" : "SOME FAILED\n");
    return ok ? 0 : 1;
}
