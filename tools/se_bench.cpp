// tools/se_bench.cpp — standalone correctness + timing harness of the ResidualUnit TAIL (reference magvit2_pytorch.py:938-940 conv1 +
// ELU, SqueezeExcite :221-240, Residual add :174) as the plan lowers it at C >= 128 in bf16:
//     mv2_pw_se (conv1 + ELU + pooling records)  ->  mv2_se_rec_fc1 + mv2_se_fc2_apply   (frames <= 256 KB: the deep levels)
//                                                 or  mv2_se_gate    + mv2_se_apply      (large frames)
// against a plain CPU reference of the whole tail (fp32 math on the bf16-rounded inputs), at the level shapes of BASELINE
// configs[1].  20 of the 146 launches of a clip are these tails (0.62 ms of 3.1 ms at batch 1): this is the harness to iterate them
// with, at ~15 s of GPU budget per run instead of a pytest session.
// Build (no GPU needed):  hipcc -w -O2 -std=c++17 tools/se_bench.cpp -o tools/se_bench -ldl
// Run on the GPU box:     tools/se_bench [path to libmv2_hip.so]
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

typedef int (*pw_se_rows_fn)(int, int64_t, int, int);
typedef int (*pw_se_fn)(int, const void*, const void*, const float*, const float*, float, void*, float*, int64_t, int, int, void*);
typedef int (*rec_fc1_fn)(const float*, const float*, const float*, float*, int, int, int, int, int, void*);
typedef int (*fc2_apply_fn)(int, const void*, const float*, const float*, const float*, const void*, void*, int, int, int, int, int, void*);
typedef int (*se_gate_fn)(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, int, void*);
typedef int (*se_apply_fn)(int, const void*, const float*, const void*, void*, int, int, int, void*);
typedef size_t (*pack_bytes_fn)(int, int, int, int, int, int);
typedef int (*pack_fn)(int, const float*, int, int, int, int, int, int, void*, size_t);

static pw_se_rows_fn p_rows; static pw_se_fn p_pw_se; static rec_fc1_fn p_rec_fc1; static fc2_apply_fn p_fc2_apply;
static se_gate_fn p_gate; static se_apply_fn p_apply; static pack_bytes_fn p_pack_bytes; static pack_fn p_pack;

static uint32_t rng_state = 2024u;
static inline float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline float bfr(float f) { return bf2f(f2bf(f)); }
static void* to_dev(const void* h, size_t n) { void* d; CK(hipMalloc(&d, n ? n : 16)); if (n) CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); return d; }

template <typename F> static float time_us(F f, int reps, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1000.f / reps;
}

static int se_hidden(int C) { return std::max(16, C / 2); }      // reference :209

// CPU reference of the tail: y2 = bf16(ELU(W1 y1 + b)); per frame softmax_p(wk . y2_p + bk) pooling, gate MLP, out = gate * y2 + x
static void cpu_tail(int frames, int P, int C, int hidden, const std::vector<uint16_t>& y1, const std::vector<uint16_t>& x, const std::vector<float>& w1r,
                     const std::vector<float>& b1c, const std::vector<float>& wk, float bk, const std::vector<float>& w1t, const std::vector<float>& b1,
                     const std::vector<float>& w2t, const std::vector<float>& b2, std::vector<float>& y2f, std::vector<float>& ref) {
    const int64_t rows = (int64_t)frames * P;
    for (int64_t m = 0; m < rows; ++m) {
        const uint16_t* in = &y1[(size_t)m * C];
        for (int n = 0; n < C; ++n) { double a = 0; const float* wr = &w1r[(size_t)n * C]; for (int k = 0; k < C; ++k) a += (double)wr[k] * bf2f(in[k]);
            const float v = (float)a + b1c[n]; y2f[(size_t)m * C + n] = bfr(v > 0 ? v : expm1f(v)); }
    }
    for (int f = 0; f < frames; ++f) {
        std::vector<double> logit(P); double mx = -1e30;
        for (int p = 0; p < P; ++p) { double a = bk; const float* yr = &y2f[((size_t)f * P + p) * C]; for (int c2 = 0; c2 < C; ++c2) a += (double)wk[c2] * yr[c2]; logit[p] = a; mx = std::max(mx, a); }
        double se = 0; std::vector<double> pooled(C, 0.0);
        for (int p = 0; p < P; ++p) { const double e = exp(logit[p] - mx); se += e; const float* yr = &y2f[((size_t)f * P + p) * C]; for (int c2 = 0; c2 < C; ++c2) pooled[c2] += e * yr[c2]; }
        for (auto& v : pooled) v /= se;
        std::vector<double> hid(hidden), gate(C);
        for (int k = 0; k < hidden; ++k) { double a = b1[k]; for (int c2 = 0; c2 < C; ++c2) a += (double)w1t[(size_t)c2 * hidden + k] * pooled[c2]; hid[k] = a > 0 ? a : 0.1 * a; }
        for (int c2 = 0; c2 < C; ++c2) { double a = b2[c2]; for (int k = 0; k < hidden; ++k) a += (double)w2t[(size_t)k * C + c2] * hid[k]; gate[c2] = 1.0 / (1.0 + exp(-a)); }
        for (int p = 0; p < P; ++p) for (int c2 = 0; c2 < C; ++c2) { const size_t i = ((size_t)f * P + p) * C + c2; ref[i] = (float)(gate[c2] * y2f[i] + bf2f(x[i])); }
    }
}

static void dump(const std::string& dir, const char* name, const void* p, size_t n) {
    FILE* f = fopen((dir + "/" + name).c_str(), "wb"); if (!f) { printf("cannot write %s/%s\n", dir.c_str(), name); exit(2); } fwrite(p, 1, n, f); fclose(f);
}
static std::string g_dump;      // --dump <dir>: write the inputs and the CPU reference of the first small case, touch no GPU

// frames x P rows of C channels; returns true when every check passed
static bool run_case(int frames, int P, int C, bool do_cpu, int reps, hipStream_t st) {
    const int64_t rows = (int64_t)frames * P;
    const int hidden = se_hidden(C);
    const int bp = p_rows(MV2_BF16, rows, P, C);
    printf("== tail C=%d frames=%d P=%d (rows %lld): rows per record %d\n", C, frames, P, (long long)rows, bp);
    if (bp <= 0) { printf("   not eligible for mv2_pw_se\n"); return true; }
    const int nchunk = P / bp;
    // ---- host data (activations and weights rounded to bf16 where the kernels read bf16)
    std::vector<uint16_t> y1((size_t)rows * C), x((size_t)rows * C);
    for (auto& v : y1) v = f2bf(frand());
    for (auto& v : x) v = f2bf(frand());
    std::vector<float> w1((size_t)C * C), b1c(C), wk(C), w1t((size_t)C * hidden), b1(hidden), w2t((size_t)hidden * C), b2(C);
    const float ws = 1.7f / sqrtf((float)C);
    for (auto& v : w1) v = frand() * ws;
    for (auto& v : b1c) v = frand() * 0.1f;
    for (auto& v : wk) v = frand() * 0.3f;
    for (auto& v : w1t) v = frand() * (1.5f / sqrtf((float)C));
    for (auto& v : b1) v = frand() * 0.1f;
    for (auto& v : w2t) v = frand() * (1.5f / sqrtf((float)hidden));
    for (auto& v : b2) v = frand() * 0.5f;
    const float bk = 0.05f;
    const size_t wbytes = p_pack_bytes(MV2_BF16, C, C, 1, 1, 1);
    std::vector<uint8_t> wp(wbytes);
    if (p_pack(MV2_BF16, w1.data(), C, C, 1, 1, 1, 1 /* fragment-major */, wp.data(), wbytes) != 0) { printf("   mv2_pack_conv3d failed\n"); return false; }
    if (!g_dump.empty()) {      // CPU reference only (validated against the Python oracle by tools/check_se_bench_reference.py)
        std::vector<float> y2f((size_t)rows * C), ref((size_t)rows * C), w1r((size_t)C * C), y1f(y1.size()), xf(x.size());
        for (size_t i = 0; i < w1r.size(); ++i) w1r[i] = bfr(w1[i]);
        for (size_t i = 0; i < y1.size(); ++i) { y1f[i] = bf2f(y1[i]); xf[i] = bf2f(x[i]); }
        cpu_tail(frames, P, C, hidden, y1, x, w1r, b1c, wk, bk, w1t, b1, w2t, b2, y2f, ref);
        const int hdr[4] = {frames, P, C, hidden};
        dump(g_dump, "hdr.i32", hdr, sizeof hdr); dump(g_dump, "y1.f32", y1f.data(), y1f.size() * 4); dump(g_dump, "x.f32", xf.data(), xf.size() * 4);
        dump(g_dump, "w1.f32", w1r.data(), w1r.size() * 4); dump(g_dump, "b1c.f32", b1c.data(), C * 4); dump(g_dump, "wk.f32", wk.data(), C * 4);
        dump(g_dump, "w1t.f32", w1t.data(), w1t.size() * 4); dump(g_dump, "b1.f32", b1.data(), hidden * 4); dump(g_dump, "w2t.f32", w2t.data(), w2t.size() * 4);
        dump(g_dump, "b2.f32", b2.data(), C * 4); dump(g_dump, "y2.f32", y2f.data(), y2f.size() * 4); dump(g_dump, "ref.f32", ref.data(), ref.size() * 4);
        printf("   dumped to %s (bk = %.3f)\n", g_dump.c_str(), bk);
        return true;
    }
    void *d_y1 = to_dev(y1.data(), y1.size() * 2), *d_x = to_dev(x.data(), x.size() * 2), *d_wp = to_dev(wp.data(), wbytes);
    void *d_b1c = to_dev(b1c.data(), C * 4), *d_wk = to_dev(wk.data(), C * 4), *d_w1t = to_dev(w1t.data(), w1t.size() * 4), *d_b1 = to_dev(b1.data(), hidden * 4);
    void *d_w2t = to_dev(w2t.data(), w2t.size() * 4), *d_b2 = to_dev(b2.data(), C * 4);
    void *d_y2, *d_out, *d_part, *d_hid, *d_gate;
    CK(hipMalloc(&d_y2, (size_t)rows * C * 2)); CK(hipMalloc(&d_out, (size_t)rows * C * 2));
    CK(hipMalloc(&d_part, (size_t)frames * nchunk * (C + 2) * 4)); CK(hipMalloc(&d_hid, (size_t)frames * hidden * 4)); CK(hipMalloc(&d_gate, (size_t)frames * C * 4));
    // the plan's choice between the two forms (plan.py _se_two_launch + the LDS bound of mv2_se_rec_fc1)
    const bool pow2 = (C & (C - 1)) == 0 && (hidden & (hidden - 1)) == 0;
    const bool two = (size_t)P * C * 2 <= 256 * 1024 && pow2 && C >= 128 && ((size_t)nchunk * (C + 3) + C + 1024) * 4 <= 64 * 1024;
    const int s1 = hidden / std::max(16, hidden / 4), s2 = C / std::max(32, C / 8);
    auto launch_pw = [&] { return p_pw_se(MV2_BF16, d_y1, d_wp, (const float*)d_b1c, (const float*)d_wk, bk, d_y2, (float*)d_part, rows, P, C, st); };
    auto launch_a = [&] { int rc = p_rec_fc1((const float*)d_part, (const float*)d_w1t, (const float*)d_b1, (float*)d_hid, frames, C, nchunk, hidden, s1, st);
                          return rc ? rc : p_fc2_apply(MV2_BF16, d_y2, (const float*)d_hid, (const float*)d_w2t, (const float*)d_b2, d_x, d_out, frames, P, C, hidden, s2, st); };
    auto launch_b = [&] { int rc = p_gate((const float*)d_part, (const float*)d_w1t, (const float*)d_b1, (const float*)d_w2t, (const float*)d_b2, (float*)d_gate, frames, C, hidden, nchunk, st);
                          return rc ? rc : p_apply(MV2_BF16, d_y2, (const float*)d_gate, d_x, d_out, frames, P, C, st); };
    bool ok = true;
    std::vector<uint16_t> o_a((size_t)rows * C), o_b((size_t)rows * C), h_y2((size_t)rows * C);
    int rc = launch_pw(); CK(hipStreamSynchronize(st));
    if (rc) { printf("   mv2_pw_se rc=%d\n", rc); return false; }
    CK(hipMemcpy(h_y2.data(), d_y2, h_y2.size() * 2, hipMemcpyDeviceToHost));
    if (two) { CK(hipMemset(d_out, 0xff, (size_t)rows * C * 2)); rc = launch_a(); CK(hipStreamSynchronize(st)); if (rc) { printf("   rec_fc1 / fc2_apply rc=%d\n", rc); ok = false; }
               CK(hipMemcpy(o_a.data(), d_out, o_a.size() * 2, hipMemcpyDeviceToHost)); }
    CK(hipMemset(d_out, 0xff, (size_t)rows * C * 2)); rc = launch_b(); CK(hipStreamSynchronize(st)); if (rc) { printf("   se_gate / se_apply rc=%d\n", rc); ok = false; }
    CK(hipMemcpy(o_b.data(), d_out, o_b.size() * 2, hipMemcpyDeviceToHost));
    if (two) {      // the two forms consume the same records: they must agree to fp32 round-off of the tiny MLP (one bf16 ulp on the output at most)
        double dmax = 0; size_t nb = 0; for (size_t i = 0; i < o_a.size(); ++i) { const double e = fabs(bf2f(o_a[i]) - bf2f(o_b[i])); dmax = std::max(dmax, e); nb += e > 2e-2 * (1 + fabs(bf2f(o_b[i]))); }
        printf("   two-launch form vs gate + apply: max|d| %.3e bad %zu  %s\n", dmax, nb, nb ? "FAIL" : "ok"); ok &= nb == 0;
    }
    if (do_cpu) {
        std::vector<float> y2f((size_t)rows * C), ref((size_t)rows * C), w1r((size_t)C * C);
        for (size_t i = 0; i < w1r.size(); ++i) w1r[i] = bfr(w1[i]);
        cpu_tail(frames, P, C, hidden, y1, x, w1r, b1c, wk, bk, w1t, b1, w2t, b2, y2f, ref);
        double e_y2 = 0; size_t nb_y2 = 0; for (size_t i = 0; i < y2f.size(); ++i) { const double e = fabs(bf2f(h_y2[i]) - y2f[i]); e_y2 = std::max(e_y2, e); nb_y2 += e > 2e-2 * (1 + fabs(y2f[i])); }
        printf("   conv1 + ELU vs CPU: max|d| %.3e bad %zu / %zu  %s\n", e_y2, nb_y2, y2f.size(), nb_y2 ? "FAIL" : "ok"); ok &= nb_y2 == 0;
        for (int form = 0; form < 2; ++form) {
            if (form == 0 && !two) continue;
            const std::vector<uint16_t>& o = form == 0 ? o_a : o_b;
            double dmax = 0, dmean = 0; size_t nb = 0;
            for (size_t i = 0; i < ref.size(); ++i) { const double e = fabs(bf2f(o[i]) - ref[i]); dmax = std::max(dmax, e); dmean += e; nb += e > 3e-2 * (1 + fabs(ref[i])); }
            printf("   tail (%s) vs CPU: max|d| %.3e mean|d| %.3e bad %zu / %zu  %s\n", form == 0 ? "rec_fc1 + fc2_apply" : "gate + apply", dmax, dmean / ref.size(), nb, ref.size(), nb ? "FAIL" : "ok");
            ok &= nb == 0;
        }
    }
    if (reps > 0) {
        const float t_pw = time_us([&] { launch_pw(); }, reps, st);
        const float t_b = time_us([&] { launch_b(); }, reps, st);
        const float t_g = time_us([&] { p_gate((const float*)d_part, (const float*)d_w1t, (const float*)d_b1, (const float*)d_w2t, (const float*)d_b2, (float*)d_gate, frames, C, hidden, nchunk, st); }, reps, st);
        printf("   time: pw_se %.1f us | gate %.1f + apply %.1f = %.1f us", t_pw, t_g, t_b - t_g, t_b);
        if (two) {
            const float t_a = time_us([&] { launch_a(); }, reps, st);
            const float t_f = time_us([&] { p_rec_fc1((const float*)d_part, (const float*)d_w1t, (const float*)d_b1, (float*)d_hid, frames, C, nchunk, hidden, s1, st); }, reps, st);
            printf(" | rec_fc1 %.1f + fc2_apply %.1f = %.1f us", t_f, t_a - t_f, t_a);
            const float t_chain = time_us([&] { launch_pw(); launch_a(); }, reps, st);
            printf(" | chain (plan's form) %.1f us", t_chain);
        } else {
            const float t_chain = time_us([&] { launch_pw(); launch_b(); }, reps, st);
            printf(" | chain (plan's form) %.1f us", t_chain);
        }
        const double bytes = (double)rows * C * 2 * 5;      // y1 in, y2 out + in, x in, out out
        printf("  [algorithmic %.1f MB]\n", bytes / 1e6);
    }
    fflush(stdout);
    for (void* p : {d_y1, d_x, d_wp, d_b1c, d_wk, d_w1t, d_b1, d_w2t, d_b2, d_y2, d_out, d_part, d_hid, d_gate}) (void)hipFree(p);
    return ok;
}

int main(int argc, char** argv) {
    std::string lib = argc > 1 ? argv[1] : "magvit2_pytorch_amd/libmv2_hip.so";
    void* h = dlopen(lib.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) { printf("dlopen %s: %s\n", lib.c_str(), dlerror()); return 2; }
    p_rows = (pw_se_rows_fn)dlsym(h, "mv2_pw_se_rows"); p_pw_se = (pw_se_fn)dlsym(h, "mv2_pw_se"); p_rec_fc1 = (rec_fc1_fn)dlsym(h, "mv2_se_rec_fc1");
    p_fc2_apply = (fc2_apply_fn)dlsym(h, "mv2_se_fc2_apply"); p_gate = (se_gate_fn)dlsym(h, "mv2_se_gate"); p_apply = (se_apply_fn)dlsym(h, "mv2_se_apply");
    p_pack_bytes = (pack_bytes_fn)dlsym(h, "mv2_pack_conv3d_bytes"); p_pack = (pack_fn)dlsym(h, "mv2_pack_conv3d");
    if (!p_rows || !p_pw_se || !p_rec_fc1 || !p_fc2_apply || !p_gate || !p_apply || !p_pack_bytes || !p_pack) { printf("missing symbols\n"); return 2; }
    const bool levels_only = argc > 2 && !strcmp(argv[2], "levels");
    if (argc > 3 && !strcmp(argv[2], "--dump")) { g_dump = argv[3]; return run_case(3, 64, 128, true, 0, nullptr) ? 0 : 1; }
    hipStream_t st; CK(hipStreamCreate(&st));
    bool ok = true;
    if (!levels_only) {      // small shapes against the CPU reference: both record sizes, both forms, several frames
        ok &= run_case(3, 64, 128, true, 0, st);
        ok &= run_case(2, 256, 128, true, 0, st);
        ok &= run_case(3, 64, 256, true, 0, st);
        ok &= run_case(2, 1024, 256, true, 0, st);
        ok &= run_case(5, 64, 512, true, 0, st);
        ok &= run_case(2, 256, 512, true, 0, st);
    }
    // level shapes of BASELINE configs[1] (B = 1): frames x H*W at C = 128 / 256 / 512
    ok &= run_case(20, 4096, 128, false, 20, st);
    ok &= run_case(20, 1024, 256, false, 20, st);
    ok &= run_case(20, 256, 512, false, 20, st);
    ok &= run_case(10, 256, 512, false, 20, st);
    ok &= run_case(5, 256, 512, false, 20, st);
    if (argc > 2 && !strcmp(argv[2], "b4")) {      // the same levels at four clips per step (BASELINE configs[2] per-GPU shape)
        ok &= run_case(80, 4096, 128, false, 10, st);
        ok &= run_case(80, 1024, 256, false, 10, st);
        ok &= run_case(80, 256, 512, false, 10, st);
        ok &= run_case(40, 256, 512, false, 10, st);
        ok &= run_case(20, 256, 512, false, 10, st);
    }
    printf(ok ? "ALL OK\n" : "SOME FAILED\n");
    return ok ? 0 : 1;
}
