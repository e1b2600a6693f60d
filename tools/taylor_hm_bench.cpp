// tools/taylor_hm_bench.cpp - the Taylor state kernel (mv2_taylor_kv, nchunk = 1) on the projection GEMM's row layout against the same
// kernel on a head-major k / v tensor (experiments library, mv2x_taylor_kv_head_major): states compared bit for bit, time of both.
// Build: hipcc -O2 -std=c++17 tools/taylor_hm_bench.cpp -o tools/taylor_hm_bench -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int (*kv_fn)(int, const void*, float*, int, int, int, int, void*);
typedef int (*kvhm_fn)(int, const void*, float*, int, int, int, void*);
typedef int (*kvmfma_fn)(const void*, float*, int, int, int, int, void*);
static uint32_t rng_state = 7u;
static inline float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
template <typename F> static float time_us(F f, int reps, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1000.f / reps;
}
int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "magvit2_pytorch_amd/libmv2_hip_exp.so";
    void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { printf("dlopen %s: %s\n", lib, dlerror()); return 2; }
    kv_fn p_kv = (kv_fn)dlsym(h, "mv2_taylor_kv"); kvhm_fn p_hm = (kvhm_fn)dlsym(h, "mv2x_taylor_kv_head_major");
    if (!p_kv || !p_hm) { printf("missing symbols (experiments library?)\n"); return 2; }
    kvmfma_fn p_mf = (kvmfma_fn)dlsym(h, "mv2x_taylor_kv_mfma");
    const bool mfma_only = argc > 2 && !strcmp(argv[2], "mfma");
    typedef int (*kvsplit_fn)(const void*, float*, int, int, int, int, void*);
    kvsplit_fn p_sp = (kvsplit_fn)dlsym(h, "mv2x_taylor_kv_split");
    const bool split_only = argc > 2 && !strcmp(argv[2], "split");
    hipStream_t st; CK(hipStreamCreate(&st));
    const int heads = 16;
    struct Shp { const char* name; int frames, n; };
    const Shp shapes[] = {{"configs[1]: 20 frames x 1024 tokens", 20, 1024}, {"batch 4: 80 frames x 1024 tokens", 80, 1024}, {"ragged: 7 frames x 933 tokens", 7, 933},
                          {"tiny: 3 frames x 5 tokens", 3, 5}, {"two chunks + tail: 3 frames x 2100 tokens", 3, 2100}};
    for (const Shp& s : shapes) {
        const int ld = 2 * heads * 8;
        std::vector<uint16_t> rows((size_t)s.frames * s.n * ld), hm(rows.size());
        for (auto& v : rows) v = f2bf(frand());
        for (int f = 0; f < s.frames; ++f) for (int t = 0; t < s.n; ++t) for (int hh = 0; hh < heads; ++hh) for (int j = 0; j < 8; ++j) {
            const size_t r = ((size_t)f * s.n + t) * ld, d = (((size_t)f * heads + hh) * s.n + t) * 16;
            hm[d + j] = rows[r + hh * 8 + j]; hm[d + 8 + j] = rows[r + (heads + hh) * 8 + j];
        }
        void *drows, *dhm; float *s0, *s1; const size_t sb = (size_t)s.frames * heads * 73 * 9 * 4;
        CK(hipMalloc(&drows, rows.size() * 2)); CK(hipMalloc(&dhm, hm.size() * 2)); CK(hipMalloc((void**)&s0, sb)); CK(hipMalloc((void**)&s1, sb));
        CK(hipMemcpy(drows, rows.data(), rows.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dhm, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(s0, 0, sb)); CK(hipMemset(s1, 0xff, sb));
        auto a = [&] { return p_kv(0, drows, s0, s.frames, s.n, heads, 1, st); };
        auto b = [&] { return p_hm(0, dhm, s1, s.frames, s.n, heads, st); };
        if (split_only) {      // the VALU kernel on nchunk token ranges per (frame, head): partial states summed on the host against the whole-frame state
            if (!p_sp) { printf("no mv2x_taylor_kv_split\n"); return 2; }
            int r0 = a(); CK(hipStreamSynchronize(st));
            std::vector<float> ref(sb / 4);
            CK(hipMemcpy(ref.data(), s0, sb, hipMemcpyDeviceToHost));
            double amax = 0; for (float v : ref) amax = fabs(v) > amax ? fabs(v) : amax;
            printf("== %-42s whole frame %5.1f us (rc %d) |", s.name, time_us(a, 30, st), r0);
            for (int nc : {2, 4, 8}) {
                float* dp; CK(hipMalloc((void**)&dp, sb * nc)); CK(hipMemset(dp, 0xff, sb * nc));
                auto m = [&] { return p_sp(drows, dp, s.frames, s.n, heads, nc, st); };
                int rm = m(); CK(hipStreamSynchronize(st));
                std::vector<float> got((size_t)nc * sb / 4);
                CK(hipMemcpy(got.data(), dp, sb * nc, hipMemcpyDeviceToHost));
                double dmax = 0; const size_t per_fh = 73 * 9;
                for (int ff = 0; ff < s.frames; ++ff) for (int hh = 0; hh < heads; ++hh) for (size_t e = 0; e < per_fh; ++e) {
                    double sum = 0; for (int z = 0; z < nc; ++z) sum += got[(((size_t)ff * nc + z) * heads + hh) * per_fh + e];
                    const double d = fabs(sum - ref[((size_t)ff * heads + hh) * per_fh + e]); if (!(d <= dmax)) dmax = d;
                }
                printf(" %d ranges: rc %d %5.1f us (sum differs by %.1e of max) |", nc, rm, time_us(m, 30, st), dmax / (amax + 1e-30));
                hipFree(dp);
            }
            printf("\n");
            hipFree(drows); hipFree(dhm); hipFree(s0); hipFree(s1);
            continue;
        }
        if (mfma_only) {       // the state on the matrix pipe (one / two bf16 terms of phi) against the VALU kernel: max |difference| relative to max |state|
            if (!p_mf) { printf("no mv2x_taylor_kv_mfma\n"); return 2; }
            int r0 = a(); CK(hipStreamSynchronize(st));
            std::vector<float> ref(sb / 4), got(sb / 4);
            CK(hipMemcpy(ref.data(), s0, sb, hipMemcpyDeviceToHost));
            double amax = 0; for (float v : ref) amax = fabs(v) > amax ? fabs(v) : amax;
            printf("== %-42s (VALU kernel rc %d, max |state| %.3g)\n", s.name, r0, amax);
            for (int split = 1; split <= 2; ++split) {
                CK(hipMemset(s1, 0xff, sb));
                auto m = [&] { return p_mf(drows, s1, s.frames, s.n, heads, split, st); };
                int rm = m(); CK(hipStreamSynchronize(st));
                CK(hipMemcpy(got.data(), s1, sb, hipMemcpyDeviceToHost));
                double dmax = 0, dsum = 0; size_t nan = 0;
                for (size_t i = 0; i < ref.size(); ++i) { const double d = fabs((double)got[i] - ref[i]); if (!(d == d)) ++nan; else { dmax = d > dmax ? d : dmax; dsum += d; } }
                printf("   %d bf16 term(s) of phi: rc %d, max |d| %.3g (%.2e of max |state|), mean |d| %.3g, nan %zu |", split, rm, dmax, dmax / (amax + 1e-30), dsum / ref.size(), nan);
                for (int pass = 0; pass < 2; ++pass) printf(" VALU %5.1f us  MFMA %5.1f us |", time_us(a, 30, st), time_us(m, 30, st));
                printf("\n");
            }
            hipFree(drows); hipFree(dhm); hipFree(s0); hipFree(s1);
            continue;
        }
        int ra = a(), rb = b(); CK(hipStreamSynchronize(st));
        std::vector<uint32_t> h0(sb / 4), h1(sb / 4);
        CK(hipMemcpy(h0.data(), s0, sb, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), s1, sb, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < h0.size(); ++i) diff += h0[i] != h1[i];
        printf("== %-38s rc %d %d, differing state words %zu of %zu\n", s.name, ra, rb, diff, h0.size());
        for (int pass = 0; pass < 2; ++pass) printf("   pass %d: row layout %6.1f us | head-major %6.1f us\n", pass, time_us(a, 30, st), time_us(b, 30, st));
        hipFree(drows); hipFree(dhm); hipFree(s0); hipFree(s1);
    }
    return 0;
}
