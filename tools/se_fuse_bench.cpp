// tools/se_fuse_bench.cpp - mv2_se_gate + mv2_se_apply (two launches) against mv2x_se_gate_apply (experiments library: every block of a
// frame repeats the gate computation, then streams its share of the frame) on the large-frame SqueezeExcite shapes of BASELINE configs[1]
// and of the batch-4 shape: outputs compared bit for bit, time per unit with the two forms interleaved.
// Build (no GPU needed):  hipcc -O2 -std=c++17 tools/se_fuse_bench.cpp -o tools/se_fuse_bench -ldl
// Run:  tools/se_fuse_bench magvit2_pytorch_amd/libmv2_hip_exp.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int (*gate_fn)(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, int, void*);
typedef int (*apply_fn)(int, const void*, const float*, const void*, void*, int, int, int, void*);
typedef int (*fused_fn)(int, const float*, const float*, const float*, const float*, const float*, const void*, const void*, void*, int, int, int, int, int, int, void*);
static uint32_t rng_state = 1u;
static inline float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static void* to_dev(const void* h, size_t n) { void* d; CK(hipMalloc(&d, n)); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); return d; }
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
    gate_fn p_gate = (gate_fn)dlsym(h, "mv2_se_gate"); apply_fn p_apply = (apply_fn)dlsym(h, "mv2_se_apply"); fused_fn p_fused = (fused_fn)dlsym(h, "mv2x_se_gate_apply");
    if (!p_gate || !p_apply || !p_fused) { printf("missing symbols (experiments library?)\n"); return 2; }
    gate_fn p_gate2 = (gate_fn)dlsym(h, "mv2x_se_gate_v2");
    const bool gate_only = argc > 2 && !strcmp(argv[2], "gate");
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Shp { const char* name; int frames, P, C, nchunk; };
    const Shp shapes[] = {{"C=64 128x128 (conv3 records)", 20, 16384, 64, 512}, {"C=128 64x64", 20, 4096, 128, 64}, {"C=256 32x32", 20, 1024, 256, 32},
                          {"B4 C=64 128x128", 80, 16384, 64, 512}, {"B4 C=128 64x64", 80, 4096, 128, 64}, {"B4 C=256 32x32", 80, 1024, 256, 32}};
    for (const Shp& s : shapes) {
        const int hidden = s.C / 2 > 16 ? s.C / 2 : 16;
        const size_t nrec = (size_t)s.frames * s.nchunk * (s.C + 2), nel = gate_only ? 1024 : (size_t)s.frames * s.P * s.C;
        std::vector<float> part(nrec), w1t((size_t)s.C * hidden), b1(hidden), w2t((size_t)hidden * s.C), b2(s.C);
        for (size_t r = 0; r < (size_t)s.frames * s.nchunk; ++r) {
            float* q = &part[r * (s.C + 2)];
            q[0] = frand() * 2.f; q[1] = 1.f + fabsf(frand()) * 30.f;
            for (int c = 0; c < s.C; ++c) q[2 + c] = frand() * q[1];
        }
        for (auto& v : w1t) v = frand() * 0.2f; for (auto& v : b1) v = frand() * 0.1f; for (auto& v : w2t) v = frand() * 0.2f; for (auto& v : b2) v = frand() * 0.1f;
        std::vector<uint16_t> y(nel), res(nel);
        for (auto& v : y) v = f2bf(frand()); for (auto& v : res) v = f2bf(frand());
        float *dpart = (float*)to_dev(part.data(), nrec * 4), *dw1 = (float*)to_dev(w1t.data(), w1t.size() * 4), *db1 = (float*)to_dev(b1.data(), b1.size() * 4),
              *dw2 = (float*)to_dev(w2t.data(), w2t.size() * 4), *db2 = (float*)to_dev(b2.data(), b2.size() * 4);
        void *dy = to_dev(y.data(), nel * 2), *dres = to_dev(res.data(), nel * 2), *dout0, *dout1; float* dgate;
        CK(hipMalloc(&dout0, nel * 2)); CK(hipMalloc(&dout1, nel * 2)); CK(hipMalloc((void**)&dgate, (size_t)s.frames * s.C * 4));
        auto two = [&] { int rc = p_gate(dpart, dw1, db1, dw2, db2, dgate, s.frames, s.C, hidden, s.nchunk, st); if (!rc) rc = p_apply(0, dy, dgate, dres, dout0, s.frames, s.P, s.C, st); return rc; };
        if (gate_only) {       // mv2_se_gate against the variant with its independent loads hoisted (same sums): gates bit for bit, us per launch
            if (!p_gate2) { printf("no mv2x_se_gate_v2\n"); return 2; }
            float* dgate2; CK(hipMalloc((void**)&dgate2, (size_t)s.frames * s.C * 4)); CK(hipMemset(dgate2, 0xff, (size_t)s.frames * s.C * 4));
            auto g1 = [&] { return p_gate(dpart, dw1, db1, dw2, db2, dgate, s.frames, s.C, hidden, s.nchunk, st); };
            auto g2 = [&] { return p_gate2(dpart, dw1, db1, dw2, db2, dgate2, s.frames, s.C, hidden, s.nchunk, st); };
            int r1 = g1(), r2 = g2(); CK(hipStreamSynchronize(st));
            std::vector<uint32_t> a((size_t)s.frames * s.C), b2v(a.size());
            CK(hipMemcpy(a.data(), dgate, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2v.data(), dgate2, a.size() * 4, hipMemcpyDeviceToHost));
            size_t diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b2v[i];
            printf("== gate only %-28s rc %d %d, differing gate words %zu of %zu:", s.name, r1, r2, diff, a.size());
            for (int pass = 0; pass < 2; ++pass) printf("  se_gate %5.1f us -> hoisted loads %5.1f us |", time_us(g1, 50, st), time_us(g2, 50, st));
            printf("\n");
            hipFree(dgate2); hipFree(dpart); hipFree(dw1); hipFree(db1); hipFree(dw2); hipFree(db2); hipFree(dy); hipFree(dres); hipFree(dout0); hipFree(dout1); hipFree(dgate);
            continue;
        }
        int rc = two(); CK(hipStreamSynchronize(st));
        if (rc) { printf("%s: two-launch form rc=%d\n", s.name, rc); continue; }
        std::vector<uint16_t> o0(nel), o1(nel);
        CK(hipMemcpy(o0.data(), dout0, nel * 2, hipMemcpyDeviceToHost));
        printf("== %-30s frames %d, %.1f MB per tensor\n", s.name, s.frames, nel * 2 / 1048576.0);
        const int nbs[] = {4, 8, 16, 32, 64};
        for (int pass = 0; pass < 2; ++pass) {
            const float t2 = time_us(two, 30, st);
            printf("   pass %d: gate + apply %6.1f us |", pass, t2);
            for (int nb : nbs) {
                CK(hipMemset(dout1, 0xff, nel * 2));
                auto one = [&] { return p_fused(0, dpart, dw1, db1, dw2, db2, dy, dres, dout1, s.frames, s.P, s.C, hidden, s.nchunk, nb, st); };
                rc = one(); CK(hipStreamSynchronize(st));
                if (rc) { printf(" nb %d rc=%d |", nb, rc); continue; }
                CK(hipMemcpy(o1.data(), dout1, nel * 2, hipMemcpyDeviceToHost));
                size_t diff = 0; for (size_t i = 0; i < nel; ++i) diff += o0[i] != o1[i];
                const float t1 = time_us(one, 30, st);
                printf(" %d blocks/frame %6.1f%s |", nb, t1, diff ? " DIFFERENT" : "");
            }
            printf("\n");
        }
        hipFree(dpart); hipFree(dw1); hipFree(db1); hipFree(dw2); hipFree(db2); hipFree(dy); hipFree(dres); hipFree(dout0); hipFree(dout1); hipFree(dgate);
    }
    return 0;
}
