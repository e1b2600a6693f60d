/* Plain-C consumer of include/mv2.h that LAUNCHES kernels: what tests/abi/abi_smoke.c cannot show without a GPU.  A host that is not
 * Python (the FFI a maintainer would bind) allocates device memory with the HIP runtime's C API, packs a weight
 * image with mv2_pack_conv3d, runs one mv2_conv (a 1x1x1 convolution = the FeedForward / projection GEMM of the path, reference
 * magvit2_pytorch.py:493-495) and one mv2_lfq launch (the quantizer call site M:1700) and compares every output element with values
 * computed here in C.  Operands are small dyadic rationals, so every partial sum is exact in fp32 whatever the order of the kernel's
 * reduction: the comparison is BIT-EXACT (except the negative ELU branch, a hardware exponential), and the 64-bit checksums FIX_* below
 * pin the expected result buffers (the fixture of this test, ~1 KB of ids + 10 KB of values condensed; the host half that produces them
 * also runs in the CPU suite: tests/test_host_logic.py builds this file with -DMV2_ABI_LAUNCH_HOST_ONLY).
 * Built with gcc and run by tests/test_parity_gpu.py::test_c_consumer_launches_kernels (-m gpu). */
#ifndef MV2_ABI_LAUNCH_HOST_ONLY
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 100; } } while (0)
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mv2.h"

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t fnv(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; }

/* the fixture: FNV-1a of the expected buffers (pre-activation fp32 of the convolution, LFQ ids, LFQ quantized values) */
#define FIX_CONV 0x3140ddd935d60d7cull
#define FIX_IDS 0x2251701bbb1868c3ull
#define FIX_Q 0x4b037b92085d0603ull

enum { M = 40, C = 64, CO = 64, CB = 10 };
static float w[CO * C], xf[M * C], bias[128], conv_ref[M * CO];
static uint16_t xb[M * C];
static float win[CB * C], bin[CB], wout_t[CB * C], bout[C], xq[M * C], q_ref[M * C], pre_ref[M * CB];
static int64_t ids_ref[M];

int main(void) {
    if (mv2_abi_version() != MV2_ABI_VERSION) { printf("abi version mismatch\n"); return 1; }
    /* ---- host half: test vectors and their expected results
     * mv2_conv: out[m][n] = elu(sum_k w[n][k] x[m][k] + bias[n]), 40 rows x 64 -> 64 channels, bf16 storage */
    for (int n = 0; n < CO; ++n) for (int k = 0; k < C; ++k) w[n * C + k] = (float)(((n * 7 + k * 3) % 17) - 8) / 16.0f;
    for (int m = 0; m < M; ++m) for (int k = 0; k < C; ++k) { xf[m * C + k] = (float)(((m * 5 + k * 11) % 13) - 6) / 8.0f; xb[m * C + k] = f2bf(xf[m * C + k]); }
    for (int n = 0; n < 128; ++n) bias[n] = n < CO ? (float)((n % 5) - 2) / 4.0f : 0.f;
    for (int m = 0; m < M; ++m) for (int n = 0; n < CO; ++n) {
        double acc = 0; for (int k = 0; k < C; ++k) acc += (double)bf2f(f2bf(w[n * C + k])) * (double)bf2f(xb[m * C + k]);
        conv_ref[m * CO + n] = (float)acc + bias[n];
    }
    /* mv2_lfq: z = W_in x + b_in (no clamp), ids from the signs (MSB first), quantized = W_out (+-1) + b_out; 40 tokens, C = 64, 10 bits, fp32 */
    for (int i = 0; i < CB; ++i) { bin[i] = (float)((i % 3) - 1) / 32.0f; for (int k = 0; k < C; ++k) { win[i * C + k] = (float)(((i * 5 + k * 7) % 11) - 5) / 8.0f; wout_t[i * C + k] = (float)(((i * 3 + k) % 9) - 4) / 16.0f; } }
    for (int k = 0; k < C; ++k) bout[k] = (float)((k % 7) - 3) / 8.0f;
    for (int m = 0; m < M; ++m) for (int k = 0; k < C; ++k) xq[m * C + k] = (float)(((m * 3 + k * 5) % 15) - 7) / 4.0f + (k == 0 ? 1.0f / 64.0f : 0.f);
    for (int m = 0; m < M; ++m) {
        int64_t id = 0; float code[CB];
        for (int i = 0; i < CB; ++i) {
            double z = bin[i]; for (int k = 0; k < C; ++k) z += (double)win[i * C + k] * (double)xq[m * C + k];
            if ((float)z == 0.f) { printf("degenerate test vector (z == 0)\n"); return 2; }
            pre_ref[m * CB + i] = (float)z; code[i] = z > 0 ? 1.f : -1.f;
            id |= (int64_t)(z > 0) << (CB - 1 - i);
        }
        ids_ref[m] = id;
        for (int k = 0; k < C; ++k) { double o = bout[k]; for (int i = 0; i < CB; ++i) o += (double)wout_t[i * C + k] * code[i]; q_ref[m * C + k] = (float)o; }
    }
    const uint64_t h_conv = fnv(conv_ref, sizeof(conv_ref)), h_ids = fnv(ids_ref, sizeof(ids_ref)), h_q = fnv(q_ref, sizeof(q_ref));
    printf("fixture checksums: conv %016llx ids %016llx quantized %016llx\n", (unsigned long long)h_conv, (unsigned long long)h_ids, (unsigned long long)h_q);
    if (h_conv != FIX_CONV || h_ids != FIX_IDS || h_q != FIX_Q) { printf("expected buffers differ from the fixture checksums\n"); return 3; }
    const size_t wbytes = mv2_pack_conv3d_bytes(MV2_BF16, CO, C, 1, 1, 1);
    if (wbytes != (size_t)128 * 128) { printf("unexpected image size %zu\n", wbytes); return 4; }
    unsigned char* wp = (unsigned char*)malloc(wbytes);
    if (mv2_pack_conv3d(MV2_BF16, w, CO, C, 1, 1, 1, 0, wp, wbytes) != MV2_OK) return 5;
    mv2_conv_desc d; memset(&d, 0, sizeof(d));
    d.dtype = MV2_BF16; d.B = 1; d.Ti = d.To = 1; d.Hi = d.Ho = 5; d.Wi = d.Wo = 8; d.Ci = C; d.co = CO; d.co_pad = 128;
    d.st = d.sh = d.sw = 1; d.kt = d.kh = d.kw = 1; d.nsteps = 1; d.seg_bytes = 128; d.check_bounds = 0; d.act = MV2_ACT_ELU; d.store = MV2_STORE_ROWS;
    d.rt = d.rh = d.rw = 1; d.out_c = CO;
    const size_t wsb = mv2_conv_workspace_bytes(&d);
#ifdef MV2_ABI_LAUNCH_HOST_ONLY
    printf("abi launch (host half) ok: workspace %zu bytes\n", wsb);
    free(wp);
    return 0;
#else
    /* ---- device half */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { printf("no HIP device\n"); return 6; }
    hipStream_t st; CK(hipStreamCreate(&st));
    mv2_seg seg; seg.elem_offset = 0; seg.sel = 1u | (1u << 8) | (1u << 16);
    void *d_x, *d_wp, *d_seg, *d_bias, *d_out, *d_ws;
    CK(hipMalloc(&d_x, sizeof(xb))); CK(hipMalloc(&d_wp, wbytes)); CK(hipMalloc(&d_seg, sizeof(seg))); CK(hipMalloc(&d_bias, sizeof(bias)));
    CK(hipMalloc(&d_out, M * CO * 2)); CK(hipMalloc(&d_ws, wsb ? wsb : 16));
    CK(hipMemcpy(d_x, xb, sizeof(xb), hipMemcpyHostToDevice)); CK(hipMemcpy(d_wp, wp, wbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_seg, &seg, sizeof(seg), hipMemcpyHostToDevice)); CK(hipMemcpy(d_bias, bias, sizeof(bias), hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0xff, M * CO * 2));
    int rc = mv2_conv(&d, d_x, sizeof(xb), d_wp, (const mv2_seg*)d_seg, (const float*)d_bias, NULL, NULL, d_out, d_ws, wsb, (void*)st);
    if (rc != MV2_OK) { printf("mv2_conv rc=%d\n", rc); return 7; }
    CK(hipStreamSynchronize(st));
    static uint16_t out[M * CO];
    CK(hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < M * CO; ++i) {
        const float v = conv_ref[i], got = bf2f(out[i]);
        /* positive branch: exactly the bf16 rounding of the exact sum; negative branch: exp(v) - 1 through the hardware exponential */
        if (v > 0.f) { if (out[i] != f2bf(v)) ++bad; }
        else { const float want = (float)(__builtin_exp((double)v) - 1.0); if (!(__builtin_fabsf(got - want) <= 0.016f * __builtin_fabsf(want) + 1e-3f)) ++bad; }
    }
    if (bad) { printf("mv2_conv: %d of %d elements differ\n", bad, M * CO); return 8; }

    void *d_xq, *d_win, *d_bin, *d_wout, *d_bout, *d_q, *d_ids, *d_pre;
    CK(hipMalloc(&d_xq, sizeof(xq))); CK(hipMalloc(&d_win, sizeof(win))); CK(hipMalloc(&d_bin, sizeof(bin))); CK(hipMalloc(&d_wout, sizeof(wout_t)));
    CK(hipMalloc(&d_bout, sizeof(bout))); CK(hipMalloc(&d_q, sizeof(xq))); CK(hipMalloc(&d_ids, M * 8)); CK(hipMalloc(&d_pre, M * CB * 4));
    CK(hipMemcpy(d_xq, xq, sizeof(xq), hipMemcpyHostToDevice)); CK(hipMemcpy(d_win, win, sizeof(win), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bin, bin, sizeof(bin), hipMemcpyHostToDevice)); CK(hipMemcpy(d_wout, wout_t, sizeof(wout_t), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bout, bout, sizeof(bout), hipMemcpyHostToDevice));
    rc = mv2_lfq(MV2_F32, d_xq, (const float*)d_win, (const float*)d_bin, (const float*)d_wout, (const float*)d_bout, d_q, (int64_t*)d_ids, (float*)d_pre,
                 M, C, CB, 1, 0.f, 1.f, (void*)st);
    if (rc != MV2_OK) { printf("mv2_lfq rc=%d\n", rc); return 9; }
    CK(hipStreamSynchronize(st));
    static int64_t ids[M]; static float q[M * C], pre[M * CB];
    CK(hipMemcpy(ids, d_ids, sizeof(ids), hipMemcpyDeviceToHost)); CK(hipMemcpy(q, d_q, sizeof(q), hipMemcpyDeviceToHost)); CK(hipMemcpy(pre, d_pre, sizeof(pre), hipMemcpyDeviceToHost));
    if (memcmp(ids, ids_ref, sizeof(ids)) != 0) { printf("mv2_lfq: ids differ\n"); return 10; }
    if (memcmp(pre, pre_ref, sizeof(pre)) != 0) { printf("mv2_lfq: projected activations differ\n"); return 11; }
    if (memcmp(q, q_ref, sizeof(q)) != 0) { printf("mv2_lfq: quantized values differ\n"); return 12; }
    if (fnv(ids, sizeof(ids)) != FIX_IDS || fnv(q, sizeof(q)) != FIX_Q) { printf("device buffers differ from the fixture checksums\n"); return 13; }
    printf("abi launch ok: mv2_conv %d x %d elements, mv2_lfq %d tokens bit-exact\n", M, CO, M);
    hipFree(d_x); hipFree(d_wp); hipFree(d_seg); hipFree(d_bias); hipFree(d_out); hipFree(d_ws);
    hipFree(d_xq); hipFree(d_win); hipFree(d_bin); hipFree(d_wout); hipFree(d_bout); hipFree(d_q); hipFree(d_ids); hipFree(d_pre);
    free(wp);
    return 0;
#endif
}
