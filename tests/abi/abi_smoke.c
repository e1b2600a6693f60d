/* Plain-C consumer of include/mv2.h: what a non-Python host (the FFI a maintainer would bind) needs from the
 * library without a GPU: the header compiles as C99, the library links, the version / sizing / packing entry points answer.
 * Built and run by tests/test_host_logic.py::test_c_consumer_builds_and_runs; no kernel is launched. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mv2.h"

int main(void) {
    if (mv2_abi_version() != MV2_ABI_VERSION) { printf("abi version mismatch\n"); return 1; }
    /* weight image of a 3x3x3 convolution 64 -> 64, bf16, plain and fragment-major: same size, same multiset of bytes */
    const int co = 64, ci = 64, kt = 3, kh = 3, kw = 3;
    const size_t n = (size_t)co * ci * kt * kh * kw, bytes = mv2_pack_conv3d_bytes(MV2_BF16, co, ci, kt, kh, kw);
    if (bytes != (size_t)27 * 128 * 128) { printf("unexpected image size %zu\n", bytes); return 2; }
    float* w = (float*)malloc(n * sizeof(float));
    unsigned char *a = (unsigned char*)malloc(bytes), *b = (unsigned char*)malloc(bytes);
    for (size_t i = 0; i < n; ++i) w[i] = (float)((int)(i % 251) - 125) / 64.0f;
    if (mv2_pack_conv3d(MV2_BF16, w, co, ci, kt, kh, kw, 0, a, bytes) != MV2_OK) return 3;
    if (mv2_pack_conv3d(MV2_BF16, w, co, ci, kt, kh, kw, 1, b, bytes) != MV2_OK) return 4;
    unsigned long sa = 0, sb = 0;
    for (size_t i = 0; i < bytes; ++i) { sa += a[i]; sb += b[i]; }
    if (sa != sb || memcmp(a, b, bytes) == 0) { printf("fragment-major image is not a permutation of the plain one\n"); return 5; }
    if (mv2_pack_conv3d(MV2_BF16, w, co, ci, kt, kh, kw, 0, a, bytes - 1) != MV2_EWORKSPACE) return 6;
    /* descriptor-level queries: the 3x3 ResidualUnit convolution at C = 64 on a (1, 20, 128, 128) clip */
    mv2_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = MV2_BF16; d.B = 1; d.Ti = d.To = 20; d.Hi = d.Ho = 128; d.Wi = d.Wo = 128; d.Ci = 64; d.co = 64; d.co_pad = 128;
    d.st = d.sh = d.sw = 1; d.pt = 2; d.ph = d.pw = 1; d.kt = d.kh = d.kw = 3; d.nsteps = 27; d.seg_bytes = 128; d.check_bounds = 1;
    d.act = MV2_ACT_ELU; d.store = MV2_STORE_ROWS; d.rt = d.rh = d.rw = 1; d.out_c = 64;
    if (!mv2_conv3_eligible(&d) || !mv2_conv3_fusable(&d)) { printf("conv3 eligibility\n"); return 7; }
    int th = 0, tw = 0;
    const int strips = mv2_conv3_tiles(&d, &th, &tw);
    if (strips != 64 * 8 || th != 2 || tw != 16) { printf("conv3 tiles %d %d %d\n", strips, th, tw); return 8; }
    if (mv2_conv3_counter_bytes(&d) == 0) return 9;
    /* malformed calls fail with an error code, they do not crash (no device is touched before validation) */
    if (mv2_conv3(&d, NULL, 0, NULL, NULL, NULL, NULL, 0, NULL, 0.f, NULL, NULL, NULL, 0, NULL, 0, NULL) == MV2_OK) return 10;
    if (mv2_lfq_stats_workspace_bytes(10, 13, 1) != 0) return 11;
    if (mv2_pw_se_rows(MV2_BF16, 81920, 4096, 128) != 64 || mv2_pw_se_rows(MV2_F32, 81920, 4096, 128) != 0) return 12;
    /* the decoder's output convolution: eligibility is a pure function of the descriptor; its weight image is kt * 18 k-steps of 1 KB */
    {
        mv2_conv_desc o = d;
        o.dtype = MV2_BF16; o.Ci = 64; o.co = 3; o.kt = 3; o.kh = 3; o.kw = 3; o.st = o.sh = o.sw = 1; o.pt = 2; o.ph = o.pw = 1; o.check_bounds = 1;
        o.To = o.Ti; o.Ho = o.Hi; o.Wo = o.Wi; o.store = MV2_STORE_NCTHW_F32; o.act = MV2_ACT_NONE; o.crop_t = 0;
        if (!mv2_conv_out_eligible(&o)) return 13;
        o.Ci = 128;
        if (mv2_conv_out_eligible(&o)) return 14;
        if (mv2_conv_out_weight_bytes(3) != 54 * 1024 || mv2_conv_out_weight_bytes(4) != 0) return 15;
        if (mv2_conv_out(&o, NULL, 0, NULL, NULL, NULL, NULL) == MV2_OK) return 16;
    }
    /* which pointwise launches take the LDS-DMA kernel: a pure function of the descriptor (feed-forward in of the C = 512 level: yes;
     * the same GEMM over a 16-row clip, a 3x3 kernel, fp32: no) */
    {
        mv2_conv_desc q = d;
        q.Ti = q.To = 20; q.Hi = q.Ho = q.Wi = q.Wo = 16; q.Ci = 512; q.co = 2816; q.co_pad = 2816; q.kt = q.kh = q.kw = 1; q.pt = q.ph = q.pw = 0;
        q.nsteps = 8; q.seg_bytes = 128; q.check_bounds = 0; q.act = MV2_ACT_GEGLU; q.store = MV2_STORE_ROWS; q.out_c = 1408;
        if (!mv2_conv_pointwise_dma(&q) || !mv2_conv_rms_ok(&q) || mv2_conv_workspace_bytes(&q) != 0) return 17;
        mv2_conv_desc r = q; r.Ti = r.To = 1; r.Hi = r.Ho = r.Wi = r.Wo = 4;
        if (mv2_conv_pointwise_dma(&r)) return 18;
        r = q; r.dtype = MV2_F32;
        if (mv2_conv_pointwise_dma(&r)) return 19;
        if (mv2_conv_pointwise_dma(&d)) return 20;
    }
    free(w); free(a); free(b);
    printf("abi smoke ok (%zu byte image, %d strips)\n", bytes, strips);
    return 0;
}
