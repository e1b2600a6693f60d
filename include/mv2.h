/*
 * mv2.h — C ABI of libmv2_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * VideoTokenizer encode -> quantize -> decode hot path.
 *
 * The reference (lucidrains/magvit2-pytorch) has no FFI / plugin interface: every FLOP is a
 * stock ATen op reached from Python.  This header is therefore the boundary a
 * maintainer would bind from `magvit2_pytorch/magvit2_pytorch.py` (ctypes stub shown in
 * INTEGRATION.md); each entry point names the reference lines whose ATen calls it replaces
 * (M = magvit2_pytorch/magvit2_pytorch.py, A = magvit2_pytorch/attend.py).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (tensor.data_ptr()) unless the name ends in `_host`;
 *  - activations are dense channel-last NDHWC: element (b,t,h,w,c) at ((((b*T+t)*H+h)*W+w)*C+c);
 *    "rows" are the B*T*H*W positions, a row is C contiguous channels;
 *  - `dtype` selects the storage + MFMA input type of activations and packed weights:
 *      MV2_BF16  bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 accumulate   (fast mode)
 *      MV2_F32   fp32 storage, v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) (id-parity mode)
 *    biases, norms' statistics, gates, softmax and quantizer arithmetic are fp32 in both;
 *  - no hidden allocation, no global state, re-entrant per stream: the caller owns all
 *    buffers, passes workspaces explicitly and gets the launch on `stream` (hipStream_t);
 *  - return value: 0 on success, a negative MV2_E* code on a rejected argument, or the
 *    positive hipError_t of a failed launch.  No exceptions cross the ABI.
 */
#ifndef MV2_H
#define MV2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV2_ABI_VERSION 2

enum { MV2_BF16 = 0, MV2_F32 = 1 };

enum {
    MV2_OK = 0,
    MV2_EINVAL = -1,      /* malformed descriptor (alignment / divisibility rules below) */
    MV2_ETOOLARGE = -2,   /* a buffer exceeds the 2^32-byte range of 32-bit buffer offsets */
    MV2_EWORKSPACE = -3,  /* workspace smaller than mv2_*_workspace_bytes() */
    MV2_ENODEVICE = -4
};

/* activation applied in the conv epilogue */
enum { MV2_ACT_NONE = 0, MV2_ACT_ELU = 1, MV2_ACT_SILU = 2, MV2_ACT_GEGLU = 3, MV2_ACT_LRELU = 4 /* leaky_relu(0.1), discriminator M:117 */ };
/* store mode of the conv epilogue */
enum {
    MV2_STORE_ROWS = 0,      /* out[row][n]                                   dense NDHWC          */
    MV2_STORE_SHUFFLE = 1,   /* depth-to-space/time: packed column n = q*Cq + c goes to
                                position (t*rt+qt, h*rh+qh, w*rw+qw), channel c                    */
    MV2_STORE_NCTHW_F32 = 2, /* fp32 NCTHW video with the first `crop_t` frames dropped            */
    MV2_STORE_ROWS_F32 = 3   /* out[row][n] as fp32 out of a bf16 launch, nothing rounded: the split-bf16
                                form of the ResidualUnit convolution (mv2_conv3 only, see mv2_split_bf16) */
};

int mv2_abi_version(void);
/* number of compute units of the current device (grid sizing); <0 on error */
int mv2_device_cus(void);

/* ---------------------------------------------------------------------------------------------
 * mv2_conv — implicit-GEMM convolution on MFMA with a fused epilogue.
 * Replaces: CausalConv3d.forward M:924-928 (F.pad + nn.Conv3d: conv_in M:1557, ResidualUnit k3
 * M:937, conv_out M:1642), nn.Conv3d 1x1x1 M:939 / M:493-495, SpatialDownsample2x conv M:776,
 * TimeDownsample2x pad+conv M:802-803, SpatialUpsample2x / TimeUpsample2x conv+SiLU+Rearrange
 * M:819-825 / M:856-862, nn.Linear to_qkv / to_out M:351-368 and the Taylor-attention linears,
 * ELU M:938,940, GEGLU M:466-469, Residual add M:174, RMSNorm scaling M:276 (as `rowscale`),
 * TokenShift M:250-253 (as a two-segment K program).
 *
 *   D[n][row] = sum_k Wp[k][n] * A[row][k]       (n = output channel, fp32 accumulate)
 * K is a host-built program of `nsteps` steps; one step is 128 bytes of a row's K extent
 * (64 bf16 / 32 fp32 elements) made of 128/seg_bytes equal segments.  Segment s of the program
 * reads seg_bytes contiguous bytes of the input at
 *      element offset  rowbase(row) + segs[s].elem_offset ,
 *      rowbase(b,to,ho,wo) = (((b*Ti + to*st - pt)*Hi + ho*sh - ph)*Wi + wo*sw - pw)*Ci
 * and contributes zeros when check_bounds != 0 and the tap (it,ih,iw) packed in segs[s].sel
 * falls outside [0,Ti)x[0,Hi)x[0,Wi) (causal front pad in time, zero "same" pad in space), or
 * when sel has bit 31 set (filler segment).  sel = 1<<it | 1<<(8+ih) | 1<<(16+iw).
 * Packed weights Wp: [nsteps][co_pad][128 bytes], row n of step s holds the weights matching
 * that step's segments, zero in padding rows / columns.  (See magvit2_pytorch_amd/pack.py.)
 *
 * Epilogue per output element v (fp32):  v *= rowscale[row] (if given); v += bias[n] (if given);
 * act; v += residual[row][n] (if given, same dtype/layout as out, MV2_STORE_ROWS only); store.
 * MV2_ACT_GEGLU: packed columns come in blocks of 64: 32 value columns then their 32 gate
 * columns; out channel j = 32*blk + i gets gelu(gate) * value and `co` is the number of OUTPUT
 * channels written (padded hidden width; padding columns have zero weights -> exact zeros).
 * ------------------------------------------------------------------------------------------- */
typedef struct mv2_seg {
    int32_t elem_offset;   /* added to rowbase, in elements */
    uint32_t sel;          /* tap selection bits, see above */
} mv2_seg;

typedef struct mv2_conv_desc {
    int32_t dtype;
    int32_t B, Ti, Hi, Wi, Ci;     /* input geometry; Ci = row length (channel stride) in elements */
    int32_t To, Ho, Wo;            /* output positions */
    int32_t co;                    /* valid packed output columns (2*hidden_pad for GEGLU) */
    int32_t co_pad;                /* rows per step in Wp, multiple of 128 */
    int32_t st, sh, sw;            /* strides */
    int32_t pt, ph, pw;            /* leading pads */
    int32_t kt, kh, kw;            /* taps per axis (<= 8 each), used for the bounds masks */
    int32_t nsteps;
    int32_t seg_bytes;             /* 16, 32, 64 or 128 */
    int32_t check_bounds;
    int32_t act;                   /* MV2_ACT_*   */
    int32_t store;                 /* MV2_STORE_* */
    int32_t rt, rh, rw;            /* MV2_STORE_SHUFFLE factors (1 otherwise) */
    int32_t crop_t;                /* MV2_STORE_NCTHW_F32: leading frames dropped */
    int32_t out_c;                 /* channels per output row (row stride of `out`) */
    int32_t splitk;                /* 0 = let the library choose, >=1 forced */
} mv2_conv_desc;

size_t mv2_conv_workspace_bytes(const mv2_conv_desc* d);
/* 1 when mv2_conv / mv2_conv_rms run this descriptor on the LDS-DMA pointwise kernel (conv.hip conv_pw_dma_kernel: bf16, 1x1x1, stride 1,
 * un-padded, whole 128-byte K steps, staged stores; GEGLU launches from 256 tiles of 128 x 128 on, every other epilogue from 32 on - the
 * kernel's own tile is 128 x 128 / 128 x 64 / 64 x 128 / 64 x 64 by launch size): same operands, same results, never splits K.
 * Introspection for tests and profiles; callers need not ask. */
int mv2_conv_pointwise_dma(const mv2_conv_desc* d);
/* 1 when mv2_conv (not mv2_conv_rms) runs this descriptor on the GATHER form of that kernel: bf16 convolutions with taps / strides / zero
 * padding whose K steps are whole 128-byte pieces of one input row (Ci % 64 == 0, seg_bytes 128), staged stores, no GEGLU, from 32 tiles of
 * 128 x 128 on: the space / time down-sampling convolutions (reference magvit2_pytorch.py:770-807).  The launch never splits K
 * (mv2_conv_workspace_bytes == 0) and its result is bit-identical to the generic kernel's un-split one.  Introspection only. */
int mv2_conv_gather_dma(const mv2_conv_desc* d);
int mv2_conv(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp, const mv2_seg* segs,
             const float* bias, const float* rowscale, const void* residual, void* out,
             void* workspace, size_t workspace_bytes, void* stream);

/* mv2_conv with the RMSNorm row scale computed INSIDE the launch (reference RMSNorm M:258-276: x / max(|x|, 1e-12) per position, the
 * gamma * sqrt(dim) factor being folded into the packed weights): rowscale[m] = 1 / max(|gathered row m|, 1e-12) over the whole
 * reduction extent, taken from the rows the GEMM stages anyway — one launch and one pass over x less than mv2_rms_rowscale + mv2_conv.
 * Needs the reduction extent to be exactly the row (no zero-padded K tail: nsteps * 128 bytes == taps * Ci * elem bytes — the caller's
 * responsibility) and mv2_conv_rms_ok(desc) == 1 (no split-K for this shape, staged 16-byte stores). */
int mv2_conv_rms_ok(const mv2_conv_desc* desc);
int mv2_conv_rms(const mv2_conv_desc* desc, const void* in, size_t in_bytes, const void* wp, const mv2_seg* segs,
                 const float* bias, const void* residual, void* out, void* stream);

/* mv2_conv_in — the encoder's input convolution (reference M:1109, 1557: CausalConv3d(channels -> 64, kt x 7 x 7)) over the pre-padded
 * channel-last clip of mv2_video_to_ndhwc (4-channel bf16 pixels): `d` is the descriptor mv2_conv takes for the same launch (Ci = 4,
 * kh = 7 row taps of kw = 1 "wide" tap each as pack.py::pack_conv_in lays them out, no padding, check_bounds = 0, input extents =
 * output extents + (kt - 1, 6, 6), Wi even, co = out_c = 64, MV2_STORE_ROWS, no activation); mv2_conv_in_eligible() says whether it
 * qualifies (else: mv2_conv).  A workgroup stages the kt halo frames of an 8 x 32 output tile in LDS once instead of gathering every
 * input row once per (time, height) tap.
 * wfrag: [kt][7 ih][2 steps s][2 channel tiles][64 lanes][8 bf16]: lane l holds, for output channel 32 tile + l % 32, the weights of
 * pixels iw = 4 s + 2 (l / 32) + {0, 1} x 4 channels of tap (it, ih) (zero for iw = 7 and for channels the clip does not have):
 * the A operand of v_mfma_f32_32x32x16_bf16; mv2_conv_in_weight_bytes(kt) bytes. */
int mv2_conv_in_eligible(const mv2_conv_desc* d);
size_t mv2_conv_in_weight_bytes(int kt);
int mv2_conv_in(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wfrag, const float* bias, void* out, void* stream);

/* mv2_conv_out — the decoder's output convolution (reference M:1127, 1642: CausalConv3d(dim -> channels, kt x 3 x 3), zero padding) as
 * an HBM stream: bf16, Ci = 64, co <= 4, kt <= 3, stride 1, 'same' in space / causal in time, fp32 NCTHW store with `crop_t` leading
 * frames dropped (MV2_STORE_NCTHW_F32).  `d` is the descriptor mv2_conv would take for the same layer (pt = kt - 1, ph = pw = 1,
 * check_bounds = 1); mv2_conv_out_eligible() says whether it qualifies (else: mv2_conv).  A workgroup stages the kt halo frames of an
 * 8 x 16 output tile in LDS once, instead of gathering every input row once per tap.
 * wfrag: [kt * 9 taps (it, ih, iw row-major)][2 channel halves][64 lanes][8 bf16]: lane l holds W[co = l % 16][ci = 32 half + 8 (l / 16) + j]
 * of the tap, zero for co >= d->co (the A operand of v_mfma_f32_16x16x32_bf16); mv2_conv_out_weight_bytes(kt) bytes. */
int mv2_conv_out_eligible(const mv2_conv_desc* d);
size_t mv2_conv_out_weight_bytes(int kt);
int mv2_conv_out(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wfrag, const float* bias, void* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * mv2_conv3 — the ResidualUnit convolution: stride-1 kt x 3 x 3 (kt <= 3) causal-in-time / "same"-in-space
 * CausalConv3d over whole 128-byte channel chunks, C_out >= 64.
 * Replaces: ResidualUnit's CausalConv3d + ELU M:937-938 (F.pad M:927 + nn.Conv3d), and — when `wp2` is given —
 * also nn.Conv3d(dim, dim, 1) + ELU M:939-940 and the softmax pooling of SqueezeExcite M:226-231 of the same
 * unit, without writing the intermediate tensor:  out = act2(W2 . act(conv(in) + bias) + bias2).
 *
 * `d` describes the convolution exactly as for mv2_conv (st = sh = sw = 1, ph = pw = 1, pt = kt - 1, check_bounds = 1,
 * nsteps = kt * 9 * C_in*esize/128); mv2_conv3_eligible() says whether a descriptor qualifies (else: mv2_conv).
 * wp3: the same [nsteps][co_pad][128 B] weights as mv2_conv's Wp (step = ((it*3 + ih)*3 + iw) * nchunk + chunk), but each
 *      group of 32 output rows is stored in MFMA-fragment order: [step][co_pad/32][k-slice 4][half 2][row 32][16 B],
 *      where the 16-byte chunk (2*k-slice + half) of row r of the plain layout is meant (pack.py pack_conv3d(frag_major)).
 * Epilogue: v += bias[n]; act (MV2_ACT_NONE or MV2_ACT_ELU); store rows (MV2_STORE_ROWS, out_c >= co).
 * Fused second stage (wp2 != NULL): bf16 only, C_out = 64 (a workgroup then owns every channel of its positions;
 *      mv2_conv3_fusable()); wp2 = the pointwise weights packed like any 1x1x1 conv ([1][128][128 B]); act2 NONE or ELU.
 *      se_part (optional, with se_wk / se_bk = SqueezeExcite to_k weight (C) and bias): the kernel also writes the
 *      pooling partials of every 2 x 16 strip of positions of every frame in mv2_se_pool's record format
 *        se_part[(frame * n + j) * (C + 2)] = {max logit, sum exp, exp-weighted channel sums[C]},
 *      n = mv2_conv3_tiles(d, &th, &tw) strips of th x tw = 2 x 16 positions per frame, j = strip_row * ceil(W/16) + strip_col.
 * Work split: one workgroup per (64-channel x 8x16-position tile, K part): launches with fewer tiles than the chip has workgroup
 *      slots (the deepest levels at batch 1) cut every tile's (chunk, time-tap) list into parts; the parts of a tile are combined
 *      in-kernel (fp32 partials in `workspace`, a ticket per tile in `counters`, summed in part order: results are bitwise
 *      reproducible).  (bf16: conv3p.hip; the exact-fp32 mode walks equal shares of the list per CU, conv3.hip - same contract.)
 *      workspace: mv2_conv3_workspace_bytes(d) bytes (0 when no tile is split), contents
 *      irrelevant.  counters: mv2_conv3_counter_bytes(d) bytes that MUST BE ZERO when the launch starts; the kernel
 *      leaves them zero, so one zero-initialised buffer can serve every mv2_conv3 launch of a stream.
 * ------------------------------------------------------------------------------------------- */
int mv2_conv3_eligible(const mv2_conv_desc* d);
int mv2_conv3_fusable(const mv2_conv_desc* d);          /* may this descriptor be launched with wp2 != NULL? */
size_t mv2_conv3_workspace_bytes(const mv2_conv_desc* d);
size_t mv2_conv3_counter_bytes(const mv2_conv_desc* d);
int mv2_conv3_tiles(const mv2_conv_desc* d, int* tile_h, int* tile_w);
int mv2_conv3(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp3, const float* bias,
              const void* wp2, const float* bias2, int act2,
              const float* se_wk, float se_bk, float* se_part, void* out, void* workspace, size_t workspace_bytes,
              uint32_t* counters, size_t counter_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * mv2_video_to_ndhwc — API-edge layout change + zero padding.
 * Replaces: pad_at_dim M:1537 and conv_in's F.pad M:927 (the 7x7x7 halo is materialised once so
 * conv_in runs without bounds checks), and the NCTHW -> channel-last rearranges.
 * in : fp32 NCTHW (B,C,T,H,W).  out: dtype NDHWC (B, T+pad_t, H+2*pad_hw, W+2*pad_hw, c_pad) with
 * zeros in every padded frame / border / channel; `tail_elems` extra zero elements follow.
 * ------------------------------------------------------------------------------------------- */
int mv2_video_to_ndhwc(int dtype, const float* in, void* out, int B, int C, int T, int H, int W,
                       int pad_t, int pad_hw, int c_pad, int tail_elems, void* stream);

/* ---------------------------------------------------------------------------------------------
 * mv2_split_bf16 — the input side of the id-exact 'split' mode (fp32 activations, contractions on the bf16 matrix pipe):
 * out[row] = [ hi(x[row]) | lo(x[row]) | hi(x[row]) ], 3 C bf16 per row, hi = bf16(x) (round to nearest even), lo = bf16(x - hi).
 * A convolution of these rows with the weight image of [ w_hi | w_hi | w_lo ] along the input channels (pack.pack_conv3d_split) is
 *     x_hi.w_hi + x_lo.w_hi + x_hi.w_lo,
 * every product exact in fp32 and accumulated in fp32: x.w to ~2^-17 relative per product instead of bf16's 2^-9 (reference call sites:
 * the ResidualUnit convolutions M:937 in front of the quantizer's sign M:1705).  rows x C fp32 in, C % 4 == 0.
 * ------------------------------------------------------------------------------------------- */
int mv2_split_bf16(const float* x, void* out, int64_t rows, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * mv2_rms_rowscale — per-row 1 / max(||x_row||_2, 1e-12)  (fp32), the data-dependent half of
 * RMSNorm M:275-276 (F.normalize); sqrt(dim)*gamma is folded into the consumer's packed weights.
 * shift_hw > 0: statistics of the TokenShift-ed row (M:250-253): channels [C/2, C) come from
 * the row `shift_hw` rows earlier (previous frame) or are zero when (row / shift_hw) % T == 0.
 * ------------------------------------------------------------------------------------------- */
int mv2_rms_rowscale(int dtype, const void* x, float* rowscale, int64_t rows, int C, int shift_hw, int T,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * SqueezeExcite (M:221-240) in three stream-ordered launches over frames = B*T, P = H*W:
 *   pool : per frame chunk online-softmax partials of  softmax_p(wk.y_p + bk)  weighted channel sums
 *          part[(f*nchunk + j)*(C+2)] = {max, sumexp, psum[C]}
 *   gate : combines the partials, 1x1 MLP (LeakyReLU 0.1) and sigmoid -> gate[f][C] (fp32);
 *          the two weight matrices are passed TRANSPOSED: w1t (C, hidden), w2t (hidden, C)
 *   apply: out = gate[f] * y + residual            (also the Residual add of M:174)
 * ------------------------------------------------------------------------------------------- */
int mv2_se_pool(int dtype, const void* y, const float* wk, float bk, float* part, int frames, int P, int C,
                int nchunk, void* stream);
int mv2_se_gate(const float* part, const float* w1t, const float* b1, const float* w2t, const float* b2,
                float* gate, int frames, int C, int hidden, int nchunk, void* stream);
int mv2_se_apply(int dtype, const void* y, const float* gate, const void* residual, void* out, int frames,
                 int P, int C, void* stream);

/* Two-launch form of the same SqueezeExcite for frames small enough to be pooled by one workgroup
 * (the caller uses it when a frame's P*C elements are <= 512 KB); results equal pool+gate+apply.
 *   pool_fc1 : grid (nslice, frames): whole-frame softmax pooling + hidden/nslice units of
 *              hid[f][k] = lrelu_0.1(W1 pooled + b1)            (hidden % nslice == 0, hidden/nslice a power of 2)
 *   fc2_apply: grid (nslice, frames): gate = sigmoid(W2 hid + b2) for C/nslice channels (a power of 2),
 *              out = gate * y + residual for those channels of every position of the frame.
 */
int mv2_se_pool_fc1(int dtype, const void* y, const float* wk, float bk, const float* w1t, const float* b1,
                    float* hid, int frames, int P, int C, int hidden, int nslice, void* stream);
/* fc1 of the two-launch form from pooling RECORDS ({max logit, sum exp, exp-weighted sums[C]} x nchunk per frame, as written by
 * mv2_pw_se, mv2_conv3's fused epilogue or mv2_se_pool) instead of a pooling pass over the tensor.  (nchunk (C + 3) + C + 1024)
 * floats must fit 64 KB of LDS. */
int mv2_se_rec_fc1(const float* part, const float* w1t, const float* b1, float* hid, int frames, int C, int nchunk,
                   int hidden, int nslice, void* stream);
int mv2_se_fc2_apply(int dtype, const void* y, const float* hid, const float* w2t, const float* b2,
                     const void* residual, void* out, int frames, int P, int C, int hidden, int nslice,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * mv2_attention — softmax attention with learned memory key/values.
 * Replaces: Attention.forward M:381-387 (split heads, prepend mem_kv) + Attend.forward A:186-243
 * (q k^T d^-1/2, right-aligned causal mask A:46-47, softmax, attn v).
 * qkv: rows x (3*heads*dim_head), column = (which*heads + head)*dim_head + d.
 * Sequence s, token i lives at row  (s / seq_inner)*seq_outer_stride + (s % seq_inner) + i*tok_stride
 *   space attention: seq_inner = 1,   seq_outer_stride = n, tok_stride = 1   (s = frame)
 *   time  attention: seq_inner = H*W, seq_outer_stride = T*H*W, tok_stride = H*W (s = pixel)
 * mem_kv: fp32 (2, heads, n_mem, dim_head).  out: rows x (heads*dim_head), same row addressing.
 * dim_head: 8, 16, 32 (the reference default, M:1076) or 64.  bf16 non-causal sequences of >= 64 tokens at dim_head 32 run on the MFMA
 * kernel; everything else (fp32, the causal time attention, the other widths) on fp32 VALU kernels.
 * ------------------------------------------------------------------------------------------- */
int mv2_attention(int dtype, const void* qkv, const float* mem_kv, void* out, int nseq, int n, int heads,
                  int dim_head, int n_mem, int causal, int seq_inner, int64_t seq_outer_stride,
                  int64_t tok_stride, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FeedForward in ONE launch (reference M:466-508 FeedForward = RMSNorm -> Conv3d 1x1x1 (dim -> 2 inner) -> GEGLU -> Conv3d 1x1x1
 * (inner -> dim), inside Residual M:174, behind TokenShift M:250-253 for the time-axis layers), bf16, C = 256 / 512:
 *   xs  = x, or with shift_hw > 0 the token-shifted row: channels [C/2, C) from the row shift_hw earlier, zero where
 *         (row / shift_hw) % T == 0 (rows % (shift_hw * T) == 0)
 *   h   = bf16( gelu(g) * v ),  [v | g] = (W1 xs) / max(|xs|, 1e-12) + b1      (sqrt(C) * gamma folded into W1's columns)
 *   out = bf16( W2 h + b2 + x )
 * w1: the mv2_conv image of the first conv ([C/64 steps][co_pad = ceil128(2 hidden_pad)][128 B], GEGLU column order: block j of 64 packed
 *     columns = 32 values then 32 gates of hidden units 32 j .. 32 j + 31) in FRAGMENT-MAJOR order (mv2_pack_conv3d's second form);
 * w2: the image of the second conv with K = hidden_pad ([hidden_pad/64 steps][C][128 B]), fragment-major; b1 (co_pad), b2 (C) fp32 in
 *     packed column order; hidden_pad % 64 == 0.  The hidden tensor never exists in memory.
 * Few-row launches split a row tile's hidden range over several workgroups: mv2_ff_workspace_bytes() of fp32 slabs and
 * mv2_ff_counter_bytes() of ticket words (zero before the first launch; every launch leaves them zero) - both may be 0 / NULL
 * when the query returns 0.  x != out.
 * ------------------------------------------------------------------------------------------- */
int mv2_ff_eligible(int dtype, int64_t rows, int C, int hidden_pad);
size_t mv2_ff_workspace_bytes(int64_t rows, int C, int hidden_pad);
size_t mv2_ff_counter_bytes(int64_t rows, int C, int hidden_pad);
int mv2_ff(int dtype, const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out,
           int64_t rows, int C, int hidden_pad, int shift_hw, int T, void* workspace, size_t workspace_bytes,
           void* counters, size_t counter_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Taylor-series linear attention (third-party TaylorSeriesLinearAttn, call site M:430), dim_head 8.
 *   kv   : per (frame, chunk, head) partial sums over tokens of phi(k)^T [v, 1]  -> part (fp32)
 *          part[((f*nchunk + j)*heads + h)*657], 657 = 73*9.  nchunk == 1: one workgroup per (frame, head) walks the whole frame and
 *          `part` receives the FINISHED states [frames][heads][657] (no partial states in memory, nothing to reduce).
 *   apply: reduces the chunks (nchunk > 1; into kvsum_ws, which is not touched and may be NULL for nchunk == 1),
 *          out = phi(q * d^-1/2) . KV / max(phi(q) . Ksum, 1e-5)
 * q: rows x (heads*8); kv: rows x (2*heads*8) with k first; out: rows x (heads*8).
 * ------------------------------------------------------------------------------------------- */
int mv2_taylor_kv(int dtype, const void* kv, float* part, int frames, int n, int heads, int nchunk,
                  void* stream);
int mv2_taylor_apply(int dtype, const void* q, const float* part, void* out, float* kvsum_ws, int frames,
                     int n, int heads, int nchunk, void* stream);
/* The same two launches on ONE head-major tensor qkv [frame][head][token][q 8 | k 8 | v 8] (24 elements per entry): what a single
 * projection GEMM of [to_q | to_kv] (both read the same RMS-normed rows, M:421-430) writes with MV2_STORE_SHUFFLE, rt = heads,
 * out_c = 24 and its packed columns ordered head-major (pack.pack_taylor_qkv).  A (frame, head) workgroup of the state kernel then reads
 * 48 contiguous bytes per token.  out: rows x (heads*8) as above. */
int mv2_taylor_kv_fused(int dtype, const void* qkv, float* part, int frames, int n, int heads, int nchunk, void* stream);
int mv2_taylor_apply_fused(int dtype, const void* qkv, const float* part, void* out, float* kvsum_ws, int frames,
                           int n, int heads, int nchunk, void* stream);
/* Any supported head width: dim_head 4, 8 or 16 (the reference's linear_attn_dim_head is a free keyword, M:1079 -> 415-419; 8 is its
 * default and the width the kernels above are tuned for — dim_head = 8 forwards to them with nchunk = 1).  phi has F = 1 + d + d^2
 * features; `state` is [frames][heads][F][d + 1] fp32 (column d = the normaliser): mv2_taylor_state_floats(dim_head) floats per
 * (frame, head), 0 for an unsupported width.  Row layouts as mv2_taylor_kv / mv2_taylor_apply with 8 -> dim_head. */
size_t mv2_taylor_state_floats(int dim_head);
int mv2_taylor_kv_d(int dtype, const void* kv, float* state, int frames, int n, int heads, int dim_head, void* stream);
int mv2_taylor_apply_d(int dtype, const void* q, const float* state, void* out, int frames, int n, int heads, int dim_head, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Quantizers (third-party vector_quantize_pytorch; call sites M:1574, M:1593, M:1700, M:1705).
 * One wavefront per token; the per-token reductions are wave shuffles.
 *  mv2_lfq : z = W_in x + b_in ; z = tanh(z/clamp)*clamp (clamp <= 0: skipped);
 *            bit_i = z_i > 0 ; ids[c] = sum_i bit_i << (cb_dim-1-i)  (int64, per codebook c);
 *            quantized = W_out (+-scale) + b_out.  prequant (optional, may be NULL): z (fp32).
 *  mv2_fsq : bounded = tanh(z + shift)*half_l - offset ; r = rint(bounded); code = r / half_w;
 *            ids[c] = sum_i (r_i + half_w_i) * basis_i (int32).
 *  *_codes : ids -> codes -> W_out codes + b_out   (indices_to_codes, M:1593).
 * W_in: (cb_total, C) fp32 = project_in.weight as stored; w_out_t: (cb_total, C) fp32 = project_out.weight
 * TRANSPOSED (a lane reads consecutive channels of one code dim); cb_total = cb_dim * num_codebooks <= 32.
 * ------------------------------------------------------------------------------------------- */
int mv2_lfq(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out_t,
            const float* b_out, void* quantized, int64_t* ids, float* prequant, int64_t rows, int C,
            int cb_dim, int num_codebooks, float clamp, float scale, void* stream);
/* ---- weight images (HOST side, host pointers): the byte-exact layouts the kernels read, from a dense fp32 kernel in the
 * reference's state_dict order (Co, Ci, kt, kh, kw) (nn.Conv3d, magvit2_pytorch.py:894-922; Conv2d / Conv1d / Linear weights are
 * the same call with the missing kernel dims = 1).
 *   plain image (mv2_conv):            [step][co_pad = ceil128(Co) rows][128 bytes];  K order = taps (it, ih, iw) row-major,
 *                                      inside a tap the Ci channels zero-padded to whole 16-byte chunks; elements of `dtype`
 *                                      (bf16 by round-to-nearest-even); unused rows / K tail are zero.
 *   fragment-major (mv2_conv3, mv2_pw_se; needs Ci * elem_size % 128 == 0): every group of 32 rows of a step becomes
 *                                      [k-slice 4][half 2][row 32][16 bytes] = 16-byte chunk (2 k-slice + half) of that row.
 * Column permutations some callers apply first (GEGLU interleave, depth-to-space order, TokenShift halves) and the RMSNorm
 * gamma * sqrt(dim) folding are row / column re-orderings of `w` before this call: INTEGRATION.md "weight images". */
size_t mv2_pack_conv3d_bytes(int dtype, int co, int ci, int kt, int kh, int kw);
int mv2_pack_conv3d(int dtype, const float* w, int co, int ci, int kt, int kh, int kw, int frag_major, void* out,
                    size_t out_bytes);

/* Conditioning (reference AdaptiveRMSNorm M:278-323, Conv3DMod M:715-753, ResidualUnitMod M:975-988, cond stems M:1341-1353).
 * The conditioning vector is per sample; everything derived from it is a small (B, n) fp32 matrix.
 * mv2_cond_linear: out[b][o] = f(sum_i w[o][i] g(in[b][i]) + bias[o]);  mode 0: Linear;  mode 1: Linear + SiLU;
 *   mode 2 (Conv3DMod demodulation): g(x) = (x + 1)^2, no bias, f(s) = rsqrt(max(s, eps)), w[o][i] = sum over taps of weight^2.
 * mv2_scale_channels: out[row][c] = act(x[row][c] * (scale[row / rows_per_sample][c] + add)), act = MV2_ACT_NONE | MV2_ACT_ELU;
 *   in place allowed (out == x). */
int mv2_cond_linear(const float* in, const float* w, const float* bias, float* out, int B, int I, int O, int mode, float eps,
                    void* stream);
int mv2_scale_channels(int dtype, const void* x, const float* scale, void* out, int64_t rows, int64_t rows_per_sample, int C,
                       float add, int act, void* stream);

/* Frame ranges (separate_first_frame_encoding, reference M:1553-1561, 1633-1639): dst[b][dst_t0 + t] = src ? src[b][src_t0 + t] : 0
 * for t < nT; src (B, src_T, frame), dst (B, dst_T, frame), a frame = frame_bytes contiguous bytes (multiple of 16). */
int mv2_copy_frames(const void* src, int src_T, int src_t0, void* dst, int dst_T, int dst_t0, int nT, int B, int64_t frame_bytes,
                    void* stream);

/* Non-constant padding (CausalConv3d pad_mode 'reflect' / 'replicate' / 'circular', reference M:899-927, Conv3DMod M:750):
 * out (B, T + pt, H + 2 ph, W + 2 pw, row) = F.pad(in (B, T, H, W, row), (pw, pw, ph, ph, pt, 0), mode) as an explicit gather of rows of
 * row_bytes (multiple of 8) bytes; mode 1 = reflect, 2 = replicate, 3 = circular; `tail_bytes` zero bytes are written behind the image.
 * The convolution that follows runs 'valid' (check_bounds = 0) on the padded image.  The reference's default 'constant' needs no call. */
int mv2_pad_ndhwc(const void* in, void* out, int B, int T, int H, int W, int64_t row_bytes, int pt, int ph, int pw, int mode,
                  int64_t tail_bytes, void* stream);

/* 'gateloop_time' (reference M:1216-1222 -> third-party gateloop_transformer.SimpleGateLoopLayer, a reconstruction: parity unpinned).
 * qkva: rows (B, T, P) x 3C = [q | kv | a], the Linear(dim, 3 dim) of the RMS-normed tokens (mv2_conv with rowscale).  Per pixel and
 * channel along time: h_t = sigmoid(a_t) h_{t-1} + kv_t (h_{-1} = 0), out_t = q_t h_t + residual_t (the Residual wrapper, M:167-174).
 * out may alias residual. */
int mv2_gateloop_scan(int dtype, const void* qkva, const void* residual, void* out, int B, int T, int64_t P, int C, void* stream);

/* ResidualUnit tail at C = 128 / 256 / 512, bf16: the 1x1x1 conv + bias + ELU (reference M:937-938) and the SqueezeExcite
 * pooling records of its output (reference M:226-231) in ONE launch (replaces mv2_conv k = 1 + mv2_se_pool).  `wp` is the
 * C x C kernel in mv2_conv3's fragment-major order (the 1x1x1 kernel packed as mv2_conv does, then pack.py::to_frag_major).
 * mv2_pw_se_rows: rows per record (32 or 64; one workgroup each), 0 when the shape is not eligible (dtype, C, P % rows).
 * se_part receives rows / mv2_pw_se_rows records of C + 2 floats {max logit, sum exp, exp-weighted channel sums[C]} in row
 * order: the record list mv2_se_gate consumes with nchunk = P / mv2_pw_se_rows.  P = rows per frame (H * W). */
int mv2_pw_se_rows(int dtype, int64_t rows, int P, int C);
int mv2_pw_se(int dtype, const void* in, const void* wp, const float* bias, const float* se_wk, float se_bk, void* out,
              float* se_part, int64_t rows, int P, int C, void* stream);

/* Training-mode LFQ (third-party LFQ.forward with self.training, reference call site M:1705, loss weights M:1067-1072, recon loss
 * M:1722).  mv2_lfq_train = mv2_lfq with the straight-through forward value z + (q - z) (prequant is mandatory: it feeds
 * the statistics).  mv2_lfq_stats: per codebook the token-SUMMED softmax probabilities over all K = 2^cb_dim codes
 * (prob = softmax_j(2 * inv_temperature * z . c_j)), the sum over tokens and codebooks of the per-token entropies (log clamped at
 * 1e-5) and the sum of squared commitment errors:  sums[num_codebooks * K + 2].  Ranks all_reduce(SUM) `sums` (RCCL) before
 * mv2_lfq_losses, which turns them into losses[4] = {per-sample entropy, batch entropy, commitment, aux} with
 * aux = (H_sample - gamma * H_batch) * w_entropy + commitment * w_commit, n_tokens = global token count.  cb_dim <= 12.
 * mv2_sq_diff_sum: sum((a - b)^2) over n fp32 elements (F.mse_loss numerator, M:1722); 16-byte aligned pointers.
 * All reductions have a fixed order: results are bitwise reproducible. */
int mv2_lfq_train(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out_t,
                  const float* b_out, void* quantized, int64_t* ids, float* prequant, int64_t rows, int C,
                  int cb_dim, int num_codebooks, float clamp, float scale, void* stream);
size_t mv2_lfq_stats_workspace_bytes(int64_t rows, int cb_dim, int num_codebooks);
int mv2_lfq_stats(const float* prequant, int64_t rows, int cb_dim, int num_codebooks, float inv_temperature, float scale,
                  float* sums, void* workspace, size_t workspace_bytes, void* stream);
int mv2_lfq_losses(const float* sums, double n_tokens, int cb_dim, int num_codebooks, float diversity_gamma,
                   float entropy_loss_weight, float commitment_loss_weight, float* losses, void* stream);
size_t mv2_sq_diff_workspace_bytes(int64_t n);
int mv2_sq_diff_sum(const float* a, const float* b, int64_t n, float* sum, void* workspace, size_t workspace_bytes, void* stream);
/* hinge discriminator loss (reference M:120-121): out[0] = mean(relu(1 + fake) + relu(1 - real)) over n fp32 logits. */
int mv2_hinge_discr_loss(const float* fake, const float* real, int n, float* out, void* stream);
int mv2_lfq_codes(int dtype, const int64_t* ids, const float* w_out_t, const float* b_out, void* out,
                  int64_t rows, int C, int cb_dim, int num_codebooks, float scale, void* stream);
int mv2_fsq(int dtype, const void* x, const float* w_in, const float* b_in, const float* w_out_t,
            const float* b_out, const float* shift, const float* half_l, const float* offset,
            const float* half_w, const int32_t* basis, void* quantized, int32_t* ids, float* prequant,
            int64_t rows, int C, int cb_dim, int num_codebooks, void* stream);
int mv2_fsq_codes(int dtype, const int32_t* ids, const float* w_out_t, const float* b_out,
                  const int32_t* levels, const float* half_w, const int32_t* basis, void* out, int64_t rows,
                  int C, int cb_dim, int num_codebooks, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Timing helper for bench.py: HIP events on the stream the kernels are launched on.
 * ------------------------------------------------------------------------------------------- */
void* mv2_event_create(void);
int mv2_event_record(void* ev, void* stream);
int mv2_event_sync(void* ev);
float mv2_event_elapsed_ms(void* start, void* stop);
void mv2_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* MV2_H */
