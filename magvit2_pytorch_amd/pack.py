"""
Host-side weight packing and K-program construction for ``mv2_conv`` (include/mv2.h).

A convolution (or Linear / 1x1x1 conv / token-shifted Linear) is described as a list of *taps*:
each tap reads ``klen`` contiguous input elements at element offset ``((it*Hi + ih)*Wi + iw)*Ci + ci_lo``
from the row base and multiplies them with a (Co, klen) weight slice.  Taps are cut into equal
*segments* of ``seg_bytes`` (the gcd of all tap byte-lengths and 128); 128/seg_bytes segments make
one 128-byte K *step*; the packed weight tensor is [nsteps][co_pad][128 bytes].

Column permutations applied at pack time (so the kernel's epilogue stores 4 consecutive channels):
  * depth-to-space / depth-to-time:  reference channel  c*Q + q   ->  packed  q*Cq + c
    (reference Rearrange 'b (c p1 p2) h w -> b c (h p1) (w p2)' magvit2_pytorch.py:824,
     'b (c p) t -> b c (t p)' :861)
  * GEGLU: value j -> 64*(j//32) + j%32, gate j -> 64*(j//32) + 32 + j%32 (reference chunk order
    x, gate = x.chunk(2, dim=1) magvit2_pytorch.py:468)
RMSNorm's  sqrt(dim) * gamma  (magvit2_pytorch.py:276) is folded into the consumer's weight columns;
the data-dependent 1/||x|| half is the kernel's ``rowscale``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

MV2_BF16, MV2_F32 = 0, 1
ACT_NONE, ACT_ELU, ACT_SILU, ACT_GEGLU, ACT_LRELU = 0, 1, 2, 3, 4
STORE_ROWS, STORE_SHUFFLE, STORE_NCTHW_F32, STORE_ROWS_F32 = 0, 1, 2, 3
FILLER_SEL = 0x80000000


def esize(dtype: torch.dtype) -> int:
    return 2 if dtype == torch.bfloat16 else 4


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return MV2_BF16
    if dtype == torch.float32:
        return MV2_F32
    raise ValueError(f'unsupported compute dtype {dtype}')


def round_up(v, m):
    return (v + m - 1) // m * m


@dataclass
class Tap:
    it: int
    ih: int
    iw: int
    ci_lo: int            # first input element of this tap inside the row (relative to channel 0 of tap position)
    weight: torch.Tensor  # (Co, klen) fp32
    filler: bool = False


@dataclass
class PackedConv:
    """dtype-specific packed weights + the geometry-independent half of the K program."""
    wp: torch.Tensor                  # (nsteps, co_pad, 128 // esize) compute dtype
    bias: Optional[torch.Tensor]      # (co_pad,) fp32 in packed column order, or None
    seg_taps: List[Tuple[int, int, int, int, bool]]   # per segment: (it, ih, iw, elem_offset_in_row, filler)
    seg_bytes: int
    nsteps: int
    co: int                           # valid packed columns
    co_pad: int
    kt: int
    kh: int
    kw: int
    k_real: int = 0                   # un-padded reduction length (algorithmic FLOP accounting)
    co_real: int = 0                  # reference output channels
    frag_major: bool = False          # wp is in mv2_conv3's MFMA-fragment order (include/mv2.h), not mv2_conv's

    def to(self, device):
        self.wp = self.wp.to(device)
        if self.bias is not None:
            self.bias = self.bias.to(device)
        return self

    def segs_tensor(self, Hi, Wi, Ci, device):
        """(nsegs, 2) int32: elem_offset, sel — geometry dependent (mv2_seg)."""
        rows = []
        for (it, ih, iw, lo, filler) in self.seg_taps:
            if filler:
                rows.append((0, FILLER_SEL - (1 << 32)))
            else:
                sel = (1 << it) | (1 << (8 + ih)) | (1 << (16 + iw))
                rows.append((((it * Hi + ih) * Wi + iw) * Ci + lo, sel))
        return torch.tensor(rows, dtype=torch.int32).to(device)


def pack_taps(taps: List[Tap], co: int, dtype: torch.dtype, bias: Optional[torch.Tensor], kt, kh, kw,
              col_perm: Optional[torch.Tensor] = None, co_packed: Optional[int] = None,
              k_real: Optional[int] = None) -> PackedConv:
    """col_perm[n_ref] = packed column of reference output channel n_ref (None = identity);
    co_packed = number of valid packed columns (defaults to co)."""
    es = esize(dtype)
    kvec = 16 // es
    co_packed = co_packed if co_packed is not None else co
    co_pad = round_up(co_packed, 128)
    lens = [round_up(t.weight.shape[1], kvec) * es for t in taps]
    seg_bytes = 128
    for lb in lens:
        seg_bytes = math.gcd(seg_bytes, lb)
    assert seg_bytes >= 16
    seg_elems = seg_bytes // es
    per_step = 128 // seg_bytes
    seg_taps = []
    cols = []            # list of (Co, seg_elems) fp32 blocks in K order
    for t, lb in zip(taps, lens):
        w = t.weight
        klen_pad = lb // es
        if w.shape[1] < klen_pad:
            w = torch.cat((w, w.new_zeros(w.shape[0], klen_pad - w.shape[1])), dim=1)
        for j in range(klen_pad // seg_elems):
            seg_taps.append((t.it, t.ih, t.iw, t.ci_lo + j * seg_elems, False))
            cols.append(w[:, j * seg_elems:(j + 1) * seg_elems])
    while len(seg_taps) % per_step:
        seg_taps.append((0, 0, 0, 0, True))
        cols.append(torch.zeros(co, seg_elems))
    nsteps = len(seg_taps) // per_step
    k_elems = 128 // es
    wk = torch.cat(cols, dim=1).float()                 # (Co_ref, nsteps * k_elems)
    full = torch.zeros(co_pad, nsteps * k_elems)
    if col_perm is None:
        full[:co] = wk
    else:
        full[col_perm] = wk
    wp = full.reshape(co_pad, nsteps, k_elems).permute(1, 0, 2).contiguous().to(dtype)
    b = None
    if bias is not None:
        b = torch.zeros(co_pad)
        if col_perm is None:
            b[:co] = bias.float()
        else:
            b[col_perm] = bias.float()
    if k_real is None:
        k_real = sum(t.weight.shape[1] for t in taps)
    return PackedConv(wp, b, seg_taps, seg_bytes, nsteps, co_packed, co_pad, kt, kh, kw, k_real, co)


# ------------------------------------------------------------------------------------------ builders
def conv_taps(w: torch.Tensor) -> List[Tap]:
    """w: (Co, Ci, kt, kh, kw) -> one tap per kernel position, full channel extent."""
    co, ci, kt, kh, kw = w.shape
    return [Tap(it, ih, iw, 0, w[:, :, it, ih, iw].float()) for it in range(kt) for ih in range(kh) for iw in range(kw)]


def to_frag_major(wp: torch.Tensor) -> torch.Tensor:
    """[nsteps][co_pad][128 B] -> mv2_conv3's weight image (include/mv2.h): every group of 32 output rows becomes
    [k-slice 4][half 2][row 32][16 B], where 16-byte chunk (2*k-slice + half) of row r of the plain layout is meant —
    one contiguous 1 KB block per (row group, k-slice) = one wave's MFMA A-fragments, loaded with one instruction."""
    ns, cp, kel = wp.shape
    e16 = kel // 8                                        # elements per 16-byte chunk
    return wp.reshape(ns, cp // 32, 32, 4, 2, e16).permute(0, 1, 3, 4, 2, 5).contiguous().reshape(ns, cp, kel)


def conv3_packable(w, dtype) -> bool:
    """weights of a convolution mv2_conv3 can run (kt <= 3, 3 x 3 in space, whole 128-byte channel chunks, C_out >= 64)."""
    co, ci, kt, kh, kw = w.shape
    return kt <= 3 and kh == 3 and kw == 3 and (ci * esize(dtype)) % 128 == 0 and co >= 64 and co % 4 == 0


def pack_conv3d(w, bias, dtype, frag_major=False):
    co, ci, kt, kh, kw = w.shape
    pc = pack_taps(conv_taps(w), co, dtype, bias, kt, kh, kw)
    if frag_major:
        assert pc.seg_bytes == 128 and pc.nsteps == kt * kh * kw * (ci * esize(dtype) // 128)
        pc.wp = to_frag_major(pc.wp)
        pc.frag_major = True
    return pc


def split_bf16_halves(x):
    """x (fp32) -> (hi, lo) as fp32 tensors holding bf16 values: hi = bf16(x), lo = bf16(x - hi); x = hi + lo to ~2^-17 relative."""
    hi = x.float().to(torch.bfloat16).float()
    lo = (x.float() - hi).to(torch.bfloat16).float()
    return hi, lo


def pack_conv3d_split(w, bias):
    """Weight image of the split-bf16 form of a k x 3 x 3 convolution (include/mv2.h mv2_split_bf16): the bf16 fragment-major image of
    [ w_hi | w_hi | w_lo ] along the input channels, to be multiplied with rows [ x_hi | x_lo | x_hi ]: x_hi.w_hi + x_lo.w_hi + x_hi.w_lo."""
    hi, lo = split_bf16_halves(w)
    return pack_conv3d(torch.cat((hi, hi, lo), dim=1), bias, torch.bfloat16, frag_major=True)


def pack_taylor_qkv(wq, wkv, heads, dtype, in_scale=None):
    """ONE projection for the linear attention's to_q and to_kv (both read the same RMS-normed rows, reference M:421-430): the packed
    columns are ordered head-major [head][q 8 | k 8 | v 8], so that the GEMM's depth-to-time store (MV2_STORE_SHUFFLE, rt = heads, out_c = 24)
    writes [frame][head][token][24] - the layout mv2_taylor_kv_fused / mv2_taylor_apply_fused read (include/mv2.h)."""
    hd = heads * 8
    assert wq.shape[0] == hd and wkv.shape[0] == 2 * hd and wq.shape[1] == wkv.shape[1]
    q, k, v = wq.float().reshape(heads, 8, -1), wkv[:hd].float().reshape(heads, 8, -1), wkv[hd:].float().reshape(heads, 8, -1)
    return pack_pointwise(torch.cat((q, k, v), dim=1).reshape(3 * hd, -1), None, dtype, in_scale=in_scale)


def pack_conv_in_frag(w):
    """mv2_conv_in's weight image (include/mv2.h): w (64, ci <= 4, kt, 7, 7) -> [kt][7 ih][2 s][2 co tiles][64 lanes][8] bf16; lane l =
    (co = 32 tile + l % 32, pixel pair l // 32); element j = (pixel iw = 4 s + 2 (l // 32) + j // 4, channel j % 4); iw = 7 and
    missing channels are zero."""
    co, ci, kt, kh, kw = w.shape
    assert co == 64 and ci <= 4 and kh == 7 and kw == 7 and 1 <= kt <= 7
    wt = torch.zeros(64, 4, kt, 7, 8)
    wt[:, :ci, :, :, :7] = w.float()
    # [tile][co32][c][it][ih][s][hi][p2]  (iw = 4 s + 2 hi + p2)  ->  [it][ih][s][tile][hi][co32][p2][c]
    img = wt.reshape(2, 32, 4, kt, 7, 2, 2, 2).permute(3, 4, 5, 0, 6, 1, 7, 2).contiguous()
    return img.reshape(kt * 7 * 2 * 2, 64, 8).to(torch.bfloat16)


def pack_conv_out(w):
    """mv2_conv_out's weight image (include/mv2.h): w (co <= 4, 64, kt, 3, 3) -> [kt * 9 taps][2 halves][64 lanes][8] bf16 with
    lane l = (co = l % 16, channel group l // 16): the A operand of v_mfma_f32_16x16x32_bf16, zero rows for co >= w.shape[0]."""
    co, ci, kt, kh, kw = w.shape
    assert ci == 64 and co <= 4 and kh == 3 and kw == 3 and 1 <= kt <= 3
    wt = torch.zeros(16, 64, kt, 3, 3)
    wt[:co] = w.float()
    # [co][half][kg][j][tap] -> [tap][half][kg][co][j]
    img = wt.reshape(16, 2, 4, 8, kt * 9).permute(4, 1, 2, 0, 3).contiguous()
    return img.reshape(kt * 9 * 2, 64, 8).to(torch.bfloat16)


def pack_pointwise(w2d, bias, dtype, in_scale: Optional[torch.Tensor] = None, k_pad: Optional[int] = None):
    """w2d: (Co, Ci).  in_scale: per-input-channel factor folded into the columns (RMSNorm gamma*sqrt(C)).
    k_pad: physical input row length when the producer padded its channels (GEGLU hidden)."""
    w = w2d.float()
    if in_scale is not None:
        w = w * in_scale.float()[None, :]
    k_real = w.shape[1]
    if k_pad is not None and k_pad > w.shape[1]:
        w = torch.cat((w, w.new_zeros(w.shape[0], k_pad - w.shape[1])), dim=1)
    return pack_taps([Tap(0, 0, 0, 0, w)], w.shape[0], dtype, bias, 1, 1, 1, k_real=k_real)


def pack_token_shift(w2d, bias, dtype, in_scale=None, **kw):
    """Linear applied to TokenShift(x) (magvit2_pytorch.py:250-253): channels [0, C/2) from frame t,
    channels [C/2, C) from frame t-1 (zero at t = 0).  kt = 2, pt = 1: it = 1 is the current frame."""
    w = w2d.float()
    if in_scale is not None:
        w = w * in_scale.float()[None, :]
    c = w.shape[1]
    h = c // 2
    taps = [Tap(1, 0, 0, 0, w[:, :h]), Tap(0, 0, 0, h, w[:, h:])]
    return pack_taps(taps, w.shape[0], dtype, bias, 2, 1, 1, **kw)


def shuffle_perm(co, q):
    """reference channel n = c*q + k  ->  packed k*cq + c."""
    cq = co // q
    n = torch.arange(co)
    return (n % q) * cq + (n // q)


def pack_shuffle_up(w2d, bias, dtype, q):
    w = w2d.float()
    co = w.shape[0]
    return pack_taps([Tap(0, 0, 0, 0, w)], co, dtype, bias, 1, 1, 1, col_perm=shuffle_perm(co, q))


def geglu_layout(inner):
    hidden_pad = round_up(inner, 64)
    j = torch.arange(inner)
    val = 64 * (j // 32) + (j % 32)
    gate = val + 32
    return hidden_pad, torch.cat((val, gate))


def pack_geglu_in(w2d, bias, dtype, in_scale=None, token_shift=False):
    """first FeedForward conv (Co = 2*inner): returns PackedConv with co = 2*hidden_pad packed columns."""
    inner = w2d.shape[0] // 2
    hidden_pad, perm = geglu_layout(inner)
    w = w2d.float()
    if in_scale is not None:
        w = w * in_scale.float()[None, :]
    if token_shift:
        c = w.shape[1]
        h = c // 2
        taps = [Tap(1, 0, 0, 0, w[:, :h]), Tap(0, 0, 0, h, w[:, h:])]
        return pack_taps(taps, w.shape[0], dtype, bias, 2, 1, 1, col_perm=perm, co_packed=2 * hidden_pad), hidden_pad
    return pack_taps([Tap(0, 0, 0, 0, w)], w.shape[0], dtype, bias, 1, 1, 1, col_perm=perm, co_packed=2 * hidden_pad), hidden_pad


def pack_conv_in(w, bias, dtype, c_pad):
    """conv_in over the pre-padded (B, T+pt, H+2p, W+2p, c_pad) video: one tap per (it, ih), the
    (iw, c) extent is contiguous in NDHWC, so K per tap = kw * c_pad elements."""
    co, ci, kt, kh, kw = w.shape
    taps = []
    for it in range(kt):
        for ih in range(kh):
            blk = torch.zeros(co, kw, c_pad)
            blk[:, :, :ci] = w[:, :, it, ih, :].float().permute(0, 2, 1)
            taps.append(Tap(it, ih, 0, 0, blk.reshape(co, kw * c_pad)))
    return pack_taps(taps, co, dtype, bias, kt, kh, 1, k_real=kt * kh * kw * ci)
