"""
TEST INFRASTRUCTURE — not product code (only tests/ may import it).

``CpuOps``: a plain-torch CPU emulation of the *kernel-level* semantics declared in include/mv2.h
(packed weights, K-program segments, epilogues, NDHWC addressing).  It has the same method surface
as ``magvit2_pytorch_amd.ops_hip.HipOps`` and is injected into the façade by tests to
  (1) validate the host logic — weight packing, K programs, plan lowering, buffer recycling — against
      the reference on CPU where no GPU exists, and
  (2) serve as the per-kernel expected value in the ``-m gpu`` unit tests of each C-ABI entry point.
It is deliberately written from the header's wording (generic gathers, no conv library calls), so a
packing bug cannot cancel against a matching bug here.  fp64 accumulation, results rounded to the
storage dtype like the kernels' stores.
"""
from __future__ import annotations

import math

import torch

ACT_NONE, ACT_ELU, ACT_SILU, ACT_GEGLU, ACT_LRELU = 0, 1, 2, 3, 4
STORE_ROWS, STORE_SHUFFLE, STORE_NCTHW_F32, STORE_ROWS_F32 = 0, 1, 2, 3


def _esize(t):
    return t.element_size()


class CpuOps:
    name = 'cpu-emulation'
    device = torch.device('cpu')
    cus = 256

    def __init__(self, acc_dtype=torch.float64):
        self.acc = acc_dtype

    def empty(self, shape, dtype):
        # NaN-poison float buffers so that any read-before-write in a plan is caught
        if dtype.is_floating_point:
            return torch.full(shape, float('nan'), dtype=dtype)
        return torch.zeros(shape, dtype=dtype)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def to_device(self, t):
        return t

    def current_stream(self):
        return 0

    def synchronize(self):
        pass

    # ------------------------------------------------------------------ conv
    def conv_workspace_bytes(self, d):
        return 0

    def conv_in_eligible(self, d) -> bool:
        """mirror of mv2_conv_in_eligible (convin.hip)"""
        return (d['dtype'] == 0 and d['Ci'] == 4 and d['co'] == 64 and d['out_c'] == 64 and d['kh'] == 7 and d['kw'] == 1 and 1 <= d['kt'] <= 7
                and d['st'] == 1 and d['sh'] == 1 and d['sw'] == 1 and d['pt'] == 0 and d['ph'] == 0 and d['pw'] == 0 and d['check_bounds'] == 0
                and d['Ti'] == d['To'] + d['kt'] - 1 and d['Hi'] == d['Ho'] + 6 and d['Wi'] == d['Wo'] + 6 and d['Wi'] % 2 == 0
                and d['store'] == STORE_ROWS and d['act'] == 0)

    def conv_in(self, d, x, in_bytes, wfrag, bias, out):
        """mv2_conv_in: the dense kernel is read back from the fragment image; 'valid' convolution over the pre-padded clip"""
        d = dict(d)
        def step(stream):
            B, Tp, Hp, Wp, kt = d['B'], d['Ti'], d['Hi'], d['Wi'], d['kt']
            # [it][ih][s][tile][hi][co32][p2][c] -> w[co][c][it][ih][iw = 4 s + 2 hi + p2]
            w = wfrag.to(self.acc).reshape(kt, 7, 2, 2, 2, 32, 2, 4).permute(3, 5, 7, 0, 1, 2, 4, 6).reshape(64, 4, kt, 7, 8)[..., :7]
            xin = x[: B * Tp * Hp * Wp * 4].reshape(B, Tp, Hp, Wp, 4).permute(0, 4, 1, 2, 3).to(self.acc)
            y = torch.nn.functional.conv3d(xin, w, None if bias is None else bias[:64].to(self.acc))      # (B, 64, To, Ho, Wo)
            y = y.permute(0, 2, 3, 4, 1).reshape(-1)
            out[: y.numel()].copy_(y.to(out.dtype))
        step.outs = [out]
        step.what = 'conv_in'
        return step

    def conv_out_eligible(self, d) -> bool:
        """mirror of mv2_conv_out_eligible (convout.hip)"""
        return (d['dtype'] == 0 and d['Ci'] == 64 and 1 <= d['co'] <= 4 and d['kh'] == 3 and d['kw'] == 3 and 1 <= d['kt'] <= 3 and d['st'] == 1
                and d['sh'] == 1 and d['sw'] == 1 and d['pt'] == d['kt'] - 1 and d['ph'] == 1 and d['pw'] == 1 and d['check_bounds'] == 1
                and d['To'] == d['Ti'] and d['Ho'] == d['Hi'] and d['Wo'] == d['Wi'] and d['store'] == STORE_NCTHW_F32 and d['act'] == 0
                and 0 <= d['crop_t'] < d['To'])

    def conv_out(self, d, x, in_bytes, wfrag, bias, out):
        """mv2_conv_out: the dense kernel is read back from the fragment image; zero padding in space, causal zeros in time"""
        d = dict(d)
        def step(stream):
            B, T, H, W, co, kt = d['B'], d['Ti'], d['Hi'], d['Wi'], d['co'], d['kt']
            # [tap][half][kg][co16][j] -> w[co][ci = 32 half + 8 kg + j][it][ih][iw]
            w = wfrag.to(self.acc).reshape(kt * 9, 2, 4, 16, 8).permute(3, 1, 2, 4, 0).reshape(16, 64, kt, 3, 3)[:co]
            xin = x[: B * T * H * W * 64].reshape(B, T, H, W, 64).permute(0, 4, 1, 2, 3).to(self.acc)
            xin = torch.nn.functional.pad(xin, (1, 1, 1, 1, kt - 1, 0))
            y = torch.nn.functional.conv3d(xin, w, None if bias is None else bias[:co].to(self.acc))
            y = y[:, :, d['crop_t']:]
            out.reshape(-1)[: y.numel()].copy_(y.reshape(-1).float())
        step.outs = [out]
        step.what = 'conv_out'
        return step

    @staticmethod
    def _pw_dma(d) -> bool:
        """mirror of pw_dma_pick / pw_dma_rule (conv.hip): which launches of mv2_conv / mv2_conv_rms run conv_pw_dma_kernel (no split-K;
        the tile - 128 x 128 for GEGLU, 128 x 64 / 64 x 128 / 64 x 64 otherwise - does not change results)"""
        if d['dtype'] != 0 or d.get('splitk', 0) > 1:
            return False
        if not (d['kt'] == d['kh'] == d['kw'] == 1 and d['st'] == d['sh'] == d['sw'] == 1 and d['pt'] == d['ph'] == d['pw'] == 0
                and d['To'] == d['Ti'] and d['Ho'] == d['Hi'] and d['Wo'] == d['Wi']):
            return False
        if d['seg_bytes'] != 128 or d['Ci'] % 64 != 0 or d['nsteps'] != d['Ci'] // 64:
            return False
        if d['store'] == STORE_NCTHW_F32 or (d['out_c'] * 2) % 16 != 0 or d['co'] % 8 != 0:
            return False
        M = d['B'] * d['To'] * d['Ho'] * d['Wo']
        tiles128 = -(-d['co'] // 128) * -(-M // 128)
        return tiles128 >= (256 if d['act'] == ACT_GEGLU else 32)

    def conv_rms_ok(self, d) -> bool:
        """mirror of mv2_conv_rms_ok (conv.hip): staged-store shapes whose launch plan keeps the reduction in one workgroup.  The
        launch plan (pick_tile / plan_launch) is restated here so that the emulated plan has the launches of the HIP plan."""
        es = 2 if d['dtype'] == 0 else 4
        if d['store'] == STORE_NCTHW_F32 or (d['out_c'] * es) % 16 != 0 or d['co'] % (16 // es) != 0:
            return False
        if self._pw_dma(d):
            return True                          # the LDS-DMA pointwise kernel never splits K
        M = d['B'] * d['To'] * d['Ho'] * d['Wo']
        co, cus = d['co'], self.cus
        blocks128 = -(-co // 128) * -(-M // 128)
        small = co >= 64 and blocks128 < 2 * cus
        if d['act'] == ACT_GEGLU:
            bco, bpos = (64, 256) if co <= 64 else ((64, 128) if small else (128, 128))
        elif small:
            bco, bpos = 64, 64
        elif co <= 32:
            bco, bpos = 32, 256
        elif co <= 64:
            bco, bpos = 64, 256
        else:
            bco, bpos = 128, 128
        blocks = -(-co // bco) * -(-M // bpos)
        sk = d.get('splitk', 0)
        if sk <= 0:
            sk = 1
            if blocks < cus and d['nsteps'] >= 16:
                sk = max(1, min((2 * cus + blocks - 1) // blocks, d['nsteps'] // 8, 16))
        sk = min(sk, d['nsteps'])
        sps = -(-d['nsteps'] // sk)
        return -(-d['nsteps'] // sps) == 1

    def conv_rms(self, d, x, in_bytes, wp, segs, bias, residual, out):
        """mv2_conv_rms: rowscale = 1 / max(|gathered row|, 1e-12), taken over the rows the GEMM gathers"""
        st = self.conv(d, x, in_bytes, wp, segs, bias, None, residual, out, None, rms_fused=True)
        st.what = 'conv_rms'
        return st

    def conv(self, d, x, in_bytes, wp, segs, bias, rowscale, residual, out, ws, rms_fused=False):
        d = dict(d)
        acc = self.acc

        def step(stream):
            es = _esize(x)
            kel = 128 // es                       # elements per K step
            seg_el = d['seg_bytes'] // es
            per_step = 128 // d['seg_bytes']
            B, Ti, Hi, Wi, Ci = d['B'], d['Ti'], d['Hi'], d['Wi'], d['Ci']
            To, Ho, Wo = d['To'], d['Ho'], d['Wo']
            M = B * To * Ho * Wo
            m = torch.arange(M)
            wo = m % Wo; r = m // Wo
            ho = r % Ho; r = r // Ho
            to = r % To; b = r // To
            t_in = to * d['st'] - d['pt']; h_in = ho * d['sh'] - d['ph']; w_in = wo * d['sw'] - d['pw']
            rowbase = (((b * Ti + t_in) * Hi + h_in) * Wi + w_in) * Ci
            flat = x.reshape(-1).to(acc)
            assert in_bytes <= x.numel() * es
            nsteps = d['nsteps']
            W = wp.to(acc)                         # (nsteps, co_pad, kel)
            D = torch.zeros(M, d['co_pad'], dtype=acc)
            ss = torch.zeros(M, dtype=acc)
            sg = segs.reshape(-1, 2)
            chunk = max(1, (1 << 22) // max(1, seg_el))
            for s in range(nsteps):
                for q in range(per_step):
                    off, sel = int(sg[s * per_step + q, 0]), int(sg[s * per_step + q, 1]) & 0xffffffff
                    if sel & 0x80000000:
                        continue
                    it = (sel & 0xff).bit_length() - 1
                    ih = ((sel >> 8) & 0xff).bit_length() - 1
                    iw = ((sel >> 16) & 0xff).bit_length() - 1
                    if d['check_bounds']:
                        ok = ((t_in + it >= 0) & (t_in + it < Ti) & (h_in + ih >= 0) & (h_in + ih < Hi)
                              & (w_in + iw >= 0) & (w_in + iw < Wi))
                    else:
                        ok = torch.ones(M, dtype=torch.bool)
                    wseg = W[s, :, q * seg_el:(q + 1) * seg_el]               # (co_pad, seg_el)
                    for c0 in range(0, M, chunk):
                        sl = slice(c0, min(M, c0 + chunk))
                        okc = ok[sl]
                        if not bool(okc.any()):
                            continue
                        idx = (rowbase[sl] + off)[:, None] + torch.arange(seg_el)[None, :]
                        idx = torch.where(okc[:, None], idx, torch.zeros_like(idx))
                        assert int(idx.max()) * es + es <= in_bytes, 'gather beyond in_bytes'
                        assert int(idx.min()) >= 0
                        a = flat[idx] * okc[:, None].to(acc)
                        D[sl] += a @ wseg.t()
                        ss[sl] += (a * a).sum(dim=1)
            if rms_fused:
                D = D * (1.0 / ss.sqrt().clamp(min=1e-12))[:, None]
            if rowscale is not None:
                D = D * rowscale[:M].to(acc)[:, None]
            if bias is not None:
                D = D + bias.to(acc)[None, :]
            co = d['co']
            if d['act'] == ACT_ELU:
                D = torch.nn.functional.elu(D)
            elif d['act'] == ACT_SILU:
                D = torch.nn.functional.silu(D)
            elif d['act'] == ACT_LRELU:
                D = torch.nn.functional.leaky_relu(D, 0.1)
            if d['act'] == ACT_GEGLU:
                hp = co // 2
                Dv = D[:, :co].reshape(M, co // 64, 2, 32)
                o = (torch.nn.functional.gelu(Dv[:, :, 1]) * Dv[:, :, 0]).reshape(M, hp)
                out[: M * hp].copy_(o.reshape(-1).to(out.dtype))
                return
            if d['store'] in (STORE_ROWS, STORE_ROWS_F32):      # ROWS_F32: `out` is an fp32 tensor, nothing is rounded
                oc = d['out_c']
                o = D[:, :co]
                if residual is not None:
                    o = o + residual[: M * oc].reshape(M, oc)[:, :co].to(acc)
                assert oc == co
                out[: M * oc].copy_(o.reshape(-1).to(out.dtype))
            elif d['store'] == STORE_SHUFFLE:
                rt, rh, rw, cq = d['rt'], d['rh'], d['rw'], d['out_c']
                o = D[:, :co].reshape(B, To, Ho, Wo, rt, rh, rw, cq).permute(0, 1, 4, 2, 5, 3, 6, 7)
                out[: M * co].copy_(o.reshape(-1).to(out.dtype))
            else:
                crop = d['crop_t']
                o = D[:, :co].reshape(B, To, Ho, Wo, co)[:, crop:].permute(0, 4, 1, 2, 3)
                out.copy_(o.to(out.dtype))
        step.outs = [o for o in [out, ws] if o is not None]
        step.what = 'conv'
        return step

    # ------------------------------------------------------------------ mv2_conv3 (include/mv2.h)
    def conv3_eligible(self, d):
        # mirrors mv2_conv3_eligible() so the CPU plan has the same launch sequence as the HIP plan
        es = 2 if d['dtype'] == 0 else 4
        return (d['dtype'] in (0, 1) and d['kh'] == 3 and d['kw'] == 3 and d['st'] == d['sh'] == d['sw'] == 1 and d['ph'] == d['pw'] == 1
                and 1 <= d['kt'] <= 3 and d['pt'] == d['kt'] - 1 and d['check_bounds'] and (d['Ci'] * es) % 128 == 0
                and d['To'] == d['Ti'] and d['Ho'] == d['Hi'] and d['Wo'] == d['Wi'] and d['nsteps'] == d['kt'] * 9 * ((d['Ci'] * es) // 128)
                and d['act'] in (ACT_NONE, ACT_ELU) and d['co'] >= 64 and d['co_pad'] % 128 == 0
                and (d['store'] == STORE_ROWS or (d['store'] == STORE_ROWS_F32 and d['dtype'] == 0))
                and d['co'] % (16 // es) == 0 and d['out_c'] % (16 // es) == 0 and d['out_c'] >= d['co'])

    def conv3_fusable(self, d):
        return bool(self.conv3_eligible(d) and d['dtype'] == 0 and d['store'] == STORE_ROWS and d['co'] == 64 and d['out_c'] == d['co'])

    def conv3_sizes(self, d):
        return 0, 0

    def conv3_tiles(self, d):
        # SE pooling records of the fused launch: one per 2 x 16 strip of positions (include/mv2.h)
        th, tw = 2, 16
        return (-(-d['Ho'] // th)) * (-(-d['Wo'] // tw)), th, tw

    @staticmethod
    def _plain_from_frag_major(wp3):
        """inverse of the weight image mv2.h describes for mv2_conv3: [step][co_pad/32][k-slice 4][half 2][row 32][16 B]
        holds 16-byte chunk (2*k-slice + half) of row r of the plain [step][co_pad][128 B] layout."""
        ns, cp, kel = wp3.shape
        e16 = kel // 8
        src = wp3.reshape(ns, cp // 32, 4, 2, 32, e16)
        plain = torch.empty(ns, cp // 32, 32, 8, e16, dtype=wp3.dtype)
        for ks in range(4):
            for half in range(2):
                plain[:, :, :, 2 * ks + half, :] = src[:, :, ks, half, :, :]
        return plain.reshape(ns, cp, kel)

    def conv3(self, d, x, in_bytes, wp3, bias, out, ws, counters, wp2=None, bias2=None, act2=0,
              se_wk=None, se_bk=0.0, se_part=None):
        """stride-1 kt x 3 x 3 causal conv (+ bias, act), optionally followed by the 1x1x1 conv (+ bias2, act2) and the per-tile
        SqueezeExcite pooling partials (include/mv2.h mv2_conv3): the intermediate is rounded to the storage dtype exactly
        like the kernel's LDS tile."""
        es = x.element_size()
        kel = 128 // es
        nchunk = d['Ci'] * es // 128
        Hi, Wi, Ci = d['Hi'], d['Wi'], d['Ci']
        rows = []
        for it in range(d['kt']):
            for ih in range(3):
                for iw in range(3):
                    for ch in range(nchunk):
                        rows.append((((it * Hi + ih) * Wi + iw) * Ci + ch * kel, (1 << it) | (1 << (8 + ih)) | (1 << (16 + iw))))
        segs = torch.tensor(rows, dtype=torch.int32)
        wp = self._plain_from_frag_major(wp3)
        M = d['B'] * d['To'] * d['Ho'] * d['Wo']
        co = d['co']
        dd = dict(d, seg_bytes=128)
        if wp2 is None:
            first = self.conv(dd, x, in_bytes, wp, segs, bias, None, None, out, None)

            def step(stream):
                first(stream)
            step.outs = [out]
            step.what = 'conv3'
            return step
        mid = torch.full((M * co,), float('nan'), dtype=x.dtype)
        first = self.conv(dd, x, in_bytes, wp, segs, bias, None, None, mid, None)
        d2 = dict(d, Ti=d['To'], Hi=d['Ho'], Wi=d['Wo'], Ci=co, kt=1, kh=1, kw=1, pt=0, ph=0, pw=0, nsteps=co * es // 128,
                  seg_bytes=128, check_bounds=0, act=act2)
        segs2 = torch.tensor([[j * (128 // es), (1 << 0) | (1 << 8) | (1 << 16)] for j in range(co * es // 128)], dtype=torch.int32)
        second = self.conv(d2, mid, mid.numel() * es, wp2, segs2, bias2, None, None, out, None)

        def pool(stream):
            # per-tile SqueezeExcite pooling partials from the stored (rounded) output, mv2_se_pool record format
            n, th, tw = self.conv3_tiles(d)
            B, T, H, W = d['B'], d['To'], d['Ho'], d['Wo']
            y = out[: M * co].reshape(B * T, H, W, co).to(self.acc)
            logits = y @ se_wk.to(self.acc) + se_bk
            pv = se_part[: B * T * n * (co + 2)].reshape(B * T, n, co + 2)
            j = 0
            for h0 in range(0, H, th):
                for w0 in range(0, W, tw):
                    lg = logits[:, h0:h0 + th, w0:w0 + tw].reshape(B * T, -1)
                    yy = y[:, h0:h0 + th, w0:w0 + tw].reshape(B * T, -1, co)
                    mx = lg.max(dim=1).values
                    e = (lg - mx[:, None]).exp()
                    pv[:, j, 0] = mx.float(); pv[:, j, 1] = e.sum(dim=1).float()
                    pv[:, j, 2:] = torch.einsum('fp,fpc->fc', e, yy).float()
                    j += 1

        def step(stream):
            first(stream)
            second(stream)
            if se_part is not None:
                pool(stream)
        step.outs = [o for o in [out, se_part] if o is not None]
        step.what = 'conv3'
        return step

    # ------------------------------------------------------------------ layout edge
    def video_to_ndhwc(self, dt, video, out, B, Cc, T, H, W, pad_t, pad_hw, c_pad, tail_elems):
        def step(stream):
            Tp, Hp, Wp = T + pad_t, H + 2 * pad_hw, W + 2 * pad_hw
            buf = torch.zeros(B, Tp, Hp, Wp, c_pad, dtype=out.dtype)
            buf[:, pad_t:, pad_hw:pad_hw + H, pad_hw:pad_hw + W, :Cc] = video.permute(0, 2, 3, 4, 1).to(out.dtype)
            out[: buf.numel()].copy_(buf.reshape(-1))
            out[buf.numel(): buf.numel() + tail_elems] = 0
        step.outs = [o for o in [out] if o is not None]
        step.what = 'video_to_ndhwc'
        return step

    def _shifted(self, x, rows, Cc, shift_hw, T):
        xr = x[: rows * Cc].reshape(rows, Cc).to(self.acc)
        if shift_hw <= 0:
            return xr
        prev = torch.zeros_like(xr)
        prev[shift_hw:] = xr[:-shift_hw]
        first = ((torch.arange(rows) // shift_hw) % T) == 0
        prev[first] = 0
        return torch.cat((xr[:, : Cc // 2], prev[:, Cc // 2:]), dim=1)

    def split_bf16(self, x, out, rows, Cc):
        """include/mv2.h mv2_split_bf16: rows of [hi | lo | hi], hi = bf16(x), lo = bf16(x - hi)"""
        def step(stream):
            xv = x[: rows * Cc].reshape(rows, Cc).float()
            hi = xv.to(torch.bfloat16)
            lo = (xv - hi.float()).to(torch.bfloat16)
            out[: rows * 3 * Cc].copy_(torch.cat((hi, lo, hi), dim=1).reshape(-1))
        step.outs = [out]
        step.what = 'split_bf16'
        return step

    def rms_rowscale(self, dt, x, rs, rows, Cc, shift_hw, T):
        def step(stream):
            xs = self._shifted(x, rows, Cc, shift_hw, T)
            rs[:rows].copy_((1.0 / xs.norm(dim=1).clamp(min=1e-12)).float())
        step.outs = [o for o in [rs] if o is not None]
        step.what = 'rms_rowscale'
        return step

    # ------------------------------------------------------------------ FeedForward in one launch (mv2_ff, csrc/ff.hip)
    def ff_eligible(self, dt, rows, Cc, hidden_pad) -> bool:
        return bool(dt == 0 and rows > 0 and Cc in (256, 512) and hidden_pad >= 64 and hidden_pad % 64 == 0 and hidden_pad <= 8192)

    def ff_sizes(self, rows, Cc, hidden_pad):
        return 0, 0

    def ff(self, dt, x, w1, b1, w2, b2, out, rows, Cc, hidden_pad, shift_hw, T, ws, counters):
        """out = round(W2 round(geglu((W1 xs) / |xs| + b1)) + b2 + x), xs = the (token-shifted) row; both weight images are fragment-major
        (include/mv2.h mv2_ff); h is rounded to the storage dtype like the kernel's LDS panel, the output once."""
        assert self.ff_eligible(dt, rows, Cc, hidden_pad)
        def step(stream):
            hp = hidden_pad
            p1 = self._plain_from_frag_major(w1)                                     # [C / 64][co_pad][64]
            W1 = p1.permute(1, 0, 2).reshape(p1.shape[1], -1)[: 2 * hp, :Cc].to(self.acc)
            p2 = self._plain_from_frag_major(w2)                                     # [hp / 64][C][64]
            W2 = p2.permute(1, 0, 2).reshape(p2.shape[1], -1)[:Cc, :hp].to(self.acc)
            xs = self._shifted(x, rows, Cc, shift_hw, T)
            rs = 1.0 / xs.norm(dim=1).clamp(min=1e-12)
            D = (xs @ W1.t()) * rs[:, None] + b1[: 2 * hp].to(self.acc)
            Dv = D.reshape(rows, hp // 32, 2, 32)                                    # packed blocks: 32 values | 32 gates
            h = (torch.nn.functional.gelu(Dv[:, :, 1]) * Dv[:, :, 0]).reshape(rows, hp).to(x.dtype).to(self.acc)
            y = (h @ W2.t() + b2[:Cc].to(self.acc)) + x[: rows * Cc].reshape(rows, Cc).to(self.acc)
            out[: rows * Cc].copy_(y.reshape(-1).to(out.dtype))
        step.outs = [out]
        step.what = 'ff'
        return step

    # ------------------------------------------------------------------ squeeze-excite
    def pw_se_rows(self, dt, rows, Pn, Cc) -> int:
        if dt != 0 or Cc not in (128, 256, 512) or rows % Pn:      # MV2_BF16 = 0
            return 0
        if Pn % 64 == 0 and rows // 64 >= 512:
            return 64
        return 32 if Pn % 32 == 0 else 0

    def pw_se(self, dt, x, wp, bias, se_wk, se_bk, out, se_part, rows, Pn, Cc):
        bp = self.pw_se_rows(dt, rows, Pn, Cc)
        assert bp
        def step(stream):
            w = self._plain_from_frag_major(wp).reshape(-1, wp.shape[1], wp.shape[2])       # [steps][co][64 ch]
            w2d = w.permute(1, 0, 2).reshape(wp.shape[1], -1)[:Cc, :Cc].to(self.acc)
            y = x[: rows * Cc].reshape(rows, Cc).to(self.acc) @ w2d.t() + bias.to(self.acc)
            y = torch.where(y > 0, y, torch.expm1(y.clamp(max=0)))
            out[: rows * Cc].copy_(y.reshape(-1).to(out.dtype))
            frames, nchunk = rows // Pn, Pn // bp
            self.se_pool(dt, out, se_wk, se_bk, se_part, frames, Pn, Cc, nchunk)(stream)
        step.outs = [out, se_part]
        step.what = 'pw_se'
        return step

    def se_pool(self, dt, y, wk, bk, part, frames, Pn, Cc, nchunk):
        def step(stream):
            yr = y[: frames * Pn * Cc].reshape(frames, Pn, Cc).to(self.acc)
            logits = yr @ wk.to(self.acc) + bk
            per = -(-Pn // nchunk)
            pv = part[: frames * nchunk * (Cc + 2)].reshape(frames, nchunk, Cc + 2)
            for j in range(nchunk):
                lo, hi = j * per, min(Pn, (j + 1) * per)
                if lo >= hi:
                    pv[:, j, 0] = float('-inf'); pv[:, j, 1:] = 0
                    continue
                lg = logits[:, lo:hi]
                mx = lg.max(dim=1).values
                e = (lg - mx[:, None]).exp()
                pv[:, j, 0] = mx.float()
                pv[:, j, 1] = e.sum(dim=1).float()
                pv[:, j, 2:] = torch.einsum('fp,fpc->fc', e, yr[:, lo:hi]).float()
        step.outs = [o for o in [part] if o is not None]
        step.what = 'se_pool'
        return step

    def se_gate(self, part, w1, b1, w2, b2, gate, frames, Cc, hidden, nchunk):
        def step(stream):
            pv = part[: frames * nchunk * (Cc + 2)].reshape(frames, nchunk, Cc + 2).to(self.acc)
            M = pv[:, :, 0].max(dim=1).values
            w = (pv[:, :, 0] - M[:, None]).exp()
            S = (pv[:, :, 1] * w).sum(dim=1)
            pooled = (pv[:, :, 2:] * w[:, :, None]).sum(dim=1) / S[:, None]
            # w1 / w2 arrive transposed (include/mv2.h): w1t (C, hidden), w2t (hidden, C)
            h = torch.nn.functional.leaky_relu(pooled @ w1.to(self.acc) + b1.to(self.acc), 0.1)
            g = torch.sigmoid(h @ w2.to(self.acc) + b2.to(self.acc))
            gate[: frames * Cc].copy_(g.reshape(-1).float())
        step.outs = [o for o in [gate] if o is not None]
        step.what = 'se_gate'
        return step

    def se_apply(self, dt, y, gate, residual, out, frames, Pn, Cc):
        def step(stream):
            n = frames * Pn * Cc
            yr = y[:n].reshape(frames, Pn, Cc).to(self.acc)
            o = yr * gate[: frames * Cc].reshape(frames, 1, Cc).to(self.acc)
            if residual is not None:
                o = o + residual[:n].reshape(frames, Pn, Cc).to(self.acc)
            out[:n].copy_(o.reshape(-1).to(out.dtype))
        step.outs = [o for o in [out] if o is not None]
        step.what = 'se_apply'
        return step

    def se_pool_fc1(self, dt, y, wk, bk, w1, b1, hid, frames, Pn, Cc, hidden, nslice):
        def step(stream):
            yr = y[: frames * Pn * Cc].reshape(frames, Pn, Cc).to(self.acc)
            a = torch.softmax(yr @ wk.to(self.acc) + bk, dim=1)
            pooled = torch.einsum('fp,fpc->fc', a, yr)
            h = torch.nn.functional.leaky_relu(pooled @ w1.to(self.acc) + b1.to(self.acc), 0.1)
            hid[: frames * hidden].copy_(h.reshape(-1).float())
        step.outs = [hid]
        step.what = 'se_pool_fc1'
        return step

    def copy_frames(self, src, src_T, src_t0, dst, dst_T, dst_t0, nT, B, frame_bytes):
        def step(stream):
            fe = frame_bytes // dst.element_size()
            d = dst.reshape(-1)[: B * dst_T * fe].reshape(B, dst_T, fe)
            if src is None:
                d[:, dst_t0: dst_t0 + nT] = 0
            else:
                d[:, dst_t0: dst_t0 + nT] = src.reshape(-1)[: B * src_T * fe].reshape(B, src_T, fe)[:, src_t0: src_t0 + nT]
        step.outs = [dst]
        step.what = 'copy_frames'
        return step

    def gateloop_scan(self, dt, qkva, residual, out, B, T, Pn, Cc):
        """mv2_gateloop_scan: h_t = sigmoid(a_t) h_{t-1} + kv_t, out_t = q_t h_t + residual_t along time, per pixel and channel"""
        def step(stream):
            n = B * T * Pn
            z = qkva.reshape(-1)[: n * 3 * Cc].reshape(B, T, Pn, 3, Cc).to(self.acc)
            r = residual.reshape(-1)[: n * Cc].reshape(B, T, Pn, Cc).to(self.acc)
            q, kv, a = z[..., 0, :], z[..., 1, :], torch.sigmoid(z[..., 2, :])
            h = torch.zeros(B, Pn, Cc, dtype=self.acc)
            o = torch.empty_like(r)
            for t in range(T):
                h = a[:, t] * h + kv[:, t]
                o[:, t] = q[:, t] * h + r[:, t]
            out.reshape(-1)[: n * Cc].copy_(o.reshape(-1).to(out.dtype))
        step.outs = [out]
        step.what = 'gateloop_scan'
        return step

    def pad_ndhwc(self, x, out, B, T, H, W, row_bytes, pt, ph, pw, mode, tail_bytes):
        """mv2_pad_ndhwc: F.pad(x, (pw, pw, ph, ph, pt, 0), mode) on an NDHWC tensor (mode 1 reflect / 2 replicate / 3 circular) + zero tail"""
        def step(stream):
            es = x.element_size()
            Cc = row_bytes // es
            src = x.reshape(-1)[: B * T * H * W * Cc].reshape(B, T, H, W, Cc)

            def idx(n, lo, hi):
                j = torch.arange(-lo, n + hi)
                if mode == 1:
                    j = j.abs()
                    j = torch.where(j >= n, 2 * (n - 1) - j, j)
                elif mode == 2:
                    j = j.clamp(0, n - 1)
                else:
                    j = j % n
                return j
            o = src[:, idx(T, pt, 0)][:, :, idx(H, ph, ph)][:, :, :, idx(W, pw, pw)]
            n = o.numel()
            flat = out.reshape(-1)
            flat[:n] = o.reshape(-1)
            flat[n: n + tail_bytes // es] = 0
        step.outs = [out]
        step.what = 'pad_ndhwc'
        return step

    def cond_linear(self, x, w, bias, out, B, I, O, mode, eps=1e-8):
        def step(stream):
            xv = x.reshape(-1)[: B * I].reshape(B, I).double()
            if mode == 2:
                s_ = ((xv + 1) ** 2) @ w.double().t()
                o = s_.clamp(min=eps).rsqrt()
            else:
                o = xv @ w.double().t() + (bias.double() if bias is not None else 0.)
                if mode == 1:
                    o = o * torch.sigmoid(o)
            out.reshape(-1)[: B * O].copy_(o.reshape(-1).float())
        step.outs = [out]
        step.what = 'cond_linear'
        return step

    def scale_channels(self, dt, x, scale, out, rows, rows_per_sample, Cc, add, act):
        def step(stream):
            B = rows // rows_per_sample
            v = x[: rows * Cc].reshape(B, rows_per_sample, Cc).to(self.acc) * (scale.reshape(-1)[: B * Cc].reshape(B, 1, Cc).to(self.acc) + add)
            if act == 1:
                v = torch.where(v > 0, v, torch.expm1(v.clamp(max=0)))
            out[: rows * Cc].copy_(v.reshape(-1).to(out.dtype))
        step.outs = [out]
        step.what = 'scale_channels'
        return step

    def se_rec_fc1(self, part, w1, b1, hid, frames, Cc, nchunk, hidden, nslice):
        def step(stream):
            pv = part[: frames * nchunk * (Cc + 2)].reshape(frames, nchunk, Cc + 2).to(self.acc)
            M = pv[:, :, 0].max(dim=1).values
            w = (pv[:, :, 0] - M[:, None]).exp()
            S = (pv[:, :, 1] * w).sum(dim=1)
            pooled = (pv[:, :, 2:] * w[:, :, None]).sum(dim=1) / S[:, None]
            h = torch.nn.functional.leaky_relu(pooled @ w1.to(self.acc) + b1.to(self.acc), 0.1)
            hid[: frames * hidden].copy_(h.reshape(-1).float())
        step.outs = [hid]
        step.what = 'se_rec_fc1'
        return step

    def se_fc2_apply(self, dt, y, hid, w2, b2, residual, out, frames, Pn, Cc, hidden, nslice):
        def step(stream):
            n = frames * Pn * Cc
            g = torch.sigmoid(hid[: frames * hidden].reshape(frames, hidden).to(self.acc) @ w2.to(self.acc) + b2.to(self.acc))
            o = y[:n].reshape(frames, Pn, Cc).to(self.acc) * g[:, None, :]
            if residual is not None:
                o = o + residual[:n].reshape(frames, Pn, Cc).to(self.acc)
            out[:n].copy_(o.reshape(-1).to(out.dtype))
        step.outs = [out]
        step.what = 'se_fc2_apply'
        return step

    # ------------------------------------------------------------------ attention
    def attention(self, dt, qkv, mem_kv, out, nseq, n, heads, dim_head, n_mem, causal, seq_inner, seq_outer_stride, tok_stride):
        def step(stream):
            ld = 3 * heads * dim_head
            s = torch.arange(nseq)
            base = (s // seq_inner) * seq_outer_stride + (s % seq_inner)
            rows = base[:, None] + torch.arange(n)[None, :] * tok_stride                # (nseq, n)
            nrows = int(rows.max()) + 1
            Q = qkv[: nrows * ld].reshape(nrows, 3, heads, dim_head).to(self.acc)[rows]  # (nseq, n, 3, h, d)
            q, k, v = Q[:, :, 0].transpose(1, 2), Q[:, :, 1].transpose(1, 2), Q[:, :, 2].transpose(1, 2)   # (nseq,h,n,d)
            mk = mem_kv[0].to(self.acc).expand(nseq, -1, -1, -1)
            mv = mem_kv[1].to(self.acc).expand(nseq, -1, -1, -1)
            k = torch.cat((mk, k), dim=2); v = torch.cat((mv, v), dim=2)
            dots = torch.einsum('shid,shjd->shij', q, k) * dim_head ** -0.5
            if causal:
                j = k.shape[2]
                mask = torch.ones((n, j), dtype=torch.bool).triu(j - n + 1)
                dots = dots.masked_fill(mask, float('-inf'))
            o = torch.einsum('shij,shjd->shid', dots.softmax(dim=-1), v).transpose(1, 2).reshape(nseq, n, heads * dim_head)
            ov = out[: nrows * heads * dim_head].reshape(nrows, heads * dim_head)
            ov[rows.reshape(-1)] = o.reshape(-1, heads * dim_head).to(out.dtype)
        step.outs = [o for o in [out] if o is not None]
        step.what = 'attention'
        return step

    @staticmethod
    def _phi(x):
        x0 = x.new_ones(x.shape[:-1] + (1,))
        x2 = (x.unsqueeze(-1) * x.unsqueeze(-2)).flatten(-2) * (0.5 ** 0.5)
        return torch.cat((x0, x, x2), dim=-1)

    def taylor_kv(self, dt, kv, part, frames, n, heads, nchunk):
        def step(stream):
            KV = kv[: frames * n * 2 * heads * 8].reshape(frames, n, 2, heads, 8).to(self.acc)
            k, v = KV[:, :, 0], KV[:, :, 1]                                             # (f, n, h, 8)
            v1 = torch.cat((v, v.new_ones(v.shape[:-1] + (1,))), dim=-1)
            ph = self._phi(k)
            per = -(-n // nchunk)
            pv = part[: frames * nchunk * heads * 657].reshape(frames, nchunk, heads, 73, 9)
            for j in range(nchunk):
                lo, hi = j * per, min(n, (j + 1) * per)
                if lo >= hi:
                    pv[:, j] = 0
                    continue
                pv[:, j] = torch.einsum('fnhd,fnhe->fhde', ph[:, lo:hi], v1[:, lo:hi]).float()
        step.outs = [o for o in [part] if o is not None]
        step.what = 'taylor_kv'
        return step

    def taylor_apply(self, dt, q, part, out, kvsum, frames, n, heads, nchunk):
        def step(stream):
            S = part[: frames * nchunk * heads * 657].reshape(frames, nchunk, heads, 73, 9).to(self.acc).sum(dim=1)
            if nchunk > 1:                       # nchunk == 1: `part` already holds the finished states, the library does not touch kvsum
                kvsum[: frames * heads * 657].copy_(S.reshape(-1).float())
            qq = q[: frames * n * heads * 8].reshape(frames, n, heads, 8).to(self.acc) * 8 ** -0.5
            acc = torch.einsum('fnhd,fhde->fnhe', self._phi(qq), S)
            o = acc[..., :8] * (1.0 / acc[..., 8:].clamp(min=1e-5))
            out[: frames * n * heads * 8].copy_(o.reshape(-1).to(out.dtype))
        step.outs = [o for o in ([out, kvsum] if nchunk > 1 else [out]) if o is not None]
        step.what = 'taylor_apply'
        return step

    @staticmethod
    def taylor_state_floats(d: int) -> int:
        """mirror of mv2_taylor_state_floats: (1 + d + d^2) features x (d + 1) columns, 0 for a width without kernels"""
        return (1 + d + d * d) * (d + 1) if d in (4, 8, 16) else 0

    def taylor_kv_d(self, dt, kv, state, frames, n, heads, d):
        """mv2_taylor_kv_d: the finished state [frames][heads][F][d + 1] at any supported dim_head (rows layout)"""
        def step(stream):
            F = 1 + d + d * d
            KV = kv[: frames * n * 2 * heads * d].reshape(frames, n, 2, heads, d).to(self.acc)
            k, v = KV[:, :, 0], KV[:, :, 1]
            v1 = torch.cat((v, v.new_ones(v.shape[:-1] + (1,))), dim=-1)
            st = torch.einsum('fnhd,fnhe->fhde', self._phi(k), v1).float()
            state[: frames * heads * F * (d + 1)].copy_(st.reshape(-1))
        step.outs = [state]
        step.what = 'taylor_kv_d'
        return step

    def taylor_apply_d(self, dt, q, state, out, frames, n, heads, d):
        def step(stream):
            F = 1 + d + d * d
            S = state[: frames * heads * F * (d + 1)].reshape(frames, heads, F, d + 1).to(self.acc)
            qq = q[: frames * n * heads * d].reshape(frames, n, heads, d).to(self.acc) * d ** -0.5
            acc = torch.einsum('fnhd,fhde->fnhe', self._phi(qq), S)
            o = acc[..., :d] * (1.0 / acc[..., d:].clamp(min=1e-5))
            out[: frames * n * heads * d].copy_(o.reshape(-1).to(out.dtype))
        step.outs = [out]
        step.what = 'taylor_apply_d'
        return step

    def _qkv_fused(self, qkv, frames, n, heads):
        """head-major fused projection output [frame][head][token][q 8 | k 8 | v 8] -> q, k, v as (f, n, h, 8)"""
        t = qkv[: frames * heads * n * 24].reshape(frames, heads, n, 3, 8).to(self.acc).permute(0, 2, 1, 3, 4)
        return t[..., 0, :], t[..., 1, :], t[..., 2, :]

    def taylor_kv_fused(self, dt, qkv, part, frames, n, heads, nchunk):
        def step(stream):
            _, k, v = self._qkv_fused(qkv, frames, n, heads)
            kv = torch.stack((k, v), dim=2).reshape(-1).to(qkv.dtype)          # rows [token][k | v]: the row-layout emulation does the rest
            self.taylor_kv(dt, kv, part, frames, n, heads, nchunk)(stream)
        step.outs = [part]
        step.what = 'taylor_kv_fused'
        return step

    def taylor_apply_fused(self, dt, qkv, part, out, kvsum, frames, n, heads, nchunk):
        def step(stream):
            q, _, _ = self._qkv_fused(qkv, frames, n, heads)
            self.taylor_apply(dt, q.reshape(-1).to(qkv.dtype), part, out, kvsum, frames, n, heads, nchunk)(stream)
        step.outs = [o for o in ([out, kvsum] if nchunk > 1 else [out]) if o is not None]
        step.what = 'taylor_apply_fused'
        return step

    # ------------------------------------------------------------------ quantizers
    def lfq(self, dt, x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, Cc, cb_dim, ncb, clamp, scale):
        def step(stream):
            z = x[: rows * Cc].reshape(rows, Cc).to(self.acc) @ w_in.to(self.acc).t() + b_in.to(self.acc)
            if clamp > 0:
                z = (z / clamp).tanh() * clamp
            pos = (z > 0).reshape(rows, ncb, cb_dim)
            mask = (2 ** torch.arange(cb_dim - 1, -1, -1)).long()
            ids[:rows].copy_((pos.long() * mask).sum(dim=-1))
            if prequant is not None:
                prequant[:rows].copy_(z.float())
            code = torch.where(z > 0, scale, -scale).to(self.acc)
            o = code @ w_out.to(self.acc) + b_out.to(self.acc)
            quantized[: rows * Cc].copy_(o.reshape(-1).to(quantized.dtype))
        step.outs = [o for o in [quantized, ids, prequant] if o is not None]
        step.what = 'lfq'
        return step

    def lfq_train(self, dt, x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, Cc, cb_dim, ncb, clamp, scale):
        assert prequant is not None
        def step(stream):
            z = x[: rows * Cc].reshape(rows, Cc).to(self.acc) @ w_in.to(self.acc).t() + b_in.to(self.acc)
            if clamp > 0:
                z = (z / clamp).tanh() * clamp
            pos = (z > 0).reshape(rows, ncb, cb_dim)
            mask = (2 ** torch.arange(cb_dim - 1, -1, -1)).long()
            ids[:rows].copy_((pos.long() * mask).sum(dim=-1))
            prequant[:rows].copy_(z.float())
            zf = z.float()
            code = (zf + (torch.where(zf > 0, scale, -scale) - zf)).to(self.acc)        # straight-through forward value
            o = code @ w_out.to(self.acc) + b_out.to(self.acc)
            quantized[: rows * Cc].copy_(o.reshape(-1).to(quantized.dtype))
        step.outs = [quantized, ids, prequant]
        step.what = 'lfq_train'
        return step

    def lfq_stats_workspace_bytes(self, rows, cb_dim, ncb) -> int:
        return min(rows, 512) * (ncb * (1 << cb_dim) + 2) * 4

    def lfq_stats(self, prequant, rows, cb_dim, ncb, inv_temperature, scale, sums, ws):
        def step(stream):
            K = 1 << cb_dim
            z = prequant[:rows].reshape(rows, ncb, cb_dim).double()
            j = torch.arange(K)
            bits = ((j[:, None] >> torch.arange(cb_dim - 1, -1, -1)[None, :]) & 1).double() * 2 - 1
            logits = 2 * inv_temperature * scale * torch.einsum('ncd,kd->nck', z, bits)
            prob = logits.softmax(dim=-1)
            ent = (-prob * prob.clamp(min=1e-5).log()).sum()
            q = torch.where(z > 0, scale, -scale)
            sums[: ncb * K].copy_(prob.sum(dim=0).reshape(-1).float())
            sums[ncb * K] = float(ent)
            sums[ncb * K + 1] = float(((z - q) ** 2).sum())
        step.outs = [sums]
        step.what = 'lfq_stats'
        return step

    def lfq_losses(self, sums, n_tokens, cb_dim, ncb, gamma, w_entropy, w_commit, losses):
        def step(stream):
            K = 1 << cb_dim
            avg = sums[: ncb * K].reshape(ncb, K).double() / n_tokens
            hb = float((-avg * avg.clamp(min=1e-5).log()).sum(dim=-1).mean())
            hs = float(sums[ncb * K]) / n_tokens / ncb
            cm = float(sums[ncb * K + 1]) / n_tokens / (ncb * cb_dim) if w_commit > 0 else 0.
            losses.copy_(torch.tensor([hs, hb, cm, (hs - gamma * hb) * w_entropy + cm * w_commit]))
        step.outs = [losses]
        step.what = 'lfq_losses'
        return step

    def sq_diff_workspace_bytes(self, n) -> int:
        return min((n + 4095) // 4096, 1024) * 4

    def sq_diff_sum(self, a, b, n, out, ws):
        def step(stream):
            out[0] = float(((a.reshape(-1)[:n].double() - b.reshape(-1)[:n].double()) ** 2).sum())
        step.outs = [out]
        step.what = 'sq_diff_sum'
        return step

    def hinge_discr_loss(self, fake, real, n, out):
        def step(stream):
            f_, r_ = fake.reshape(-1)[:n].double(), real.reshape(-1)[:n].double()
            out[0] = float((torch.relu(1 + f_) + torch.relu(1 - r_)).mean())
        step.outs = [out]
        step.what = 'hinge_discr_loss'
        return step

    def lfq_codes(self, dt, ids, w_out, b_out, out, rows, Cc, cb_dim, ncb, scale):
        def step(stream):
            mask = (2 ** torch.arange(cb_dim - 1, -1, -1)).long()
            bits = (ids[:rows].long().reshape(rows, ncb, 1) & mask) != 0
            code = torch.where(bits, scale, -scale).to(self.acc).reshape(rows, ncb * cb_dim)
            o = code @ w_out.to(self.acc) + b_out.to(self.acc)
            out[: rows * Cc].copy_(o.reshape(-1).to(out.dtype))
        step.outs = [o for o in [out] if o is not None]
        step.what = 'lfq_codes'
        return step

    def fsq(self, dt, x, w_in, b_in, w_out, b_out, consts, quantized, ids, prequant, rows, Cc, cb_dim, ncb):
        def step(stream):
            z = x[: rows * Cc].reshape(rows, Cc).to(self.acc) @ w_in.to(self.acc).t() + b_in.to(self.acc)
            z = z.reshape(rows, ncb, cb_dim).float()
            bounded = (z + consts['shift']).tanh() * consts['half_l'] - consts['offset']
            r = bounded.round()
            code = r / consts['half_w']
            lvl = (code * consts['half_w'] + consts['half_w']) * consts['basis']
            ids[:rows].copy_(lvl.sum(dim=-1).to(torch.int32))
            if prequant is not None:
                prequant[:rows].copy_(bounded.reshape(rows, -1))
            o = code.reshape(rows, -1).to(self.acc) @ w_out.to(self.acc) + b_out.to(self.acc)
            quantized[: rows * Cc].copy_(o.reshape(-1).to(quantized.dtype))
        step.outs = [o for o in [quantized, ids, prequant] if o is not None]
        step.what = 'fsq'
        return step

    def fsq_codes(self, dt, ids, w_out, b_out, consts, out, rows, Cc, cb_dim, ncb):
        def step(stream):
            lv = (ids[:rows].reshape(rows, ncb, 1) // consts['basis']) % consts['levels']
            code = ((lv - consts['half_w']) / consts['half_w']).reshape(rows, -1).to(self.acc)
            o = code @ w_out.to(self.acc) + b_out.to(self.acc)
            out[: rows * Cc].copy_(o.reshape(-1).to(out.dtype))
        step.outs = [o for o in [out] if o is not None]
        step.what = 'fsq_codes'
        return step
