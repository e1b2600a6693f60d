"""
Execution plans: the layer-spec tuple lowered to a flat list of kernel launches.

``PackedModel`` holds the dtype-specific packed weights of one state_dict (built once per dtype /
device).  ``Plan`` lowers encode / quantize / decode for a fixed input geometry (B, T, H, W) into
*steps* — closures ``step(stream)`` produced by an ops backend (``ops_hip.HipOps`` in the product;
tests inject the CPU emulation of the kernel semantics from ``oracle/ndhwc_ops.py`` to validate this
host logic where no GPU exists).  All activations are NDHWC; buffers are allocated once at plan
build (static addresses -> HIP-graph capturable) and recycled through a size-keyed free list so the
working set stays inside the 256 MiB Infinity Cache where it can.

Reference control flow being lowered: VideoTokenizer.encode magvit2_pytorch.py:1522-1576,
decode :1597-1649, decode_from_code_indices :1578-1595, forward (eval branches) :1656-1720.
"""
from __future__ import annotations

import contextlib
import dataclasses
import math
from typing import Dict, List, Optional

import torch

from . import pack as P
from .spec import Stage, parse_layers, se_hidden, ff_inner, has_cond, cond_expanded


@dataclasses.dataclass
class LoweringOptions:
    """Which kernel form the lowering picks where two exist.  The defaults are the product; the other settings exist so that the lockstep
    tests reach every kernel form and for A/B timing.  Explicit (``with lowering_options(...)``), never read from the environment: a plan
    bakes them in when it is built, and a stray variable must not change which kernel runs."""
    fuse_pw: bool = True           # bf16 C = 64: the unit's 1x1x1 conv + ELU inside mv2_conv3
    fuse_pool: bool = True         # ... and the SqueezeExcite pooling records from that epilogue
    pw_se: bool = True             # mv2_pw_se (1x1x1 conv + ELU + pooling records) instead of mv2_conv + a pooling pass
    se_two_launch: bool = True     # small frames: mv2_se_pool_fc1 / mv2_se_fc2_apply instead of pool / gate / apply
    conv_rms: bool = True          # RMSNorm folded into the following GEMM (mv2_conv_rms)
    fuse_ff: bool = True           # FeedForward (RMSNorm + ff1 + GEGLU + ff2 + residual) as one launch (mv2_ff) at C = 256 / 512
    # ... from this many rows on.  Measured on MI355X (profiles/r05_findings.md, one launch against the two it replaces): 20480 rows at
    # C = 256: 46 against 53 us, 5120 rows at C = 512: 39 / 44, 81920 rows: 125 / 200, 20480 rows at C = 512: 151 / 159, 5120 token-shifted
    # rows: 43 / 51 - but 1280 token-shifted rows (20 row tiles for 256 CUs): 32 against 27.  0 forces the kernel (lockstep tests).
    fuse_ff_min_rows: int = 2560
    fuse_qkv: bool = True          # linear attention: to_q and to_kv as ONE GEMM with a head-major store (mv2_taylor_*_fused) instead of two GEMMs
    conv_in: bool = True           # convin.hip instead of the generic gather kernel
    conv_out: bool = True          # convout.hip instead of the generic gather kernel
    convin_cpad: int = 4           # channels per input pixel of mv2_video_to_ndhwc (4: 8-byte bf16 pixels, 8: the round-1 layout)


OPTIONS = LoweringOptions()


@contextlib.contextmanager
def lowering_options(**kw):
    """Temporarily replace fields of ``OPTIONS`` (PackedModel / Plan objects built inside the block keep what they read)."""
    global OPTIONS
    bad = set(kw) - {fl.name for fl in dataclasses.fields(LoweringOptions)}
    if bad:
        raise TypeError(f'unknown lowering option(s): {sorted(bad)}')
    if 'convin_cpad' in kw and kw['convin_cpad'] not in (4, 8):
        raise ValueError('convin_cpad must be 4 or 8')
    prev = OPTIONS
    OPTIONS = dataclasses.replace(prev, **kw)
    try:
        yield OPTIONS
    finally:
        OPTIONS = prev


PAD_MODES = {'constant': 0, 'reflect': 1, 'replicate': 2, 'circular': 3}      # mv2_pad_ndhwc mode codes (0: no call, the gathers pad with zeros)


class PackedModel:
    def __init__(self, cfg: dict, sd: Dict[str, torch.Tensor], dtype: torch.dtype, ops, split_k3: bool = False):
        """split_k3 (fp32 storage only): the ResidualUnit k x 3 x 3 convolutions additionally get the split-bf16 weight image
        (pack.pack_conv3d_split) and their launches run x_hi.w_hi + x_lo.w_hi + x_hi.w_lo on the bf16 matrix pipe with fp32 rows in and
        out (mv2_split_bf16 + mv2_conv3 with MV2_STORE_ROWS_F32) instead of the exact-fp32 MFMA: the id-exact mode that is not 16 x slower
        on its dominant launches (tools/split_bf16_study.py: pre-quantizer error ~1e-6 against a sign margin of 1e-4)."""
        self.cfg, self.dtype, self.ops = cfg, dtype, ops
        self.split_k3 = bool(split_k3) and dtype == torch.float32
        self.dt = P.dtype_code(dtype)
        self.stages: List[Stage] = parse_layers(cfg['layers'], cfg['init_dim'], cfg['max_dim'])
        # 4-channel pixels: 8-byte (bf16) / 16-byte (fp32) rows; 16-byte gathers at 8-byte alignment verified on gfx950
        self.c_in_pad = OPTIONS.convin_cpad
        if cfg['channels'] > self.c_in_pad:          # 5 .. 64 input channels: pixels of 8, 16, ... channels (whole 16-byte chunks in bf16; 8 = the round-1 layout)
            self.c_in_pad = (cfg['channels'] + 7) // 8 * 8
        f = lambda k: sd[k].detach().float().cpu()
        dev = ops.to_device
        self.convs: Dict[str, P.PackedConv] = {}
        self.f32: Dict[str, torch.Tensor] = {}
        self.hidden_pad: Dict[str, int] = {}
        self.frag: Dict[str, torch.Tensor] = {}

        def put(name, pc):
            self.convs[name] = pc.to(ops.device)

        put('conv_in', P.pack_conv_in(f('conv_in.conv.weight'), f('conv_in.conv.bias'), dtype, self.c_in_pad))
        put('conv_out', P.pack_conv3d(f('conv_out.conv.weight'), f('conv_out.conv.bias'), dtype))
        wi = f('conv_in.conv.weight')
        if dtype == torch.bfloat16 and self.c_in_pad == 4 and wi.shape[0] == 64 and tuple(wi.shape[3:]) == (7, 7) and wi.shape[2] <= 7:
            self.frag['conv_in'] = dev(P.pack_conv_in_frag(wi))            # mv2_conv_in's fragment image (convin.hip)
        wo = f('conv_out.conv.weight')
        if dtype == torch.bfloat16 and wo.shape[1] == 64 and wo.shape[0] <= 4 and tuple(wo.shape[3:]) == (3, 3) and wo.shape[2] <= 3:
            self.frag['conv_out'] = dev(P.pack_conv_out(wo))               # mv2_conv_out's fragment image (convout.hip)
        self.separate_first_frame = bool(cfg.get('separate_first_frame_encoding'))
        if self.separate_first_frame:      # SameConv2d pair = the same packers with one time tap (reference :1113-1120)
            put('conv_in_first_frame', P.pack_conv_in(f('conv_in_first_frame.weight')[:, :, None], f('conv_in_first_frame.bias'), dtype, self.c_in_pad))
            put('conv_out_first_frame', P.pack_conv3d(f('conv_out_first_frame.weight')[:, :, None], f('conv_out_first_frame.bias'), dtype))

        def residual_unit(pre):
            w3 = f(pre + '.fn.0.conv.weight')
            put(pre + '.k3', P.pack_conv3d(w3, f(pre + '.fn.0.conv.bias'), dtype, frag_major=P.conv3_packable(w3, dtype)))
            if self.split_k3 and P.conv3_packable(w3, torch.bfloat16):
                put(pre + '.k3s', P.pack_conv3d_split(w3, f(pre + '.fn.0.conv.bias')))
            put(pre + '.k1', P.pack_pointwise(f(pre + '.fn.2.weight').flatten(1), f(pre + '.fn.2.bias'), dtype))
            pc1 = self.convs[pre + '.k1']
            if dtype == torch.bfloat16 and pc1.co in (128, 256, 512) and pc1.seg_bytes == 128 and pc1.co_pad == pc1.co:
                self.frag[pre + '.k1'] = P.to_frag_major(pc1.wp)       # mv2_pw_se's weight image (same bytes, fragment order)
            c = sd[pre + '.fn.2.weight'].shape[0]
            self.f32[pre + '.se.wk'] = dev(f(pre + '.fn.4.to_k.weight').reshape(c).contiguous())
            self.f32[pre + '.se.bk'] = f(pre + '.fn.4.to_k.bias').reshape(())          # host scalar
            # transposed for coalesced thread-per-output reads: w1t (C, hidden), w2t (hidden, C)
            self.f32[pre + '.se.w1'] = dev(f(pre + '.fn.4.net.0.weight').flatten(1).t().contiguous())
            self.f32[pre + '.se.b1'] = dev(f(pre + '.fn.4.net.0.bias'))
            self.f32[pre + '.se.w2'] = dev(f(pre + '.fn.4.net.2.weight').flatten(1).t().contiguous())
            self.f32[pre + '.se.b2'] = dev(f(pre + '.fn.4.net.2.bias'))

        def adaptive(pre, dim):
            # AdaptiveRMSNorm (reference M:278-323): gamma = to_gamma(cond) is per sample, so only sqrt(dim) folds into the weights
            self.f32[pre + '.to_gamma.w'] = dev(f(pre + '.to_gamma.weight').contiguous())
            self.f32[pre + '.to_gamma.b'] = dev(f(pre + '.to_gamma.bias'))
            return torch.full((dim,), dim ** 0.5)

        def feed_forward(pre, dim, token_shift, cond=False):
            g = adaptive(pre + '.norm', dim) if cond else f(pre + '.norm.gamma').reshape(dim) * (dim ** 0.5)
            pc, hp = P.pack_geglu_in(f(pre + '.net.0.weight').flatten(1), f(pre + '.net.0.bias'), dtype, in_scale=g,
                                     token_shift=token_shift)
            put(pre + '.ff1', pc)
            self.hidden_pad[pre] = hp
            put(pre + '.ff2', P.pack_pointwise(f(pre + '.net.2.weight').flatten(1), f(pre + '.net.2.bias'), dtype, k_pad=hp))
            pc2 = self.convs[pre + '.ff2']
            if (dtype == torch.bfloat16 and not cond and dim in (256, 512) and pc.seg_bytes == 128 and pc2.seg_bytes == 128 and pc2.co_pad == dim
                    and pc.nsteps == dim // 64 and pc2.nsteps == hp // 64 and pc.bias is not None and pc2.bias is not None):
                # mv2_ff's weight images: the same bytes in fragment order (a token-shifted first conv has the same K order: current-frame
                # half, then previous-frame half)
                self.frag[pre + '.ff1'] = P.to_frag_major(pc.wp)
                self.frag[pre + '.ff2'] = P.to_frag_major(pc2.wp)

        def residual_unit_mod(pre):
            # ResidualUnitMod / Conv3DMod (reference M:680-753, 946-988): shared weights, per-sample input modulation and
            # output demodulation; wsq[o][i] = sum over taps of w^2 feeds the demodulation factors
            w3 = f(pre + '.conv.weights')
            put(pre + '.k3', P.pack_conv3d(w3, torch.zeros(w3.shape[0]), dtype, frag_major=P.conv3_packable(w3, dtype)))
            put(pre + '.k1', P.pack_pointwise(f(pre + '.conv_out.weight').flatten(1), f(pre + '.conv_out.bias'), dtype))
            self.f32[pre + '.wsq'] = dev(w3.pow(2).sum(dim=(2, 3, 4)).contiguous())
            self.f32[pre + '.to_cond.w'] = dev(f(pre + '.to_cond.weight').contiguous())
            self.f32[pre + '.to_cond.b'] = dev(f(pre + '.to_cond.bias'))

        def attention(pre, dim, token_shift, cond=False):
            g = adaptive(pre + '.norm', dim) if cond else f(pre + '.norm.gamma') * (dim ** 0.5)
            packer = P.pack_token_shift if token_shift else P.pack_pointwise
            put(pre + '.qkv', packer(f(pre + '.to_qkv.0.weight'), None, dtype, in_scale=g))
            put(pre + '.out', P.pack_pointwise(f(pre + '.to_out.1.weight'), None, dtype))
            self.f32[pre + '.mem_kv'] = dev(f(pre + '.mem_kv').contiguous())

        def linear_attention(pre, dim):
            g = f(pre + '.norm.gamma') * (dim ** 0.5)
            put(pre + '.q', P.pack_pointwise(f(pre + '.attn.to_q.0.weight'), None, dtype, in_scale=g))
            put(pre + '.kv', P.pack_pointwise(f(pre + '.attn.to_kv.0.weight'), None, dtype, in_scale=g))
            heads = cfg['linear_attn_heads']
            if OPTIONS.fuse_qkv and cfg['linear_attn_dim_head'] == 8 and (24 * P.esize(dtype)) % 16 == 0:
                put(pre + '.qkv', P.pack_taylor_qkv(f(pre + '.attn.to_q.0.weight'), f(pre + '.attn.to_kv.0.weight'), heads, dtype, in_scale=g))
            put(pre + '.out', P.pack_pointwise(f(pre + '.attn.to_out.0.weight'), None, dtype))

        for st in self.stages:
            for pre, enc in ((st.enc_prefix, True), (st.dec_prefix, False)):
                if st.kind == 'residual':
                    residual_unit(pre)
                elif st.kind == 'consecutive_residual':
                    for j in range(st.n):
                        residual_unit(f'{pre}.{j}')
                elif st.kind == 'compress_space':
                    if enc:
                        w = f(pre + '.conv.weight')
                        put(pre, P.pack_conv3d(w[:, :, None], f(pre + '.conv.bias'), dtype))
                    else:
                        put(pre, P.pack_shuffle_up(f(pre + '.net.0.weight').flatten(1), f(pre + '.net.0.bias'), dtype, 4))
                elif st.kind == 'compress_time':
                    if enc:
                        w = f(pre + '.conv.weight')
                        put(pre, P.pack_conv3d(w[:, :, :, None, None], f(pre + '.conv.bias'), dtype))
                    else:
                        put(pre, P.pack_shuffle_up(f(pre + '.net.0.weight').flatten(1), f(pre + '.net.0.bias'), dtype, 2))
                elif st.kind == 'attend_space':
                    attention(pre + '.0.fn', st.dim, False)
                    feed_forward(pre + '.1.fn', st.dim, False)
                elif st.kind == 'linear_attend_space':
                    linear_attention(pre + '.0.fn', st.dim)
                    feed_forward(pre + '.1.fn', st.dim, False)
                elif st.kind == 'attend_time':
                    attention(pre + '.0.fn.fn', st.dim, True)
                    feed_forward(pre + '.1.fn.fn', st.dim, True)
                elif st.kind == 'gateloop_time':
                    g = f(pre + '.fn.fn.norm.gamma') * (st.dim ** 0.5)
                    put(pre + '.fn.fn.qkva', P.pack_pointwise(f(pre + '.fn.fn.to_qkva.0.weight'), None, dtype, in_scale=g))
                elif st.kind == 'cond_residual':
                    residual_unit_mod(pre)
                elif st.kind == 'cond_attend_space':
                    attention(pre + '.0.fn', st.dim, False, cond=True)
                    feed_forward(pre + '.1.fn', st.dim, False, cond=True)
                elif st.kind == 'cond_attend_time':
                    attention(pre + '.0.fn.fn', st.dim, True, cond=True)
                    feed_forward(pre + '.1.fn.fn', st.dim, True, cond=True)
        self.has_cond = has_cond(cfg)
        if self.has_cond:
            self.dim_cond, self.dim_cond_e = cfg['dim_cond'], cond_expanded(cfg)
            for which in ('encoder_cond_in', 'decoder_cond_in'):
                self.f32[which + '.w'] = dev(f(which + '.0.weight').contiguous())
                self.f32[which + '.b'] = dev(f(which + '.0.bias'))

        # quantizer (fp32 weights; per-token wave kernels)
        self.latent_dim = self.stages[-1].dim_out if self.stages else cfg['init_dim']
        self.use_fsq = bool(cfg['use_fsq'])
        self.ncb = cfg['num_codebooks']
        if self.use_fsq:
            levels = list(cfg['fsq_levels'])
            self.cb_dim = len(levels)
            L = torch.tensor(levels, dtype=torch.int32)
            half_l = (L - 1) * (1 + 1e-3) / 2
            offset = torch.where(L % 2 == 0, 0.5, 0.0)
            self.fsq = dict(levels=dev(L), basis=dev(torch.cumprod(torch.tensor([1] + levels[:-1]), dim=0, dtype=torch.int32)),
                            half_l=dev(half_l.float()), offset=dev(offset.float()), shift=dev((offset / half_l).atanh().float()),
                            half_w=dev((L // 2).float()))
        else:
            self.cb_dim = int(math.log2(cfg['codebook_size']))
        cbt = self.cb_dim * self.ncb
        if 'quantizers.project_in.weight' in sd:
            self.q_w_in, self.q_b_in = dev(f('quantizers.project_in.weight').contiguous()), dev(f('quantizers.project_in.bias'))
            self.q_w_out, self.q_b_out = dev(f('quantizers.project_out.weight').t().contiguous()), dev(f('quantizers.project_out.bias'))   # (cbt, C): mv2.h w_out_t
        else:   # dim == codebook dims: identity projections
            eye = torch.eye(cbt)
            self.q_w_in, self.q_b_in, self.q_w_out, self.q_b_out = dev(eye), dev(torch.zeros(cbt)), dev(eye.clone()), dev(torch.zeros(cbt))


class _Arena:
    """size-keyed recycling of plan buffers (static addresses; liveness handled by the builder)."""

    def __init__(self, ops):
        self.ops = ops
        self.free: Dict[tuple, list] = {}
        self.total_bytes = 0

    def get(self, numel, dtype):
        key = (int(numel), dtype)
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        self.total_bytes += int(numel) * torch.empty((), dtype=dtype).element_size()
        return self.ops.empty((int(numel),), dtype)

    def put(self, t):
        self.free.setdefault((t.numel(), t.dtype), []).append(t)


class Act:
    """an NDHWC activation: flat tensor + geometry."""
    __slots__ = ('t', 'B', 'T', 'H', 'W', 'C')

    def __init__(self, t, B, T, H, W, C):
        self.t, self.B, self.T, self.H, self.W, self.C = t, B, T, H, W, C

    @property
    def rows(self):
        return self.B * self.T * self.H * self.W


def _se_two_launch(Pn, C, hidden, dtype) -> bool:
    """Eligibility of mv2_se_pool_fc1 / mv2_se_fc2_apply (include/mv2.h): a frame of at most 256 KB (the 16x16
    levels: measured 21.5 us against 27 us for pool + gate + apply at C = 512; at 512 KB frames the redundant
    pooling loses, 40 us against 24 us) and power-of-two widths.  lowering_options(se_two_launch=False) forces the three-launch form."""
    if not OPTIONS.se_two_launch:
        return False
    es = 2 if dtype == torch.bfloat16 else 4
    pow2 = lambda v: v > 0 and (v & (v - 1)) == 0
    return Pn * C * es <= 256 * 1024 and pow2(C) and pow2(hidden) and C >= 128 and hidden >= 16


class Plan:
    def __init__(self, pm: PackedModel, B: int, T: int, H: int, W: int, want_prequant: bool = False, train: bool = False,
                 headless: bool = False):
        """headless = the reference's video_contains_first_frame=False (M:1530-1538, 1604, 1646): no time_padding zero frames in
        front of the clip, nothing cropped after conv_out, the first frame is not encoded separately."""
        self.pm, self.ops, self.cfg = pm, pm.ops, pm.cfg
        self.dtype, self.dt = pm.dtype, pm.dt
        self.B, self.T, self.H, self.W = B, T, H, W
        self.arena = _Arena(self.ops)
        self.tdf = 2 ** sum(s.kind == 'compress_time' for s in pm.stages)
        self.headless = bool(headless)
        self.time_padding = 0 if self.headless else self.tdf - 1
        self.train = train and not pm.use_fsq        # training-mode LFQ: straight-through forward value + entropy / commitment aux loss
        want_prequant = want_prequant or self.train
        self.want_prequant = want_prequant
        self.step_names: List[str] = []
        self.step_flops: Dict[int, float] = {}      # id(step) -> algorithmic FLOPs (conv launches)
        self.step_flops_exec: Dict[int, float] = {}  # id(step) -> FLOPs the launch executes where that differs (mv2_conv3 skips the causal-front taps)
        self._ticket_buf = None
        ch = self.cfg['channels']
        ops = self.ops
        # static I/O
        self.video_in = ops.empty((B, ch, T, H, W), torch.float32)
        self.recon = ops.empty((B, ch, T, H, W), torch.float32)
        self.sep = pm.separate_first_frame and not self.headless
        if self.sep:
            assert T >= 2, 'separate_first_frame_encoding needs at least two frames (the reference fails on a single frame: conv_in gets an empty clip)'
            # edge buffers: the facade splits the clip into its first frame and the rest and joins the two reconstructions
            self.video_first, self.video_rest = ops.empty((B, ch, 1, H, W), torch.float32), ops.empty((B, ch, T - 1, H, W), torch.float32)
            self.recon_first, self.recon_rest = ops.empty((B, ch, 1, H, W), torch.float32), ops.empty((B, ch, T - 1, H, W), torch.float32)
        # conditioning: the caller's (B, dim_cond) vector and the two stem outputs (reference M:1341-1353, 1545-1548, 1613-1616)
        self.cond_in = self.cond_e = self.cond_d = None
        if pm.has_cond:
            self.cond_in = ops.empty((B, pm.dim_cond), torch.float32)
            self.cond_e = ops.empty((B, pm.dim_cond_e), torch.float32)
            self.cond_d = ops.empty((B, pm.dim_cond_e), torch.float32)
        # ---- encode
        self.enc_steps: List = []
        x = self._lower_encode(self.enc_steps)
        self.latent = x
        # ---- quantize
        self.q_steps: List = []
        self.ids = ops.empty((x.rows, pm.ncb), torch.int32 if pm.use_fsq else torch.int64)
        self.prequant = ops.empty((x.rows, pm.cb_dim * pm.ncb), torch.float32) if want_prequant else None
        q = Act(self.arena.get(x.rows * x.C, self.dtype), x.B, x.T, x.H, x.W, x.C)
        # lfq_spherical (third-party LFQ(spherical=True), reference M:1070, 1372): input and quantized value are l2-normalised per
        # codebook.  Normalising keeps every sign, so the ids are those of the plain quantizer and the quantized value is
        # +-1/sqrt(cb_dim) per code dim: the eval path is the same kernel with that scale.  The training-mode statistics read the
        # NORMALISED activations with the same scale: the facade normalises `prequant` (rows x code dims, a few KB) between the forward
        # launches and mv2_lfq_stats.
        self.lfq_scale = pm.cb_dim ** -0.5 if (self.cfg.get('lfq_spherical') and not pm.use_fsq) else 1.
        if pm.use_fsq:
            self.q_steps.append(ops.fsq(self.dt, x.t, pm.q_w_in, pm.q_b_in, pm.q_w_out, pm.q_b_out, pm.fsq, q.t, self.ids,
                                        self.prequant, x.rows, x.C, pm.cb_dim, pm.ncb))
        else:
            fwd = ops.lfq_train if self.train else ops.lfq
            self.q_steps.append(fwd(self.dt, x.t, pm.q_w_in, pm.q_b_in, pm.q_w_out, pm.q_b_out, q.t, self.ids, self.prequant,
                                    x.rows, x.C, pm.cb_dim, pm.ncb, self.cfg['lfq_soft_clamp_input_value'] or 0., self.lfq_scale))
        self.quantized = q
        # ---- losses (forward(return_loss=True), reference M:1705-1722): per-rank sums on device; the facade all-reduces
        # `lfq_sums` across ranks before the finalising launch
        self.n_tokens = x.rows
        self.loss_steps: List = []
        n = self.video_in.numel()
        self.sq_sum = ops.empty((1,), torch.float32)
        self.sq_sum_rest = None
        sq_ws = ops.empty((max(1, ops.sq_diff_workspace_bytes(n) // 4),), torch.float32)
        if self.sep:
            # the reconstruction lives in two buffers (first frame | rest, M:1633-1639): two sums of squared differences
            n1 = self.video_first.numel()
            self.sq_sum_rest = ops.empty((1,), torch.float32)
            self.loss_steps.append(ops.sq_diff_sum(self.video_first, self.recon_first, n1, self.sq_sum, sq_ws))
            self.loss_steps.append(ops.sq_diff_sum(self.video_rest, self.recon_rest, n - n1, self.sq_sum_rest, sq_ws))
        else:
            self.loss_steps.append(ops.sq_diff_sum(self.video_in, self.recon, n, self.sq_sum, sq_ws))
        self.n_recon_steps = len(self.loss_steps)
        self.lfq_sums = self.lfq_losses = None
        if self.train:
            K = 1 << pm.cb_dim
            self.lfq_sums = ops.empty((pm.ncb * K + 2,), torch.float32)
            self.lfq_losses = ops.empty((4,), torch.float32)
            st_ws = ops.empty((max(1, ops.lfq_stats_workspace_bytes(x.rows, pm.cb_dim, pm.ncb) // 4),), torch.float32)
            self.loss_steps.append(ops.lfq_stats(self.prequant, x.rows, pm.cb_dim, pm.ncb, self.cfg.get('lfq_inv_temperature', 100.), self.lfq_scale,
                                                 self.lfq_sums, st_ws))
        # ---- ids -> quantized (decode_from_code_indices)
        self.ids_in = ops.empty((x.rows, pm.ncb), torch.int32 if pm.use_fsq else torch.int64)
        if pm.use_fsq:
            self.codes_steps = [ops.fsq_codes(self.dt, self.ids_in, pm.q_w_out, pm.q_b_out, pm.fsq, q.t, x.rows, x.C, pm.cb_dim, pm.ncb)]
        else:
            self.codes_steps = [ops.lfq_codes(self.dt, self.ids_in, pm.q_w_out, pm.q_b_out, q.t, x.rows, x.C, pm.cb_dim, pm.ncb, self.lfq_scale)]
        # ---- decode
        self.dec_steps: List = []
        self._lower_decode(self.dec_steps, q)

    # ------------------------------------------------------------------ helpers
    def _conv(self, steps, name, pc: P.PackedConv, x: Act, *, To=None, Ho=None, Wo=None, st=1, sh=1, sw=1, pt=0, ph=0, pw=0,
              check_bounds=True, act=P.ACT_NONE, store=P.STORE_ROWS, shuffle=(1, 1, 1), crop_t=0, out_c=None, rowscale=None,
              residual=None, out_t=None, in_bytes=None, in_geom=None, rms_shift=None):
        """rms_shift (None / False / True): the launch is preceded by RMSNorm's row scale (True: of the TokenShift-ed row).  It is computed
        inside the GEMM launch where the kernel can (mv2_conv_rms), by a separate mv2_rms_rowscale launch otherwise."""
        B, Ti, Hi, Wi, Ci = in_geom if in_geom is not None else (x.B, x.T, x.H, x.W, x.C)
        To = Ti if To is None else To
        Ho = Hi if Ho is None else Ho
        Wo = Wi if Wo is None else Wo
        rt, rh, rw = shuffle
        if out_c is None:
            out_c = pc.co // 2 if act == P.ACT_GEGLU else (pc.co // (rt * rh * rw) if store == P.STORE_SHUFFLE else pc.co)
        d = dict(dtype=self.dt, B=B, Ti=Ti, Hi=Hi, Wi=Wi, Ci=Ci, To=To, Ho=Ho, Wo=Wo, co=pc.co, co_pad=pc.co_pad, st=st, sh=sh,
                 sw=sw, pt=pt, ph=ph, pw=pw, kt=pc.kt, kh=pc.kh, kw=pc.kw, nsteps=pc.nsteps, seg_bytes=pc.seg_bytes,
                 check_bounds=int(check_bounds), act=act, store=store, rt=rt, rh=rh, rw=rw, crop_t=crop_t, out_c=out_c, splitk=0)
        rows_out = B * To * Ho * Wo
        if store == P.STORE_NCTHW_F32:
            assert out_t is not None
            out = None
        else:
            out_rows = rows_out * rt * rh * rw
            if out_t is None:
                out_t = self.arena.get(out_rows * out_c, self.dtype)
            out = Act(out_t, B, To * rt, Ho * rh, Wo * rw, out_c)
        segs = pc.segs_tensor(Hi, Wi, Ci, self.ops.device)
        if rms_shift is not None:
            assert rowscale is None
            # mv2_conv_rms sums the squares of what the K program GATHERS; that is the real row only when every non-filler segment is
            # fully real.  pack_taps rounds a tap up to a 16-byte chunk: a token-shifted bf16 layer of width 8 gathers 8 elements per
            # half-row tap, i.e. 4 elements of a neighbouring row under zero weights - the GEMM is right, the fused 1 / |row| is not
            # (width 16 and above are exact).  Such layers take the separate mv2_rms_rowscale launch.
            es = 2 if self.dtype == torch.bfloat16 else 4
            gather_exact = pc.k_real * es == sum(1 for sgt in pc.seg_taps if not sgt[4]) * pc.seg_bytes
            if gather_exact and self.ops.conv_rms_ok(d) and OPTIONS.conv_rms:
                if in_bytes is None:
                    in_bytes = x.t.numel() * x.t.element_size()
                steps.append(self.ops.conv_rms(d, x.t, in_bytes, pc.wp, segs, pc.bias, None if residual is None else residual.t, out_t))
                steps[-1].label = name
                self.step_flops[id(steps[-1])] = 2.0 * rows_out * pc.co_real * pc.k_real
                self.step_names.append(name)
                return out
            rowscale = self._rms(steps, x, bool(rms_shift))
            out_ = self._conv(steps, name, pc, x, To=To, Ho=Ho, Wo=Wo, st=st, sh=sh, sw=sw, pt=pt, ph=ph, pw=pw, check_bounds=check_bounds, act=act,
                              store=store, shuffle=shuffle, crop_t=crop_t, out_c=out_c, rowscale=rowscale, residual=residual, out_t=out_t,
                              in_bytes=in_bytes, in_geom=in_geom)
            self.arena.put(rowscale)
            return out_
        wsb = self.ops.conv_workspace_bytes(d)
        ws = self.arena.get((wsb + 3) // 4, torch.float32) if wsb > 0 else None
        if in_bytes is None:
            in_bytes = x.t.numel() * x.t.element_size()
        steps.append(self.ops.conv(d, x.t, in_bytes, pc.wp, segs, pc.bias, rowscale, None if residual is None else residual.t,
                                   out_t, ws))
        steps[-1].label = name
        self.step_flops[id(steps[-1])] = 2.0 * rows_out * pc.co_real * pc.k_real
        self.step_names.append(name)
        if ws is not None:
            self.arena.put(ws)
        return out

    def _conv3(self, steps, pre, x: Act, k, fuse_k1: bool, se_part_for=None, act=P.ACT_ELU):
        """The ResidualUnit's k x 3 x 3 causal conv + ELU through mv2_conv3; with fuse_k1 also the 1x1x1 conv + ELU (and, with
        se_part_for, the SqueezeExcite pooling partials of that unit) in the same launch.
        Returns (output, (partials, records per frame) or None)."""
        pm, ops = self.pm, self.ops
        pc, pc2 = pm.convs[pre + '.k3'], pm.convs[pre + '.k1']
        d = dict(dtype=self.dt, B=x.B, Ti=x.T, Hi=x.H, Wi=x.W, Ci=x.C, To=x.T, Ho=x.H, Wo=x.W, co=pc.co, co_pad=pc.co_pad, st=1, sh=1,
                 sw=1, pt=k - 1, ph=k // 2, pw=k // 2, kt=pc.kt, kh=pc.kh, kw=pc.kw, nsteps=pc.nsteps, seg_bytes=pc.seg_bytes,
                 check_bounds=1, act=act, store=P.STORE_ROWS, rt=1, rh=1, rw=1, crop_t=0, out_c=pc.co, splitk=0)
        assert ops.conv3_eligible(d), 'fragment-major weights but the descriptor is not mv2_conv3-eligible'
        out = Act(self.arena.get(x.rows * pc.co, self.dtype), x.B, x.T, x.H, x.W, pc.co)
        ws_bytes, cnt_bytes = ops.conv3_sizes(d)
        assert not fuse_k1 or ops.conv3_fusable(d)
        ws = self.arena.get((ws_bytes + 3) // 4, torch.float32) if ws_bytes > 0 else None
        kw, pw_part = {}, None
        if fuse_k1:
            kw = dict(wp2=pc2.wp, bias2=pc2.bias, act2=P.ACT_ELU)
            if se_part_for is not None:
                n, _, _ = ops.conv3_tiles(d)
                part = self.arena.get(x.B * x.T * n * (pc.co + 2), torch.float32)
                kw.update(se_wk=pm.f32[se_part_for + '.se.wk'], se_bk=float(pm.f32[se_part_for + '.se.bk']), se_part=part)
                pw_part = (part, n)
        steps.append(ops.conv3(d, x.t, x.t.numel() * x.t.element_size(), pc.wp, pc.bias, out.t, ws, self._tickets(cnt_bytes), **kw))
        label = pre + ('.k3+k1' if fuse_k1 else '.k3')
        steps[-1].label = label
        fl = 2.0 * x.rows * pc.co_real * pc.k_real
        fl2 = 2.0 * x.rows * pc2.co_real * pc2.k_real if fuse_k1 else 0.0
        self.step_flops[id(steps[-1])] = fl + fl2
        # the kernel skips the time taps that fall into the causal zero padding (frame t has min(t, kt - 1) + 1 valid taps): the
        # reference multiplies those zeros, so they count as algorithmic FLOPs but not as matrix-pipe work
        valid_taps = sum(min(t, pc.kt - 1) + 1 for t in range(x.T))
        self.step_flops_exec[id(steps[-1])] = fl * valid_taps / float(x.T * pc.kt) + fl2
        self.step_names.append(label)
        if ws is not None:
            self.arena.put(ws)
        return out, pw_part

    def _conv3_split(self, steps, pre, x: Act, k, act=P.ACT_ELU) -> Act:
        """The ResidualUnit's k x 3 x 3 causal conv + ELU of an fp32 plan on the bf16 matrix pipe: mv2_split_bf16 turns the fp32 rows into
        [hi | lo | hi] bf16 rows, mv2_conv3 multiplies them with the [w_hi | w_hi | w_lo] image and stores fp32 rows (MV2_STORE_ROWS_F32)."""
        pm, ops = self.pm, self.ops
        pc = pm.convs[pre + '.k3s']
        xs = self.arena.get(x.rows * 3 * x.C, torch.bfloat16)
        steps.append(ops.split_bf16(x.t, xs, x.rows, x.C))
        steps[-1].label = pre + '.split'
        self.step_names.append('split_bf16')
        d = dict(dtype=P.dtype_code(torch.bfloat16), B=x.B, Ti=x.T, Hi=x.H, Wi=x.W, Ci=3 * x.C, To=x.T, Ho=x.H, Wo=x.W, co=pc.co, co_pad=pc.co_pad,
                 st=1, sh=1, sw=1, pt=k - 1, ph=k // 2, pw=k // 2, kt=pc.kt, kh=pc.kh, kw=pc.kw, nsteps=pc.nsteps, seg_bytes=pc.seg_bytes,
                 check_bounds=1, act=act, store=P.STORE_ROWS_F32, rt=1, rh=1, rw=1, crop_t=0, out_c=pc.co, splitk=0)
        assert ops.conv3_eligible(d), 'the split-bf16 descriptor is not mv2_conv3-eligible'
        out = Act(self.arena.get(x.rows * pc.co, torch.float32), x.B, x.T, x.H, x.W, pc.co)
        ws_bytes, cnt_bytes = ops.conv3_sizes(d)
        ws = self.arena.get((ws_bytes + 3) // 4, torch.float32) if ws_bytes > 0 else None
        steps.append(ops.conv3(d, xs, xs.numel() * 2, pc.wp, pc.bias, out.t, ws, self._tickets(cnt_bytes)))
        steps[-1].label = pre + '.k3(split)'
        fl = 2.0 * x.rows * pc.co_real * (pc.k_real // 3)           # algorithmic FLOPs: the convolution's, not the three split products'
        self.step_flops[id(steps[-1])] = fl
        valid_taps = sum(min(t, pc.kt - 1) + 1 for t in range(x.T))
        self.step_flops_exec[id(steps[-1])] = 3.0 * fl * valid_taps / float(x.T * pc.kt)
        self.step_names.append(pre + '.k3(split)')
        for t in (xs, ws):
            if t is not None:
                self.arena.put(t)
        return out

    def _tickets(self, nbytes):
        """One zero-initialised ticket buffer shared by every mv2_conv3 launch of the plan (each launch leaves it zero)."""
        n = (int(nbytes) + 3) // 4
        if n == 0:
            return None
        if self._ticket_buf is None:
            # upper bound of the tile count of any level: 8 x 16 tiles of the input resolution, <= 4 channel tiles
            cap = self.B * (self.T + self.time_padding) * (-(-self.H // 8)) * (-(-self.W // 16)) * 4
            self._ticket_buf = self.ops.zeros((max(cap, n),), torch.int32)
        assert self._ticket_buf.numel() >= n, 'ticket buffer too small for this level'
        return self._ticket_buf

    def _rms(self, steps, x: Act, shift: bool):
        rs = self.arena.get(x.rows, torch.float32)
        steps.append(self.ops.rms_rowscale(self.dt, x.t, rs, x.rows, x.C, x.H * x.W if shift else 0, x.T))
        self.step_names.append('rms_rowscale')
        return rs

    def _residual_unit(self, steps, pre, x: Act) -> Act:
        pm, ops = self.pm, self.ops
        k = self.cfg['residual_conv_kernel_size']
        frames, Pn, C = x.B * x.T, x.H * x.W, x.C
        hidden = se_hidden(C)
        two = _se_two_launch(Pn, C, hidden, self.dtype)
        pc3, pc1 = pm.convs[pre + '.k3'], pm.convs[pre + '.k1']
        pw_part = None
        if (pre + '.k3s') in pm.convs:
            # fp32 storage, split-bf16 products (id-exact mode): the rest of the unit stays on the fp32 kernels
            y1 = self._conv3_split(steps, pre, x, k)
            y2 = self._conv(steps, pre + '.k1', pc1, y1, act=P.ACT_ELU, check_bounds=False)
            self.arena.put(y1.t)
        elif pc3.frag_major:
            # mv2_conv3; bf16 at C = 64 (a workgroup owns every channel of its positions): conv1 + ELU (+ SE pooling partials) ride along
            fuse = self.dtype == torch.bfloat16 and pc3.co == 64 and pc1.co == pc3.co and pc1.kt == 1 and OPTIONS.fuse_pw
            y1, pw_part = self._conv3(steps, pre, x, k, fuse, se_part_for=(pre if fuse and not two and OPTIONS.fuse_pool else None))
            bp = 0
            if not fuse and (pre + '.k1') in pm.frag and OPTIONS.pw_se:
                bp = ops.pw_se_rows(self.dt, x.rows, Pn, C)
            if fuse:
                y2 = y1
            elif bp:
                # conv1 + ELU + SqueezeExcite pooling records in one launch (pwse.hip); the gate + apply launches follow
                y2 = Act(self.arena.get(x.rows * C, self.dtype), x.B, x.T, x.H, x.W, C)
                nchunk = Pn // bp
                part = self.arena.get(frames * nchunk * (C + 2), torch.float32)
                steps.append(ops.pw_se(self.dt, y1.t, pm.frag[pre + '.k1'], pc1.bias, pm.f32[pre + '.se.wk'], float(pm.f32[pre + '.se.bk']),
                                       y2.t, part, x.rows, Pn, C))
                steps[-1].label = pre + '.k1+pool'
                self.step_flops[id(steps[-1])] = 2.0 * x.rows * C * C
                self.step_names.append(pre + '.k1+pool')
                self.arena.put(y1.t)
                pw_part = (part, nchunk)
                if two and (nchunk * (C + 3) + C + 1024) * 4 <= 64 * 1024:
                    # small frames: fc1 from the records over hidden-unit slices, then fc2 + apply over channel slices
                    hid = self.arena.get(frames * hidden, torch.float32)
                    out = Act(self.arena.get(x.rows * C, self.dtype), x.B, x.T, x.H, x.W, C)
                    s1, s2 = hidden // max(16, hidden // 4), C // max(32, C // 8)
                    steps.append(ops.se_rec_fc1(part, pm.f32[pre + '.se.w1'], pm.f32[pre + '.se.b1'], hid, frames, C, nchunk, hidden, s1))
                    steps.append(ops.se_fc2_apply(self.dt, y2.t, hid, pm.f32[pre + '.se.w2'], pm.f32[pre + '.se.b2'], x.t, out.t,
                                                  frames, Pn, C, hidden, s2))
                    self.step_names += ['se_rec_fc1', 'se_fc2_apply']
                    for t in (hid, part, y2.t, x.t):
                        self.arena.put(t)
                    return out
                two = False
            else:
                y2 = self._conv(steps, pre + '.k1', pc1, y1, act=P.ACT_ELU, check_bounds=False)
                self.arena.put(y1.t)
        else:
            y1 = self._conv(steps, pre + '.k3', pc3, x, pt=k - 1, ph=k // 2, pw=k // 2, act=P.ACT_ELU)
            y2 = self._conv(steps, pre + '.k1', pc1, y1, act=P.ACT_ELU, check_bounds=False)
            self.arena.put(y1.t)
        if two:
            # frames small enough for one workgroup to pool: two wide launches instead of three latency chains
            hid = self.arena.get(frames * hidden, torch.float32)
            out = Act(self.arena.get(x.rows * C, self.dtype), x.B, x.T, x.H, x.W, C)
            s1, s2 = hidden // max(16, hidden // 4), C // max(32, C // 8)
            steps.append(ops.se_pool_fc1(self.dt, y2.t, pm.f32[pre + '.se.wk'], float(pm.f32[pre + '.se.bk']), pm.f32[pre + '.se.w1'],
                                         pm.f32[pre + '.se.b1'], hid, frames, Pn, C, hidden, s1))
            steps.append(ops.se_fc2_apply(self.dt, y2.t, hid, pm.f32[pre + '.se.w2'], pm.f32[pre + '.se.b2'], x.t, out.t,
                                          frames, Pn, C, hidden, s2))
            self.step_names += ['se_pool_fc1', 'se_fc2_apply']
            for t in (hid, y2.t, x.t):
                self.arena.put(t)
            return out
        gate = self.arena.get(frames * C, torch.float32)
        if pw_part is not None:
            part, nchunk = pw_part                       # written by the fused conv kernel
        else:
            nchunk = max(1, min(Pn // 16 if Pn >= 16 else 1, -(-1024 // frames)))   # ~1024 blocks, >= 16 rows per chunk
            part = self.arena.get(frames * nchunk * (C + 2), torch.float32)
            steps.append(ops.se_pool(self.dt, y2.t, pm.f32[pre + '.se.wk'], float(pm.f32[pre + '.se.bk']), part, frames, Pn, C, nchunk))
            self.step_names.append('se_pool')
        steps.append(ops.se_gate(part, pm.f32[pre + '.se.w1'], pm.f32[pre + '.se.b1'], pm.f32[pre + '.se.w2'], pm.f32[pre + '.se.b2'],
                                 gate, frames, C, se_hidden(C), nchunk))
        out = Act(self.arena.get(x.rows * C, self.dtype), x.B, x.T, x.H, x.W, C)
        steps.append(ops.se_apply(self.dt, y2.t, gate, x.t, out.t, frames, Pn, C))
        self.step_names += ['se_gate', 'se_apply']
        for t in (part, gate, y2.t, x.t):
            self.arena.put(t)
        return out

    def _modulated(self, steps, pre, x: Act, cond, w_key, b_key, add) -> Act:
        """x * (Linear(cond) + add) per sample and channel (AdaptiveRMSNorm's gamma, Conv3DMod's input modulation)."""
        pm, ops = self.pm, self.ops
        g = self.arena.get(x.B * x.C, torch.float32)
        steps.append(ops.cond_linear(cond, pm.f32[pre + w_key], pm.f32[pre + b_key], g, x.B, pm.dim_cond_e, x.C, 0))
        xs = Act(self.arena.get(x.rows * x.C, self.dtype), x.B, x.T, x.H, x.W, x.C)
        steps.append(ops.scale_channels(self.dt, x.t, g, xs.t, x.rows, x.T * x.H * x.W, x.C, add, P.ACT_NONE))
        self.step_names += ['cond_linear', 'scale_channels']
        return xs, g

    def _residual_unit_mod(self, steps, pre, x: Act, cond) -> Act:
        """ResidualUnitMod (reference M:975-988): x + ELU(conv_out(ELU(demod * conv3(x * (to_cond(cond) + 1)))))."""
        pm, ops = self.pm, self.ops
        k = self.cfg['residual_conv_kernel_size']
        pc3, pc1 = pm.convs[pre + '.k3'], pm.convs[pre + '.k1']
        xs, c = self._modulated(steps, pre, x, cond, '.to_cond.w', '.to_cond.b', 1.)
        if pc3.frag_major:
            y1, _ = self._conv3(steps, pre, xs, k, False, act=P.ACT_NONE)
        else:
            y1 = self._conv(steps, pre + '.k3', pc3, xs, pt=k - 1, ph=k // 2, pw=k // 2)
        self.arena.put(xs.t)
        dm = self.arena.get(x.B * x.C, torch.float32)
        steps.append(ops.cond_linear(c, pm.f32[pre + '.wsq'], None, dm, x.B, x.C, x.C, 2, 1e-8))
        steps.append(ops.scale_channels(self.dt, y1.t, dm, y1.t, x.rows, x.T * x.H * x.W, x.C, 0., P.ACT_ELU))
        self.step_names += ['cond_linear', 'scale_channels']
        out = self._conv(steps, pre + '.k1', pc1, y1, act=P.ACT_ELU, residual=x, check_bounds=False)
        for t in (c, dm, y1.t, x.t):
            self.arena.put(t)
        return out

    def _feed_forward(self, steps, pre, x: Act, shift: bool, cond=None) -> Act:
        pm, ops = self.pm, self.ops
        hp = pm.hidden_pad.get(pre, 0)
        if (cond is None and OPTIONS.fuse_ff and x.rows >= OPTIONS.fuse_ff_min_rows and (pre + '.ff1') in pm.frag
                and ops.ff_eligible(self.dt, x.rows, x.C, hp)):
            # the whole block in one launch (ff.hip): the hidden tensor stays on chip
            pc1, pc2 = pm.convs[pre + '.ff1'], pm.convs[pre + '.ff2']
            out = Act(self.arena.get(x.rows * x.C, self.dtype), x.B, x.T, x.H, x.W, x.C)
            wsb, cntb = ops.ff_sizes(x.rows, x.C, hp)
            ws = self.arena.get((wsb + 3) // 4, torch.float32) if wsb > 0 else None
            steps.append(ops.ff(self.dt, x.t, pm.frag[pre + '.ff1'], pc1.bias, pm.frag[pre + '.ff2'], pc2.bias, out.t, x.rows, x.C, hp,
                                x.H * x.W if shift else 0, x.T, ws, self._tickets(cntb) if wsb > 0 else None))
            steps[-1].label = pre + '.ff'
            self.step_flops[id(steps[-1])] = 2.0 * x.rows * (pc1.co_real * pc1.k_real + pc2.co_real * pc2.k_real)
            self.step_names.append(pre + '.ff')
            if ws is not None:
                self.arena.put(ws)
            self.arena.put(x.t)
            return out
        kw = dict(pt=1, check_bounds=True) if shift else dict(check_bounds=False)
        if cond is None:       # RMSNorm's row scale rides in the GEMM launch (mv2_conv_rms) where the kernel can
            h = self._conv(steps, pre + '.ff1', pm.convs[pre + '.ff1'], x, act=P.ACT_GEGLU, rms_shift=shift, **kw)
        else:                  # the scale is taken on x, the GEMM reads the modulated copy
            rs = self._rms(steps, x, shift)
            xin, g = self._modulated(steps, pre + '.norm', x, cond, '.to_gamma.w', '.to_gamma.b', 0.)
            self.arena.put(g)
            h = self._conv(steps, pre + '.ff1', pm.convs[pre + '.ff1'], xin, act=P.ACT_GEGLU, rowscale=rs, **kw)
            self.arena.put(xin.t)
            self.arena.put(rs)
        out = self._conv(steps, pre + '.ff2', pm.convs[pre + '.ff2'], h, residual=x, check_bounds=False)
        self.arena.put(h.t)
        self.arena.put(x.t)
        return out

    def _attention(self, steps, pre, x: Act, time_axis: bool, cond=None) -> Act:
        pm, ops, cfg = self.pm, self.ops, self.cfg
        heads, dh = cfg['attn_heads'], cfg['attn_dim_head']
        kw = dict(pt=1, check_bounds=True) if time_axis else dict(check_bounds=False)
        if cond is None:
            qkv = self._conv(steps, pre + '.qkv', pm.convs[pre + '.qkv'], x, rms_shift=time_axis, **kw)
        else:
            rs = self._rms(steps, x, time_axis)
            xin, g = self._modulated(steps, pre + '.norm', x, cond, '.to_gamma.w', '.to_gamma.b', 0.)
            self.arena.put(g)
            qkv = self._conv(steps, pre + '.qkv', pm.convs[pre + '.qkv'], xin, rowscale=rs, **kw)
            self.arena.put(xin.t)
            self.arena.put(rs)
        ao = Act(self.arena.get(x.rows * heads * dh, self.dtype), x.B, x.T, x.H, x.W, heads * dh)
        mem = pm.f32[pre + '.mem_kv']
        hw = x.H * x.W
        if time_axis:
            nseq, ntok = x.B * hw, x.T
            steps.append(ops.attention(self.dt, qkv.t, mem, ao.t, nseq, ntok, heads, dh, mem.shape[2], True, hw, x.T * hw, hw))
        else:
            nseq, ntok = x.B * x.T, hw
            steps.append(ops.attention(self.dt, qkv.t, mem, ao.t, nseq, ntok, heads, dh, mem.shape[2], False, 1, hw, 1))
        steps[-1].label = pre + '.attn'
        # algorithmic FLOPs of sim = q k^T and out = attn v over the full (n, n + mem) score matrix (the reference computes the causal
        # half too and masks it, attend.py:225-243)
        self.step_flops[id(steps[-1])] = 4.0 * nseq * heads * ntok * (ntok + mem.shape[2]) * dh
        self.step_names.append('attention')
        self.arena.put(qkv.t)
        out = self._conv(steps, pre + '.out', pm.convs[pre + '.out'], ao, residual=x, check_bounds=False)
        self.arena.put(ao.t)
        self.arena.put(x.t)
        return out

    def _linear_attention_any_width(self, steps, pre, x: Act, heads, dh) -> Act:
        """linear_attn_dim_head 4 / 16 (the reference takes any, M:1079; 8 is its default and has the tuned kernels): to_q and to_kv as two
        GEMMs in the row layouts, the whole-frame state kernel and the apply kernel at that width (mv2_taylor_kv_d / mv2_taylor_apply_d)."""
        pm, ops = self.pm, self.ops
        nfl = ops.taylor_state_floats(dh)
        if nfl == 0:
            raise NotImplementedError(f'linear_attn_dim_head={dh}: the Taylor attention kernels are built for 4, 8 or 16')
        frames, n = x.B * x.T, x.H * x.W
        q = self._conv(steps, pre + '.q', pm.convs[pre + '.q'], x, rms_shift=False, check_bounds=False)
        kv = self._conv(steps, pre + '.kv', pm.convs[pre + '.kv'], x, rms_shift=False, check_bounds=False)
        state = self.arena.get(frames * heads * nfl, torch.float32)
        o = Act(self.arena.get(x.rows * heads * dh, self.dtype), x.B, x.T, x.H, x.W, heads * dh)
        steps.append(ops.taylor_kv_d(self.dt, kv.t, state, frames, n, heads, dh))
        steps.append(ops.taylor_apply_d(self.dt, q.t, state, o.t, frames, n, heads, dh))
        self.step_names += ['taylor_kv_d', 'taylor_apply_d']
        for t in (state, q.t, kv.t):
            self.arena.put(t)
        out = self._conv(steps, pre + '.out', pm.convs[pre + '.out'], o, residual=x, check_bounds=False)
        self.arena.put(o.t)
        self.arena.put(x.t)
        return out

    def _gateloop(self, steps, pre, x: Act) -> Act:
        """'gateloop_time' (reference M:1216-1222; third-party SimpleGateLoopLayer, reconstruction): RMSNorm folded into the Linear
        (gamma * sqrt(dim) in the weights, 1 / |x| as the epilogue rowscale), [q | kv | a] = one pointwise GEMM, then the gated
        scan along time with the residual add (mv2_gateloop_scan)."""
        pm, ops = self.pm, self.ops
        qkva = self._conv(steps, pre + '.qkva', pm.convs[pre + '.qkva'], x, rms_shift=False, check_bounds=False)
        out = Act(self.arena.get(x.rows * x.C, self.dtype), x.B, x.T, x.H, x.W, x.C)
        steps.append(ops.gateloop_scan(self.dt, qkva.t, x.t, out.t, x.B, x.T, x.H * x.W, x.C))
        self.step_names.append('gateloop_scan')
        self.arena.put(qkva.t)
        self.arena.put(x.t)
        return out

    def _linear_attention(self, steps, pre, x: Act) -> Act:
        pm, ops, cfg = self.pm, self.ops, self.cfg
        heads, dh = cfg['linear_attn_heads'], cfg['linear_attn_dim_head']
        if dh != 8:
            return self._linear_attention_any_width(steps, pre, x, heads, dh)
        frames, n = x.B * x.T, x.H * x.W
        # one workgroup per (frame, head) accumulates the whole frame's state (mv2_taylor_kv with nchunk = 1): no partial states in HBM and no
        # reduce launch.  Only where frames x heads leaves most of the chip idle (single images, one or two frames) does the chunked form
        # (nchunk > 1: partial states + reduce) pay: measured 21.5 -> 10.5 us at 3 frames x 2100 tokens, slower from 20 frames on
        # (profiles/r04_findings.md section 13)
        fh, cus = frames * heads, int(getattr(ops, 'cus', 256))
        nchunk = 1 if 2 * fh >= cus else max(1, min(-(-cus // fh), n // 32))
        part = self.arena.get(frames * nchunk * heads * 657, torch.float32)
        kvsum = self.arena.get(frames * heads * 657, torch.float32) if nchunk > 1 else None
        o = Act(self.arena.get(x.rows * heads * dh, self.dtype), x.B, x.T, x.H, x.W, heads * dh)
        if (pre + '.qkv') in pm.convs:
            # to_q and to_kv as ONE GEMM whose depth-to-time store writes [frame][head][token][q | k | v] (round 6): one launch less, and the
            # state kernel streams 48 contiguous bytes per token instead of two 16-byte pieces of a 512-byte row
            qkv = self._conv(steps, pre + '.qkv', pm.convs[pre + '.qkv'], x, rms_shift=False, check_bounds=False, store=P.STORE_SHUFFLE,
                             shuffle=(heads, 1, 1), out_c=24)
            steps.append(ops.taylor_kv_fused(self.dt, qkv.t, part, frames, n, heads, nchunk))
            steps.append(ops.taylor_apply_fused(self.dt, qkv.t, part, o.t, kvsum, frames, n, heads, nchunk))
            self.step_names += ['taylor_kv_fused', 'taylor_apply_fused']
            free = (part, kvsum, qkv.t)
        else:
            q = self._conv(steps, pre + '.q', pm.convs[pre + '.q'], x, rms_shift=False, check_bounds=False)
            kv = self._conv(steps, pre + '.kv', pm.convs[pre + '.kv'], x, rms_shift=False, check_bounds=False)
            steps.append(ops.taylor_kv(self.dt, kv.t, part, frames, n, heads, nchunk))
            steps.append(ops.taylor_apply(self.dt, q.t, part, o.t, kvsum, frames, n, heads, nchunk))
            self.step_names += ['taylor_kv', 'taylor_apply']
            free = (part, kvsum, q.t, kv.t)
        for t in free:
            if t is not None:
                self.arena.put(t)
        out = self._conv(steps, pre + '.out', pm.convs[pre + '.out'], o, residual=x, check_bounds=False)
        self.arena.put(o.t)
        self.arena.put(x.t)
        return out

    # ------------------------------------------------------------------ encode / decode lowering
    def _lower_encode(self, steps) -> Act:
        pm, ops, cfg = self.pm, self.ops, self.cfg
        B, T, H, W = self.B, self.T, self.H, self.W
        kt, kh, kw = cfg['input_conv_kernel_size']
        assert kh == kw and kh % 2 == 1
        if pm.has_cond:
            steps.append(ops.cond_linear(self.cond_in, pm.f32['encoder_cond_in.w'], pm.f32['encoder_cond_in.b'], self.cond_e, B,
                                         pm.dim_cond, pm.dim_cond_e, 1))
            self.step_names.append('cond_in')
        pad_hw, cp = kh // 2, pm.c_in_pad
        Hp, Wp = H + 2 * pad_hw, W + 2 * pad_hw
        tail = 64 * cp

        mode = PAD_MODES[cfg.get('pad_mode', 'constant')]

        def conv_in(name, video, frames, pad_t, To, zero_t=0):
            """pad_t = every frame in front of the clip: `zero_t` zero frames (the tokenizer's time_padding, M:1537) + the
            convolution's own kt - 1, which follow pad_mode (M:925-927) like its spatial halo."""
            Tp = frames + pad_t
            padded = self.arena.get(B * Tp * Hp * Wp * cp + tail, self.dtype)
            own_t = pad_t - zero_t
            if mode and name == 'conv_in' and own_t < frames + zero_t:      # non-constant padding (M:925: constant when the clip is not longer than the time padding)
                es = padded.element_size()
                plain = self.arena.get(B * (frames + zero_t) * H * W * cp, self.dtype)
                steps.append(ops.video_to_ndhwc(self.dt, video, plain, B, cfg['channels'], frames, H, W, zero_t, 0, cp, 0))
                steps.append(ops.pad_ndhwc(plain, padded, B, frames + zero_t, H, W, cp * es, own_t, pad_hw, pad_hw, mode, tail * es))
                self.step_names += ['video_to_ndhwc', 'pad_ndhwc']
                self.arena.put(plain)
            else:
                steps.append(ops.video_to_ndhwc(self.dt, video, padded, B, cfg['channels'], frames, H, W, pad_t, pad_hw, cp, tail))
                self.step_names.append('video_to_ndhwc')
            pc = pm.convs[name]
            d = dict(dtype=self.dt, B=B, Ti=Tp, Hi=Hp, Wi=Wp, Ci=cp, To=To, Ho=H, Wo=W, co=pc.co, co_pad=pc.co_pad, st=1, sh=1, sw=1, pt=0, ph=0,
                     pw=0, kt=pc.kt, kh=pc.kh, kw=pc.kw, nsteps=pc.nsteps, seg_bytes=pc.seg_bytes, check_bounds=0, act=P.ACT_NONE,
                     store=P.STORE_ROWS, rt=1, rh=1, rw=1, crop_t=0, out_c=pc.co, splitk=0)
            if name in pm.frag and ops.conv_in_eligible(d) and OPTIONS.conv_in:
                # the halo-staged form (convin.hip): the input frames of a tile go to LDS once instead of one gather per (it, ih) tap
                y = Act(self.arena.get(B * To * H * W * pc.co, self.dtype), B, To, H, W, pc.co)
                steps.append(ops.conv_in(d, padded, padded.numel() * padded.element_size(), pm.frag[name], pc.bias, y.t))
                steps[-1].label = name
                self.step_flops[id(steps[-1])] = 2.0 * B * To * H * W * pc.co_real * pc.k_real
                self.step_names.append(name)
                self.arena.put(padded)
                return y
            y = self._conv(steps, name, pc, Act(padded, B, Tp, Hp, Wp, cp), To=To, Ho=H, Wo=W, check_bounds=False,
                           in_bytes=padded.numel() * padded.element_size())
            self.arena.put(padded)
            return y

        if self.sep:
            # reference M:1553-1561: first frame through its own 2-D conv, frames 1.. through conv_in (causal zero padding right
            # before frame 1), then [time_padding zero frames | first | rest] packed into one latent
            tp = self.time_padding
            ff = conv_in('conv_in_first_frame', self.video_first, 1, 0, 1)
            rest = conv_in('conv_in', self.video_rest, T - 1, kt - 1, T - 1)
            x = Act(self.arena.get(B * (T + tp) * H * W * ff.C, self.dtype), B, T + tp, H, W, ff.C)
            fb = H * W * ff.C * ff.t.element_size()
            if tp:
                steps.append(ops.copy_frames(None, 0, 0, x.t, T + tp, 0, tp, B, fb))
            steps.append(ops.copy_frames(ff.t, 1, 0, x.t, T + tp, tp, 1, B, fb))
            steps.append(ops.copy_frames(rest.t, T - 1, 0, x.t, T + tp, tp + 1, T - 1, B, fb))
            self.step_names += ['copy_frames'] * (3 if tp else 2)
            self.arena.put(ff.t)
            self.arena.put(rest.t)
        else:
            x = conv_in('conv_in', self.video_in, T, self.time_padding + (kt - 1), T + self.time_padding, zero_t=self.time_padding)
        for st in pm.stages:
            x = self._stage(steps, st, st.enc_prefix, x, True)
        return x

    def _stage(self, steps, st: Stage, pre, x: Act, enc: bool) -> Act:
        pm = self.pm
        if st.kind == 'residual':
            return self._residual_unit(steps, pre, x)
        if st.kind == 'consecutive_residual':
            for j in range(st.n):
                x = self._residual_unit(steps, f'{pre}.{j}', x)
            return x
        if st.kind == 'compress_space':
            if enc:
                out = self._conv(steps, pre, pm.convs[pre], x, Ho=(x.H + 2 - 3) // 2 + 1, Wo=(x.W + 2 - 3) // 2 + 1, sh=2, sw=2, ph=1, pw=1)
            else:
                out = self._conv(steps, pre, pm.convs[pre], x, act=P.ACT_SILU, store=P.STORE_SHUFFLE, shuffle=(1, 2, 2), check_bounds=False)
            self.arena.put(x.t)
            return out
        if st.kind == 'compress_time':
            if enc:
                out = self._conv(steps, pre, pm.convs[pre], x, To=(x.T - 1) // 2 + 1, st=2, pt=2)
            else:
                out = self._conv(steps, pre, pm.convs[pre], x, act=P.ACT_SILU, store=P.STORE_SHUFFLE, shuffle=(2, 1, 1), check_bounds=False)
            self.arena.put(x.t)
            return out
        if st.kind == 'attend_space':
            x = self._attention(steps, pre + '.0.fn', x, False)
            return self._feed_forward(steps, pre + '.1.fn', x, False)
        if st.kind == 'linear_attend_space':
            x = self._linear_attention(steps, pre + '.0.fn', x)
            return self._feed_forward(steps, pre + '.1.fn', x, False)
        if st.kind == 'attend_time':
            x = self._attention(steps, pre + '.0.fn.fn', x, True)
            return self._feed_forward(steps, pre + '.1.fn.fn', x, True)
        if st.kind == 'gateloop_time':
            return self._gateloop(steps, pre + '.fn.fn', x)
        cond = self.cond_e if enc else self.cond_d
        if st.kind == 'cond_residual':
            return self._residual_unit_mod(steps, pre, x, cond)
        if st.kind == 'cond_attend_space':
            x = self._attention(steps, pre + '.0.fn', x, False, cond=cond)
            return self._feed_forward(steps, pre + '.1.fn', x, False, cond=cond)
        if st.kind == 'cond_attend_time':
            x = self._attention(steps, pre + '.0.fn.fn', x, True, cond=cond)
            return self._feed_forward(steps, pre + '.1.fn.fn', x, True, cond=cond)
        raise ValueError(st.kind)

    def _lower_decode(self, steps, q: Act):
        pm, cfg = self.pm, self.cfg
        # The decoder recycles its input buffer like any other activation: every decode is preceded by a
        # step that rewrites `quantized` (the quantizer, the ids->codes kernel, or an explicit copy).
        x = Act(q.t, q.B, q.T, q.H, q.W, q.C)
        if pm.has_cond:
            steps.append(self.ops.cond_linear(self.cond_in, pm.f32['decoder_cond_in.w'], pm.f32['decoder_cond_in.b'], self.cond_d, self.B,
                                              pm.dim_cond, pm.dim_cond_e, 1))
            self.step_names.append('cond_in')
        for st in reversed(pm.stages):
            x = self._stage(steps, st, st.dec_prefix, x, False)
        kt, kh, kw = cfg['output_conv_kernel_size']
        if self.sep:
            # reference M:1633-1639: frame time_padding through the 2-D conv, the frames after it through conv_out
            tp, T = self.time_padding, self.T
            fb = x.H * x.W * x.C * x.t.element_size()
            xff = Act(self.arena.get(x.B * x.H * x.W * x.C, self.dtype), x.B, 1, x.H, x.W, x.C)
            xr = Act(self.arena.get(x.B * (T - 1) * x.H * x.W * x.C, self.dtype), x.B, T - 1, x.H, x.W, x.C)
            steps.append(self.ops.copy_frames(x.t, x.T, tp, xff.t, 1, 0, 1, x.B, fb))
            steps.append(self.ops.copy_frames(x.t, x.T, tp + 1, xr.t, T - 1, 0, T - 1, x.B, fb))
            self.step_names += ['copy_frames'] * 2
            self._conv(steps, 'conv_out_first_frame', pm.convs['conv_out_first_frame'], xff, ph=kh // 2, pw=kw // 2, store=P.STORE_NCTHW_F32,
                       out_t=self.recon_first, out_c=4)
            self._conv_out(steps, xr, kt, kh, kw, 0, self.recon_rest)
            return
        self._conv_out(steps, x, kt, kh, kw, self.time_padding, self.recon)

    def _conv_out(self, steps, x: Act, kt, kh, kw, crop_t, out_t):
        """conv_out (CausalConv3d, M:1127, 1642): zero padding comes free from out-of-range gathers; any other pad_mode is an
        explicit gather into a padded image followed by the 'valid' convolution (M:925-927)."""
        pm, cfg = self.pm, self.cfg
        mode = PAD_MODES[cfg.get('pad_mode', 'constant')]
        if mode and kt - 1 < x.T:
            es = x.t.element_size()
            pt_, ph_, pw_ = kt - 1, kh // 2, kw // 2
            padded = self.arena.get(x.B * (x.T + pt_) * (x.H + 2 * ph_) * (x.W + 2 * pw_) * x.C, self.dtype)
            steps.append(self.ops.pad_ndhwc(x.t, padded, x.B, x.T, x.H, x.W, x.C * es, pt_, ph_, pw_, mode, 0))
            self.step_names.append('pad_ndhwc')
            xp = Act(padded, x.B, x.T + pt_, x.H + 2 * ph_, x.W + 2 * pw_, x.C)
            self._conv(steps, 'conv_out', pm.convs['conv_out'], xp, To=x.T, Ho=x.H, Wo=x.W, check_bounds=False, store=P.STORE_NCTHW_F32,
                       crop_t=crop_t, out_t=out_t, out_c=4)
            self.arena.put(padded)
            return
        pc = pm.convs['conv_out']
        d = dict(dtype=self.dt, B=x.B, Ti=x.T, Hi=x.H, Wi=x.W, Ci=x.C, To=x.T, Ho=x.H, Wo=x.W, co=pc.co, co_pad=pc.co_pad, st=1, sh=1, sw=1,
                 pt=kt - 1, ph=kh // 2, pw=kw // 2, kt=pc.kt, kh=pc.kh, kw=pc.kw, nsteps=pc.nsteps, seg_bytes=pc.seg_bytes, check_bounds=1,
                 act=P.ACT_NONE, store=P.STORE_NCTHW_F32, rt=1, rh=1, rw=1, crop_t=crop_t, out_c=4, splitk=0)
        if 'conv_out' in pm.frag and self.ops.conv_out_eligible(d) and OPTIONS.conv_out:
            # the HBM-stream form (convout.hip): the halo frames of a tile are staged once instead of being gathered once per tap
            steps.append(self.ops.conv_out(d, x.t, x.t.numel() * x.t.element_size(), pm.frag['conv_out'], pc.bias, out_t))
            steps[-1].label = 'conv_out'
            self.step_flops[id(steps[-1])] = 2.0 * x.rows * pc.co_real * pc.k_real
            self.step_names.append('conv_out')
            return
        self._conv(steps, 'conv_out', pc, x, pt=kt - 1, ph=kh // 2, pw=kw // 2, store=P.STORE_NCTHW_F32,
                   crop_t=crop_t, out_t=out_t, out_c=4)

    # ------------------------------------------------------------------ execution
    def run(self, steps, stream=None):
        s = self.ops.current_stream() if stream is None else stream
        for st in steps:
            st(s)
