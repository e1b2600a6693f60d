"""
``Discriminator`` — the reference's image discriminator (magvit2_pytorch.py:590-676, DiscriminatorBlock :549-588) on the
same HIP kernels as the tokenizer: forward values only (no backward, hence no gradient penalty).  Parameter names and shapes
are the reference's (``blocks.{i}.0.{conv_res, net.0, net.2, downsample.1}``, ``blocks.{i}.1.{0,1}.fn.*``,
``to_logits.{0,3}``), so the ``discr.*`` part of a reference checkpoint loads unchanged.

Lowering (all through ``mv2_conv`` unless noted), per block at resolution H x W:
  res  = 1x1 conv, stride 2 (stride 1 in the last block)              M:557, 576
  h    = leaky_relu(conv3x3(x)); h = leaky_relu(conv3x3(h))           M:559-564 (MV2_ACT_LRELU epilogue)
  x    = (conv1x1(space_to_depth(h)) + res) / sqrt 2                  M:568-571, 580-587: the rearrange + 1x1 conv is ONE
         2x2 stride-2 convolution (weight[o][c*4 + p1*2 + p2] -> tap (p1, p2) of channel c); 1/sqrt 2 is folded into the
         weights and biases of both summands, and through the positively homogeneous leaky_relu in the last block
  x    = x + LinearSpaceAttention(x); x = x + FeedForward(x)          M:632-645 (Taylor attention kernels, GEGLU epilogue)
to_logits: conv3x3 + leaky_relu, then Linear(C*F*F -> 1) as an F x F 'valid' convolution storing fp32 (M:654-666); F = image_size //
2 ** num_layers is 4 for power-of-two sizes and 4..7 otherwise (e.g. 96 -> 6).
``antialiased_downsample=True`` needs kornia's filter3d (absent here and in the reference's default): raises.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
from torch import nn

from . import pack as P
from . import spec as S
from .plan import Act, Plan, _Arena


def discriminator_dims(dim, image_size, channels=3, max_dim=512):
    num_layers = int(math.log2(image_size) - 2)                       # reference :607
    dims = [channels] + [min((dim * 4) * (2 ** i), max_dim) for i in range(num_layers + 1)]
    return list(zip(dims[:-1], dims[1:]))


def discriminator_last_fmap(image_size):
    """side of the feature map to_logits flattens: image_size // 2 ** num_layers (reference :654-657); 4 for power-of-two sizes, 4..7 otherwise"""
    return image_size // (2 ** int(math.log2(image_size) - 2))


def discriminator_param_specs(dim, image_size, channels=3, max_dim=512, linear_attn_heads=16, linear_attn_dim_head=8) -> List[S.ParamSpec]:
    sp: List[S.ParamSpec] = []
    pairs = discriminator_dims(dim, image_size, channels, max_dim)
    for i, (cin, cout) in enumerate(pairs):
        b = f'blocks.{i}.0'
        sp += S._conv(b + '.conv_res', (cout, cin, 1, 1))
        sp += S._conv(b + '.net.0', (cout, cin, 3, 3))
        sp += S._conv(b + '.net.2', (cout, cout, 3, 3))
        if i != len(pairs) - 1:
            sp += S._conv(b + '.downsample.1', (cout, cout * 4, 1, 1))
        a = f'blocks.{i}.1'
        sp += S._linear_attention(a + '.0.fn', cout, linear_attn_heads, linear_attn_dim_head)
        inner = S.ff_inner(cout)
        sp += [S.ParamSpec(a + '.1.fn.norm.gamma', (cout, 1, 1), ('ones',))]           # images=True: Conv2d / (dim, 1, 1), reference :481-483
        sp += S._conv(a + '.1.fn.net.0', (inner * 2, cout, 1, 1))
        sp += S._conv(a + '.1.fn.net.2', (cout, inner, 1, 1))
    last = pairs[-1][1]
    sp += S._conv('to_logits.0', (last, last, 3, 3))
    sp += S._conv('to_logits.3', (1, last * discriminator_last_fmap(image_size) ** 2))
    return sp


class _DiscrPacked:
    """packed weights of one dtype; duck-types PackedModel for the Plan lowering helpers."""

    def __init__(self, hp: dict, sd: Dict[str, torch.Tensor], dtype, ops):
        self.hp, self.ops, self.dtype, self.dt = hp, ops, dtype, P.dtype_code(dtype)
        self.cfg = dict(linear_attn_heads=hp['linear_attn_heads'], linear_attn_dim_head=hp['linear_attn_dim_head'],
                        attn_heads=8, attn_dim_head=32, channels=hp['channels'])
        self.convs: Dict[str, P.PackedConv] = {}
        self.f32: Dict[str, torch.Tensor] = {}
        self.hidden_pad: Dict[str, int] = {}
        self.frag: Dict[str, torch.Tensor] = {}
        self.has_cond, self.c_in_pad = False, 4
        f = lambda k: sd[k].detach().float().cpu()
        s = 2 ** -0.5

        def put(name, pc):
            self.convs[name] = pc.to(ops.device)

        self.pairs = discriminator_dims(hp['dim'], hp['image_size'], hp['channels'], hp['max_dim'])
        for i, (cin, cout) in enumerate(self.pairs):
            b, last = f'blocks.{i}.0', i == len(self.pairs) - 1
            wres, bres = f(b + '.conv_res.weight')[:, :, None] * s, f(b + '.conv_res.bias') * s
            w0, b0 = f(b + '.net.0.weight')[:, :, None], f(b + '.net.0.bias')
            if i == 0:      # 3-4 channel pixels of the padded input image: (iw, c) extents are contiguous, as for conv_in
                put(b + '.res', P.pack_conv_in(wres, bres, dtype, self.c_in_pad))
                put(b + '.n0', P.pack_conv_in(w0, b0, dtype, self.c_in_pad))
            else:
                put(b + '.res', P.pack_conv3d(wres, bres, dtype))
                put(b + '.n0', P.pack_conv3d(w0, b0, dtype))
            w2, b2 = f(b + '.net.2.weight')[:, :, None], f(b + '.net.2.bias')
            if last:        # leaky_relu(z) / sqrt 2 = leaky_relu(z / sqrt 2)
                w2, b2 = w2 * s, b2 * s
            put(b + '.n2', P.pack_conv3d(w2, b2, dtype))
            if not last:    # 'b c (h p1) (w p2) -> b (c p1 p2) h w' + 1x1 conv == 2x2 stride-2 conv
                wd = f(b + '.downsample.1.weight').reshape(cout, cout, 2, 2)[:, :, None] * s
                put(b + '.down', P.pack_conv3d(wd, f(b + '.downsample.1.bias') * s, dtype))
            a = f'blocks.{i}.1'
            pre = a + '.0.fn'
            g = f(pre + '.norm.gamma') * (cout ** 0.5)
            put(pre + '.q', P.pack_pointwise(f(pre + '.attn.to_q.0.weight'), None, dtype, in_scale=g))
            put(pre + '.kv', P.pack_pointwise(f(pre + '.attn.to_kv.0.weight'), None, dtype, in_scale=g))
            put(pre + '.out', P.pack_pointwise(f(pre + '.attn.to_out.0.weight'), None, dtype))
            pre = a + '.1.fn'
            g = f(pre + '.norm.gamma').reshape(cout) * (cout ** 0.5)
            pc, hpad = P.pack_geglu_in(f(pre + '.net.0.weight').flatten(1), f(pre + '.net.0.bias'), dtype, in_scale=g)
            put(pre + '.ff1', pc)
            self.hidden_pad[pre] = hpad
            put(pre + '.ff2', P.pack_pointwise(f(pre + '.net.2.weight').flatten(1), f(pre + '.net.2.bias'), dtype, k_pad=hpad))
        cl = self.pairs[-1][1]
        put('to_logits.0', P.pack_conv3d(f('to_logits.0.weight')[:, :, None], f('to_logits.0.bias'), dtype))
        fm = discriminator_last_fmap(hp['image_size'])         # Rearrange('b ... -> b (...)') flattens (c, h, w): Linear == fm x fm 'valid' conv
        put('to_logits.3', P.pack_conv3d(f('to_logits.3.weight').reshape(1, cl, fm, fm)[:, :, None], f('to_logits.3.bias'), dtype))


class _DiscrPlan(Plan):
    """launch list of one discriminator forward for a fixed batch; reuses Plan's conv / attention / feed-forward lowering."""

    def __init__(self, pm: _DiscrPacked, B: int):            # noqa: deliberately not calling Plan.__init__ (different graph)
        self.pm, self.ops, self.cfg = pm, pm.ops, pm.cfg
        self.dtype, self.dt = pm.dtype, pm.dt
        H = W = pm.hp['image_size']
        self.B, self.T, self.H, self.W = B, 1, H, W
        self.arena = _Arena(self.ops)
        self.time_padding = 0
        self.step_names: List[str] = []
        self.step_flops: Dict[int, float] = {}
        self._ticket_buf = None
        ops, ch, cp = self.ops, pm.hp['channels'], pm.c_in_pad
        self.images_in = ops.empty((B, ch, 1, H, W), torch.float32)
        self.logits = ops.empty((B, 1, 1, 1, 1), torch.float32)        # NCTHW store of a one-channel 1 x 1 x 1 'video'
        steps = self.steps = []
        Hp, Wp, tail = H + 2, W + 2, 64 * cp
        padded = self.arena.get(B * Hp * Wp * cp + tail, self.dtype)
        steps.append(ops.video_to_ndhwc(self.dt, self.images_in, padded, B, ch, 1, H, W, 0, 1, cp, tail))
        self.step_names.append('video_to_ndhwc')
        x = Act(padded, B, 1, Hp, Wp, cp)
        in_bytes = padded.numel() * padded.element_size()
        n = len(pm.pairs)
        for i, (cin, cout) in enumerate(pm.pairs):
            b, last = f'blocks.{i}.0', i == n - 1
            Ho, Wo = (H, W) if last else (H // 2, W // 2)
            st = 1 if last else 2
            if i == 0:      # x is the zero-padded image: tap origins are shifted by the one-pixel halo
                res = self._conv(steps, b + '.res', pm.convs[b + '.res'], x, To=1, Ho=Ho, Wo=Wo, sh=st, sw=st, ph=-1, pw=-1,
                                 check_bounds=False, in_bytes=in_bytes)
                h = self._conv(steps, b + '.n0', pm.convs[b + '.n0'], x, To=1, Ho=H, Wo=W, check_bounds=False, act=P.ACT_LRELU,
                               in_bytes=in_bytes)
            else:
                res = self._conv(steps, b + '.res', pm.convs[b + '.res'], x, Ho=Ho, Wo=Wo, sh=st, sw=st, check_bounds=False)
                h = self._conv(steps, b + '.n0', pm.convs[b + '.n0'], x, ph=1, pw=1, act=P.ACT_LRELU)
            self.arena.put(x.t)
            if last:
                x = self._conv(steps, b + '.n2', pm.convs[b + '.n2'], h, ph=1, pw=1, act=P.ACT_LRELU, residual=res)
            else:
                h2 = self._conv(steps, b + '.n2', pm.convs[b + '.n2'], h, ph=1, pw=1, act=P.ACT_LRELU)
                x = self._conv(steps, b + '.down', pm.convs[b + '.down'], h2, Ho=Ho, Wo=Wo, sh=2, sw=2, check_bounds=False, residual=res)
                self.arena.put(h2.t)
            self.arena.put(h.t)
            self.arena.put(res.t)
            a = f'blocks.{i}.1'
            x = self._linear_attention(steps, a + '.0.fn', x)
            x = self._feed_forward(steps, a + '.1.fn', x, False)
            H, W = Ho, Wo
        h = self._conv(steps, 'to_logits.0', pm.convs['to_logits.0'], x, ph=1, pw=1, act=P.ACT_LRELU)
        self.arena.put(x.t)
        fm = discriminator_last_fmap(pm.hp['image_size'])
        assert (h.H, h.W) == (fm, fm), 'to_logits flattens the feature map the last block leaves'
        self._conv(steps, 'to_logits.3', pm.convs['to_logits.3'], h, Ho=1, Wo=1, check_bounds=False, store=P.STORE_NCTHW_F32,
                   out_t=self.logits, out_c=4)
        self.arena.put(h.t)


class Discriminator(nn.Module):
    def __init__(self, *, dim, image_size, channels=3, max_dim=512, attn_heads=8, attn_dim_head=32, linear_attn_dim_head=8,
                 linear_attn_heads=16, ff_mult=4, antialiased_downsample=False, seed=None):
        super().__init__()
        if antialiased_downsample:
            raise NotImplementedError('antialiased_downsample needs kornia.filters.filter3d (absent); the reference default is False')
        assert ff_mult == 4 and channels <= 4
        image_size = image_size if isinstance(image_size, int) else min(image_size)
        assert image_size >= 8, 'the reference builds int(log2(image_size) - 2) >= 1 blocks'
        nl = int(math.log2(image_size) - 2)
        # every block but the last halves the map through 'b c (h p1) (w p2) -> b (c p1 p2) h w' (reference :568): sizes that are not a
        # multiple of 2 ** num_layers construct in the reference too and fail in its forward; the same is deferred to forward() here
        self._geometry_error = None if image_size % (2 ** nl) == 0 else (
            f'image_size {image_size} is not a multiple of 2 ** {nl}: the reference discriminator cannot halve an odd feature map either')
        self.hp = dict(dim=dim, image_size=image_size, channels=channels, max_dim=max_dim, linear_attn_heads=linear_attn_heads,
                       linear_attn_dim_head=linear_attn_dim_head)
        gen = torch.Generator().manual_seed(seed if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        self._param_names = []
        from .tokenizer import _register
        for ps in discriminator_param_specs(**self.hp):
            _register(self, ps.name, S.init_tensor(ps, gen), ps.buffer)
            self._param_names.append(ps.name)
        self.compute_dtype = None
        self._ops = None
        self._packed: Dict[tuple, _DiscrPacked] = {}
        self._plans: Dict[tuple, _DiscrPlan] = {}
        self._weights_version = 0

    # ------------------------------------------------------------------ backend (same conventions as VideoTokenizer)
    @property
    def device(self):
        return next(self.parameters()).device

    def _set_backend(self, ops):
        self._ops = ops
        self._packed.clear()
        self._plans.clear()

    def _backend(self):
        dev = self.device
        if self._ops is not None and getattr(self._ops, 'name', '') == 'hip' and self._ops.device != dev:
            self._set_backend(None)
        if self._ops is None:
            from .ops_hip import HipOps           # raises if the library / a HIP device is missing: no CPU fallback
            if dev.type != 'cuda':
                raise RuntimeError('Discriminator parameters are on %s: move the module to a HIP device; no CPU fallback' % dev)
            self._ops = HipOps(dev)
        return self._ops

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._packed.clear()
        self._plans.clear()
        self._weights_version += 1
        return out

    def refresh_weights(self):
        self._packed.clear()
        self._plans.clear()
        self._weights_version += 1

    def _dtype(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        w = self.get_parameter('to_logits.0.weight')
        return torch.bfloat16 if (torch.is_autocast_enabled() or w.dtype == torch.bfloat16) else torch.float32

    @torch.no_grad()
    def forward(self, x):
        """images (B, C, H, W) -> logits (B,) fp32 (reference Discriminator.forward M:670-676)."""
        if self._geometry_error:
            raise ValueError(self._geometry_error)
        assert x.ndim == 4 and x.shape[1] == self.hp['channels'] and x.shape[-2:] == (self.hp['image_size'],) * 2
        ops, dt = self._backend(), self._dtype()
        kpm = (dt, self._weights_version)
        if kpm not in self._packed:
            sd = {n: self.get_parameter(n) for n in self._param_names}
            self._packed = {kpm: _DiscrPacked(self.hp, sd, dt, ops)}
            self._plans.clear()
        kp = (dt, self._weights_version, x.shape[0])
        if kp not in self._plans:
            if len(self._plans) >= 4:
                self._plans.pop(next(iter(self._plans)))
            self._plans[kp] = _DiscrPlan(self._packed[kpm], x.shape[0])
        plan = self._plans[kp]
        plan.images_in.copy_(x.unsqueeze(2))
        if getattr(ops, 'name', '') == 'hip':
            with torch.cuda.device(ops.device):
                plan.run(plan.steps)
        else:
            plan.run(plan.steps)
        return plan.logits.reshape(-1).clone()
