"""
TEST INFRASTRUCTURE — not product code. Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package.

CPU restatements (plain torch, fp32) of the three third-party modules the
reference hot path calls but does not vendor:

  * ``LFQ`` / ``FSQ``  from ``vector-quantize-pytorch >= 1.14.39``
        (reference: setup.py:34; import magvit2_pytorch.py:21; ctor
        magvit2_pytorch.py:1364-1382; calls :1574, :1593, :1700, :1705)
  * ``TaylorSeriesLinearAttn`` from ``taylor-series-linear-attention >= 0.1.5``
        (reference: setup.py:35; import magvit2_pytorch.py:34; ctor :415-419; call :430)

Neither package is present under /root/reference nor installable here (no
network), and the reference pins only a lower bound, no lockfile.  These
classes restate the packages' *published* eval-mode algorithm:

  LFQ  : project_in Linear(dim -> log2 K * n_codebooks, bias) ->
         soft clamp  x = tanh(x / v) * v  (v = soft_clamp_input_value) ->
         fp32: q = where(x > 0, +scale, -scale)   (strict >) ->
         ids = sum_i [q_i > 0] * 2^(d-1-i)  (int32 products, torch.sum -> int64) ->
         eval: x = q -> project_out Linear(bias).
         indices_to_codes: bits ((id & mask) != 0) -> bits*2*scale - scale -> project_out.
  FSQ  : project_in -> bound(z) = tanh(z + shift) * half_l - offset with
         half_l = (L-1)(1+1e-3)/2, offset = 0.5 for even L, shift = atanh(offset/half_l)
         -> round (half to even) -> / (L // 2); ids = sum((zhat*hw + hw) * basis) -> int32,
         basis = cumprod([1] + L[:-1]).
  Taylor: q,k,v projections (bias-free; q scaled by d^-1/2), feature map
         phi(x) = [1, x, vec(x (x) x) / sqrt 2], kv = sum_n phi(k)^T v,
         out = phi(q) kv / clamp(phi(q) . sum_m phi(k), min=1e-5), to_out bias-free.

PARITY UNPINNED for these three modules: the reference has no tests / golden
vectors and the upstream packages cannot be executed here.  Everything else on
the hot path is pinned by running the reference's own code (oracle/ref_loader.py).
State-dict key names follow upstream so that reference checkpoints load.
"""
from __future__ import annotations

from math import log2, ceil
from collections import namedtuple

import torch
from torch import nn
import torch.nn.functional as F

Return = namedtuple('Return', ['quantized', 'indices', 'entropy_aux_loss'])
LossBreakdown = namedtuple('LossBreakdown', ['per_sample_entropy', 'batch_entropy', 'commitment'])


def _channel_last(x):
    # 'b d ... -> b (...) d'
    b, d = x.shape[:2]
    spatial = x.shape[2:]
    return x.reshape(b, d, -1).transpose(1, 2), spatial


def _channel_first(x, spatial):
    # 'b (...) d -> b d ...'
    b, n, d = x.shape
    return x.transpose(1, 2).reshape(b, d, *spatial)


class LFQ(nn.Module):
    def __init__(
        self,
        *,
        dim=None,
        codebook_size=None,
        entropy_loss_weight=0.1,
        commitment_loss_weight=0.25,
        diversity_gamma=1.,
        straight_through_activation=nn.Identity(),
        num_codebooks=1,
        keep_num_codebooks_dim=None,
        codebook_scale=1.,
        frac_per_sample_entropy=1.,
        has_projections=None,
        projection_has_bias=True,
        soft_clamp_input_value=None,
        cosine_sim_project_in=False,
        cosine_sim_project_in_scale=None,
        channel_first=None,
        experimental_softplus_entropy_loss=False,
        entropy_loss_offset=5.,
        spherical=False,
        force_quantization_f32=True,
    ):
        super().__init__()
        assert dim is not None or codebook_size is not None
        assert codebook_size is None or log2(codebook_size).is_integer(), \
            'your codebook size must be a power of 2 for lookup free quantization'
        codebook_size = codebook_size if codebook_size is not None else 2 ** dim
        self.codebook_size = codebook_size
        codebook_dim = int(log2(codebook_size))
        codebook_dims = codebook_dim * num_codebooks
        dim = dim if dim is not None else codebook_dims
        has_projections = has_projections if has_projections is not None else dim != codebook_dims
        assert not cosine_sim_project_in, 'not restated (unused by the scoped configs)'
        # spherical (RECONSTRUCTION from memory of upstream, parity unpinned like the rest of this class): inputs, quantized values, code
        # book and looked-up codes are l2-normalised per codebook and multiplied by codebook_scale: `maybe_l2norm`
        self.spherical = spherical
        self.maybe_l2norm = (lambda t: F.normalize(t, dim=-1) * self.codebook_scale) if spherical else (lambda t: t)

        self.project_in = nn.Linear(dim, codebook_dims, bias=projection_has_bias) if has_projections else nn.Identity()
        self.project_out = nn.Linear(codebook_dims, dim, bias=projection_has_bias) if has_projections else nn.Identity()
        self.has_projections = has_projections

        self.dim = dim
        self.codebook_dim = codebook_dim
        self.num_codebooks = num_codebooks
        keep_num_codebooks_dim = keep_num_codebooks_dim if keep_num_codebooks_dim is not None else num_codebooks > 1
        assert not (num_codebooks > 1 and not keep_num_codebooks_dim)
        self.keep_num_codebooks_dim = keep_num_codebooks_dim
        self.channel_first = channel_first
        self.activation = straight_through_activation
        self.diversity_gamma = diversity_gamma
        self.entropy_loss_weight = entropy_loss_weight
        self.codebook_scale = codebook_scale
        self.commitment_loss_weight = commitment_loss_weight
        self.soft_clamp_input_value = soft_clamp_input_value
        assert soft_clamp_input_value is None or soft_clamp_input_value >= codebook_scale
        self.force_quantization_f32 = force_quantization_f32

        self.register_buffer('mask', 2 ** torch.arange(codebook_dim - 1, -1, -1))
        self.register_buffer('zero', torch.tensor(0.), persistent=False)

    @property
    def dtype(self):
        return self.project_out.weight.dtype if self.has_projections else torch.float32

    def indices_to_codes(self, indices, project_out=True):
        is_img_or_video = indices.ndim >= (3 + int(self.keep_num_codebooks_dim))
        should_transpose = self.channel_first if self.channel_first is not None else is_img_or_video
        if not self.keep_num_codebooks_dim:
            indices = indices.unsqueeze(-1)
        bits = ((indices[..., None].int() & self.mask) != 0).to(self.dtype)
        codes = bits * self.codebook_scale * 2 - self.codebook_scale
        codes = self.maybe_l2norm(codes)
        codes = codes.flatten(-2)                      # '... c d -> ... (c d)'
        if project_out:
            codes = self.project_out(codes)
        if should_transpose:
            codes = codes.movedim(-1, 1)               # 'b ... d -> b d ...'
        return codes

    def forward(self, x, inv_temperature=100., return_loss_breakdown=False, mask=None):
        """eval path as before; training path (RECONSTRUCTION of the upstream entropy / commitment losses, parity unpinned):
        straight-through value x + (q - x), prob = softmax(2 * inv_temperature * x . codebook) over all 2^d codes,
        per-sample entropy (log clamped at 1e-5) averaged over tokens and codebooks, entropy of the token-averaged
        distribution (all-reduced mean under torch.distributed), aux = (H_sample - gamma * H_batch) * w_entropy +
        mse(x, q) * w_commit."""
        assert mask is None
        is_img_or_video = x.ndim >= 4
        should_transpose = self.channel_first if self.channel_first is not None else is_img_or_video
        if should_transpose:
            x, spatial = _channel_last(x)
        assert x.shape[-1] == self.dim
        x = self.project_in(x)
        if self.soft_clamp_input_value is not None:
            v = self.soft_clamp_input_value
            x = (x / v).tanh() * v
        b, n, _ = x.shape
        x = x.reshape(b, n, self.num_codebooks, self.codebook_dim)
        x = self.maybe_l2norm(x)
        orig_dtype = x.dtype
        x = x.float()
        codebook_value = torch.ones_like(x) * self.codebook_scale
        quantized = torch.where(x > 0, codebook_value, -codebook_value)
        indices = ((quantized > 0).int() * self.mask.int()).sum(dim=-1)   # int64, 'b n c'
        quantized = self.maybe_l2norm(quantized)
        per_sample_entropy = batch_entropy = commit_loss = entropy_aux_loss = self.zero
        if self.training:
            original_input = x
            x = x + (quantized - x).detach()                                  # straight-through estimator
            all_codes = torch.arange(self.codebook_size, device=x.device)
            bits = ((all_codes[..., None].int() & self.mask.int()) != 0).float()
            codebook = self.maybe_l2norm(bits * self.codebook_scale * 2 - self.codebook_scale)   # (K, d): every sign pattern, MSB first
            distance = -2 * torch.einsum('...id,jd->...ij', original_input, codebook)
            prob = (-distance * inv_temperature).softmax(dim=-1)              # (b, n, c, K)
            prob = prob.reshape(b * n, self.num_codebooks, self.codebook_size)
            ent = lambda pr: (-pr * pr.clamp(min=1e-5).log()).sum(dim=-1)
            per_sample_entropy = ent(prob).mean()
            avg_prob = prob.mean(dim=0)                                       # (c, K): distribution over the batch's tokens
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(avg_prob)
                avg_prob = avg_prob / dist.get_world_size()
            batch_entropy = ent(avg_prob).mean()
            entropy_aux_loss = per_sample_entropy - self.diversity_gamma * batch_entropy
            if self.commitment_loss_weight > 0:
                commit_loss = F.mse_loss(original_input, quantized.detach())
        else:
            x = quantized
        x = x.type(orig_dtype)
        x = x.reshape(b, n, -1)
        x = self.project_out(x)
        if should_transpose:
            x = _channel_first(x, spatial)
            indices = indices.reshape(b, *spatial, self.num_codebooks)
        if not self.keep_num_codebooks_dim:
            indices = indices.squeeze(-1)
        aux_loss = entropy_aux_loss * self.entropy_loss_weight + commit_loss * self.commitment_loss_weight
        ret = Return(x, indices, aux_loss)
        if not return_loss_breakdown:
            return ret
        return ret, LossBreakdown(per_sample_entropy, batch_entropy, commit_loss)


class FSQ(nn.Module):
    def __init__(
        self,
        levels,
        dim=None,
        num_codebooks=1,
        keep_num_codebooks_dim=None,
        scale=None,
        allowed_dtypes=(torch.float32, torch.float64),
        channel_first=False,
        projection_has_bias=True,
        return_indices=True,
        force_quantization_f32=True,
    ):
        super().__init__()
        levels = list(levels)
        self.register_buffer('_levels', torch.tensor(levels, dtype=torch.int32), persistent=False)
        self.register_buffer('_basis', torch.cumprod(torch.tensor([1] + levels[:-1]), dim=0, dtype=torch.int32),
                             persistent=False)
        self.scale = scale
        codebook_dim = len(levels)
        self.codebook_dim = codebook_dim
        effective_codebook_dim = codebook_dim * num_codebooks
        self.num_codebooks = num_codebooks
        self.effective_codebook_dim = effective_codebook_dim
        keep_num_codebooks_dim = keep_num_codebooks_dim if keep_num_codebooks_dim is not None else num_codebooks > 1
        assert not (num_codebooks > 1 and not keep_num_codebooks_dim)
        self.keep_num_codebooks_dim = keep_num_codebooks_dim
        self.dim = dim if dim is not None else len(levels) * num_codebooks
        self.channel_first = channel_first
        has_projections = self.dim != effective_codebook_dim
        self.project_in = nn.Linear(self.dim, effective_codebook_dim, bias=projection_has_bias) if has_projections else nn.Identity()
        self.project_out = nn.Linear(effective_codebook_dim, self.dim, bias=projection_has_bias) if has_projections else nn.Identity()
        self.has_projections = has_projections
        self.return_indices = return_indices
        self.codebook_size = int(self._levels.prod().item())
        self.allowed_dtypes = allowed_dtypes
        self.force_quantization_f32 = force_quantization_f32

    def bound(self, z, eps=1e-3):
        half_l = (self._levels - 1) * (1 + eps) / 2
        offset = torch.where(self._levels % 2 == 0, 0.5, 0.0)
        shift = (offset / half_l).atanh()
        return (z + shift).tanh() * half_l - offset

    def quantize(self, z):
        quantized = self.bound(z).round()
        half_width = self._levels // 2
        return quantized / half_width

    def _scale_and_shift(self, zhat_normalized):
        half_width = self._levels // 2
        return (zhat_normalized * half_width) + half_width

    def _scale_and_shift_inverse(self, zhat):
        half_width = self._levels // 2
        return (zhat - half_width) / half_width

    def codes_to_indices(self, zhat):
        assert zhat.shape[-1] == self.codebook_dim
        zhat = self._scale_and_shift(zhat)
        return (zhat * self._basis).sum(dim=-1).to(torch.int32)

    def indices_to_level_indices(self, indices):
        indices = indices.unsqueeze(-1)
        return (indices // self._basis) % self._levels

    def _indices_to_codes(self, indices):
        return self._scale_and_shift_inverse(self.indices_to_level_indices(indices))

    def indices_to_codes(self, indices):
        is_img_or_video = indices.ndim >= (3 + int(self.keep_num_codebooks_dim))
        codes = self._indices_to_codes(indices)
        if self.keep_num_codebooks_dim:
            codes = codes.flatten(-2)
        codes = self.project_out(codes)
        if is_img_or_video or self.channel_first:
            codes = codes.movedim(-1, 1)
        return codes

    def forward(self, z):
        is_img_or_video = z.ndim >= 4
        need_move = is_img_or_video or self.channel_first
        if need_move:
            z, spatial = _channel_last(z)
        assert z.shape[-1] == self.dim
        z = self.project_in(z)
        b, n, _ = z.shape
        z = z.reshape(b, n, self.num_codebooks, self.codebook_dim)
        orig_dtype = z.dtype
        if self.force_quantization_f32 and orig_dtype not in self.allowed_dtypes:
            z = z.float()
        codes = self.quantize(z)
        indices = self.codes_to_indices(codes) if self.return_indices else None
        codes = codes.reshape(b, n, -1).type(orig_dtype)
        out = self.project_out(codes)
        if need_move:
            out = _channel_first(out, spatial)
            if indices is not None:
                indices = indices.reshape(b, *spatial, self.num_codebooks)
        if not self.keep_num_codebooks_dim and indices is not None:
            indices = indices.squeeze(-1)
        return out, indices


def second_taylor_expansion(x):
    """phi(x) = [1, x, vec(x (x) x) * sqrt(0.5)]  — last dim d -> 1 + d + d*d."""
    x0 = x.new_ones(x.shape[:-1] + (1,))
    x2 = (x.unsqueeze(-1) * x.unsqueeze(-2)).flatten(-2) * (0.5 ** 0.5)
    return torch.cat((x0, x, x2), dim=-1)


class _Seq(nn.Sequential):
    """nn.Sequential of one Linear so the weight key is '<name>.0.weight' like upstream
    (upstream: nn.Sequential(nn.Linear(...), Rearrange(...)))."""


class TaylorSeriesLinearAttn(nn.Module):
    def __init__(self, dim, *, dim_head=16, heads=8, causal=False, one_headed_kv=False, rotary_emb=False,
                 combine_heads=True, gate_value_heads=False, prenorm=False, shift_tokens=False, dropout=0.):
        super().__init__()
        assert not (causal or one_headed_kv or rotary_emb or gate_value_heads or prenorm or shift_tokens), \
            'not restated (the reference constructs it with dim, dim_head, heads only: magvit2_pytorch.py:415-419)'
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        dim_inner = dim_head * heads
        self.to_q = _Seq(nn.Linear(dim, dim_inner, bias=False))
        self.to_kv = _Seq(nn.Linear(dim, dim_inner * 2, bias=False))
        self.to_out = _Seq(nn.Linear(dim_inner, dim, bias=False)) if combine_heads else nn.Identity()

    def forward(self, x, mask=None, context=None, eps=1e-5):
        b, n, _ = x.shape
        h, d = self.heads, self.dim_head
        q = self.to_q(x).reshape(b, n, h, d).transpose(1, 2)                       # b h n d
        kv = self.to_kv(x).reshape(b, n, 2, h, d).permute(2, 0, 3, 1, 4)          # kv b h n d
        k, v = kv[0], kv[1]
        q = q * self.scale
        q, k = second_taylor_expansion(q), second_taylor_expansion(k)
        kvs = torch.einsum('bhnd,bhne->bhde', k, v)
        qk_inv = 1. / torch.einsum('bhnd,bhmd->bhn', q, k).clamp(min=eps)
        out = torch.einsum('bhnd,bhde,bhn->bhne', q, kvs, qk_inv)
        out = out.transpose(1, 2).reshape(b, n, h * d)
        return self.to_out(out)


# ----------------------------------------------------------------------------- gateloop-transformer (>= 0.2.2): SimpleGateLoopLayer
class _GateLoopRMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return F.normalize(x, dim=-1) * self.scale * self.gamma


class SimpleGateLoopLayer(nn.Module):
    """RECONSTRUCTION from memory of gateloop_transformer.SimpleGateLoopLayer (package absent from /root/reference, pinned only as
    `gateloop-transformer>=0.2.2` in setup.py:28): **parity unpinned**, parameter names `norm.gamma`, `to_qkva.0.weight` unverified.
    prenorm RMSNorm -> Linear(dim, 3 dim, bias=False) split as [q | kv | a] -> per feature along the sequence
    h_t = sigmoid(a_t) h_{t-1} + kv_t (an associative scan upstream), out_t = q_t h_t.  The reference call site
    (magvit2_pytorch.py:1216-1222) constructs it with `dim` only and wraps it in Residual + ToTimeSequence."""

    def __init__(self, dim, prenorm=True, use_heinsen=False, use_jax_associative_scan=False, post_ln=False, reverse=False):
        super().__init__()
        assert not post_ln and not reverse
        self.dim = dim
        self.norm = _GateLoopRMSNorm(dim) if prenorm else nn.Identity()
        self.to_qkva = nn.Sequential(nn.Linear(dim, dim * 3, bias=False))

    def forward(self, x, cache=None, return_cache=False):
        assert cache is None and not return_cache
        x = self.norm(x)
        q, kv, a = self.to_qkva(x).chunk(3, dim=-1)          # 'b n (qkva d)': q first, then kv, then a
        a = a.sigmoid()
        h = torch.zeros_like(kv[:, 0])
        out = []
        for t in range(x.shape[1]):
            h = a[:, t] * h + kv[:, t]
            out.append(q[:, t] * h)
        return torch.stack(out, dim=1)
