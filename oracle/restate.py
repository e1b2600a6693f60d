"""
TEST INFRASTRUCTURE — not product code.  Only tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg may import this; the product path never does.

A self-contained CPU restatement (plain torch, functional, fp32 or fp64) of the reference's
VideoTokenizer eval hot path.  It needs nothing from /root/reference, so it travels to the
GPU box as the checker and as the ``cpu_baseline`` ("port").  Each function cites the
reference lines it follows (M = magvit2_pytorch/magvit2_pytorch.py, A = attend.py).

Pinning: ``tests/test_oracle_vs_reference.py`` (runs only where /root/reference exists)
executes the *reference's own code* (oracle/ref_loader.py) on the same weights/inputs and
requires this restatement to agree (ids identical, tensors to fp32 round-off), and
``tests/golden/`` holds outputs of the reference itself (oracle/make_golden.py) that this
file is checked against on every box.  The LFQ / FSQ / Taylor-attention arithmetic is a
reconstruction of un-vendored third-party packages: PARITY UNPINNED for those three pieces
(see oracle/thirdparty.py), pinned for everything else.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- spec parsing
def parse_layers(layers, init_dim, max_dim):
    """Layer-spec grammar of M:1138-1318, reduced to the scoped (un-conditioned) layer types.
    Returns a list of dicts {kind, dim, dim_out, n} in encoder order."""
    out = []
    dim = init_dim
    for ld in layers:
        kind, *params = ld if isinstance(ld, tuple) else (ld,)
        dim_out = dim
        n = 1
        if kind == 'residual':
            pass
        elif kind == 'consecutive_residual':
            n, = params
        elif kind in ('compress_space', 'compress_time'):
            dim_out = params[0] if len(params) > 0 and params[0] is not None else dim * 2
            dim_out = min(dim_out, max_dim)
        elif kind in ('attend_space', 'linear_attend_space', 'attend_time', 'gateloop_time', 'cond_residual', 'cond_attend_space', 'cond_attend_time'):
            pass
        else:
            raise ValueError(f'unknown / unscoped layer type {kind}')
        out.append(dict(kind=kind, dim=dim, dim_out=dim_out, n=n))
        dim = dim_out
    return out


# ----------------------------------------------------------------------------- blocks
def causal_conv3d(x, w, b, stride=1, pad_mode='constant'):
    """M:892-928: front-pad time by (kt-1) + (1-stride), same-pad H/W; `pad_mode` as F.pad's mode (the reference default is
    constant zeros; the tokenizer passes its pad_mode to conv_in and conv_out only, M:1109, 1127), falling back to constant
    when the clip is not longer than the time padding (M:925)."""
    kt, kh, kw = w.shape[2:]
    tp = (kt - 1) + (1 - stride)
    mode = pad_mode if tp < x.shape[2] else 'constant'
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, tp, 0), mode=mode)
    return F.conv3d(x, w, b, stride=(stride, 1, 1))


def squeeze_excite(x, p, pre):
    """M:221-240 on a 5-D tensor: per (b, t) frame softmax-pooled context -> 2-layer gate."""
    B, C, T, H, W = x.shape
    fr = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H * W)
    wk = p[pre + 'to_k.weight'].reshape(C)
    logits = torch.einsum('c,ncp->np', wk, fr) + p[pre + 'to_k.bias']
    attn = logits.softmax(dim=-1)
    pooled = torch.einsum('np,ncp->nc', attn, fr)
    w1 = p[pre + 'net.0.weight'].flatten(1)
    w2 = p[pre + 'net.2.weight'].flatten(1)
    h = F.leaky_relu(pooled @ w1.t() + p[pre + 'net.0.bias'], 0.1)
    g = torch.sigmoid(h @ w2.t() + p[pre + 'net.2.bias'])
    g = g.reshape(B, T, C).permute(0, 2, 1)[..., None, None]
    return x * g


def residual_unit(x, p, pre):
    """M:930-944: x + SE(ELU(conv1x1x1(ELU(causal_conv3d_k(x)))))."""
    y = F.elu(causal_conv3d(x, p[pre + 'fn.0.conv.weight'], p[pre + 'fn.0.conv.bias']))
    y = F.elu(F.conv3d(y, p[pre + 'fn.2.weight'], p[pre + 'fn.2.bias']))
    y = squeeze_excite(y, p, pre + 'fn.4.')
    return y + x


def spatial_down(x, p, pre):
    """M:770-780: per-frame Conv2d k3 s2 p1."""
    B, C, T, H, W = x.shape
    fr = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    w = p[pre + 'conv.weight']
    k = w.shape[-1]
    o = F.conv2d(fr, w, p[pre + 'conv.bias'], stride=2, padding=k // 2)
    return o.reshape(B, T, *o.shape[1:]).permute(0, 2, 1, 3, 4)


def time_down(x, p, pre):
    """M:796-807: per-pixel causal Conv1d k3 s2, left pad k-1."""
    B, C, T, H, W = x.shape
    seq = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, T)
    w = p[pre + 'conv.weight']
    seq = F.pad(seq, (w.shape[-1] - 1, 0))
    o = F.conv1d(seq, w, p[pre + 'conv.bias'], stride=2)
    return o.reshape(B, H, W, o.shape[1], o.shape[2]).permute(0, 3, 4, 1, 2)


def spatial_up(x, p, pre):
    """M:838-846: Conv2d 1x1 C -> 4C', SiLU, 'b (c p1 p2) h w -> b c (h p1) (w p2)'."""
    B, C, T, H, W = x.shape
    fr = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    o = F.silu(F.conv2d(fr, p[pre + 'net.0.weight'], p[pre + 'net.0.bias']))
    co = o.shape[1] // 4
    o = o.reshape(B * T, co, 2, 2, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B * T, co, H * 2, W * 2)
    return o.reshape(B, T, co, H * 2, W * 2).permute(0, 2, 1, 3, 4)


def time_up(x, p, pre):
    """M:875-883: Conv1d k1 C -> 2C', SiLU, 'b (c p) t -> b c (t p)'."""
    B, C, T, H, W = x.shape
    seq = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, T)
    o = F.silu(F.conv1d(seq, p[pre + 'net.0.weight'], p[pre + 'net.0.bias']))
    co = o.shape[1] // 2
    o = o.reshape(B * H * W, co, 2, T).permute(0, 1, 3, 2).reshape(B * H * W, co, T * 2)
    return o.reshape(B, H, W, co, T * 2).permute(0, 3, 4, 1, 2)


def rmsnorm_last(x, gamma):
    """M:275-276 (channel-last): F.normalize(x, dim=-1) * sqrt(dim) * gamma."""
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * gamma


def attend(q, k, v, causal):
    """A:186-243 (einsum path): softmax(q k^T d^-1/2 [+ right-aligned causal mask]) v."""
    n, j = q.shape[-2], k.shape[-2]
    dots = torch.einsum('bhid,bhjd->bhij', q, k) * (q.shape[-1] ** -0.5)
    if causal and n > 1:
        mask = torch.ones((n, j), dtype=torch.bool).triu(j - n + 1)            # A:46-47
        dots = dots.masked_fill(mask, -torch.finfo(dots.dtype).max)
    return torch.einsum('bhij,bhjd->bhid', dots.softmax(dim=-1), v)


def adaptive_gamma(cond, p, pre):
    """AdaptiveRMSNorm M:306-323 without to_bias (the tokenizer never builds it): gamma = to_gamma(cond), (B, dim)."""
    return cond @ p[pre + 'to_gamma.weight'].t() + p[pre + 'to_gamma.bias']


def attention_tokens(tok, p, pre, heads, causal, gamma=None):
    """M:371-388 on tokens (b', n, C): norm -> qkv -> prepend mem_kv -> attend -> out.  gamma (b', 1, C): AdaptiveRMSNorm."""
    b, n, C = tok.shape
    xn = rmsnorm_last(tok, p[pre + 'norm.gamma'] if gamma is None else gamma)
    qkv = xn @ p[pre + 'to_qkv.0.weight'].t()
    d = qkv.shape[-1] // (3 * heads)
    qkv = qkv.reshape(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)                 # qkv b h n d
    q, k, v = qkv[0], qkv[1], qkv[2]
    mk, mv = p[pre + 'mem_kv'][0], p[pre + 'mem_kv'][1]
    k = torch.cat((mk.expand(b, -1, -1, -1), k), dim=-2)
    v = torch.cat((mv.expand(b, -1, -1, -1), v), dim=-2)
    o = attend(q, k, v, causal)
    o = o.transpose(1, 2).reshape(b, n, heads * d)
    return o @ p[pre + 'to_out.1.weight'].t()


def space_attention(x, p, pre, heads, cond=None):
    """M:444-454: tokens = H*W per (b, t) frame.  cond (B, dim_cond): the sample's adaptive gamma for each of its frames —
    the INTENT of 'cond_attend_space'; the reference itself only runs this with T = 1 (AdaptiveRMSNorm asserts
    cond.shape[0] == B*T, M:309) and never through VideoTokenizer (its Sequential drops the keyword, M:1256-1259)."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 2, 3, 4, 1).reshape(B * T, H * W, C)
    g = None if cond is None else adaptive_gamma(cond, p, pre + 'norm.').repeat_interleave(T, dim=0)[:, None, :]
    o = attention_tokens(tok, p, pre, heads, causal=False, gamma=g)
    return o.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)


def time_attention(x, p, pre, heads, cond=None):
    """M:456-464: tokens = T per (b, h, w) pixel, causal.  cond: see space_attention (the reference runs it at H = W = 1 only)."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 3, 4, 2, 1).reshape(B * H * W, T, C)
    g = None if cond is None else adaptive_gamma(cond, p, pre + 'norm.').repeat_interleave(H * W, dim=0)[:, None, :]
    o = attention_tokens(tok, p, pre, heads, causal=True, gamma=g)
    return o.reshape(B, H, W, T, C).permute(0, 4, 3, 1, 2)


def taylor_feature(x):
    x0 = x.new_ones(x.shape[:-1] + (1,))
    x2 = (x.unsqueeze(-1) * x.unsqueeze(-2)).flatten(-2) * (0.5 ** 0.5)
    return torch.cat((x0, x, x2), dim=-1)


def linear_space_attention(x, p, pre, heads, dim_head):
    """M:432-442 -> M:421-430 -> third-party TaylorSeriesLinearAttn (reconstruction)."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 2, 3, 4, 1).reshape(B * T, H * W, C)
    xn = rmsnorm_last(tok, p[pre + 'norm.gamma'])
    b, n, _ = xn.shape
    q = (xn @ p[pre + 'attn.to_q.0.weight'].t()).reshape(b, n, heads, dim_head).transpose(1, 2)
    kv = (xn @ p[pre + 'attn.to_kv.0.weight'].t()).reshape(b, n, 2, heads, dim_head).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    q = taylor_feature(q * dim_head ** -0.5)
    k = taylor_feature(k)
    kvs = torch.einsum('bhnd,bhne->bhde', k, v)
    qk_inv = 1. / torch.einsum('bhnd,bhd->bhn', q, k.sum(dim=-2)).clamp(min=1e-5)
    o = torch.einsum('bhnd,bhde,bhn->bhne', q, kvs, qk_inv)
    o = o.transpose(1, 2).reshape(b, n, heads * dim_head) @ p[pre + 'attn.to_out.0.weight'].t()
    return o.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)


def conv3d_mod(x, w, cond, eps=1e-8):
    """Conv3DMod.forward M:715-753 (StyleGAN2 modulation): per-sample weights w * (cond + 1) on the input channels, demodulated
    to unit norm per output channel, causal zero padding, one grouped convolution."""
    B = x.shape[0]
    kt, kh, kw = w.shape[2:]
    wb = w[None] * (cond[:, None, :, None, None, None] + 1)                     # (B, o, i, kt, kh, kw)
    inv = wb.pow(2).sum(dim=(2, 3, 4, 5), keepdim=True).clamp(min=eps).rsqrt()
    wb = (wb * inv).reshape(B * w.shape[0], *w.shape[1:])
    xg = F.pad(x.reshape(1, B * x.shape[1], *x.shape[2:]), (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0))
    return F.conv3d(xg, wb, groups=B).reshape(B, w.shape[0], *x.shape[2:])


def residual_unit_mod(x, p, pre, cond):
    """ResidualUnitMod.forward M:975-988: to_cond -> Conv3DMod -> ELU -> 1x1x1 -> ELU -> + x."""
    c = cond @ p[pre + 'to_cond.weight'].t() + p[pre + 'to_cond.bias']
    h = F.elu(conv3d_mod(x, p[pre + 'conv.weights'], c))
    h = F.elu(F.conv3d(h, p[pre + 'conv_out.weight'], p[pre + 'conv_out.bias']))
    return h + x


def feed_forward(x, p, pre, cond=None):
    """M:499-508 / GEGLU M:466-469: channel-first (Adaptive)RMSNorm -> 1x1x1 -> gelu(gate) * x -> 1x1x1."""
    C = x.shape[1]
    gamma = p[pre + 'norm.gamma'] if cond is None else adaptive_gamma(cond, p, pre + 'norm.')[:, :, None, None, None]
    xn = F.normalize(x, dim=1) * (C ** 0.5) * gamma
    h = F.conv3d(xn, p[pre + 'net.0.weight'], p[pre + 'net.0.bias'])
    a, gate = h.chunk(2, dim=1)
    h = F.gelu(gate) * a
    return F.conv3d(h, p[pre + 'net.2.weight'], p[pre + 'net.2.bias'])


def gateloop_time(x, p, pre):
    """'gateloop_time' (M:1216-1222): ToTimeSequence(Residual(SimpleGateLoopLayer(dim))) — per pixel a sequence over the frames.
    Third-party layer restated from oracle/thirdparty.py (reconstruction, parity unpinned): RMSNorm (gamma * sqrt(dim)) -> Linear(dim, 3 dim)
    = [q | kv | a] -> h_t = sigmoid(a_t) h_{t-1} + kv_t, out_t = q_t h_t; + x."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 3, 4, 2, 1).reshape(B * H * W, T, C)
    n = rmsnorm_last(tok, p[pre + 'norm.gamma'])
    q, kv, a = (n @ p[pre + 'to_qkva.0.weight'].t()).chunk(3, dim=-1)
    a = a.sigmoid()
    h = torch.zeros_like(kv[:, 0])
    out = []
    for t in range(T):
        h = a[:, t] * h + kv[:, t]
        out.append(q[:, t] * h)
    o = torch.stack(out, dim=1) + tok
    return o.reshape(B, H, W, T, C).permute(0, 4, 3, 1, 2)


def token_shift(x):
    """M:250-253: second half of the channels delayed by one frame, zero at t = 0."""
    a, s = x.chunk(2, dim=1)
    s = F.pad(s, (0, 0, 0, 0, 1, -1))
    return torch.cat((a, s), dim=1)


# ----------------------------------------------------------------------------- quantizers
def _project(t, p, which):
    """quantizers.project_in / project_out: a Linear, or the identity when the latent width already equals the code width
    (third-party `has_projections = dim != codebook_dims`: nn.Identity, no parameters in the state_dict)."""
    w = p.get(f'quantizers.{which}.weight')
    return t if w is None else t @ w.t() + p[f'quantizers.{which}.bias']


def _code_dtype(p):
    w = p.get('quantizers.project_out.weight')
    return torch.float32 if w is None else w.dtype


def lfq_quantize(x, p, codebook_dim, num_codebooks=1, clamp=10., scale=1., train=False, spherical=False):
    """third-party LFQ.forward eval path (reconstruction; see oracle/thirdparty.py).  spherical: the projected input and the
    quantized value are l2-normalised per codebook (ids unchanged: normalisation keeps every sign; quantized = +-scale / sqrt(d))."""
    B, C = x.shape[:2]
    sp = x.shape[2:]
    tok = x.reshape(B, C, -1).transpose(1, 2)
    z = _project(tok, p, 'project_in')
    if clamp is not None:
        z = (z / clamp).tanh() * clamp
    z = z.reshape(B, -1, num_codebooks, codebook_dim)
    if spherical:
        z = F.normalize(z, dim=-1) * scale
    z = z.float()
    pos = z > 0
    mask = (2 ** torch.arange(codebook_dim - 1, -1, -1)).int()
    ids = (pos.int() * mask).sum(dim=-1)                                       # int64
    q = torch.where(pos, torch.full_like(z, scale), torch.full_like(z, -scale))
    if spherical:
        q = F.normalize(q, dim=-1) * scale
    if train:
        q = z + (q - z)                                                        # straight-through value (forward: q up to fp32 rounding)
    q = q.to(x.dtype).reshape(B, -1, num_codebooks * codebook_dim)
    out = _project(q, p, 'project_out')
    out = out.transpose(1, 2).reshape(B, C, *sp)
    ids = ids.reshape(B, *sp, num_codebooks)
    if num_codebooks == 1:
        ids = ids.squeeze(-1)
    return out, ids, z.reshape(B, *sp, num_codebooks * codebook_dim)


def lfq_train_losses(z, codebook_dim, num_codebooks, inv_temperature=100., diversity_gamma=2.5, entropy_loss_weight=0.1,
                     commitment_loss_weight=1., scale=1.):
    """third-party LFQ.forward TRAINING branch (reconstruction, parity unpinned; see oracle/thirdparty.py): from the
    pre-sign activations z (..., num_codebooks * codebook_dim) -> (per_sample_entropy, batch_entropy, commit_loss, aux_loss).
    Single-process form (torch.distributed: the token-averaged distribution is mean-all-reduced before its entropy)."""
    z = z.reshape(-1, num_codebooks, codebook_dim).float()
    K = 2 ** codebook_dim
    mask = (2 ** torch.arange(codebook_dim - 1, -1, -1))
    codebook = (((torch.arange(K)[:, None] & mask) != 0).float() * 2 - 1) * scale                # (K, d), MSB first
    prob = (2 * inv_temperature * torch.einsum('ncd,kd->nck', z, codebook)).softmax(dim=-1)
    ent = lambda pr: (-pr * pr.clamp(min=1e-5).log()).sum(dim=-1)
    per_sample = ent(prob).mean()
    batch = ent(prob.mean(dim=0)).mean()
    q = torch.where(z > 0, torch.full_like(z, scale), torch.full_like(z, -scale))
    commit = F.mse_loss(z, q) if commitment_loss_weight > 0 else z.new_zeros(())
    aux = (per_sample - diversity_gamma * batch) * entropy_loss_weight + commit * commitment_loss_weight
    return per_sample, batch, commit, aux


def lfq_indices_to_codes(ids, p, codebook_dim, num_codebooks=1, scale=1., spherical=False):
    dtype = _code_dtype(p)
    if num_codebooks == 1:
        ids = ids.unsqueeze(-1)
    mask = (2 ** torch.arange(codebook_dim - 1, -1, -1)).int()
    bits = ((ids[..., None].int() & mask) != 0).to(dtype)
    codes = bits * scale * 2 - scale
    if spherical:
        codes = F.normalize(codes, dim=-1) * scale
    out = _project(codes.flatten(-2), p, 'project_out')
    return out.movedim(-1, 1)


def _fsq_consts(levels, dtype):
    L = torch.tensor(levels, dtype=torch.int32)
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1])), dim=0, dtype=torch.int32)
    half_l = (L - 1) * (1 + 1e-3) / 2
    offset = torch.where(L % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    return L, basis, half_l, offset, shift, L // 2


def fsq_quantize(x, p, levels):
    """third-party FSQ.forward (reconstruction)."""
    B, C = x.shape[:2]
    sp = x.shape[2:]
    tok = x.reshape(B, C, -1).transpose(1, 2)
    z = _project(tok, p, 'project_in')
    L, basis, half_l, offset, shift, hw = _fsq_consts(levels, z.dtype)
    bounded = (z + shift).tanh() * half_l - offset
    codes = bounded.round() / hw
    ids = ((codes * hw + hw) * basis).sum(dim=-1).to(torch.int32)
    out = _project(codes.to(x.dtype), p, 'project_out')
    return out.transpose(1, 2).reshape(B, C, *sp), ids.reshape(B, *sp), bounded.reshape(B, *sp, len(levels))


def fsq_indices_to_codes(ids, p, levels):
    L, basis, half_l, offset, shift, hw = _fsq_consts(levels, torch.float32)
    lv = (ids.unsqueeze(-1) // basis) % L
    codes = ((lv - hw) / hw).to(_code_dtype(p))
    out = _project(codes, p, 'project_out')
    return out.movedim(-1, 1)


# ----------------------------------------------------------------------------- the tokenizer
class OracleTokenizer:
    """Functional restatement of VideoTokenizer eval paths (M:1522-1720) over a plain
    ``state_dict`` with the reference's key names."""

    def __init__(self, state_dict, *, image_size, layers, codebook_size=None, init_dim=64, max_dim=float('inf'),
                 channels=3, num_codebooks=1, use_fsq=False, fsq_levels=None, attn_dim_head=32, attn_heads=8,
                 linear_attn_dim_head=8, linear_attn_heads=16, lfq_soft_clamp_input_value=10., dim_cond=None,
                 separate_first_frame_encoding=False, pad_mode='constant', lfq_spherical=False, dtype=torch.float32, **_ignored):
        self.p = {k: (v.detach().to(dtype) if v.is_floating_point() else v.detach().clone())
                  for k, v in state_dict.items() if not k.startswith('discr.')}
        self.dtype = dtype
        self.image_size = image_size
        self.channels = channels
        self.stages = parse_layers(layers, init_dim, max_dim)
        self.attn_heads = attn_heads
        self.lin_heads, self.lin_dim_head = linear_attn_heads, linear_attn_dim_head
        self.use_fsq = use_fsq
        self.fsq_levels = list(fsq_levels) if fsq_levels is not None else None
        self.num_codebooks = num_codebooks
        self.codebook_dim = int(math.log2(codebook_size)) if codebook_size is not None else None
        self.clamp = lfq_soft_clamp_input_value
        self.time_downsample_factor = 2 ** sum(s['kind'] == 'compress_time' for s in self.stages)
        self.time_padding = self.time_downsample_factor - 1
        self.fmap_size = image_size // (2 ** sum(s['kind'] == 'compress_space' for s in self.stages))
        self.separate_first_frame = bool(separate_first_frame_encoding)
        self.pad_mode = pad_mode
        self.lfq_spherical = bool(lfq_spherical)
        self.dim_cond = dim_cond
        self.has_cond = any(st['kind'].startswith('cond_') for st in self.stages)
        self.taps = None            # optional dict name -> tensor for layer-wise debugging

    def _tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x
        return x

    # stage dispatch, shared by encoder (enc=True) and decoder
    def _cond_in(self, cond, which):
        """M:1341-1353, 1545-1548: Linear(dim_cond, dim_cond * expansion) + SiLU, separate stems for encoder and decoder."""
        assert (not self.has_cond) or cond is not None, '`cond` must be passed into tokenizer forward method since conditionable layers were specified'
        if cond is None:
            return None
        assert cond.shape[1] == self.dim_cond
        return F.silu(cond.to(self.dtype) @ self.p[which + '.0.weight'].t() + self.p[which + '.0.bias'])

    def _stage(self, x, st, pre, enc, cond=None, cond_raw=None):
        p, kind = self.p, st['kind']
        if kind == 'cond_residual':
            return residual_unit_mod(x, p, pre + '.', cond)
        if kind == 'cond_attend_space':       # AdaptiveRMSNorm is built on the raw dim_cond (M:1247-1254): needs expansion factor 1
            x = space_attention(x, p, pre + '.0.fn.', self.attn_heads, cond=cond) + x
            return feed_forward(x, p, pre + '.1.fn.', cond=cond) + x
        if kind == 'cond_attend_time':
            x = time_attention(token_shift(x), p, pre + '.0.fn.fn.', self.attn_heads, cond=cond) + x
            return feed_forward(token_shift(x), p, pre + '.1.fn.fn.', cond=cond) + x
        if kind == 'residual':
            return residual_unit(x, p, pre + '.')
        if kind == 'consecutive_residual':
            for j in range(st['n']):
                x = residual_unit(x, p, f'{pre}.{j}.')
            return x
        if kind == 'compress_space':
            return spatial_down(x, p, pre + '.') if enc else spatial_up(x, p, pre + '.')
        if kind == 'compress_time':
            return time_down(x, p, pre + '.') if enc else time_up(x, p, pre + '.')
        if kind == 'attend_space':
            x = space_attention(x, p, pre + '.0.fn.', self.attn_heads) + x
            return feed_forward(x, p, pre + '.1.fn.') + x
        if kind == 'linear_attend_space':
            x = linear_space_attention(x, p, pre + '.0.fn.', self.lin_heads, self.lin_dim_head) + x
            return feed_forward(x, p, pre + '.1.fn.') + x
        if kind == 'attend_time':
            x = time_attention(token_shift(x), p, pre + '.0.fn.fn.', self.attn_heads) + x
            return feed_forward(token_shift(x), p, pre + '.1.fn.fn.') + x
        if kind == 'gateloop_time':
            return gateloop_time(x, p, pre + '.fn.fn.')
        raise ValueError(kind)

    def encode(self, video, cond=None, video_contains_first_frame=True):
        """M:1522-1576.  NB the trailing LayerNorm of encoder_layers is never executed (zip truncation, M:1565).
        video_contains_first_frame=False ("headless" clips, M:1530-1538): no time_padding zero frames in front and the first
        frame is not treated separately; conv_in still pads causally on its own."""
        x = video.to(self.dtype)
        tp = self.time_padding if video_contains_first_frame else 0
        if self.separate_first_frame and video_contains_first_frame:
            # M:1553-1561: first frame through its own SameConv2d, the REMAINING frames through conv_in (whose causal zero
            # padding starts right before frame 1 — it never sees the first frame), then time_padding zero frames in front
            w2 = self.p['conv_in_first_frame.weight']
            ff = F.conv2d(x[:, :, 0], w2, self.p['conv_in_first_frame.bias'], padding=(w2.shape[-2] // 2, w2.shape[-1] // 2))
            rest = causal_conv3d(x[:, :, 1:], self.p['conv_in.conv.weight'], self.p['conv_in.conv.bias'], pad_mode=self.pad_mode)
            x = torch.cat((ff.unsqueeze(2), rest), dim=2)
            x = self._tap('conv_in', F.pad(x, (0, 0, 0, 0, self.time_padding, 0)))
        else:
            x = F.pad(x, (0, 0, 0, 0, tp, 0))                                   # M:1534-1537
            x = self._tap('conv_in', causal_conv3d(x, self.p['conv_in.conv.weight'], self.p['conv_in.conv.bias'], pad_mode=self.pad_mode))
        ce = self._cond_in(cond, 'encoder_cond_in')
        for i, st in enumerate(self.stages):
            x = self._tap(f'enc{i}', self._stage(x, st, f'encoder_layers.{i}', True, cond=ce))
        return x

    def quantize(self, x, train=False):
        if self.use_fsq:
            q, ids, pre = fsq_quantize(x, self.p, self.fsq_levels)
        else:
            q, ids, pre = lfq_quantize(x, self.p, self.codebook_dim, self.num_codebooks, self.clamp, train=train, spherical=self.lfq_spherical)
        self._tap('prequant', pre)
        return self._tap('quantized', q), ids

    def indices_to_codes(self, ids):
        if self.use_fsq:
            return fsq_indices_to_codes(ids, self.p, self.fsq_levels)
        return lfq_indices_to_codes(ids, self.p, self.codebook_dim, self.num_codebooks, spherical=self.lfq_spherical)

    def decode(self, q, cond=None, video_contains_first_frame=True):
        """M:1597-1649 (headless: nothing is cropped, M:1646)."""
        x = q
        n = len(self.stages)
        cd = self._cond_in(cond, 'decoder_cond_in')
        for j, st in enumerate(reversed(self.stages)):
            x = self._tap(f'dec{j}', self._stage(x, st, f'decoder_layers.{j}', False, cond=cd))
        if self.separate_first_frame and video_contains_first_frame:             # M:1633-1639
            tp = self.time_padding
            w2 = self.p['conv_out_first_frame.weight']
            outff = F.conv2d(x[:, :, tp], w2, self.p['conv_out_first_frame.bias'], padding=(w2.shape[-2] // 2, w2.shape[-1] // 2))
            out = causal_conv3d(x[:, :, tp + 1:], self.p['conv_out.conv.weight'], self.p['conv_out.conv.bias'], pad_mode=self.pad_mode)
            return torch.cat((outff.unsqueeze(2), out), dim=2)
        x = causal_conv3d(x, self.p['conv_out.conv.weight'], self.p['conv_out.conv.bias'], pad_mode=self.pad_mode)
        return x[:, :, self.time_padding:] if video_contains_first_frame else x  # M:1646-1647

    def decode_from_code_indices(self, codes, cond=None, video_contains_first_frame=True):
        """M:1578-1595."""
        assert codes.dtype in (torch.long, torch.int32)
        if codes.ndim == 2:
            f2 = self.fmap_size ** 2
            assert codes.shape[-1] % f2 == 0
            codes = codes.reshape(codes.shape[0], -1, self.fmap_size, self.fmap_size)
        return self.decode(self.indices_to_codes(codes), cond=cond, video_contains_first_frame=video_contains_first_frame)

    def tokenize(self, video, cond=None):
        return self.forward(video, cond=cond, return_codes=True)

    def forward(self, video, cond=None, return_codes=False, return_recon=False, video_contains_first_frame=True):
        """M:1656-1720 eval branches."""
        assert video.ndim in (4, 5)
        assert video.shape[-2:] == (self.image_size, self.image_size)
        if video.ndim == 4:
            video, video_contains_first_frame = video.unsqueeze(2), True        # M:1683-1685
        vcff = video_contains_first_frame
        assert (video.shape[2] - int(vcff)) % self.time_downsample_factor == 0   # M:1691
        with torch.no_grad():
            x = self.encode(video, cond=cond, video_contains_first_frame=vcff)
            q, ids = self.quantize(x)
            if return_codes and not return_recon:
                return ids
            recon = self.decode(q, cond=cond, video_contains_first_frame=vcff)
        if return_codes:
            return ids, recon
        return recon

    def forward_loss(self, video, training=True, lfq_entropy_loss_weight=0.1, lfq_commitment_loss_weight=1., lfq_diversity_gamma=2.5,
                     quantizer_aux_loss_weight=1., cond=None, video_contains_first_frame=True, discr_params=None, discr_kw=None,
                     adversarial_loss_weight=1., multiscale_discrs=(), multiscale_adversarial_loss_weight=1., vgg=None,
                     perceptual_loss_weight=0.):
        """M:1656-1896 with return_loss=True: total = mse(video, recon) + aux * quantizer_aux_loss_weight
        (+ perceptual * w + gen_loss * adaptive_weight * adversarial_loss_weight + multiscale terms); aux = 0 in eval mode and
        for FSQ (M:1700-1705).  Forward values only, so the adaptive weights are the constant 1 of M:1832 / 1861 (the
        gradient-norm ratio of M:1819-1842 exists only in training mode WITH a VGG).
        discr_params: the `discr.*` entries -> the generator's hinge loss -mean(D(random recon frame)) (M:1826-1832);
        multiscale_discrs: only their NUMBER matters to the reference, which never calls them on this branch (M:1856:
        `fake_logits = recon_video_frames`), so each contributes -mean(picked recon frames);
        vgg: a callable on (B, 3, H, W) frames (M:1785-1805), eval mode only.
        Returns (total, dict(recon_loss, aux, per_sample_entropy, batch_entropy, commitment, perceptual, gen_loss, multiscale_gen_losses))."""
        if video.ndim == 4:
            video, video_contains_first_frame = video.unsqueeze(2), True
        vcff = video_contains_first_frame
        with torch.no_grad():
            x = self.encode(video, cond=cond, video_contains_first_frame=vcff)
            train = training and not self.use_fsq
            self.taps = self.taps if self.taps is not None else {}
            q, ids = self.quantize(x, train=train)
            recon = self.decode(q, cond=cond, video_contains_first_frame=vcff)
            recon_loss = F.mse_loss(video, recon)
            zero = recon_loss.new_zeros(())
            if train:
                ps, bt, cm, aux = lfq_train_losses(self.taps['prequant'], self.codebook_dim, self.num_codebooks, diversity_gamma=lfq_diversity_gamma,
                                                   entropy_loss_weight=lfq_entropy_loss_weight, commitment_loss_weight=lfq_commitment_loss_weight,
                                                   scale=self.codebook_dim ** -0.5 if self.lfq_spherical else 1.)
            else:
                ps = bt = cm = aux = zero
            batch, frames = video.shape[0], video.shape[2]
            bi = torch.arange(batch)
            pick = lambda vid, idx: vid[bi, :, idx[:, 0]]                                    # pick_video_frame M:88-97
            perceptual, fi = zero, None
            if vgg is not None and perceptual_loss_weight > 0.:                              # M:1785-1805
                assert not training, 'training mode + VGG needs the gradient-norm adaptive weight (autograd)'
                fi = torch.randn((batch, frames)).topk(1, dim=-1).indices
                a, b = pick(video, fi), pick(recon, fi)
                if self.channels == 1:
                    a, b = a.expand(-1, 3, -1, -1), b.expand(-1, 3, -1, -1)
                elif self.channels == 4:
                    a, b = a[:, :3], b[:, :3]
                perceptual = F.mse_loss(vgg(a), vgg(b))
            gen_loss, frames_picked = zero, None
            has_gan = discr_params is not None and adversarial_loss_weight > 0.
            if has_gan:                                                                      # M:1826-1832
                fi = torch.randn((batch, frames)).topk(1, dim=-1).indices
                frames_picked = pick(recon, fi)
                kw = dict(dict(image_size=self.image_size, channels=self.channels), **(discr_kw or {}))
                gen_loss = -discriminator_forward(frames_picked, discr_params, 'discr.', **kw).mean()      # hinge_gen_loss M:123-124
            ms = []
            if len(multiscale_discrs) > 0 and multiscale_adversarial_loss_weight > 0.:      # M:1851-1869
                if frames_picked is None and fi is not None:
                    frames_picked = pick(recon, fi)
                if frames_picked is None:
                    raise NameError('the reference reads `frame_indices` before assignment here (M:1853) unless has_gan or use_vgg')
                ms = [-frames_picked.mean() for _ in multiscale_discrs]
        total = recon_loss + aux * quantizer_aux_loss_weight + perceptual * perceptual_loss_weight + gen_loss * adversarial_loss_weight
        if ms:
            total = total + sum(ms) * multiscale_adversarial_loss_weight
        return total, dict(recon_loss=recon_loss, aux=aux, per_sample_entropy=ps, batch_entropy=bt, commitment=cm,
                           perceptual=perceptual, gen_loss=gen_loss, multiscale_gen_losses=ms)

    def forward_discr_loss(self, video, discr_params, *, dim, channels=3, max_dim=512, cond=None, video_contains_first_frame=True,
                           multiscale_discrs=(), multiscale_adversarial_loss_weight=1.):
        """M:1730-1782 with apply_gradient_penalty=False: one random frame per sample
        (torch.randn((batch, frames)).topk(1) on the CPU default generator — seed it before the call to compare), real and
        reconstructed frame through the image discriminator, hinge loss; every multiscale discriminator (any callable on the
        whole clip, M:1756-1763) adds hinge(D(recon), D(video)) * multiscale_adversarial_loss_weight.
        discr_params: the `discr.*` entries.  Returns (total, real_logits, fake_logits)."""
        if video.ndim == 4:
            video, video_contains_first_frame = video.unsqueeze(2), True
        with torch.no_grad():
            recon = self.forward(video, cond=cond, return_recon=True, video_contains_first_frame=video_contains_first_frame)
            batch, frames = video.shape[0], video.shape[2]
            frame_indices = torch.randn((batch, frames)).topk(1, dim=-1).indices
            bi = torch.arange(batch)
            real, fake = video[bi, :, frame_indices[:, 0]], recon[bi, :, frame_indices[:, 0]]
            kw = dict(dim=dim, image_size=self.image_size, channels=channels, max_dim=max_dim)
            real_logits = discriminator_forward(real, discr_params, 'discr.', **kw)
            fake_logits = discriminator_forward(fake, discr_params, 'discr.', **kw)
            total = hinge_discr_loss(fake_logits, real_logits)
            if multiscale_adversarial_loss_weight > 0.:
                for d in multiscale_discrs:
                    total = total + hinge_discr_loss(d(recon), d(video)) * multiscale_adversarial_loss_weight
        return total, real_logits, fake_logits

    __call__ = forward


# ----------------------------------------------------------------------------- discriminator
def discriminator_dims(dim, image_size, channels=3, max_dim=512):
    """Discriminator.__init__ M:605-630: (in, out) channels per block; the last block does not downsample."""
    num_layers = int(math.log2(image_size) - 2)
    dims = [channels] + [min((dim * 4) * (2 ** i), max_dim) for i in range(num_layers + 1)]
    return list(zip(dims[:-1], dims[1:]))


def discriminator_forward(x, p, pre, *, dim, image_size, channels=3, max_dim=512, linear_attn_heads=16, linear_attn_dim_head=8):
    """Discriminator.forward M:670-676 (antialiased_downsample=False, the default): per block DiscriminatorBlock M:575-588
    (1x1 stride-2 residual conv; conv3x3 + leaky_relu(0.1) twice; space-to-depth + 1x1 conv; (x + res) / sqrt 2), then
    Residual(LinearSpaceAttention) + Residual(FeedForward(images=True)); to_logits: conv3x3, leaky_relu, flatten, Linear -> (B,)."""
    pairs = discriminator_dims(dim, image_size, channels, max_dim)
    for i, (cin, cout) in enumerate(pairs):
        b = f'{pre}blocks.{i}.0.'
        last = i == len(pairs) - 1
        res = F.conv2d(x, p[b + 'conv_res.weight'], p[b + 'conv_res.bias'], stride=1 if last else 2)
        h = F.leaky_relu(F.conv2d(x, p[b + 'net.0.weight'], p[b + 'net.0.bias'], padding=1), 0.1)
        h = F.leaky_relu(F.conv2d(h, p[b + 'net.2.weight'], p[b + 'net.2.bias'], padding=1), 0.1)
        if not last:
            B, C, H, W = h.shape
            h = h.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, C * 4, H // 2, W // 2)   # 'b c (h p1) (w p2) -> b (c p1 p2) h w'
            h = F.conv2d(h, p[b + 'downsample.1.weight'], p[b + 'downsample.1.bias'])
        x = (h + res) * (2 ** -0.5)
        a = f'{pre}blocks.{i}.1.'
        x5 = x.unsqueeze(2)
        x5 = linear_space_attention(x5, p, a + '0.fn.', linear_attn_heads, linear_attn_dim_head) + x5
        pf = {k: (v.unsqueeze(-1) if (k.startswith(a + '1.fn.') and v.ndim >= 3) else v) for k, v in p.items() if k.startswith(a + '1.fn.')}
        x5 = feed_forward(x5, pf, a + '1.fn.') + x5                    # images=True: Conv2d / (dim, 1, 1) gamma, same arithmetic
        x = x5.squeeze(2)
    h = F.leaky_relu(F.conv2d(x, p[pre + 'to_logits.0.weight'], p[pre + 'to_logits.0.bias'], padding=1), 0.1)
    return h.flatten(1) @ p[pre + 'to_logits.3.weight'].t().squeeze(-1) + p[pre + 'to_logits.3.bias']


def hinge_discr_loss(fake, real):
    """M:120-121."""
    return (F.relu(1 + fake) + F.relu(1 - real)).mean()
