"""
Batch sharding for the tokenizer hot path: one process per GPU, clips are independent units.

Every op on the path is per-sample (no BatchNorm; SqueezeExcite / softmax / attention reductions stay
inside a clip; the eval quantizer has no cross-sample state — SURVEY.md §8e), so the data path needs
NO collective: each rank encodes / decodes its own clips with replicated weights.  The only exchange
is optional result gathering off the critical path — code ids (10 KB per 17-frame clip) via one
all_gather, and a few fp32 metric sums via one all_reduce — over RCCL/xGMI (``backend='nccl'`` is RCCL
on ROCm); both are latency-bound single small messages, so no bucketing is needed.  The reference has
no equivalent (its only parallelism is accelerate/DDP training, trainer.py:44-46, 97-102).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, rank: int, world_size: int):
    """contiguous, balanced split of n clips: the first n % world ranks take one extra."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(video: torch.Tensor):
    rank, ws = world()
    lo, hi = shard_bounds(video.shape[0], rank, ws)
    return video[lo:hi]


def gather_codes(local_codes: torch.Tensor, total: int) -> torch.Tensor:
    """all_gather of the per-rank code ids into the global (total, ...) tensor (ragged-safe)."""
    rank, ws = world()
    if ws == 1:
        return local_codes
    counts = [shard_bounds(total, r, ws)[1] - shard_bounds(total, r, ws)[0] for r in range(ws)]
    mx = max(counts)
    pad = local_codes.new_zeros((mx,) + tuple(local_codes.shape[1:]))
    pad[: local_codes.shape[0]] = local_codes
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def reduce_metrics(sums: Dict[str, float], device=None) -> Dict[str, float]:
    """one all_reduce(SUM) of a few fp32 scalars (e.g. squared-error sum, element count)."""
    rank, ws = world()
    if ws == 1:
        return dict(sums)
    keys = sorted(sums)
    t = torch.tensor([float(sums[k]) for k in keys], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: float(v) for k, v in zip(keys, t.tolist())}


@torch.no_grad()
def tokenize_sharded(tokenizer, video: torch.Tensor, gather: bool = True):
    """each rank tokenizes its shard of the global batch; optionally gathers all ids everywhere."""
    local = shard_batch(video)
    if local.shape[0] == 0:
        # more ranks than clips: this rank has nothing to encode but still joins the collective (an empty shard must not
        # build a zero-size plan while the other ranks wait in all_gather)
        T = video.shape[2] if video.ndim == 5 else 1
        lt = (T - 1) // tokenizer.time_downsample_factor + 1
        tail = (lt, tokenizer.fmap_size, tokenizer.fmap_size) + ((tokenizer.cfg['num_codebooks'],) if tokenizer.cfg['num_codebooks'] > 1 else ())
        dev = tokenizer.device if tokenizer.device.type == 'cuda' else video.device
        codes = torch.zeros((0,) + tail, dtype=torch.int32 if tokenizer.use_fsq else torch.int64, device=dev)
    else:
        codes = tokenizer.tokenize(local.to(tokenizer.device) if tokenizer.device.type == 'cuda' else local)
    return gather_codes(codes, video.shape[0]) if gather else codes


@torch.no_grad()
def reconstruction_metrics_sharded(tokenizer, video: torch.Tensor) -> Dict[str, float]:
    """global MSE / PSNR of forward(video) over a sharded batch: local sums + one tiny all_reduce."""
    local = shard_batch(video)
    on_gpu = tokenizer.device.type == 'cuda'
    if local.shape[0] == 0:                      # empty shard: contribute zeros, still join the all_reduce
        se, n = 0.0, 0.0
    else:
        dev_in = local.to(tokenizer.device) if on_gpu else local
        recon = tokenizer(dev_in, return_recon=True)
        ref = dev_in.unsqueeze(2) if dev_in.ndim == 4 else dev_in
        se, n = (recon.float() - ref.float()).pow(2).sum().item(), float(ref.numel())
    tot = reduce_metrics(dict(se=se, n=n), device=tokenizer.device if on_gpu else None)
    mse = tot['se'] / max(tot['n'], 1.0)
    return dict(mse=mse, psnr=float('inf') if mse == 0 else -10.0 * torch.log10(torch.tensor(mse)).item())


@torch.no_grad()
def discriminator_metrics_sharded(tokenizer, video: torch.Tensor) -> Dict[str, float]:
    """Global-batch hinge discriminator loss and mean logits over a sharded batch (the "discriminator / metrics reduction" of the
    brief): every rank runs forward(return_discr_loss=True, apply_gradient_penalty=False) on its clips — one random frame per clip
    through the image discriminator (reference magvit2_pytorch.py:1730-1750) — and contributes per-clip SUMS; one all_reduce of four
    scalars.  The hinge loss is a mean over clips, so sum / count reproduces the single-process value on the whole batch for the
    same frame picks; empty shards contribute zeros and still join the collective."""
    local = shard_batch(video)
    on_gpu = tokenizer.device.type == 'cuda'
    sums = dict(hinge=0.0, real=0.0, fake=0.0, n=0.0)
    if local.shape[0] > 0:
        dev_in = local.to(tokenizer.device) if on_gpu else local
        total, br = tokenizer(dev_in, return_discr_loss=True, apply_gradient_penalty=False)
        n = float(local.shape[0])
        sums = dict(hinge=float(br.discr_loss) * n, n=n, real=0.0, fake=0.0)
        lg = getattr(tokenizer, '_last_discr_logits', None)
        if lg is not None:
            sums.update(real=float(lg[0].sum()), fake=float(lg[1].sum()))
    tot = reduce_metrics(sums, device=tokenizer.device if on_gpu else None)
    n = max(tot['n'], 1.0)
    return dict(discr_loss=tot['hinge'] / n, real_logits_mean=tot['real'] / n, fake_logits_mean=tot['fake'] / n, clips=tot['n'])
