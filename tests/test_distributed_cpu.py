"""N > 1 path on CPU: two gloo processes shard a batch of clips (no data-path collective), tokenize
their shards and all_gather the ids / all_reduce the metric sums — must equal the single-process
result.  (On the GPU box the same code runs over RCCL: backend 'nccl'.)  The per-rank tokenizer here
executes through the CPU emulation of the kernel semantics, so this exercises sharding + gathering,
not the kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer
from magvit2_pytorch_amd import distributed as D
from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
from oracle.ndhwc_ops import CpuOps

KW = dict(image_size=32, init_dim=8, max_dim=32, codebook_size=256, layers=README_LAYERS)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_tok():
    cfg = make_cfg(**KW)
    tok = VideoTokenizer(**KW, seed=0)
    tok.load_state_dict(init_state_dict(cfg, seed=0, randomize_se=True))
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    return tok


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    D.init_from_env('gloo')
    tok = _make_tok()
    video = torch.randn(total, 3, 5, 32, 32, generator=torch.Generator().manual_seed(3))     # same global batch on every rank
    codes = D.tokenize_sharded(tok, video, gather=True)
    metrics = D.reconstruction_metrics_sharded(tok, video)
    lo, hi = D.shard_bounds(total, rank, world)
    q.put((rank, codes.numpy(), metrics, (lo, hi)))     # by value: a torch tensor in an mp.Queue is a shared-memory handle that dies with this process
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('total', [4, 3, 1])      # 3: ragged split (2 + 1); 1: rank 1 holds an EMPTY shard and still joins the collectives
def test_two_rank_sharding_matches_single_process(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tok = _make_tok()
    video = torch.randn(total, 3, 5, 32, 32, generator=torch.Generator().manual_seed(3))
    ref_codes = tok.tokenize(video)
    recon = tok(video, return_recon=True)
    ref_mse = (recon - video).pow(2).mean().item()
    bounds = sorted(r[3] for r in res)
    assert bounds[0][0] == 0 and bounds[-1][1] == total and bounds[0][1] == bounds[1][0]
    for rank, codes, metrics, _ in res:
        codes = torch.from_numpy(codes)
        assert codes.shape == ref_codes.shape and torch.equal(codes, ref_codes)
        assert abs(metrics['mse'] - ref_mse) < 1e-6 * max(1.0, ref_mse)


def test_shard_bounds_partition():
    for n in range(0, 20):
        for w in (1, 2, 3, 8):
            cuts = [D.shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


# ------------------------------------------------------------------------------- training-mode LFQ aux loss across ranks
def _loss_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    D.init_from_env('gloo')
    tok = VideoTokenizer(**KW, use_gan=False, perceptual_loss_weight=0., seed=0)
    tok.load_state_dict(init_state_dict(make_cfg(**KW), seed=0, randomize_se=True))
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    tok.train()
    video = torch.randn(4, 3, 5, 32, 32, generator=torch.Generator().manual_seed(3))
    lo, hi = D.shard_bounds(4, rank, world)
    total, br = tok(video[lo:hi], return_loss=True)
    qb = br.quantizer_loss_breakdown
    q.put((rank, float(total), float(br.recon_loss), float(br.lfq_aux_loss), float(qb.per_sample_entropy), float(qb.batch_entropy),
           float(qb.commitment)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_lfq_aux_loss_is_the_global_batch_loss():
    """The token-summed code distribution is all-reduced: every rank reports the aux loss of the GLOBAL batch (what the
    third-party quantizer's all-reduced mean distribution gives for equal shards), its own shard's reconstruction loss."""
    from oracle.restate import OracleTokenizer
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loss_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd = init_state_dict(make_cfg(**KW), seed=0, randomize_se=True)
    video = torch.randn(4, 3, 5, 32, 32, generator=torch.Generator().manual_seed(3))
    orc = OracleTokenizer(sd, **make_cfg(**KW))
    _, whole = orc.forward_loss(video, training=True)
    for rank, total, recon, aux, ps, bt, cm in res:
        _, part = orc.forward_loss(video[2 * rank: 2 * rank + 2], training=True)
        assert abs(aux - float(whole['aux'])) <= 1e-5
        assert abs(ps - float(whole['per_sample_entropy'])) <= 1e-5 and abs(bt - float(whole['batch_entropy'])) <= 1e-5
        assert abs(cm - float(whole['commitment'])) <= 1e-5
        assert abs(recon - float(part['recon_loss'])) <= 1e-5
        assert abs(total - (recon + aux)) <= 1e-6
    assert res[0][3] == res[1][3]                                    # identical on both ranks


# ------------------------------------------------------------------------------- discriminator / metrics reduction across ranks
GAN_KW = dict(image_size=16, init_dim=8, max_dim=16, codebook_size=64, layers=('residual', 'compress_space', 'compress_time', 'residual'))


def _gan_tok():
    tok = VideoTokenizer(**GAN_KW, use_gan=True, perceptual_loss_weight=0., seed=0)
    tok.load_state_dict(init_state_dict(make_cfg(**GAN_KW), seed=0, randomize_se=True))
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    return tok


def _discr_worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    D.init_from_env('gloo')
    tok = _gan_tok()
    video = torch.randn(total, 3, 5, 16, 16, generator=torch.Generator().manual_seed(3))
    torch.manual_seed(100 + rank)                       # the frame pick draws from the CPU default generator (reference :1736)
    q.put((rank, D.discriminator_metrics_sharded(tok, video)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('total', [3, 1])      # ragged (2 + 1) and one empty shard
def test_two_rank_discriminator_metrics(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_discr_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expected: the clip-weighted mean of the per-shard losses, each with its rank's frame picks
    tok = _gan_tok()
    video = torch.randn(total, 3, 5, 16, 16, generator=torch.Generator().manual_seed(3))
    hinge = real = fake = 0.
    for r in range(2):
        lo, hi = D.shard_bounds(total, r, 2)
        if hi > lo:
            torch.manual_seed(100 + r)
            _, br = tok(video[lo:hi], return_discr_loss=True, apply_gradient_penalty=False)
            hinge += float(br.discr_loss) * (hi - lo)
            real += float(tok._last_discr_logits[0].sum())
            fake += float(tok._last_discr_logits[1].sum())
    for r in range(2):
        m = res[r]
        assert m['clips'] == total and abs(m['discr_loss'] - hinge / total) <= 1e-6
        assert abs(m['real_logits_mean'] - real / total) <= 1e-6 and abs(m['fake_logits_mean'] - fake / total) <= 1e-6
    assert res[0] == res[1]
