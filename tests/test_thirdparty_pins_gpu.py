"""GPU checks of the three THIRD-PARTY pieces of the path (vector-quantize-pytorch LFQ / FSQ, taylor-series-linear-attention)
against DERIVED vectors — closed-form expectations worked out from the published algorithms, NOT outputs of the upstream
packages (they are absent from /root/reference and from this image: parity for these modules stays "unpinned", see
oracle/thirdparty.py).  What these tests pin is that the HIP kernels implement the algebra the papers state:

  * LFQ (Yu et al. 2023, "Language Model Beats Diffusion", lookup-free quantisation): code_i = sign bit i of the id, MSB first
    (mask = 2^(d-1-i)), codes in {-1, +1}; exhaustive ids -> codes -> ids identity over all 2^10 = 1024 ids.
  * FSQ (Mentzer et al. 2023, "Finite Scalar Quantization"): levels [8,5,5,5] -> 1000 codes, basis = cumprod([1] + L[:-1]) =
    [1, 8, 40, 200], per-dim values (l - floor(L/2)) / floor(L/2); exhaustive ids -> codes -> ids identity over all 1000 ids
    (through the exact pre-image of the bounding function), plus the paper's level table and a worked id.
  * Taylor-series linear attention (2nd-order expansion of exp(q.k)): phi(x) = [1, x, vec(x (x) x)/sqrt(2)],
    out = phi(q d^-1/2) . sum_n phi(k_n)^T v_n / max(phi(q d^-1/2) . sum_n phi(k_n), 1e-5); hand-computed for 4 tokens.
Everything goes through the C ABI (libmv2_hip.so via ctypes)."""
import math
import os
import socket

import numpy as np
import pytest
import torch

from magvit2_pytorch_amd import pack as P

pytestmark = pytest.mark.gpu


def _ops():
    from magvit2_pytorch_amd.ops_hip import HipOps
    return HipOps('cuda:0')


def test_lfq_all_1024_ids_round_trip_and_bit_order():
    ops = _ops()
    d, C, n = 10, 32, 1024
    dev = ops.device
    ids = torch.arange(n, dtype=torch.int64, device=dev).reshape(n, 1)
    sel = torch.zeros(d, C, device=dev)
    sel[torch.arange(d), torch.arange(d)] = 1.0                      # code dim i <-> channel i
    zero_c, zero_d = torch.zeros(C, device=dev), torch.zeros(d, device=dev)
    out = torch.empty(n * C, dtype=torch.float32, device=dev)
    st = ops.current_stream()
    ops.lfq_codes(P.MV2_F32, ids, sel, zero_c, out, n, C, d, 1, 1.0)(st)
    ops.synchronize()
    codes = out.reshape(n, C)[:, :d].cpu()
    i = torch.arange(n)[:, None]
    expect = (((i >> (d - 1 - torch.arange(d))[None, :]) & 1) * 2 - 1).float()      # MSB first, {-1, +1}
    assert torch.equal(codes, expect)
    assert torch.equal(out.reshape(n, C)[:, d:].cpu(), torch.zeros(n, C - d))
    # back through the forward kernel (project_in selects the same channels, soft clamp 10 as in the reference's ctor)
    q = torch.empty(n * C, dtype=torch.float32, device=dev)
    ids2 = torch.empty(n, 1, dtype=torch.int64, device=dev)
    ops.lfq(P.MV2_F32, out, sel, zero_d, sel, zero_c, q, ids2, None, n, C, d, 1, 10.0, 1.0)(st)
    ops.synchronize()
    assert torch.equal(ids2.cpu(), ids.cpu())
    assert torch.equal(q.reshape(n, C)[:, :d].cpu(), expect)         # eval path: quantized = codes


def _fsq_consts(levels, dev):
    L = torch.tensor(levels, dtype=torch.int32)
    half_l = (L - 1) * (1 + 1e-3) / 2
    offset = torch.where(L % 2 == 0, 0.5, 0.0)
    return dict(levels=L.to(dev), basis=torch.cumprod(torch.tensor([1] + levels[:-1]), dim=0, dtype=torch.int32).to(dev),
                half_l=half_l.float().to(dev), offset=offset.float().to(dev), shift=(offset / half_l).atanh().float().to(dev),
                half_w=(L // 2).float().to(dev)), half_l.double(), offset.double()


def test_fsq_all_1000_ids_round_trip_and_level_table():
    ops = _ops()
    levels = [8, 5, 5, 5]
    d, C, n = 4, 32, 1000
    dev = ops.device
    consts, half_l, offset = _fsq_consts(levels, dev)
    assert consts['basis'].cpu().tolist() == [1, 8, 40, 200]                      # cumprod([1] + L[:-1])
    ids = torch.arange(n, dtype=torch.int32, device=dev).reshape(n, 1)
    sel = torch.zeros(d, C, device=dev)
    sel[torch.arange(d), torch.arange(d)] = 1.0
    zero_c, zero_d = torch.zeros(C, device=dev), torch.zeros(d, device=dev)
    out = torch.empty(n * C, dtype=torch.float32, device=dev)
    st = ops.current_stream()
    ops.fsq_codes(P.MV2_F32, ids, sel, zero_c, consts, out, n, C, d, 1)(st)
    ops.synchronize()
    codes = out.reshape(n, C)[:, :d].cpu().double()
    # level table of the paper: digit l_i = (id // basis_i) % L_i, value (l_i - floor(L_i/2)) / floor(L_i/2)
    L = torch.tensor(levels)
    hw = (L // 2).double()
    digits = (torch.arange(n)[:, None] // torch.tensor([1, 8, 40, 200])[None, :]) % L[None, :]
    expect = (digits.double() - hw) / hw
    assert torch.equal(codes, expect)
    assert sorted(set(codes[:, 0].tolist())) == [-1.0, -0.75, -0.5, -0.25, 0.0, 0.25, 0.5, 0.75]      # L = 8
    assert sorted(set(codes[:, 1].tolist())) == [-1.0, -0.5, 0.0, 0.5, 1.0]                            # L = 5
    assert codes[177].tolist() == [-0.75, 0.0, 1.0, -1.0]           # 177 = 1 + 2*8 + 4*40 + 0*200
    # forward kernel on the exact pre-image of every code: z = atanh((level + offset) / half_l) - shift
    lv = expect * hw                                                 # integer levels (centred)
    z = torch.atanh((lv + offset[None, :]) / half_l[None, :]) - torch.atanh(offset / half_l)[None, :]
    x = torch.zeros(n, C, dtype=torch.float32)
    x[:, :d] = z.float()
    q = torch.empty(n * C, dtype=torch.float32, device=dev)
    ids2 = torch.empty(n, 1, dtype=torch.int32, device=dev)
    ops.fsq(P.MV2_F32, x.to(dev).reshape(-1), sel, zero_d, sel, zero_c, consts, q, ids2, None, n, C, d, 1)(st)
    ops.synchronize()
    assert torch.equal(ids2.cpu(), ids.cpu())
    assert torch.equal(q.reshape(n, C)[:, :d].cpu().double(), expect)


@pytest.mark.parametrize('n,nchunk', [(4, 1), (4, 2), (133, 1), (133, 3)])
def test_taylor_linear_attention_hand_computed_four_tokens(n, nchunk):
    """n = 4: the hand-computable case; n = 133: a ragged frame (the whole-frame kernel's 32 token groups x 4 tokens per trip with a masked
    tail).  nchunk = 1 is the whole-frame state kernel (no partial states, no reduce), nchunk > 1 the chunked form that stays in the ABI."""
    ops = _ops()
    heads, dh, frames = 16, 8, 1
    g = torch.Generator().manual_seed(11)
    q = torch.randn(n, heads * dh, generator=g) * 0.7
    kv = torch.randn(n, 2 * heads * dh, generator=g) * 0.7
    dev = ops.device
    part = torch.empty(frames * nchunk * heads * 657, dtype=torch.float32, device=dev)
    kvsum = torch.empty(frames * heads * 657, dtype=torch.float32, device=dev)
    out = torch.empty(n * heads * dh, dtype=torch.float32, device=dev)
    st = ops.current_stream()
    ops.taylor_kv(P.MV2_F32, kv.to(dev).reshape(-1), part, frames, n, heads, nchunk)(st)
    ops.taylor_apply(P.MV2_F32, q.to(dev).reshape(-1), part, out, kvsum, frames, n, heads, nchunk)(st)
    ops.synchronize()
    got = out.reshape(n, heads, dh).cpu().double().numpy()

    def phi(x):                                                      # (dh,) -> (1 + dh + dh^2,)
        return np.concatenate(([1.0], x, np.outer(x, x).reshape(-1) / math.sqrt(2.0)))
    qn, kvn = q.double().numpy(), kv.double().numpy()
    exp = np.zeros((n, heads, dh))
    for h in range(heads):
        k = kvn[:, h * dh:(h + 1) * dh]
        v = kvn[:, heads * dh + h * dh: heads * dh + (h + 1) * dh]
        KV = sum(np.outer(phi(k[t]), v[t]) for t in range(n))       # (73, 8)
        Ks = sum(phi(k[t]) for t in range(n))                        # (73,)
        for t in range(n):
            f = phi(qn[t, h * dh:(h + 1) * dh] * dh ** -0.5)
            exp[t, h] = f @ KV / max(f @ Ks, 1e-5)
    assert np.abs(got - exp).max() <= (2e-5 if n <= 8 else 2e-4), np.abs(got - exp).max()
    # the same attention through the FUSED head-major tensor [frame][head][token][q 8 | k 8 | v 8] (mv2_taylor_kv_fused / _apply_fused: what
    # the single q | kv projection GEMM writes): same closed form, and bit-identical to the row-layout launches (same arithmetic, other addresses)
    kk, vv = kv[:, :heads * dh].reshape(n, heads, dh), kv[:, heads * dh:].reshape(n, heads, dh)
    qkv = torch.stack((q.reshape(n, heads, dh), kk, vv), dim=2).permute(1, 0, 2, 3).contiguous()          # (heads, n, 3, 8), one frame
    part2, out2 = torch.empty_like(part), torch.empty_like(out)
    ops.taylor_kv_fused(P.MV2_F32, qkv.to(dev).reshape(-1), part2, frames, n, heads, nchunk)(st)
    ops.taylor_apply_fused(P.MV2_F32, qkv.to(dev).reshape(-1), part2, out2, kvsum, frames, n, heads, nchunk)(st)
    ops.synchronize()
    assert torch.equal(out2, out) and torch.equal(part2, part)


# ------------------------------------------------------------------------------- training-mode LFQ losses and the gateloop scan: closed forms
def test_lfq_training_losses_hand_computed_three_tokens():
    """mv2_lfq_stats + mv2_lfq_losses against a case worked out by hand - Yu et al. 2023 section 3 (entropy penalty
    E[H(f(z))] - gamma H(E[f(z)]) over the implicit codebook {-1, +1}^d) with the upstream conventions the reference's call site relies on
    (M:1705; weights M:1067-1072): soft assignment prob_j = softmax_j(2 * inv_temperature * z . c_j), log clamped at 1e-5, commitment =
    mean squared distance to the chosen code.  d = 2 -> K = 4 codes, MSB first: c_0 = (-1, -1), c_1 = (-1, +1), c_2 = (+1, -1), c_3 = (+1, +1).
    Three tokens chosen so that the logits are the small integers (-1, -3, 3, 1), (2, -2, 2, -2) and (-4, 0, 0, 4) at inv_temperature 100."""
    ops = _ops()
    dev = ops.device
    z = np.array([[0.010, -0.005], [0.0, -0.010], [0.010, 0.010]], dtype=np.float64)
    codes = np.array([[-1, -1], [-1, 1], [1, -1], [1, 1]], dtype=np.float64)
    inv_t, gamma, w_ent, w_commit = 100.0, 1.0, 0.1, 0.25
    logits = 2.0 * inv_t * z @ codes.T
    assert np.allclose(logits, [[-1, -3, 3, 1], [2, -2, 2, -2], [-4, 0, 0, 4]])
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    prob = e / e.sum(axis=1, keepdims=True)
    ent = lambda p_: float(-(p_ * np.log(np.maximum(p_, 1e-5))).sum())
    h_sample = np.mean([ent(prob[i]) for i in range(3)])
    h_batch = ent(prob.mean(axis=0))
    q = np.where(z > 0, 1.0, -1.0)                                   # strict x > 0: the zero activation quantises to -1
    commit = float(((z - q) ** 2).mean())
    aux = (h_sample - gamma * h_batch) * w_ent + commit * w_commit
    pre = torch.tensor(z, dtype=torch.float32, device=dev).contiguous()
    sums = torch.zeros(4 + 2, dtype=torch.float32, device=dev)
    ws = torch.zeros(max(1, ops.lfq_stats_workspace_bytes(3, 2, 1) // 4), dtype=torch.float32, device=dev)
    losses = torch.zeros(4, dtype=torch.float32, device=dev)
    st = ops.current_stream()
    ops.lfq_stats(pre, 3, 2, 1, inv_t, 1.0, sums, ws)(st)
    ops.lfq_losses(sums, 3.0, 2, 1, gamma, w_ent, w_commit, losses)(st)
    ops.synchronize()
    got_sums, got = sums.cpu().double().numpy(), losses.cpu().double().numpy()
    assert np.abs(got_sums[:4] - prob.sum(axis=0)).max() <= 2e-6, (got_sums[:4], prob.sum(axis=0))
    assert abs(got_sums[4] - 3 * h_sample) <= 5e-6 and abs(got_sums[5] - ((z - q) ** 2).sum()) <= 5e-6
    assert np.abs(got - np.array([h_sample, h_batch, commit, aux])).max() <= 5e-6, (got, [h_sample, h_batch, commit, aux])
    # two codebooks of the same tokens: every mean is unchanged (entropies average over codebooks, the commitment over all code dims)
    pre2 = torch.tensor(np.concatenate([z, z], axis=1), dtype=torch.float32, device=dev).contiguous()
    sums2 = torch.zeros(2 * 4 + 2, dtype=torch.float32, device=dev)
    ws2 = torch.zeros(max(1, ops.lfq_stats_workspace_bytes(3, 2, 2) // 4), dtype=torch.float32, device=dev)
    ops.lfq_stats(pre2, 3, 2, 2, inv_t, 1.0, sums2, ws2)(st)
    ops.lfq_losses(sums2, 3.0, 2, 2, gamma, w_ent, w_commit, losses)(st)
    ops.synchronize()
    assert np.abs(losses.cpu().double().numpy() - np.array([h_sample, h_batch, commit, aux])).max() <= 5e-6


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gateloop_scan_hand_computed_four_frames(dt):
    """mv2_gateloop_scan against the recurrence written out by hand: per pixel and channel along time
    h_t = sigmoid(a_t) h_{t-1} + kv_t (h_{-1} = 0), out_t = q_t h_t + residual_t (gated linear recurrence of Katsch 2023, "GateLoop", in
    the real-valued SimpleGateLoopLayer form the reference imports at M:32 / 1221).  4 frames, 3 pixels, 8 channels, 2 clips: the scan
    must run along TIME (row stride = pixels) and restart at every clip."""
    ops = _ops()
    dev = ops.device
    B, T, Pn, Cc = 2, 4, 3, 8
    rng = np.random.RandomState(11)
    q, kv, a, res = (rng.uniform(-1, 1, size=(B, T, Pn, Cc)) for _ in range(4))
    if dt == torch.bfloat16:      # the kernel reads bf16 operands: the hand computation starts from the same rounded values
        rb = lambda x: torch.tensor(x).to(torch.bfloat16).double().numpy()
        q, kv, a, res = rb(q), rb(kv), rb(a), rb(res)
    exp = np.zeros((B, T, Pn, Cc))
    for b in range(B):
        for p_ in range(Pn):
            for c in range(Cc):
                h = 0.0
                for t in range(T):
                    h = h / (1.0 + math.exp(-a[b, t, p_, c])) + kv[b, t, p_, c]
                    exp[b, t, p_, c] = q[b, t, p_, c] * h + res[b, t, p_, c]
    # frame 0 has no history: out_0 = q_0 kv_0 + residual_0 whatever the gate says
    assert np.allclose(exp[:, 0], q[:, 0] * kv[:, 0] + res[:, 0])
    qkva = torch.tensor(np.concatenate([q, kv, a], axis=-1).reshape(B * T * Pn, 3 * Cc)).to(dt).to(dev).contiguous()
    rt = torch.tensor(res.reshape(B * T * Pn, Cc)).to(dt).to(dev).contiguous()
    out = torch.empty_like(rt)
    ops.gateloop_scan(P.dtype_code(dt), qkva, rt, out, B, T, Pn, Cc)(ops.current_stream())
    ops.synchronize()
    got = out.cpu().double().numpy().reshape(B, T, Pn, Cc)
    tol = 2e-6 if dt == torch.float32 else 2 ** -7 * np.abs(exp) + 1e-2
    assert (np.abs(got - exp) <= tol).all(), np.abs(got - exp).max()


# ------------------------------------------------------------------------------- RCCL: first contact when a multi-GPU box appears
def _nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    import torch.distributed as dist
    from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer
    from magvit2_pytorch_amd import distributed as D
    from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
    D.init_from_env('nccl')
    kw = dict(image_size=32, init_dim=16, max_dim=64, codebook_size=256, layers=README_LAYERS)
    tok = VideoTokenizer(**kw, seed=0)
    tok.load_state_dict(init_state_dict(make_cfg(**kw), seed=0, randomize_se=True))
    tok = tok.to(f'cuda:{rank}')
    tok.compute_dtype = torch.float32
    video = torch.randn(4, 3, 5, 32, 32, generator=torch.Generator().manual_seed(3))        # same global batch on every rank
    codes = D.tokenize_sharded(tok, video, gather=True)                                       # all_gather over RCCL
    metrics = D.reconstruction_metrics_sharded(tok, video)                                    # all_reduce over RCCL
    q.put((rank, codes.cpu().numpy(), metrics))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two visible HIP devices (RCCL smoke)')
def test_two_rank_rccl_smoke():
    """2 ranks, backend 'nccl' (= RCCL): sharded tokenize + ids all_gather + metric all_reduce must equal the 1-GPU result."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])
    assert abs(res[0][2]['mse'] - res[1][2]['mse']) < 1e-12
    from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer
    from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
    kw = dict(image_size=32, init_dim=16, max_dim=64, codebook_size=256, layers=README_LAYERS)
    tok = VideoTokenizer(**kw, seed=0)
    tok.load_state_dict(init_state_dict(make_cfg(**kw), seed=0, randomize_se=True))
    tok = tok.to('cuda:0')
    tok.compute_dtype = torch.float32
    video = torch.randn(4, 3, 5, 32, 32, generator=torch.Generator().manual_seed(3))
    single = tok.tokenize(video.cuda()).cpu().numpy()
    assert np.array_equal(res[0][1], single)
    # ... and bench.py's own N = 2 control path (respawn under torch.distributed.run, rendezvous, barriers, per-rank times, gather_codes,
    # JSON) on the two devices, so that the first SCALE run exercises code a GPU has executed once
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--config', '2',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, cwd=root)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['scaling'] == 'weak' and len(line['per_rank_ms_per_step']) == 2
