#!/usr/bin/env python
"""
bench.py — encode+decode throughput of the VideoTokenizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): video frames/sec, encode+decode, 128^2 x 17 frames, bf16.  A step is one
``forward(video, return_codes=True, return_recon=True)`` over this rank's batch of synthetic clips
(BASELINE.json configs[1]: (1,3,17,128,128), README layer spec, LFQ K=1024; weights random-init with
the reference's init families, seed 0, SqueezeExcite output layer re-randomised so residual branches
are live).  Inputs are resident in HBM before the timed region.  Clips are independent units: ranks
shard the batch with no data-path collective ("weak" scaling: per-GPU batch fixed).

Extra objects on the JSON line:
  roofline     — the dominant kernel (conv_kernel: every MFMA implicit-GEMM launch of the step).
                 achieved = sum of ALGORITHMIC flops (2*M*Cout*K, un-padded) of the conv launches of one
                 step / sum of their durations, each measured live with HIP events on the launch stream.
  cpu_baseline — the CPU oracle (oracle/restate.py, torch fp32) timed on this box's host cores on the
                 same workload (rank 0, N = 1 only), reported next to the GPU number.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get('RANK', '0')) == 0:
        print(f'[bench +{time.perf_counter() - _T0:6.1f}s] {msg}', file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch-per-gpu', type=int, default=1, help='clips per GPU per step (configs[1] = 1)')
    ap.add_argument('--frames', type=int, default=17)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--breakdown', default='', help='write the per-launch timing table to this json file')
    return ap.parse_args()


def per_launch_profile(tok, video, ops):
    """One eager pass with a HIP event pair around every launch (events recorded on the launch stream)."""
    plan, key = tok._plan(video.shape[0], video.shape[2])
    plan.video_in.copy_(video)
    lib = ops.lib
    stream = ops.current_stream()
    steps = plan.enc_steps + plan.q_steps + plan.dec_steps
    for st in steps:                          # warm
        st(stream)
    ops.synchronize()
    # every launch is issued REPS times back to back between its two events: the queue stays full, so the
    # event delta / REPS is the kernel's own duration (+ one inter-kernel gap) rather than the host's
    # per-call latency (which dominates a single eager launch of a < 30 us kernel).  Launches are
    # idempotent (no in-place ops), so the data flow is unchanged.
    REPS = 5
    evs = [lib.mv2_event_create() for _ in range(len(steps) + 1)]
    lib.mv2_event_record(evs[0], stream)
    for i, st in enumerate(steps):
        for _ in range(REPS):
            st(stream)
        lib.mv2_event_record(evs[i + 1], stream)
    lib.mv2_event_sync(evs[-1])
    rows = []
    for i, st in enumerate(steps):
        ms = lib.mv2_event_elapsed_ms(evs[i], evs[i + 1]) / REPS
        rows.append(dict(i=i, what=getattr(st, 'what', '?'), label=getattr(st, 'label', getattr(st, 'what', '?')),
                         ms=ms, flops=plan.step_flops.get(id(st), 0.0)))
    for e in evs:
        lib.mv2_event_destroy(e)
    return rows


def main():
    args = parse()
    from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer
    from magvit2_pytorch_amd import distributed as D
    from magvit2_pytorch_amd.spec import init_state_dict, make_cfg

    rank, world, local = D.init_from_env('nccl' if int(os.environ.get('WORLD_SIZE', '1')) > 1 else None)
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs a HIP device: the product path has no CPU fallback'
    dev = torch.device(f'cuda:{local}')
    torch.cuda.set_device(dev)

    kw = dict(image_size=128, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=0, randomize_se=True)
    tok = VideoTokenizer(**kw, seed=0)
    tok.load_state_dict(sd)
    tok = tok.to(dev)
    tok.compute_dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    tok.use_hip_graph = not args.no_graph
    B, T = args.batch_per_gpu, args.frames
    video = torch.randn(B, 3, T, 128, 128, generator=torch.Generator().manual_seed(1 + rank)).to(dev)

    log(f'model + input ready on {dev} ({args.dtype}, B={B}, T={T})')

    def step():
        return tok(video, return_codes=True, return_recon=True)

    for _ in range(max(args.warmup, 3)):            # >= 3: eager run, graph capture, first replay
        step()
    # untimed spin-up: a fresh box needs a few hundred ms of load before the GPU holds its working clock
    # (observed: a 20-step timed loop right after 3 warm-up steps ran 17 % slower than the same kernels a
    # moment later).  Not part of the W warm-up steps' semantics, never timed.
    torch.cuda.synchronize(dev)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < (0.0 if os.environ.get('MV2_BENCH_NO_SPIN') else 0.5):
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
    log('warm-up done (eager, graph capture, replay, 0.5 s clock spin-up)')
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codes, recon = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
        # the one exchange the path has: gather ids for the global batch (off the timed path)
        allc = D.gather_codes(codes, B * world)
        assert allc.shape[0] == B * world
        # every rank leaves the group together; rank 0 alone goes on to the per-launch profile and the JSON line
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    assert torch.isfinite(recon).all()

    if rank != 0:
        return
    frames_total = world * B * T * args.steps
    value = frames_total / el
    log(f'timed region: {el / args.steps * 1e3:.3f} ms/step -> {value:.1f} frames/s')

    # ---- roofline of the dominant kernel (rank 0)
    ops = tok._backend()
    rows = per_launch_profile(tok, video, ops)
    log('per-launch event profile done')
    conv = [r for r in rows if r['what'] == 'mv2_conv']
    conv_ms = sum(r['ms'] for r in conv)
    conv_flops = sum(r['flops'] for r in conv)
    all_ms = sum(r['ms'] for r in rows)
    peak = 2500.0 if args.dtype == 'bf16' else 157.3
    # dominant kernel: conv_halo_kernel = the ResidualUnit 3x3 causal convs (81 % of the path's FLOPs)
    conv = [r for r in rows if r['what'] in ('mv2_conv', 'mv2_conv_pw')]
    k3 = [r for r in conv if '.k3' in r['label']]      # incl. the fused '.k3+k1' launches (their FLOPs include the 1x1x1)
    k3_ms = sum(r['ms'] for r in k3)
    k3_flops = sum(r['flops'] for r in k3)
    achieved = k3_flops / (k3_ms * 1e-3) / 1e12 if k3_ms > 0 else 0.0
    all_conv = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # HBM traffic of the dominant kernel: from the committed rocprofv3 PMC passes (own --pmc runs, FETCH_SIZE x2
    # correction for wide coalesced reads on gfx950, see profiles/README.md) - not collectable inside this process
    traffic = None
    tf = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
    if os.path.exists(tf) and args.dtype == 'bf16' and B == 1:
        try:
            traffic = json.load(open(tf)).get('conv_halo_kernel_hbm_bytes_per_launch')
        except Exception:
            traffic = None
    roofline = dict(bound='mfma', kernel='conv_halo_kernel (ResidualUnit CausalConv3d k3, reference magvit2_pytorch.py:937)',
                    achieved=round(achieved, 2), peak=peak, unit='TFLOP/s', frac=round(achieved / peak, 4), traffic=traffic,
                    launches=len(k3), flops_per_launch=k3_flops / max(len(k3), 1), avg_launch_ms=round(k3_ms / max(len(k3), 1), 5),
                    all_conv_launches=dict(launches=len(conv), flops_per_step=conv_flops, ms_per_step=round(conv_ms, 4),
                                           achieved=round(all_conv, 2), frac=round(all_conv / peak, 4)),
                    all_kernels_ms_per_step=round(all_ms, 4))
    if args.breakdown:
        os.makedirs(os.path.dirname(os.path.abspath(args.breakdown)), exist_ok=True)
        agg = {}
        for r in rows:
            a = agg.setdefault(r['label'] if r['what'] == 'mv2_conv' else r['what'], dict(ms=0.0, flops=0.0, n=0))
            a['ms'] += r['ms']; a['flops'] += r['flops']; a['n'] += 1
        json.dump(dict(rows=rows, by_label=agg), open(args.breakdown, 'w'), indent=1)

    # ---- CPU baseline: the oracle restatement on this box's host cores, same workload
    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1:
        from oracle.restate import OracleTokenizer
        # cores this process may actually run on (a container's cpu_count() can overstate it wildly,
        # and oversubscribed oneDNN threads make the baseline look absurdly slow)
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        ncores = max(1, min(ncores, int(os.environ.get('MV2_CPU_BASELINE_THREADS', '64'))))
        torch.set_num_threads(ncores)
        log(f'cpu baseline: oracle on {ncores} threads (cpu_count={os.cpu_count()})')
        orc = OracleTokenizer(sd, **cfg)
        vc = video[:1].cpu()
        t1 = time.perf_counter()
        oc, orc_recon = orc(vc, return_codes=True, return_recon=True)                 # warm-up
        ts = [time.perf_counter() - t1]
        log(f'cpu baseline: warm-up {ts[0]:.2f} s')
        budget = 30.0                                                              # bounded sample (seconds of CPU wall)
        while len(ts) < 4 and sum(ts) + ts[-1] < budget:
            t1 = time.perf_counter()
            oc, orc_recon = orc(vc, return_codes=True, return_recon=True)
            ts.append(time.perf_counter() - t1)
        timed = ts[1:] if len(ts) > 1 else ts
        med = sorted(timed)[len(timed) // 2]
        mism = (codes[:1].cpu() != oc).float().mean().item()
        derr = (recon[:1].cpu() - orc_recon).abs().mean().item()
        cpu_baseline = dict(value=round(T / med, 3), unit='frames/s', cores=ncores, kind='port',
                            sample=f'1 clip (1,3,{T},128,128) fp32, {len(timed)} timed run(s) after warm-up, median {med:.2f} s/clip '
                                   f'(bounded to ~{budget:.0f} s of CPU time)',
                            ids_mismatch_vs_gpu=round(mism, 5), recon_mean_abs_diff_vs_gpu=round(derr, 6))

    out = dict(
        metric='video frames/sec encode+decode, 128x128x17f', value=round(value, 2), unit='frames/s', n_gpus=world,
        steps=args.steps, warmup=max(args.warmup, 3), ms_per_step=round(el / args.steps * 1e3, 4), higher_is_better=True,
        scaling='weak', vs_baseline=None, dtype=args.dtype, data='synthetic',
        config=dict(workload=f'BASELINE configs[1]: ({B},3,{T},128,128) clip(s) per GPU, README layer spec, LFQ K=1024, '
                             f'forward(return_codes, return_recon), random-init weights seed 0',
                    global_batch=B * world, frames_per_clip=T, parallelism=f'dp{world} (clip sharding, no data-path collective)',
                    hip_graph=not args.no_graph),
        roofline=roofline, cpu_baseline=cpu_baseline)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
