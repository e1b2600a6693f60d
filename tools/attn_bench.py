"""Harness of mv2_attention's MFMA kernel (csrc/attn.hip attn_mfma_kernel) through the ctypes binding: BASELINE configs[3]'s space attention
(144 frames x 8 heads, 1024 queries x (1024 + 4 memory) keys, head dim 32, bf16) and the configs[1] shape (20 frames, 256 + 4 keys), each
against torch's fp32 softmax attention on the same bf16 operands (a subset of the frames at the large shape) and timed with HIP events.
GPU box only.   python tools/attn_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magvit2_pytorch_amd import pack as P                      # noqa: E402
from magvit2_pytorch_amd import ops_hip                         # noqa: E402
from magvit2_pytorch_amd.ops_hip import HipOps                  # noqa: E402

if os.environ.get('ATTN_LIB'):                                   # A/B: another build of the library (e.g. the previous kernel)
    ops_hip.LIB_PATH = os.path.abspath(os.environ['ATTN_LIB'])

ops = HipOps('cuda:0')
dev = ops.device
heads, d, n_mem = 8, 32, 4


def run(frames, n, check_frames, reps=20, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(3)
    qkv = (torch.randn(frames * n, 3 * heads * d, generator=g) * scale).to(torch.bfloat16).to(dev)
    mem = (torch.randn(2, heads, n_mem, d, generator=g) * scale).to(dev)
    out = torch.full((frames * n, heads * d), float('nan'), dtype=torch.bfloat16, device=dev)
    st = ops.current_stream()
    step = ops.attention(P.MV2_BF16, qkv, mem, out, frames, n, heads, d, n_mem, 0, 1, n, 1)      # seq = frame, tokens contiguous
    step(st); ops.synchronize()
    # reference on a subset of frames
    q, k, v = qkv[: check_frames * n].float().reshape(check_frames, n, 3, heads, d).permute(2, 0, 3, 1, 4)      # (3, f, h, n, d)
    mk = mem[0].to(torch.bfloat16).float()[None].expand(check_frames, -1, -1, -1)
    mv = mem[1].to(torch.bfloat16).float()[None].expand(check_frames, -1, -1, -1)
    k = torch.cat((mk, k), dim=2); v = torch.cat((mv, v), dim=2)
    att = (q @ k.transpose(-1, -2) * d ** -0.5).softmax(dim=-1)
    ref = (att @ v).permute(0, 2, 1, 3).reshape(check_frames * n, heads * d)
    got = out[: check_frames * n].float()
    err = (got - ref).abs()
    assert torch.isfinite(out).all(), 'non-finite output'
    lib = ops.lib
    e0, e1 = lib.mv2_event_create(), lib.mv2_event_create()
    for _ in range(3):
        step(st)
    lib.mv2_event_record(e0, st)
    for _ in range(reps):
        step(st)
    lib.mv2_event_record(e1, st); lib.mv2_event_sync(e1)
    us = lib.mv2_event_elapsed_ms(e0, e1) / reps * 1e3
    fl = 4.0 * n * (n + n_mem) * d * heads * frames
    print(f'frames {frames} n {n} scale {scale}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TF/s = {fl / us * 1e-6 / 2500:.3f} of 2.5 PF | max|d| {err.max().item():.3e} '
          f'mean|d| {err.mean().item():.3e} (|ref| mean {ref.abs().mean().item():.3e})', flush=True)
    return err.max().item()


if __name__ == '__main__':
    for sc in (1.0, 3.0):
        run(144, 1024, 4, scale=sc)
    run(20, 256, 20)
    run(3, 133, 3)            # ragged: one partial tile, one partial query block
    run(2, 64, 2)
