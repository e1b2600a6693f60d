#!/usr/bin/env python
"""VERDICT r3 item 6 — Winograd F(2x2, 3x3) per time tap for the bf16 k x 3 x 3 ResidualUnit convolutions: the NUMERICAL half of the decision,
measured on the CPU (the structural half - accumulator and operand budgets on gfx950 - is argued in profiles/r04_findings.md).

One ResidualUnit convolution (CausalConv3d 3x3x3, C -> C, reference magvit2_pytorch.py:937) on bf16-rounded activations and weights:
  exact    : fp64 accumulation of the bf16 operands (what both forms approximate)
  direct   : bf16 operands, fp32 accumulation, one rounding of the output to bf16          (the product kernel, mv2_conv3)
  winograd : per time tap U = G g G^T and V = B^T d B computed in fp32 and ROUNDED TO bf16 (they are the MFMA operands), products
             accumulated in fp32 over channels and time taps, output transform A^T M A in fp32, one rounding of the output to bf16
Reported per channel width: mean |error| against `exact` before and after the output rounding, as multiples of the direct form's.
The bf16 mode's drift gates (tests/test_parity_gpu.py) allow 1.10 x on the smooth halves of the path and the verdict's bar was
"<= 1.05 x the yardstick means": a per-layer error that doubles cannot pass either."""
import math
import sys
import torch

torch.manual_seed(0)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


def run(C, H=16, W=16, T=3):
    x = bf(torch.randn(T, C, H + 2, W + 2, dtype=torch.float32) * 0.6)          # zero-free halo: interior tiles only (edges are not the question)
    w = bf(torch.randn(C, C, 3, 3, 3, dtype=torch.float32) * (1.0 / math.sqrt(27 * C / 3)))   # the reference's init scale, O(1) outputs
    xd, wd = x.double(), w.double()
    # output frame = last frame, all three time taps valid
    exact = sum(torch.nn.functional.conv2d(xd[t][None], wd[:, :, t]) for t in range(T))[0]                      # (C, H, W)
    direct32 = sum(torch.nn.functional.conv2d(x[t][None], w[:, :, t]) for t in range(T))[0]                      # fp32 accumulation of exact bf16 products
    # Winograd: tiles of 2 x 2 outputs
    nh, nw = H // 2, W // 2
    M = torch.zeros(C, nh, nw, 4, 4, dtype=torch.float32)
    for t in range(T):
        U = torch.einsum('ij,ocjk,lk->ocil', G, wd[:, :, t], G)                                                  # (Co, Ci, 4, 4) in fp64 ...
        U = bf(U.float())                                                                                         # ... stored as bf16 MFMA operands
        patches = xd[t].unfold(1, 4, 2).unfold(2, 4, 2)                                                           # (Ci, nh, nw, 4, 4)
        V = torch.einsum('ij,chwjk,lk->chwil', BT, patches, BT)
        V = bf(V.float())
        M += torch.einsum('ocil,chwil->ohwil', U, V)                                                             # fp32 accumulation (the matrix pipe)
    Y = torch.einsum('ij,ohwjk,lk->ohwil', AT.float(), M, AT.float())                                             # (Co, nh, nw, 2, 2)
    wino32 = Y.permute(0, 1, 3, 2, 4).reshape(C, H, W)
    rms = exact.pow(2).mean().sqrt().item()
    e = lambda a: (a.double() - exact).abs().mean().item()
    return dict(C=C, rms=rms, direct_pre=e(direct32), wino_pre=e(wino32), direct_bf16=e(bf(direct32)), wino_bf16=e(bf(wino32)))


if __name__ == '__main__':
    print('C    out rms   direct fp32-acc   winograd fp32-acc   | after bf16 rounding: direct   winograd   ratio')
    for C in (64, 128, 256):
        r = run(C)
        print(f'{r["C"]:<4d} {r["rms"]:.3f}     {r["direct_pre"]:.3e}         {r["wino_pre"]:.3e}           | {r["direct_bf16"]:.3e}   {r["wino_bf16"]:.3e}   {r["wino_bf16"] / r["direct_bf16"]:.2f} x')
        sys.stdout.flush()
