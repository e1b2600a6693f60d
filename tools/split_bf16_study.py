"""CPU study behind the id-exact 'split' mode: can the encoder's big contractions run as split-bf16 products on the
bf16 matrix pipe - x = x_hi + x_lo, w = w_hi + w_lo (each half bf16), x.w ~ x_hi.w_hi + x_lo.w_hi + x_hi.w_lo accumulated in fp32 - and
still reproduce the fp32 oracle's code ids under the rule of tests/test_parity_gpu.py::test_fp32_mode_vs_reference_golden (ids equal
wherever the oracle's pre-sign margin exceeds 1e-4; <= 0.5 % differing overall)?

TEST INFRASTRUCTURE (imports oracle/): the oracle restatement is run twice on a golden fixture's weights and clip, once as is and once with
the chosen contractions replaced by the three-term split form (products of bf16 values are exact in fp32, the CPU convolution accumulates in
fp32 like the MFMA does); nothing else changes.

    python tools/split_bf16_study.py [fixture ...]           default: mid_lfq cfg2_video128x17
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import restate as R                      # noqa: E402
from _util import Golden                             # noqa: E402


def halves(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def split3(fn, x, w, b, terms=3):
    """fn(x, w, None) as the split-bf16 sum: hi.hi + lo.hi + hi.lo (+ lo.lo with terms == 4), bias added once in fp32."""
    xh, xl = halves(x)
    wh, wl = halves(w)
    y = fn(xh, wh, None) + (fn(xl, wh, None) + fn(xh, wl, None))
    if terms == 4:
        y = y + fn(xl, wl, None)
    return y if b is None else y + b.reshape(1, -1, *([1] * (y.ndim - 2)))


def run(name, level, terms=3):
    g = Golden(name)
    sd = g.weights()
    v = g.video()
    orc = R.OracleTokenizer(sd, **g.cfg)
    orc.taps = {}
    t0 = time.time()
    ref_codes = orc(v, return_codes=True)
    t_ref = time.time() - t0
    pre_ref = orc.taps['prequant'].clone()
    assert torch.equal(ref_codes, g.codes), 'the restatement does not reproduce the golden ids on this box'

    orig_cc, orig_conv3d, orig_conv2d, orig_conv1d = R.causal_conv3d, F.conv3d, F.conv2d, F.conv1d
    state = {'in_res': False}

    def cc_split(x, w, b, stride=1, pad_mode='constant'):
        # level 1: the ResidualUnit's k x 3 x 3 convolutions only (C -> C, kernel > 1); level 2: + conv_in / conv_out-type causal convs
        is_res = w.shape[0] == w.shape[1] and w.shape[-1] == 3
        if (level >= 1 and is_res) or level >= 2:
            return split3(lambda a, ww, bb: orig_cc(a, ww, bb, stride, pad_mode), x, w, b, terms)
        return orig_cc(x, w, b, stride, pad_mode)

    def conv_nd_split(orig):
        def f(x, w, b=None, *a, **k):
            if level >= 3 and x.dtype == torch.float32:      # level 3: every convolution (1x1x1, down-sampling, up-sampling projections)
                return split3(lambda xx, ww, bb: orig(xx, ww, bb, *a, **k), x, w, b, terms)
            return orig(x, w, b, *a, **k)
        return f

    R.causal_conv3d = cc_split
    if level >= 3:
        F.conv2d, F.conv1d = conv_nd_split(orig_conv2d), conv_nd_split(orig_conv1d)
        # F.conv3d is what causal_conv3d itself calls: patch only the module-level uses inside restate (1x1x1 convs) through a flag
        R.F = type('Fp', (), {})()
        for k in dir(F):
            if not k.startswith('__'):
                setattr(R.F, k, getattr(F, k))
        R.F.conv3d = conv_nd_split(orig_conv3d)
    try:
        orc2 = R.OracleTokenizer(sd, **g.cfg)
        orc2.taps = {}
        t0 = time.time()
        codes = orc2(v, return_codes=True)
        t_split = time.time() - t0
        pre = orc2.taps['prequant']
    finally:
        R.causal_conv3d = orig_cc
        F.conv2d, F.conv1d = orig_conv2d, orig_conv1d
        R.F = F
    diff = codes != ref_codes
    frac = diff.float().mean().item()
    if not g.cfg['use_fsq']:
        safe = pre_ref.abs().amin(dim=-1) > 1e-4
        safe = safe.reshape(diff.shape) if safe.shape != diff.shape else safe
        bad = int((diff & safe).sum())
        unsafe = 1.0 - safe.float().mean().item()
    else:
        bad, unsafe = -1, float('nan')
    d = (pre - pre_ref).abs()
    print(f'[{name}] level {level} terms {terms}: ids mismatch {frac:.5%} ({int(diff.sum())} of {diff.numel()}), outside the 1e-4 margin band: {bad}; '
          f'tokens inside the band {unsafe:.4%}; pre-quantizer |d| max {d.max().item():.3e} mean {d.mean().item():.3e} (|z| mean {pre_ref.abs().mean().item():.3e}); '
          f'cpu {t_ref:.1f}s / {t_split:.1f}s', flush=True)
    return frac, bad


if __name__ == '__main__':
    torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '8')))
    names = sys.argv[1:] or ['mid_lfq', 'cfg2_video128x17']
    for n in names:
        for level in (1, 2, 3):
            run(n, level)
