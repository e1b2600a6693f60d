"""Harness of mv2_pw_se (csrc/pwse.hip: ResidualUnit 1x1x1 conv + ELU + SqueezeExcite pooling records) on the GPU box: the C / 32-wave form
against the four-wave form of round 4 (experiments library, MV2_PW_SE_NW=4) - rows must be bit-identical, records equal to summation
order - and HIP-event timings of both on the launches of configs[1] / configs[2].   usage: python tools/pwse_bench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                                             # noqa: E402
import magvit2_pytorch_amd.ops_hip as O                  # noqa: E402
from magvit2_pytorch_amd import pack as P                # noqa: E402

O.LIB_PATH = O.LIB_PATH.replace('libmv2_hip.so', 'libmv2_hip_exp.so')
ops = O.HipOps('cuda:0')
lib, dev, BF, DT = ops.lib, ops.device, torch.bfloat16, P.MV2_BF16


def time_step(step, reps=30):
    st = ops.current_stream()
    step(st); ops.synchronize()
    e0, e1 = lib.mv2_event_create(), lib.mv2_event_create()
    lib.mv2_event_record(e0, st)
    for _ in range(reps):
        step(st)
    lib.mv2_event_record(e1, st)
    lib.mv2_event_sync(e1)
    ms = lib.mv2_event_elapsed_ms(e0, e1) / reps
    lib.mv2_event_destroy(e0); lib.mv2_event_destroy(e1)
    return ms * 1e3


def case(name, C, rows, Pn):
    g = torch.Generator().manual_seed(C + rows)
    w = torch.randn(C, C, generator=g) * C ** -0.5
    b = torch.randn(C, generator=g) * 0.1
    pc = P.pack_pointwise(w, b, BF).to(dev)
    frag = P.to_frag_major(pc.wp)
    wk = (torch.randn(C, generator=g) * C ** -0.5).to(dev)
    x = (torch.randn(rows, C, generator=g)).to(BF).to(dev).reshape(-1)
    bp = ops.pw_se_rows(DT, rows, Pn, C)
    assert bp
    nrec = rows // bp
    res = {}
    for nw in ('4', 'wide'):
        os.environ['MV2_PW_SE_NW'] = '4' if nw == '4' else '0'
        out = torch.full((rows * C,), float('nan'), dtype=BF, device=dev)
        part = torch.full((nrec * (C + 2),), float('nan'), dtype=torch.float32, device=dev)
        step = ops.pw_se(DT, x, frag, pc.bias, wk, 0.05, out, part, rows, Pn, C)
        step(ops.current_stream()); ops.synchronize()
        res[nw] = (out, part, time_step(step))
    (o4, p4, t4), (ow, pw_, tw) = res['4'], res['wide']
    same = torch.equal(o4, ow)
    rel = ((p4 - pw_).abs() / (p4.abs() + 1e-3)).max().item()
    ok = same and torch.isfinite(pw_).all().item() and rel < 2e-3      # records: fp32 sums over rows in another grouping
    print(f'{name:30s} C={C} rows={rows:6d} bp={bp}  rows identical: {same}  records max rel diff {rel:.2e}  four waves {t4:6.1f} us, C/32 waves {tw:6.1f} us' + ('  ok' if ok else '  FAIL'), flush=True)
    return ok


def main():
    ok = True
    for nm, C, rows, Pn in [('B1 C=128 64x64 T20', 128, 81920, 4096), ('B1 C=256 32x32 T20', 256, 20480, 1024), ('B1 C=512 16x16 T20', 512, 5120, 256),
                            ('B1 C=512 T10', 512, 2560, 256), ('B1 C=512 T5', 512, 1280, 256), ('ragged frames C=256 P=96', 256, 96 * 7, 96),
                            ('B4 C=256', 256, 81920, 1024), ('B4 C=512 T20', 512, 20480, 256), ('B4 C=512 T10', 512, 10240, 256), ('B4 C=512 T5', 512, 5120, 256)]:
        ok &= case(nm, C, rows, Pn)
    for bpv in ('32', '64'):      # the row-tile rule (pw_se_bp): 64-row tiles below 512 of them?
        os.environ['MV2_PW_SE_BP'] = bpv
        for nm, C, rows, Pn in [('B4 C=512 T20', 512, 20480, 256), ('B4 C=512 T10', 512, 10240, 256), ('B1 C=256 T20', 256, 20480, 1024), ('B1 C=512 T20', 512, 5120, 256)]:
            ok &= case(f'bp={bpv} {nm}', C, rows, Pn)
    os.environ.pop('MV2_PW_SE_BP', None)
    print('ALL OK' if ok else 'SOME FAILED')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
