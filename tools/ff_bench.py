"""Harness of mv2_ff (csrc/ff.hip: FeedForward as one launch) on the GPU box: every shape against the two-launch form it replaces
(mv2_conv_rms with the GEGLU epilogue + mv2_conv with the residual: the validated path) element by element, small shapes also against the
CPU emulation of the kernel semantics (oracle/ndhwc_ops.py), then HIP-event timings of both forms.
usage: python tools/ff_bench.py [exp] [quick | sweep [prof]]      exp: load libmv2_hip_exp.so (its knobs: MV2_FF_S parts per row tile, MV2_FF_ROT chunk
rotation, MV2_FF_NW=8 the eight-wave C = 256 form, MV2_FF_ABL compile-time ablations 1 / 2 / 3 / 7 / 8 / 11 / 15, MV2_FF_PROF phase clocks)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                                             # noqa: E402
import magvit2_pytorch_amd.ops_hip as O                  # noqa: E402
from magvit2_pytorch_amd import pack as P                # noqa: E402

if 'exp' in sys.argv[1:]:
    O.LIB_PATH = O.LIB_PATH.replace('libmv2_hip.so', 'libmv2_hip_exp.so')
QUICK = 'quick' in sys.argv[1:]
ops = O.HipOps('cuda:0')
lib = ops.lib
dev = ops.device
BF = torch.bfloat16
DT = P.MV2_BF16


def time_steps(steps, reps=20):
    st = ops.current_stream()
    for s in steps:
        s(st)
    ops.synchronize()
    e0, e1 = lib.mv2_event_create(), lib.mv2_event_create()
    lib.mv2_event_record(e0, st)
    for _ in range(reps):
        for s in steps:
            s(st)
    lib.mv2_event_record(e1, st)
    lib.mv2_event_sync(e1)
    ms = lib.mv2_event_elapsed_ms(e0, e1) / reps
    lib.mv2_event_destroy(e0); lib.mv2_event_destroy(e1)
    return ms * 1e3


def case(name, C, rows, shift_hw=0, T=1, seed=0, check_cpu=False, time_it=True):
    g = torch.Generator().manual_seed(seed)
    inner = int(C * 4 * 2 / 3)
    w1 = torch.randn(2 * inner, C, generator=g) * C ** -0.5
    b1 = torch.randn(2 * inner, generator=g) * 0.1
    w2 = torch.randn(C, inner, generator=g) * inner ** -0.5
    b2 = torch.randn(C, generator=g) * 0.1
    gamma = (1 + 0.1 * torch.randn(C, generator=g)) * C ** 0.5
    pc1, hp = P.pack_geglu_in(w1, b1, BF, in_scale=gamma, token_shift=shift_hw > 0)
    pc2 = P.pack_pointwise(w2, b2, BF, k_pad=hp)
    pc1.to(dev); pc2.to(dev)
    f1, f2 = P.to_frag_major(pc1.wp), P.to_frag_major(pc2.wp)
    x = (torch.randn(rows, C, generator=g) * 1.3).to(BF).to(dev).reshape(-1)
    # ---- the two-launch form, as plan.Plan._feed_forward lowers it
    if shift_hw:
        frames = rows // shift_hw
        geo = dict(B=frames // T, Ti=T, Hi=1, Wi=shift_hw, To=T, Ho=1, Wo=shift_hw)
    else:
        geo = dict(B=1, Ti=1, Hi=1, Wi=rows, To=1, Ho=1, Wo=rows)

    def desc(pc, Ci, act, out_c, pt, check):
        return dict(dtype=DT, **geo, Ci=Ci, co=pc.co, co_pad=pc.co_pad, st=1, sh=1, sw=1, pt=pt, ph=0, pw=0, kt=pc.kt, kh=pc.kh, kw=pc.kw,
                    nsteps=pc.nsteps, seg_bytes=pc.seg_bytes, check_bounds=check, act=act, store=P.STORE_ROWS, rt=1, rh=1, rw=1, crop_t=0,
                    out_c=out_c, splitk=0)
    d1 = desc(pc1, C, P.ACT_GEGLU, pc1.co // 2, 1 if shift_hw else 0, 1 if shift_hw else 0)
    d2 = desc(pc2, hp, P.ACT_NONE, pc2.co, 0, 0)
    h = torch.empty(rows * hp, dtype=BF, device=dev)
    ref = torch.empty(rows * C, dtype=BF, device=dev)
    out = torch.full((rows * C,), float('nan'), dtype=BF, device=dev)
    segs1 = pc1.segs_tensor(geo['Hi'], geo['Wi'], C, dev)
    segs2 = pc2.segs_tensor(geo['Hi'], geo['Wi'], hp, dev)
    two = []
    if ops.conv_rms_ok(d1):
        two.append(ops.conv_rms(d1, x, x.numel() * 2, pc1.wp, segs1, pc1.bias, None, h))
    else:
        rs = torch.empty(rows, dtype=torch.float32, device=dev)
        two.append(ops.rms_rowscale(DT, x, rs, rows, C, shift_hw, T))
        wsb = ops.conv_workspace_bytes(d1)
        ws1 = torch.empty(max(1, wsb // 4), dtype=torch.float32, device=dev) if wsb else None
        two.append(ops.conv(d1, x, x.numel() * 2, pc1.wp, segs1, pc1.bias, rs, None, h, ws1))
    wsb = ops.conv_workspace_bytes(d2)
    ws2 = torch.empty(max(1, wsb // 4), dtype=torch.float32, device=dev) if wsb else None
    two.append(ops.conv(d2, h, h.numel() * 2, pc2.wp, segs2, pc2.bias, None, x, ref, ws2))
    # ---- one launch
    assert ops.ff_eligible(DT, rows, C, hp)
    wsb, cntb = ops.ff_sizes(rows, C, hp)
    ws = torch.empty(max(1, wsb // 4), dtype=torch.float32, device=dev) if wsb else None
    cnt = torch.zeros(max(1, cntb // 4), dtype=torch.int32, device=dev) if wsb else None
    one = [ops.ff(DT, x, f1, pc1.bias, f2, pc2.bias, out, rows, C, hp, shift_hw, T, ws, cnt)]
    st = ops.current_stream()
    for s in two:
        s(st)
    one[0](st)
    ops.synchronize()
    a, b = out.float(), ref.float()
    assert torch.isfinite(a).all(), f'{name}: non-finite output'
    diff = (a - b).abs()
    nz = int((diff > 0).sum())
    scale = b.abs().mean().item()
    line = f'{name:34s} C={C} rows={rows:6d} hp={hp} slab={wsb >> 10:6d}K  vs two-launch: {nz:7d} / {a.numel()} differ, max|d| {diff.max().item():.3e} (mean|ref| {scale:.3f})'
    ok = diff.max().item() <= 0.04 * max(1.0, b.abs().max().item()) and nz <= 0.02 * a.numel()      # 1-ulp bf16 flips from the RMS scale's summation order
    if cnt is not None:
        ok = ok and int(cnt.abs().sum()) == 0
        line += f' tickets zero: {int(cnt.abs().sum()) == 0}'
    # run-to-run determinism of the split form
    out2 = torch.empty_like(out)
    ops.ff(DT, x, f1, pc1.bias, f2, pc2.bias, out2, rows, C, hp, shift_hw, T, ws, cnt)(st)
    ops.synchronize()
    rr = torch.equal(out, out2)
    ok = ok and rr
    line += f' rerun identical: {rr}'
    if check_cpu:
        from oracle.ndhwc_ops import CpuOps
        cpu = CpuOps()
        oc = torch.empty(rows * C, dtype=BF)
        cpu.ff(DT, x.cpu(), f1.cpu(), pc1.bias.cpu(), f2.cpu(), pc2.bias.cpu(), oc, rows, C, hp, shift_hw, T, None, None)(0)
        dc = (out.cpu().float() - oc.float()).abs()
        line += f' | vs CPU emulation max|d| {dc.max().item():.3e} (>1 ulp: {int((dc > 0.02 * oc.float().abs().clamp(min=0.25)).sum())})'
        ok = ok and int((dc > 0.02 * oc.float().abs().clamp(min=0.25)).sum()) == 0
    if time_it:
        t2, t1 = time_steps(two), time_steps(one)
        fl = 2.0 * rows * (2 * inner * C + inner * C)
        line += f' | two launches {t2:7.1f} us, one {t1:7.1f} us ({fl / t1 / 1e6 / 2500:.3f} of 2.5 PF)'
    print(line + ('  ok' if ok else '  FAIL'), flush=True)
    return ok


def sweep():
    """experiments library: the knobs of mv2_ff (read by the library per call) on the launches of configs[1] / configs[2]"""
    shapes = [('C=256 B1', 256, 20480, 0, 1), ('C=512 B1', 512, 5120, 0, 1), ('C=512 T5 shift B1', 512, 1280, 256, 5),
              ('C=256 B4', 256, 81920, 0, 1), ('C=512 B4', 512, 20480, 0, 1)]
    settings = [dict(), dict(MV2_FF_ROT='0'), dict(MV2_FF_NW='8'), dict(MV2_FF_NW='8', MV2_FF_S='2'), dict(MV2_FF_S='1'), dict(MV2_FF_S='2'), dict(MV2_FF_S='3'), dict(MV2_FF_S='4'),
                dict(MV2_FF_S='6'), dict(MV2_FF_ABL='1'), dict(MV2_FF_ABL='2'), dict(MV2_FF_ABL='3'),
                dict(MV2_FF_ABL='3', MV2_FF_S='1'), dict(MV2_FF_ABL='7', MV2_FF_S='1'), dict(MV2_FF_ABL='8', MV2_FF_S='1'), dict(MV2_FF_ABL='11', MV2_FF_S='1'),
                dict(MV2_FF_ABL='15', MV2_FF_S='1')]
    knobs = ('MV2_FF_ROT', 'MV2_FF_D', 'MV2_FF_S', 'MV2_FF_ABL', 'MV2_FF_PROF', 'MV2_FF_NW')
    if 'prof' in sys.argv[1:]:      # phase clocks of the kernel (the library prints one line per launch: the last one of a case is warm)
        settings = [dict(MV2_FF_PROF='1'), dict(MV2_FF_PROF='1', MV2_FF_S='1'), 
                    dict(MV2_FF_PROF='1', MV2_FF_ABL='2', MV2_FF_S='1'), dict(MV2_FF_PROF='1', MV2_FF_ABL='3', MV2_FF_S='1')]
    for name, C, rows, sh, T in shapes:
        for st in settings:
            for k in knobs:
                os.environ.pop(k, None)
            os.environ.update(st)
            try:
                case(f'{name} {st}', C, rows, shift_hw=sh, T=T)
            except Exception as e:
                print(f'{name} {st}: {e!r}'[:200], flush=True)
    for k in knobs:
        os.environ.pop(k, None)


def main():
    if 'sweep' in sys.argv[1:]:
        sweep()
        return 0
    ok = True
    # correctness first: small / ragged / shifted shapes, also against the CPU emulation
    ok &= case('small C=256', 256, 192, check_cpu=True, time_it=False)
    ok &= case('ragged rows C=256', 256, 1000, check_cpu=True, time_it=False)
    ok &= case('small C=512', 512, 128, check_cpu=True, time_it=False)
    ok &= case('ragged rows C=512', 512, 333, check_cpu=True, time_it=False)
    ok &= case('token shift C=512 T=5 hw=64', 512, 5 * 64 * 2, shift_hw=64, T=5, check_cpu=True, time_it=False)
    ok &= case('token shift C=256 T=3 hw=32', 256, 3 * 32 * 4, shift_hw=32, T=3, check_cpu=True, time_it=False)
    if not ok:
        print('SOME FAILED (small shapes): not timing')
        return 1
    # the launches of configs[1] (B = 1) and of configs[2]'s per-GPU shape (B = 4)
    ok &= case('enc5/dec8 C=256 T20 32x32', 256, 20480)
    ok &= case('enc8/dec5 C=512 T20 16x16', 512, 5120)
    ok &= case('enc13/dec0 C=512 T5 shift', 512, 1280, shift_hw=256, T=5)
    if not QUICK:
        ok &= case('B4 C=256', 256, 81920)
        ok &= case('B4 C=512 T20', 512, 20480)
        ok &= case('B4 C=512 T5 shift', 512, 5120, shift_hw=256, T=5)
    print('ALL OK' if ok else 'SOME FAILED')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
