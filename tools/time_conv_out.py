"""times mv2_conv_out on the decoder's shape at B = 1 and B = 4 (GPU box)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from magvit2_pytorch_amd import pack as PK
import magvit2_pytorch_amd.ops_hip as _o
if os.environ.get('MV2_EXP_LIB'):
    _o.LIB_PATH = _o.LIB_PATH.replace('libmv2_hip.so', 'libmv2_hip_exp.so')
from magvit2_pytorch_amd.ops_hip import HipOps
ops = HipOps('cuda:0'); dev = ops.device
for B in (1, 4):
    T, H, W, co, kt, crop = 20, 128, 128, 3, 3, 3
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * T * H * W * 64, generator=g).to(torch.bfloat16).to(dev)
    wf = PK.pack_conv_out(torch.randn(co, 64, kt, 3, 3, generator=g) / 40).to(dev)
    bias = torch.zeros(4, device=dev)
    d = dict(dtype=PK.MV2_BF16, B=B, Ti=T, Hi=H, Wi=W, Ci=64, To=T, Ho=H, Wo=W, co=co, co_pad=128, st=1, sh=1, sw=1, pt=kt - 1, ph=1, pw=1, kt=kt, kh=3, kw=3,
             nsteps=kt * 9, seg_bytes=128, check_bounds=1, act=PK.ACT_NONE, store=PK.STORE_NCTHW_F32, rt=1, rh=1, rw=1, crop_t=crop, out_c=4, splitk=0)
    out = torch.empty(B * co * (T - crop) * H * W, device=dev)
    step = ops.conv_out(d, x, x.numel() * 2, wf, bias, out)
    st = ops.current_stream()
    for _ in range(5): step(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): step(st)
    e1.record(); torch.cuda.synchronize()
    print('conv_out B=%d: %.1f us per launch' % (B, e0.elapsed_time(e1) * 1e3 / 50))
