"""Probe for an intra-clip frame-chunk pipeline: how long do TWO half-length clips in flight on two streams take against ONE full-length clip?
(The launches of a two-chunk pipeline over the frames of one clip are about those of two half-length clips.)  python tools/halfclip_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                                                             # noqa: E402
from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer            # noqa: E402
from magvit2_pytorch_amd.spec import init_state_dict, make_cfg           # noqa: E402
from magvit2_pytorch_amd.streaming import ClipLanes                      # noqa: E402

kw = dict(image_size=128, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS)
cfg = make_cfg(**kw)
tok = VideoTokenizer(**kw, seed=0)
tok.load_state_dict(init_state_dict(cfg, seed=0, randomize_se=True))
tok = tok.cuda()
tok.compute_dtype = torch.bfloat16


def timed(fn, n=100):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for T in (17, 9, 5):
    v = torch.randn(1, 3, T, 128, 128, generator=torch.Generator().manual_seed(T)).cuda()
    print(f'one clip of {T} frames ({T + 3} with the time padding): {timed(lambda: tok(v, return_codes=True, return_recon=True)):.3f} ms', flush=True)
lanes = ClipLanes(tok, lanes=2)
for T in (9, 5):
    vs = [torch.randn(1, 3, T, 128, 128, generator=torch.Generator().manual_seed(10 * T + i)).cuda() for i in range(2)]
    print(f'two clips of {T} frames in flight on two streams: {timed(lambda: list(lanes.map(vs, return_codes=True, return_recon=True))):.3f} ms per pair', flush=True)
lanes4 = ClipLanes(tok, lanes=4)
vs = [torch.randn(1, 3, 5, 128, 128, generator=torch.Generator().manual_seed(70 + i)).cuda() for i in range(4)]
print(f'four clips of 5 frames in flight on four streams: {timed(lambda: list(lanes4.map(vs, return_codes=True, return_recon=True))):.3f} ms per four', flush=True)
