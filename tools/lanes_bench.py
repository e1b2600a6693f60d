"""How should a batch of independent clips be scheduled on one GPU?  One call on the whole batch against sub-batches in flight on
streaming.ClipLanes (replicas of the tokenizer on their own HIP streams).  usage: python tools/lanes_bench.py [clips=4]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                                                             # noqa: E402
from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer            # noqa: E402
from magvit2_pytorch_amd.spec import init_state_dict, make_cfg           # noqa: E402
from magvit2_pytorch_amd.streaming import ClipLanes                      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = dict(image_size=128, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS)
cfg = make_cfg(**kw)
tok = VideoTokenizer(**kw, seed=0)
tok.load_state_dict(init_state_dict(cfg, seed=0, randomize_se=True))
tok = tok.cuda()
tok.compute_dtype = torch.bfloat16
v = torch.randn(N, 3, 17, 128, 128, generator=torch.Generator().manual_seed(1)).cuda()


def timed(fn, n=60):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, out


ms, ref = timed(lambda: tok(v, return_codes=True, return_recon=True))
print(f'one call on {N} clips: {ms:.3f} ms  ({N * 17 / ms * 1e3:.0f} frames/s)', flush=True)
for lanes, per in ((2, N // 2), (N, 1), (2, 1), (4, 1), (3, 1)):
    if per < 1 or lanes > N and per == 1 and lanes != N:
        continue
    cl = ClipLanes(tok, lanes=lanes)
    chunks = [v[i:i + per] for i in range(0, N, per)]
    ms, outs = timed(lambda: list(cl.map(chunks, return_codes=True, return_recon=True)))
    codes = torch.cat([o[0] for o in outs])
    print(f'{lanes} lanes, {per} clip(s) per call, {len(chunks)} calls: {ms:.3f} ms  ({N * 17 / ms * 1e3:.0f} frames/s)  ids equal the one-call result: {torch.equal(codes, ref[0])}', flush=True)
    del cl
