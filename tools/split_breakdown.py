"""Per-launch event table of the id-exact 'split' mode's encoder (compute_dtype = 'split': fp32 storage, ResidualUnit convolutions as split-bf16
products) on BASELINE configs[1], next to the exact-fp32 mode's: where the mode's time is.  GPU box only.
    python tools/split_breakdown.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
from magvit2_pytorch_amd import README_LAYERS      # noqa: E402

kw = dict(image_size=128, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS)
tok, cfg, sd = bench.make_tokenizer(kw, 'cuda:0', 'bf16', True)
video = torch.randn(1, 3, 17, 128, 128, generator=torch.Generator().manual_seed(1)).cuda()
out = {}
for mode in ('split', torch.float32):
    tok.compute_dtype = mode
    tok(video, return_codes=True)
    rows = bench.per_launch_profile(tok, video, tok._backend(), tokenize_only=True)
    agg = {}
    for r in rows:
        key = r['what'] if r['what'] not in ('mv2_conv', 'mv2_conv3', 'mv2_conv_rms') else r['label']
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += r['ms']
    tot = sum(r['ms'] for r in rows)
    print(f'#### {mode}: encoder + quantizer {len(rows)} launches, {tot:.3f} ms')
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f'   {ms * 1e3:8.1f} us  x{n:<3d} {k}')
    out[str(mode)] = rows
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
