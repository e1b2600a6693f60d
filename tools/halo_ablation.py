"""Per-tap component ablation of the halo conv kernel (TIMING ONLY: the ablated kernels compute wrong results).
One process, the environment switch is flipped between per-launch event profiles (conv_halo.hip re-reads
MV2_HALO_ABL / MV2_HALO_EW on every launch).  Prints the duration of every ResidualUnit 3x3 launch per variant.
usage (GPU box): python tools/halo_ablation.py [out.json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                    # noqa: E402
from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer                   # noqa: E402
from magvit2_pytorch_amd.spec import init_state_dict, make_cfg                  # noqa: E402

VARIANTS = [('default', {}), ('no barrier', {'MV2_HALO_ABL': '4'}), ('no MFMA', {'MV2_HALO_ABL': '1'}),
            ('no LDS frag reads', {'MV2_HALO_ABL': '2'}), ('no LDS writes', {'MV2_HALO_ABL': '3'}),
            ('no weight loads', {'MV2_HALO_ABL': '5'}), ('early weight write', {'MV2_HALO_EW': '1'}), ('default again', {})]


def main():
    kw = dict(image_size=128, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS)
    cfg = make_cfg(**kw)
    tok = VideoTokenizer(**kw, seed=0)
    tok.load_state_dict(init_state_dict(cfg, seed=0, randomize_se=True))
    tok = tok.cuda()
    tok.compute_dtype = torch.bfloat16
    tok.use_hip_graph = False
    video = torch.randn(1, 3, 17, 128, 128, generator=torch.Generator().manual_seed(1)).cuda()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.7:                  # plan build + clock spin-up
        tok(video, return_codes=True, return_recon=True)
        torch.cuda.synchronize()
    ops = tok._backend()
    out = {}
    for name, env in VARIANTS:
        for k in ('MV2_HALO_ABL', 'MV2_HALO_EW'):
            os.environ.pop(k, None)
        os.environ.update(env)
        rows = bench.per_launch_profile(tok, video, ops)
        out[name] = {r['label']: round(r['ms'] * 1e3, 1) for r in rows if '.k3' in r['label']}
        print(f'{name:20s} sum k3 = {sum(out[name].values()):7.1f} us   ' +
              ' '.join(f'{v:5.1f}' for v in list(out[name].values())[:11]), flush=True)
    for k in ('MV2_HALO_ABL', 'MV2_HALO_EW'):
        os.environ.pop(k, None)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
