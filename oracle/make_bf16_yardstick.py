"""
TEST INFRASTRUCTURE — not product code.  Authoring container only (needs /root/reference).

The "stated fp tolerance" of the bf16 GPU path is the reference's OWN bf16 drift: the reference VideoTokenizer run under
``torch.autocast('cpu', dtype=torch.bfloat16)`` against the same module in fp32, same weights, same clip.  One clip gives one
sample of a chaotic quantity (a 1-ulp change upstream moves which tokens flip), so the yardstick is taken over several data
seeds and the GPU test compares MEANS over the same seeds (tests/test_parity_gpu.py::test_bf16_drift_within_reference_yardstick).

    python oracle/make_bf16_yardstick.py [case ...]  # (re)computes the named cases (default: all) and merges them into tests/golden/bf16_yardstick.json

Per case and data seed: ids mismatch (fraction of tokens), recon mean|d| and max|d| of autocast-bf16 vs fp32, both from the
reference executed in place (oracle/ref_loader.py).
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from magvit2_pytorch_amd.spec import README_LAYERS, init_state_dict, make_cfg   # noqa: E402
from oracle.ref_loader import make_reference_tokenizer                          # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'bf16_yardstick.json')

# name: (ctor kwargs, clip shape, weight seed, data seeds) — weights / first data seed are those of the golden fixtures and of bench.py
CASES = {
    'mid_lfq': (dict(image_size=64, init_dim=16, max_dim=128, codebook_size=1024, layers=README_LAYERS), (1, 3, 9, 64, 64), 5, [6, 101, 102, 103, 104, 105]),
    'cfg2_video128x17': (dict(image_size=128, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS), (1, 3, 17, 128, 128), 0,
                         [1, 101, 102, 103, 104, 105]),
    # the two largest fixtures (BASELINE configs[3] and [4]; shapes, weight and data seed of tests/golden/cfg4_* / cfg5_*): ONE draw each
    # (CPU minutes per clip) - it anchors the single-clip caps of tests/test_parity_gpu.py to the reference instead of to measured HIP values
    'cfg4_video256x33': (dict(image_size=256, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS), (2, 3, 33, 256, 256), 0, [1]),
    'cfg5_fsq128x65': (dict(image_size=128, init_dim=64, max_dim=512, use_fsq=True, fsq_levels=[8, 5, 5, 5], layers=README_LAYERS), (2, 3, 65, 128, 128), 0, [1]),
}


def main():
    torch.set_num_threads(os.cpu_count())
    res = dict(torch_version=torch.__version__, note='reference VideoTokenizer, CPU: autocast(bfloat16) vs fp32; see oracle/make_bf16_yardstick.py', cases={})
    if os.path.exists(OUT):
        with open(OUT) as f:
            res['cases'] = json.load(f).get('cases', {})
    todo = sys.argv[1:] or list(CASES)
    for name, (kw, shape, wseed, dseeds) in CASES.items():
        if name not in todo:
            continue
        cfg = make_cfg(**kw)
        sd = init_state_dict(cfg, seed=wseed, randomize_se=True)
        ref = make_reference_tokenizer(**kw)
        ref.load_state_dict(sd, strict=False)
        rows = []
        for ds in dseeds:
            v = torch.randn(*shape, generator=torch.Generator().manual_seed(ds))
            t0 = time.time()
            with torch.no_grad():
                c32, r32 = ref(v, return_codes=True, return_recon=True)
                with torch.autocast('cpu', dtype=torch.bfloat16):
                    c16, r16 = ref(v, return_codes=True, return_recon=True)
                # the two SMOOTH halves (no id flips involved): the encoder up to the quantizer's projection, and the decoder teacher-forced
                # with the fp32 ids — what the GPU test's regression gates are taken on
                z32 = ref.quantizers.project_in(ref.encode(v).movedim(1, -1))
                with torch.autocast('cpu', dtype=torch.bfloat16):
                    z16 = ref.quantizers.project_in(ref.encode(v).movedim(1, -1))
                    d16 = ref.decode_from_code_indices(c32)
            d = (r16.float() - r32).abs()
            dd = (d16.float() - r32).abs()
            dz = (z16.float() - z32).abs()
            rows.append(dict(data_seed=ds, ids_mismatch=float((c16 != c32).float().mean()), recon_mean_abs=float(d.mean()), recon_max_abs=float(d.max()),
                             decoder_only_mean_abs=float(dd.mean()), prequant_mean_abs=float(dz.mean()), prequant_ref_mean_abs_value=float(z32.abs().mean())))
            print(name, rows[-1], f'{time.time() - t0:.1f}s', flush=True)
        n = len(rows)
        res['cases'][name] = dict(kw={k: (list(x) if isinstance(x, tuple) else x) for k, x in kw.items()}, shape=list(shape), weight_seed=wseed, rows=rows,
                                  mean_ids_mismatch=sum(r['ids_mismatch'] for r in rows) / n, mean_recon_mean_abs=sum(r['recon_mean_abs'] for r in rows) / n,
                                  mean_decoder_only_mean_abs=sum(r['decoder_only_mean_abs'] for r in rows) / n,
                                  mean_prequant_mean_abs=sum(r['prequant_mean_abs'] for r in rows) / n)
        with open(OUT, 'w') as f:
            json.dump(res, f, indent=1)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
