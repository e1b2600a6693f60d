#!/usr/bin/env python3
"""Aggregates the rocprofv3 --pmc passes of tools/gpu_profile.sh into the two judged summaries:
  <tag>_pmc_per_kernel_avg_per_launch.csv  counter values per kernel name, averaged per launch, all passes side by side
  <tag>_traffic.json                       L2<->fabric bytes per conv3_kernel launch (bench.py's roofline.traffic source)
usage: tools/pmc_aggregate.py gpurun_out r02 profiles"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(src, tag, dst):
    per = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))       # kernel -> counter -> [sum, dispatches]
    launches = defaultdict(dict)
    for pas in ('pmc_sq', 'pmc_fetch', 'pmc_write'):
        files = glob.glob(os.path.join(src, f'{tag}_{pas}', '**', '*counter_collection.csv'), recursive=True)
        seen = defaultdict(set)
        for fn in files:
            with open(fn) as f:
                for row in csv.DictReader(f):
                    k, c = row['Kernel_Name'], row['Counter_Name']
                    per[k][c][0] += float(row['Counter_Value'])
                    seen[k].add((fn, row['Dispatch_Id']))
        for k, s in seen.items():
            launches[k][pas] = len(s)
            for c in per[k]:
                pass
    counters = sorted({c for k in per for c in per[k]})
    passes = ('pmc_fetch', 'pmc_sq', 'pmc_write')
    which = {}
    for pas in passes:
        for fn in glob.glob(os.path.join(src, f'{tag}_{pas}', '**', '*counter_collection.csv'), recursive=True):
            with open(fn) as f:
                for row in csv.DictReader(f):
                    which[row['Counter_Name']] = pas
    out_csv = os.path.join(dst, f'{tag}_pmc_per_kernel_avg_per_launch.csv')
    with open(out_csv, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel'] + counters + [f'launches_{tag}_{p}' for p in passes])
        for k in sorted(per, key=lambda k: -per[k].get('SQ_WAVE_CYCLES', [0, 0])[0]):
            row = [k]
            for c in counters:
                n = launches[k].get(which.get(c, ''), 0)
                row.append(round(per[k][c][0] / n, 1) if n else '')
            w.writerow(row + [launches[k].get(p, 0) for p in passes])
    # traffic per conv3 launch
    fs = ws = 0.0
    nf = nw = 0
    for k in per:
        if 'conv3p_kernel' in k or 'conv3_kernel' in k:      # mv2_conv3: the bf16 position-tile kernel (conv3p.hip) / the fp32 K-split kernel (conv3.hip)
            fs += per[k]['FETCH_SIZE'][0]; nf += launches[k].get('pmc_fetch', 0)
            ws += per[k]['WRITE_SIZE'][0]; nw += launches[k].get('pmc_write', 0)
    if nf and nw:
        fetch_kb, write_kb = fs / nf, ws / nw
        import hashlib
        h = hashlib.sha256()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for fn in ('conv3.hip', 'conv3_knobs.h', 'conv3p.hip', 'conv3p.h', 'conv_common.h', 'common.h'):      # bench.py::kernel_source_hash
            with open(os.path.join(root, 'magvit2_pytorch_amd', 'csrc', fn), 'rb') as f:
                h.update(f.read())
        js = dict(conv3_kernel_hbm_bytes_per_launch=int((2 * fetch_kb + write_kb) * 1024), launches_averaged=nf,
                  kernel_source_sha256=h.hexdigest(),
                  fetch_size_kb_avg=round(fetch_kb, 1), write_size_kb_avg=round(write_kb, 1),
                  note='rocprofv3 --pmc passes of `bench.py --steps 20 --warmup 3 --no-graph` (B=1, bf16), averaged over all mv2_conv3 '
                       'launches (22 per forward); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a 16 B/lane coalesced '
                       'stream), WRITE_SIZE as reported. Counts L2<->fabric bytes, i.e. MALL hits are included. Algorithmic bytes per launch '
                       '~31 MB (11.2 MB in + 11.2 MB out + 8.5 MB weights on average).')
        with open(os.path.join(dst, f'{tag}_traffic.json'), 'w') as f:
            json.dump(js, f, indent=1)
        print(json.dumps(js)[:300])
    print('wrote', out_csv)


if __name__ == '__main__':
    main(*sys.argv[1:4])
