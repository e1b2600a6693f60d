"""Summary of a rocprofv3 kernel trace of bench.py in graph mode: the last 20 replays of the captured step are located as runs of kernels
starting with mv2's video_to_ndhwc kernel; per kernel position: mean duration and mean idle gap to the previous kernel's end."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
starts = [i for i, n in enumerate(names) if 'video_to_ndhwc' in n]
# a replay = from one video_to_ndhwc to the next; keep those with the modal kernel count
lens = defaultdict(int)
for a, b in zip(starts, starts[1:]):
    lens[b - a] += 1
modal = max(lens, key=lens.get)
steps = [(a, b) for a, b in zip(starts, starts[1:]) if b - a == modal][-20:]
print(f'{len(rows)} kernels traced; replay length (kernels incl. torch copies) {modal}; {len(steps)} replays analysed')
dur = defaultdict(list); gap = defaultdict(list)
tot_busy = []; tot_span = []
for a, b in steps:
    busy = 0
    for k in range(a, b):
        s, e = int(rows[k]['Start_Timestamp']), int(rows[k]['End_Timestamp'])
        dur[k - a].append(e - s); busy += e - s
        if k > a:
            gap[k - a].append(s - int(rows[k - 1]['End_Timestamp']))
    tot_busy.append(busy)
    tot_span.append(int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp']))
n = len(steps)
print(f'per replay: span first start -> last end {sum(tot_span) / n / 1e3:.1f} us, sum of kernel durations {sum(tot_busy) / n / 1e3:.1f} us, '
      f'sum of gaps {(sum(tot_span) - sum(tot_busy)) / n / 1e3:.1f} us')
allg = [g for k in gap for g in gap[k]]
allg.sort()
print(f'gap between consecutive kernels: mean {sum(allg) / len(allg) / 1e3:.2f} us, median {allg[len(allg) // 2] / 1e3:.2f}, p90 {allg[int(len(allg) * .9)] / 1e3:.2f}, max {allg[-1] / 1e3:.2f}; negative (overlap) {sum(g < 0 for g in allg)} of {len(allg)}')
byname = defaultdict(lambda: [0, 0.0, 0.0])
a0 = steps[-1][0]
for k in range(modal):
    nm = names[a0 + k].split('(')[0][-60:]
    byname[nm][0] += 1; byname[nm][1] += sum(dur[k]) / n; byname[nm][2] += (sum(gap[k]) / n if k in gap else 0)
print(f'{"kernel":62s} {"n":>3s} {"sum dur us":>10s} {"avg us":>8s} {"sum gap-before us":>10s}')
for nm, (c, d, g) in sorted(byname.items(), key=lambda kv: -kv[1][1]):
    print(f'{nm:62s} {c:3d} {d / 1e3:10.1f} {d / c / 1e3:8.2f} {g / 1e3:10.1f}')
print('\nper position:')
for k in range(modal):
    print(f'{k:3d} {names[a0 + k].split("(")[0][-50:]:50s} dur {sum(dur[k]) / n / 1e3:7.2f}  gap-before {(sum(gap[k]) / n / 1e3 if k in gap else 0):6.2f}')
