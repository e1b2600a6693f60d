"""Is the step bound by what the kernels can do, or by what the chip may draw?  The same captured step (BASELINE configs[1], bf16, one clip)
replayed (a) back to back, as bench.py times it, and (b) with the device left idle between replays (host sleep of 2 / 5 / 20 ms): every
replay bracketed by its own event pair.  If the replays behind an idle gap are faster, the sustained figure is set by the power / thermal
management over the step, not by the sum of its kernels' schedules (profiles/r06_findings.md section 2 found the same for mv2_conv3 alone).
usage (GPU box): python tools/duty_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    kw, B, T, size, _, _ = bench.workload(1, tiny=False)
    tok, cfg, sd = bench.make_tokenizer(kw, dev, 'bf16', True)
    video = torch.randn(B, 3, T, size, size, generator=torch.Generator().manual_seed(1)).to(dev)

    def step():
        return tok(video, return_codes=True, return_recon=True)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:      # soak, as bench.py does
        step()
    torch.cuda.synchronize()

    def run(n, gap_s):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            if gap_s > 0:
                torch.cuda.synchronize()
                time.sleep(gap_s)
            a.record()
            step()
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        return ms[len(ms) // 2], ms[0], ms[-1], sum(ms) / len(ms)

    for rep in range(2):
        for gap in (0.0, 0.002, 0.005, 0.020, 0.0):
            med, lo, hi, mean = run(200 if gap < 0.01 else 100, gap)
            print(f'pass {rep}: idle gap {gap * 1e3:5.1f} ms -> per replay: median {med:.3f} ms  min {lo:.3f}  max {hi:.3f}  mean {mean:.3f}', flush=True)


if __name__ == '__main__':
    main()
