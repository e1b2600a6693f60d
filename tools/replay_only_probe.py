"""What the facade's three device copies per call (clip in, reconstruction and ids out) and the graph <-> eager transitions cost: the
captured step of BASELINE configs[1] timed (a) through forward() as bench.py does and (b) as bare replays of the same graph back to back.
usage (GPU box): python tools/replay_only_probe.py"""
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
    step = lambda: tok(video, return_codes=True, return_recon=True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    graphs = [g for g in tok._graphs.values() if not isinstance(g, str)]
    assert len(graphs) == 1
    g = graphs[0]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        step()
    torch.cuda.synchronize()

    def timed(fn, n=300):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    for rep in range(3):
        print(f'pass {rep}: forward() {timed(step):.4f} ms   bare graph replay {timed(g.replay):.4f} ms', flush=True)


if __name__ == '__main__':
    main()
