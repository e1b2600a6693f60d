"""
Host-fed tokenization with the PCIe transfers off the critical path.

``VideoTokenizer.tokenize(video)`` takes a device tensor; a server that receives clips in host memory pays
H2D (3.3 MB per 17-frame 128x128 fp32 clip, ~0.05-0.1 ms) + compute (3.2 ms at batch 1) + D2H of the ids (10 KB) in
sequence.  ``tokenize_stream`` keeps ``depth`` clips in flight: the upload of clip i+1 and the download of the ids of
clip i-1 run on a side HIP stream while clip i computes on the caller's stream (events order the hand-offs; the plan's
static input buffer is filled by a device-to-device copy on the compute stream, so the captured HIP graph is replayed
unchanged).  On a CPU backend (tests: the kernel emulation) it degrades to the plain loop.

The reference has no equivalent (it is called with device tensors, trainer.py moves batches with accelerate).
"""
from __future__ import annotations

from collections import deque
from typing import Iterable, Iterator

import torch


@torch.no_grad()
def tokenize_stream(tokenizer, clips: Iterable[torch.Tensor], depth: int = 2, cond: Iterable[torch.Tensor] | None = None) -> Iterator[torch.Tensor]:
    """yields the code ids (host tensors) of every clip of `clips` (host tensors (B, C, T, H, W), ideally pinned), in order"""
    assert depth >= 1
    tokenizer.eval()
    dev = tokenizer.device
    conds = iter(cond) if cond is not None else None
    if dev.type != 'cuda':                                        # CPU backend (kernel emulation in the tests): nothing to overlap
        for v in clips:
            yield tokenizer.tokenize(v, cond=next(conds) if conds is not None else None).cpu()
        return
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    staged = deque()      # (device clip, device cond, upload-done event)
    done = deque()        # (host ids, download-done event)
    it = iter(clips)

    def upload():
        try:
            v = next(it)
        except StopIteration:
            return False
        c = next(conds) if conds is not None else None
        with torch.cuda.stream(side):
            dv = v.to(dev, non_blocking=True)
            dc = c.to(dev, non_blocking=True) if c is not None else None
            ev = torch.cuda.Event()
            ev.record(side)
        staged.append((dv, dc, ev))
        return True

    for _ in range(depth):
        if not upload():
            break
    while staged:
        dv, dc, ev = staged.popleft()
        main.wait_event(ev)                                       # the clip has landed
        ids = tokenizer.tokenize(dv, cond=dc)                     # D2D into the plan's input buffer + graph replay, on `main`
        dv.record_stream(main)
        if dc is not None:
            dc.record_stream(main)                                # consumed on `main` by the copy into plan.cond_in
        computed = torch.cuda.Event()
        computed.record(main)
        with torch.cuda.stream(side):
            side.wait_event(computed)
            host = torch.empty(ids.shape, dtype=ids.dtype, pin_memory=True)
            host.copy_(ids, non_blocking=True)
            ids.record_stream(side)
            got = torch.cuda.Event()
            got.record(side)
        done.append((host, got))
        upload()                                                  # next upload overlaps the compute just queued
        while len(done) > depth:
            h, g = done.popleft()
            g.synchronize()
            yield h
    while done:
        h, g = done.popleft()
        g.synchronize()
        yield h
