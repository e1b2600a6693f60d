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


class ClipLanes:
    """Several independent calls in flight: `lanes` replicas of an eval tokenizer (same weights, own plan buffers and HIP graphs), one HIP
    stream each, clips dealt round-robin.

    Why: at one clip per call the path is a chain of ~130 dependent launches of ~20 us each and the chip idles through their latency
    phases (profiles/r05_findings.md section 5); a second, independent chain on a second stream fills them: measured on MI355X,
    configs[1], two lanes: 3.91 ms per pair of clips against 2 x 2.64 ms (1.35 x the frames/s; `bench.py`'s `two_streams` object).  Batching
    the clips into one call (`forward` on a (B, ...) tensor) is still better where the caller HAS the clips together (4 clips: 1.79 ms
    per clip); lanes are for requests that arrive one by one.  Results are those of the plain calls (same kernels, same plans).

    Stream semantics: a lane's stream waits for the caller's current stream before each call (the clip may have been produced there), and
    `map` makes the caller's stream wait for a result's event before yielding it, so consumers stay stream-ordered.  On a CPU backend
    (tests: the kernel emulation) it degrades to the plain loop.  The reference has no equivalent."""

    def __init__(self, tokenizer, lanes: int = 2):
        assert lanes >= 1
        tokenizer.eval()
        self.device = tokenizer.device
        self.on_gpu = self.device.type == 'cuda'
        self.toks = [tokenizer]
        # NOTE: copy_for_eval() calls refresh_weights() on the ORIGINAL, which drops its packed weights, plans and captured
        # graphs: lane 0 re-packs and re-captures on its next call.  Build the lanes before warming up (bench.py and the tests do), or call
        # the tokenizer once afterwards.  Launch-time sizing queries (mv2_ff's workgroup split, mv2_conv3's work list) follow the CURRENT
        # device: the lanes' calls run under torch.cuda.device(tokenizer.device), like every call of the tokenizer itself.
        for _ in range(lanes - 1):
            dup = tokenizer.copy_for_eval()
            dup.compute_dtype, dup.use_hip_graph = tokenizer.compute_dtype, tokenizer.use_hip_graph
            if not self.on_gpu and getattr(tokenizer, '_ops', None) is not None:
                dup._set_backend(tokenizer._ops)                      # test hook: the injected emulation backend serves every lane
            self.toks.append(dup)
        self.streams = [torch.cuda.Stream(self.device) for _ in self.toks] if self.on_gpu else [None] * lanes

    @staticmethod
    def _tensors(out):
        if torch.is_tensor(out):
            yield out
        elif isinstance(out, (tuple, list)):
            for o in out:
                yield from ClipLanes._tensors(o)

    @torch.no_grad()
    def map(self, clips: Iterable[torch.Tensor], method: str = 'forward', **kw) -> Iterator:
        """yields `getattr(tokenizer, method)(clip, **kw)` for every clip, in order, with up to `lanes` calls in flight"""
        n = len(self.toks)
        pending = deque()

        def finish():
            out, ev = pending.popleft()
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in self._tensors(out):
                    t.record_stream(cur)                              # allocated on the lane's stream, consumed on the caller's
            return out

        for i, v in enumerate(clips):
            tok = self.toks[i % n]
            fn = tok if method == 'forward' else getattr(tok, method)
            if not self.on_gpu:
                pending.append((fn(v, **kw), None))
            else:
                s = self.streams[i % n]
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    out = fn(v, **kw)
                    ev = torch.cuda.Event()
                    ev.record(s)
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(s)
                pending.append((out, ev))
            if len(pending) >= n:
                yield finish()
        while pending:
            yield finish()
