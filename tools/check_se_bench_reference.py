"""Checks the CPU reference inside tools/se_bench.cpp against the Python oracle, with no GPU:
    tools/se_bench magvit2_pytorch_amd/libmv2_hip.so --dump /tmp/sedump && python tools/check_se_bench_reference.py /tmp/sedump
(1) oracle/restate.py's conv1 + ELU + squeeze_excite + residual add on the dumped inputs (fp32 math on the bf16-rounded values, y2 rounded
to bf16 as the kernels store it) must reproduce the harness's `ref`; (2) the kernel-semantics emulation (oracle/ndhwc_ops.py: mv2_pw_se on
the fragment-major image that mv2_pack_conv3d produced, then gate + apply) must agree with it to bf16 round-off — so a disagreement
between the harness and the GPU next round points at the kernel, not at the harness."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.restate import squeeze_excite      # noqa: E402

d = sys.argv[1]
ld = lambda n: torch.from_numpy(np.fromfile(os.path.join(d, n), dtype=np.float32).copy())
frames, P, C, hidden = np.fromfile(os.path.join(d, 'hdr.i32'), dtype=np.int32).tolist()
y1, x = ld('y1.f32').reshape(frames, P, C), ld('x.f32').reshape(frames, P, C)
w1, b1c, wk = ld('w1.f32').reshape(C, C), ld('b1c.f32'), ld('wk.f32')
w1t, b1, w2t, b2 = ld('w1t.f32').reshape(C, hidden), ld('b1.f32'), ld('w2t.f32').reshape(hidden, C), ld('b2.f32')
y2_h, ref_h = ld('y2.f32').reshape(frames, P, C), ld('ref.f32').reshape(frames, P, C)
bk = 0.05

y2 = F.elu(y1.double() @ w1.double().t() + b1c.double()).float().bfloat16().float()
print('conv1 + ELU   : max|d| %.3e' % (y2 - y2_h).abs().max().item())
# oracle squeeze_excite takes (B, C, T, H, W): frames as T, positions as H x 1
p = {'se.to_k.weight': wk.reshape(1, C, 1, 1), 'se.to_k.bias': torch.tensor([bk]), 'se.net.0.weight': w1t.t().reshape(hidden, C, 1, 1),
     'se.net.0.bias': b1, 'se.net.2.weight': w2t.t().reshape(C, hidden, 1, 1), 'se.net.2.bias': b2}
y5 = y2_h.permute(2, 0, 1).reshape(1, C, frames, P, 1)
out = squeeze_excite(y5.double(), {k: v.double() for k, v in p.items()}, 'se.') + x.permute(2, 0, 1).reshape(1, C, frames, P, 1).double()
out = out.reshape(C, frames, P).permute(1, 2, 0).float()
e = (out - ref_h).abs().max().item()
print('whole tail    : max|d| %.3e' % e)
assert (y2 - y2_h).abs().max().item() <= 8e-3 and e <= 1e-5, 'the harness reference disagrees with the oracle'
print('se_bench reference == oracle: ok')
