"""
ctypes binding of libmv2_hip.so (include/mv2.h) — the only compute backend of the product path.

There is deliberately NO CPU fallback: if the library is missing, cannot be loaded, or no HIP
device is present, constructing ``HipOps`` raises.  Every method returns a *step*: a closure
``step(stream_ptr)`` that launches the kernel(s) on the given HIP stream with arguments frozen at
plan-build time (device pointers of statically allocated torch tensors), so a forward pass is a
flat list of C calls that can be captured into a HIP graph.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import pack as P

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmv2_hip.so')


class Mv2Error(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'dtype', 'B', 'Ti', 'Hi', 'Wi', 'Ci', 'To', 'Ho', 'Wo', 'co', 'co_pad', 'st', 'sh', 'sw', 'pt', 'ph', 'pw',
        'kt', 'kh', 'kw', 'nsteps', 'seg_bytes', 'check_bounds', 'act', 'store', 'rt', 'rh', 'rw', 'crop_t', 'out_c',
        'splitk')]


_SIGS = {
    'mv2_abi_version': (C.c_int, []),
    'mv2_device_cus': (C.c_int, []),
    'mv2_conv_workspace_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'mv2_conv': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mv2_conv_pointwise_dma': (C.c_int, [C.POINTER(ConvDesc)]),
    'mv2_conv_gather_dma': (C.c_int, [C.POINTER(ConvDesc)]),
    'mv2_conv_rms_ok': (C.c_int, [C.POINTER(ConvDesc)]),
    'mv2_conv_rms': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mv2_conv_in_eligible': (C.c_int, [C.POINTER(ConvDesc)]),
    'mv2_conv_in_weight_bytes': (C.c_size_t, [C.c_int]),
    'mv2_conv_in': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mv2_conv_out_eligible': (C.c_int, [C.POINTER(ConvDesc)]),
    'mv2_conv_out_weight_bytes': (C.c_size_t, [C.c_int]),
    'mv2_conv_out': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mv2_conv3_eligible': (C.c_int, [C.POINTER(ConvDesc)]),
    'mv2_conv3_fusable': (C.c_int, [C.POINTER(ConvDesc)]),
    'mv2_conv3_workspace_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'mv2_conv3_counter_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'mv2_conv3_tiles': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'mv2_conv3': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mv2_video_to_ndhwc': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 9 + [C.c_void_p]),
    'mv2_rms_rowscale': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_split_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'mv2_se_pool': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_se_gate': (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]),
    'mv2_se_apply': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_se_pool_fc1': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    'mv2_se_fc2_apply': (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]),
    'mv2_attention': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_int64, C.c_int64, C.c_void_p]),
    'mv2_taylor_kv': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_taylor_apply': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_taylor_kv_fused': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_taylor_apply_fused': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_taylor_state_floats': (C.c_size_t, [C.c_int]),
    'mv2_taylor_kv_d': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_taylor_apply_d': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_lfq': (C.c_int, [C.c_int] + [C.c_void_p] * 8 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    'mv2_pack_conv3d_bytes': (C.c_size_t, [C.c_int] * 6),
    'mv2_pack_conv3d': (C.c_int, [C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_size_t]),
    'mv2_copy_frames': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    'mv2_gateloop_scan': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    'mv2_pad_ndhwc': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    'mv2_cond_linear': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    'mv2_scale_channels': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    'mv2_se_rec_fc1': (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    'mv2_pw_se_rows': (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    'mv2_pw_se': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                            C.c_int, C.c_void_p]),
    'mv2_ff_eligible': (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    'mv2_ff_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    'mv2_ff_counter_bytes': (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    'mv2_ff': (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mv2_lfq_train': (C.c_int, [C.c_int] + [C.c_void_p] * 8 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    'mv2_lfq_stats_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    'mv2_lfq_stats': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mv2_lfq_losses': (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'mv2_sq_diff_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'mv2_sq_diff_sum': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mv2_hinge_discr_loss': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'mv2_lfq_codes': (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    'mv2_fsq': (C.c_int, [C.c_int] + [C.c_void_p] * 13 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_fsq_codes': (C.c_int, [C.c_int] + [C.c_void_p] * 7 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'mv2_event_create': (C.c_void_p, []),
    'mv2_event_record': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mv2_event_sync': (C.c_int, [C.c_void_p]),
    'mv2_event_elapsed_ms': (C.c_float, [C.c_void_p, C.c_void_p]),
    'mv2_event_destroy': (None, [C.c_void_p]),
}
EXPORTS = tuple(_SIGS)
_lib = None


def load_library():
    """Loads libmv2_hip.so and declares every prototype of include/mv2.h.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Mv2Error(f'{LIB_PATH} not built: run `python -m magvit2_pytorch_amd.build` (hipcc, gfx950). '
                       'There is no CPU fallback for the product path.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _check(rc, what):
    if rc != 0:
        raise Mv2Error(f'{what} failed with code {rc}')


def _mk(fn, args, what, keep, outs):
    def step(stream, _fn=fn, _args=args, _what=what, _keep=keep):
        rc = _fn(*_args, stream)
        if rc != 0:
            raise Mv2Error(f'{_what} failed with code {rc}')
    step.outs = [o for o in outs if o is not None]      # tensors this launch writes (lockstep parity runner)
    step.what = what
    return step


class HipOps:
    """Kernel launcher over the C ABI.  All tensors must live on ``device`` (a HIP device)."""
    name = 'hip'

    def __init__(self, device=None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise Mv2Error('no HIP device visible: the product path has no CPU fallback')
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        with torch.cuda.device(self.device):
            cus = self.lib.mv2_device_cus()
        if cus <= 0:
            raise Mv2Error(f'mv2_device_cus() = {cus}')
        self.cus = cus

    # ---- allocation (plumbing only)
    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def to_device(self, t):
        return t.to(self.device)

    # ---- steps
    def conv_workspace_bytes(self, d: dict) -> int:
        desc = ConvDesc(**d)
        with torch.cuda.device(self.device):
            return int(self.lib.mv2_conv_workspace_bytes(C.byref(desc)))

    def conv(self, d: dict, x, in_bytes, wp, segs, bias, rowscale, residual, out, ws):
        desc = ConvDesc(**d)
        fn = self.lib.mv2_conv
        args = (C.byref(desc), _ptr(x), in_bytes, _ptr(wp), _ptr(segs), _ptr(bias), _ptr(rowscale), _ptr(residual),
                _ptr(out), _ptr(ws), 0 if ws is None else ws.numel() * ws.element_size())
        keep = (desc, x, wp, segs, bias, rowscale, residual, out, ws)

        return _mk(fn, args, 'mv2_conv', keep, [out, ws])

    def conv_pointwise_dma(self, d: dict) -> bool:
        """does mv2_conv / mv2_conv_rms run this descriptor on the LDS-DMA pointwise kernel (introspection for tests)"""
        desc = ConvDesc(**d)
        return bool(self.lib.mv2_conv_pointwise_dma(C.byref(desc)))

    def conv_gather_dma(self, d: dict) -> bool:
        """does mv2_conv run this descriptor on the gather form of the LDS-DMA kernel (taps / strides / padding; introspection for tests)"""
        desc = ConvDesc(**d)
        return bool(self.lib.mv2_conv_gather_dma(C.byref(desc)))

    def conv_rms_ok(self, d: dict) -> bool:
        desc = ConvDesc(**d)
        return bool(self.lib.mv2_conv_rms_ok(C.byref(desc)))

    def conv_rms(self, d: dict, x, in_bytes, wp, segs, bias, residual, out):
        desc = ConvDesc(**d)
        fn = self.lib.mv2_conv_rms
        args = (C.byref(desc), _ptr(x), in_bytes, _ptr(wp), _ptr(segs), _ptr(bias), _ptr(residual), _ptr(out))
        return _mk(fn, args, 'mv2_conv_rms', (desc, x, wp, segs, bias, residual, out), [out])

    def conv_in_eligible(self, d: dict) -> bool:
        desc = ConvDesc(**d)
        return bool(self.lib.mv2_conv_in_eligible(C.byref(desc)))

    def conv_in(self, d: dict, x, in_bytes, wfrag, bias, out):
        desc = ConvDesc(**d)
        assert wfrag.numel() * wfrag.element_size() == self.lib.mv2_conv_in_weight_bytes(d['kt'])
        fn = self.lib.mv2_conv_in
        args = (C.byref(desc), _ptr(x), in_bytes, _ptr(wfrag), _ptr(bias), _ptr(out))
        return _mk(fn, args, 'mv2_conv_in', (desc, x, wfrag, bias, out), [out])

    def conv_out_eligible(self, d: dict) -> bool:
        desc = ConvDesc(**d)
        return bool(self.lib.mv2_conv_out_eligible(C.byref(desc)))

    def conv_out(self, d: dict, x, in_bytes, wfrag, bias, out):
        desc = ConvDesc(**d)
        assert wfrag.numel() * wfrag.element_size() == self.lib.mv2_conv_out_weight_bytes(d['kt'])
        fn = self.lib.mv2_conv_out
        args = (C.byref(desc), _ptr(x), in_bytes, _ptr(wfrag), _ptr(bias), _ptr(out))
        return _mk(fn, args, 'mv2_conv_out', (desc, x, wfrag, bias, out), [out])

    def conv3_eligible(self, d: dict) -> bool:
        desc = ConvDesc(**d)
        return bool(self.lib.mv2_conv3_eligible(C.byref(desc)))

    def conv3_fusable(self, d: dict) -> bool:
        desc = ConvDesc(**d)
        return bool(self.lib.mv2_conv3_fusable(C.byref(desc)))

    def conv3_sizes(self, d: dict):
        """(slab workspace bytes, ticket-counter bytes) of mv2_conv3 for this descriptor on this device."""
        desc = ConvDesc(**d)
        with torch.cuda.device(self.device):
            return int(self.lib.mv2_conv3_workspace_bytes(C.byref(desc))), int(self.lib.mv2_conv3_counter_bytes(C.byref(desc)))

    def conv3_tiles(self, d: dict):
        """(strips per frame, 2, 16): the SqueezeExcite pooling records written by the fused kernel."""
        desc = ConvDesc(**d)
        th, tw = C.c_int(0), C.c_int(0)
        n = self.lib.mv2_conv3_tiles(C.byref(desc), C.byref(th), C.byref(tw))
        if n <= 0:
            raise Mv2Error(f'mv2_conv3_tiles failed: {n}')
        return n, th.value, tw.value

    def conv3(self, d: dict, x, in_bytes, wp3, bias, out, ws, counters, wp2=None, bias2=None, act2=0,
              se_wk=None, se_bk=0.0, se_part=None):
        desc = ConvDesc(**d)
        fn = self.lib.mv2_conv3
        nb = lambda t: 0 if t is None else t.numel() * t.element_size()
        args = (C.byref(desc), _ptr(x), in_bytes, _ptr(wp3), _ptr(bias), _ptr(wp2), _ptr(bias2), int(act2),
                _ptr(se_wk), float(se_bk), _ptr(se_part), _ptr(out), _ptr(ws), nb(ws), _ptr(counters), nb(counters))
        keep = (desc, x, wp3, bias, wp2, bias2, se_wk, se_part, out, ws, counters)
        return _mk(fn, args, 'mv2_conv3', keep, [out, se_part])

    def split_bf16(self, x, out, rows, Cc):
        fn, args = self.lib.mv2_split_bf16, (_ptr(x), _ptr(out), rows, Cc)
        return _mk(fn, args, 'mv2_split_bf16', (x, out), [out])

    def copy_frames(self, src, src_T, src_t0, dst, dst_T, dst_t0, nT, B, frame_bytes):
        fn, args = self.lib.mv2_copy_frames, (_ptr(src), src_T, src_t0, _ptr(dst), dst_T, dst_t0, nT, B, frame_bytes)
        return _mk(fn, args, 'mv2_copy_frames', (src, dst), [dst])

    def gateloop_scan(self, dt, qkva, residual, out, B, T, Pn, Cc):
        fn, args = self.lib.mv2_gateloop_scan, (dt, _ptr(qkva), _ptr(residual), _ptr(out), B, T, Pn, Cc)
        return _mk(fn, args, 'mv2_gateloop_scan', (qkva, residual, out), [out])

    def pad_ndhwc(self, x, out, B, T, H, W, row_bytes, pt, ph, pw, mode, tail_bytes):
        fn, args = self.lib.mv2_pad_ndhwc, (_ptr(x), _ptr(out), B, T, H, W, row_bytes, pt, ph, pw, mode, tail_bytes)
        return _mk(fn, args, 'mv2_pad_ndhwc', (x, out), [out])

    def cond_linear(self, x, w, bias, out, B, I, O, mode, eps=1e-8):
        fn, args = self.lib.mv2_cond_linear, (_ptr(x), _ptr(w), _ptr(bias), _ptr(out), B, I, O, mode, float(eps))
        return _mk(fn, args, 'mv2_cond_linear', (x, w, bias, out), [out])

    def scale_channels(self, dt, x, scale, out, rows, rows_per_sample, Cc, add, act):
        fn, args = self.lib.mv2_scale_channels, (dt, _ptr(x), _ptr(scale), _ptr(out), rows, rows_per_sample, Cc, float(add), act)
        return _mk(fn, args, 'mv2_scale_channels', (x, scale, out), [out])

    def se_rec_fc1(self, part, w1, b1, hid, frames, Cc, nchunk, hidden, nslice):
        fn, args = self.lib.mv2_se_rec_fc1, (_ptr(part), _ptr(w1), _ptr(b1), _ptr(hid), frames, Cc, nchunk, hidden, nslice)
        return _mk(fn, args, 'mv2_se_rec_fc1', (part, w1, b1, hid), [hid])

    def pw_se_rows(self, dt, rows, Pn, Cc) -> int:
        return int(self.lib.mv2_pw_se_rows(dt, rows, Pn, Cc))

    def pw_se(self, dt, x, wp, bias, se_wk, se_bk, out, se_part, rows, Pn, Cc):
        fn = self.lib.mv2_pw_se
        args = (dt, _ptr(x), _ptr(wp), _ptr(bias), _ptr(se_wk), float(se_bk), _ptr(out), _ptr(se_part), rows, Pn, Cc)
        return _mk(fn, args, 'mv2_pw_se', (x, wp, bias, se_wk, out, se_part), [out, se_part])

    def ff_eligible(self, dt, rows, Cc, hidden_pad) -> bool:
        return bool(self.lib.mv2_ff_eligible(dt, rows, Cc, hidden_pad))

    def ff_sizes(self, rows, Cc, hidden_pad):
        """(slab workspace bytes, ticket-counter bytes) of mv2_ff for this shape on this device."""
        with torch.cuda.device(self.device):
            return int(self.lib.mv2_ff_workspace_bytes(rows, Cc, hidden_pad)), int(self.lib.mv2_ff_counter_bytes(rows, Cc, hidden_pad))

    def ff(self, dt, x, w1, b1, w2, b2, out, rows, Cc, hidden_pad, shift_hw, T, ws, counters):
        fn = self.lib.mv2_ff
        nb = lambda t: 0 if t is None else t.numel() * t.element_size()
        args = (dt, _ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(out), rows, Cc, hidden_pad, shift_hw, T,
                _ptr(ws), nb(ws), _ptr(counters), nb(counters))
        return _mk(fn, args, 'mv2_ff', (x, w1, b1, w2, b2, out, ws, counters), [out])

    def video_to_ndhwc(self, dt, video, out, B, Cc, T, H, W, pad_t, pad_hw, c_pad, tail_elems):
        fn, args = self.lib.mv2_video_to_ndhwc, (dt, _ptr(video), _ptr(out), B, Cc, T, H, W, pad_t, pad_hw, c_pad, tail_elems)
        keep = (video, out)
        return _mk(fn, args, 'mv2_video_to_ndhwc', keep, [out])

    def rms_rowscale(self, dt, x, rs, rows, Cc, shift_hw, T):
        fn, args = self.lib.mv2_rms_rowscale, (dt, _ptr(x), _ptr(rs), rows, Cc, shift_hw, T)
        keep = (x, rs)
        return _mk(fn, args, 'mv2_rms_rowscale', keep, [rs])

    def se_pool(self, dt, y, wk, bk, part, frames, Pn, Cc, nchunk):
        fn, args = self.lib.mv2_se_pool, (dt, _ptr(y), _ptr(wk), float(bk), _ptr(part), frames, Pn, Cc, nchunk)
        keep = (y, wk, part)
        return _mk(fn, args, 'mv2_se_pool', keep, [part])

    def se_gate(self, part, w1, b1, w2, b2, gate, frames, Cc, hidden, nchunk):
        fn, args = self.lib.mv2_se_gate, (_ptr(part), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(gate), frames, Cc, hidden, nchunk)
        keep = (part, w1, b1, w2, b2, gate)
        return _mk(fn, args, 'mv2_se_gate', keep, [gate])

    def se_apply(self, dt, y, gate, residual, out, frames, Pn, Cc):
        fn, args = self.lib.mv2_se_apply, (dt, _ptr(y), _ptr(gate), _ptr(residual), _ptr(out), frames, Pn, Cc)
        keep = (y, gate, residual, out)
        return _mk(fn, args, 'mv2_se_apply', keep, [out])

    def se_pool_fc1(self, dt, y, wk, bk, w1, b1, hid, frames, Pn, Cc, hidden, nslice):
        fn = self.lib.mv2_se_pool_fc1
        args = (dt, _ptr(y), _ptr(wk), float(bk), _ptr(w1), _ptr(b1), _ptr(hid), frames, Pn, Cc, hidden, nslice)
        keep = (y, wk, w1, b1, hid)
        return _mk(fn, args, 'mv2_se_pool_fc1', keep, [hid])

    def se_fc2_apply(self, dt, y, hid, w2, b2, residual, out, frames, Pn, Cc, hidden, nslice):
        fn = self.lib.mv2_se_fc2_apply
        args = (dt, _ptr(y), _ptr(hid), _ptr(w2), _ptr(b2), _ptr(residual), _ptr(out), frames, Pn, Cc, hidden, nslice)
        keep = (y, hid, w2, b2, residual, out)
        return _mk(fn, args, 'mv2_se_fc2_apply', keep, [out])

    def attention(self, dt, qkv, mem_kv, out, nseq, n, heads, dim_head, n_mem, causal, seq_inner, seq_outer_stride, tok_stride):
        fn = self.lib.mv2_attention
        args = (dt, _ptr(qkv), _ptr(mem_kv), _ptr(out), nseq, n, heads, dim_head, n_mem, int(causal), seq_inner,
                seq_outer_stride, tok_stride)
        keep = (qkv, mem_kv, out)
        return _mk(fn, args, 'mv2_attention', keep, [out])

    def taylor_kv(self, dt, kv, part, frames, n, heads, nchunk):
        fn, args = self.lib.mv2_taylor_kv, (dt, _ptr(kv), _ptr(part), frames, n, heads, nchunk)
        keep = (kv, part)
        return _mk(fn, args, 'mv2_taylor_kv', keep, [part])

    def taylor_apply(self, dt, q, part, out, kvsum, frames, n, heads, nchunk):
        fn, args = self.lib.mv2_taylor_apply, (dt, _ptr(q), _ptr(part), _ptr(out), _ptr(kvsum), frames, n, heads, nchunk)
        keep = (q, part, out, kvsum)
        return _mk(fn, args, 'mv2_taylor_apply', keep, [out, kvsum] if nchunk > 1 else [out])      # nchunk == 1: kvsum is not written

    def taylor_state_floats(self, dim_head: int) -> int:
        return int(self.lib.mv2_taylor_state_floats(dim_head))

    def taylor_kv_d(self, dt, kv, state, frames, n, heads, dim_head):
        fn, args = self.lib.mv2_taylor_kv_d, (dt, _ptr(kv), _ptr(state), frames, n, heads, dim_head)
        keep = (kv, state)
        return _mk(fn, args, 'mv2_taylor_kv_d', keep, [state])

    def taylor_apply_d(self, dt, q, state, out, frames, n, heads, dim_head):
        fn, args = self.lib.mv2_taylor_apply_d, (dt, _ptr(q), _ptr(state), _ptr(out), frames, n, heads, dim_head)
        keep = (q, state, out)
        return _mk(fn, args, 'mv2_taylor_apply_d', keep, [out])

    def taylor_kv_fused(self, dt, kv, part, frames, n, heads, nchunk):
        fn, args = self.lib.mv2_taylor_kv_fused, (dt, _ptr(kv), _ptr(part), frames, n, heads, nchunk)
        keep = (kv, part)
        return _mk(fn, args, 'mv2_taylor_kv_fused', keep, [part])

    def taylor_apply_fused(self, dt, q, part, out, kvsum, frames, n, heads, nchunk):
        fn, args = self.lib.mv2_taylor_apply_fused, (dt, _ptr(q), _ptr(part), _ptr(out), _ptr(kvsum), frames, n, heads, nchunk)
        keep = (q, part, out, kvsum)
        return _mk(fn, args, 'mv2_taylor_apply_fused', keep, [out, kvsum] if nchunk > 1 else [out])      # nchunk == 1: kvsum is not written

    def lfq(self, dt, x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, Cc, cb_dim, ncb, clamp, scale):
        fn = self.lib.mv2_lfq
        args = (dt, _ptr(x), _ptr(w_in), _ptr(b_in), _ptr(w_out), _ptr(b_out), _ptr(quantized), _ptr(ids), _ptr(prequant),
                rows, Cc, cb_dim, ncb, float(clamp), float(scale))
        keep = (x, w_in, b_in, w_out, b_out, quantized, ids, prequant)
        return _mk(fn, args, 'mv2_lfq', keep, [quantized, ids, prequant])

    def lfq_train(self, dt, x, w_in, b_in, w_out, b_out, quantized, ids, prequant, rows, Cc, cb_dim, ncb, clamp, scale):
        fn = self.lib.mv2_lfq_train
        args = (dt, _ptr(x), _ptr(w_in), _ptr(b_in), _ptr(w_out), _ptr(b_out), _ptr(quantized), _ptr(ids), _ptr(prequant),
                rows, Cc, cb_dim, ncb, float(clamp), float(scale))
        keep = (x, w_in, b_in, w_out, b_out, quantized, ids, prequant)
        return _mk(fn, args, 'mv2_lfq_train', keep, [quantized, ids, prequant])

    def lfq_stats_workspace_bytes(self, rows, cb_dim, ncb) -> int:
        return int(self.lib.mv2_lfq_stats_workspace_bytes(rows, cb_dim, ncb))

    def lfq_stats(self, prequant, rows, cb_dim, ncb, inv_temperature, scale, sums, ws):
        fn = self.lib.mv2_lfq_stats
        args = (_ptr(prequant), rows, cb_dim, ncb, float(inv_temperature), float(scale), _ptr(sums), _ptr(ws),
                ws.numel() * ws.element_size())
        return _mk(fn, args, 'mv2_lfq_stats', (prequant, sums, ws), [sums])

    def lfq_losses(self, sums, n_tokens, cb_dim, ncb, gamma, w_entropy, w_commit, losses):
        fn = self.lib.mv2_lfq_losses
        args = (_ptr(sums), float(n_tokens), cb_dim, ncb, float(gamma), float(w_entropy), float(w_commit), _ptr(losses))
        return _mk(fn, args, 'mv2_lfq_losses', (sums, losses), [losses])

    def sq_diff_workspace_bytes(self, n) -> int:
        return int(self.lib.mv2_sq_diff_workspace_bytes(n))

    def sq_diff_sum(self, a, b, n, out, ws):
        fn = self.lib.mv2_sq_diff_sum
        args = (_ptr(a), _ptr(b), n, _ptr(out), _ptr(ws), ws.numel() * ws.element_size())
        return _mk(fn, args, 'mv2_sq_diff_sum', (a, b, out, ws), [out])

    def hinge_discr_loss(self, fake, real, n, out):
        fn, args = self.lib.mv2_hinge_discr_loss, (_ptr(fake), _ptr(real), n, _ptr(out))
        return _mk(fn, args, 'mv2_hinge_discr_loss', (fake, real, out), [out])

    def lfq_codes(self, dt, ids, w_out, b_out, out, rows, Cc, cb_dim, ncb, scale):
        fn, args = self.lib.mv2_lfq_codes, (dt, _ptr(ids), _ptr(w_out), _ptr(b_out), _ptr(out), rows, Cc, cb_dim, ncb, float(scale))
        keep = (ids, w_out, b_out, out)
        return _mk(fn, args, 'mv2_lfq_codes', keep, [out])

    def fsq(self, dt, x, w_in, b_in, w_out, b_out, consts, quantized, ids, prequant, rows, Cc, cb_dim, ncb):
        fn = self.lib.mv2_fsq
        args = (dt, _ptr(x), _ptr(w_in), _ptr(b_in), _ptr(w_out), _ptr(b_out), _ptr(consts['shift']), _ptr(consts['half_l']),
                _ptr(consts['offset']), _ptr(consts['half_w']), _ptr(consts['basis']), _ptr(quantized), _ptr(ids),
                _ptr(prequant), rows, Cc, cb_dim, ncb)
        keep = (x, w_in, b_in, w_out, b_out, consts, quantized, ids, prequant)
        return _mk(fn, args, 'mv2_fsq', keep, [quantized, ids, prequant])

    def fsq_codes(self, dt, ids, w_out, b_out, consts, out, rows, Cc, cb_dim, ncb):
        fn = self.lib.mv2_fsq_codes
        args = (dt, _ptr(ids), _ptr(w_out), _ptr(b_out), _ptr(consts['levels']), _ptr(consts['half_w']), _ptr(consts['basis']),
                _ptr(out), rows, Cc, cb_dim, ncb)
        keep = (ids, w_out, b_out, consts, out)
        return _mk(fn, args, 'mv2_fsq_codes', keep, [out])

    # ---- stream / timing plumbing
    def current_stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def synchronize(self):
        torch.cuda.synchronize(self.device)
