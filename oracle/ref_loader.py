"""
TEST INFRASTRUCTURE — not product code.

Loads the reference's own ``magvit2_pytorch.py`` + ``attend.py`` so the *reference itself* can
be executed on CPU.  It is used to (a) pin ``oracle/restate.py``, (b) generate the golden
fixtures under tests/golden/ (``oracle/make_golden.py``) and (c) as the checker / CPU baseline
on the GPU box (``__graft_entry__.smoke()``, ``bench.py``'s ``cpu_baseline`` leg).

Where it reads from: ``/root/reference`` in the authoring container (read in place).  That path
does not exist on the GPU box, so ``__graft_entry__.build()`` places a build-time copy of the
three files it needs under ``oracle/_ref/`` (git-ignored: never part of the repository's
history; it travels with the gpurun snapshot exactly like the built ``.so``), and this loader
falls back to that copy.  ``MV2_REFERENCE_ROOT`` overrides both.

The reference is not importable as a package here: its
``__init__`` pulls trainer -> data -> cv2 / ema_pytorch / ..., and the model file imports
beartype, torchvision, kornia, gateloop_transformer, vector_quantize_pytorch and
taylor_series_linear_attention, none of which are installed.  We register ``sys.modules``
stubs for the non-arithmetic ones, route the two arithmetic ones to
``oracle/thirdparty.py`` (restatements, parity unpinned), create an empty package object
that bypasses ``__init__.py`` and exec version.py / attend.py / magvit2_pytorch.py by path.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
import typing
import warnings

REF_COPY_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')
REF_FILES = ('magvit2_pytorch.py', 'attend.py', 'version.py')


def _resolve_root():
    env = os.environ.get('MV2_REFERENCE_ROOT')
    cands = [env, '/root/reference', REF_COPY_ROOT]
    for c in cands:
        if c and os.path.isfile(os.path.join(c, 'magvit2_pytorch', 'magvit2_pytorch.py')):
            return c
        if c and c is env:
            warnings.warn(f'MV2_REFERENCE_ROOT={env!r} does not contain magvit2_pytorch/magvit2_pytorch.py: ignored')
    return '/root/reference'


REFERENCE_ROOT = _resolve_root()
_PKG_DIR = os.path.join(REFERENCE_ROOT, 'magvit2_pytorch')


def reference_available() -> bool:
    return os.path.isfile(os.path.join(_PKG_DIR, 'magvit2_pytorch.py'))


def reference_origin() -> str:
    """'/root/reference' (authoring container), 'oracle/_ref' (build-time copy) or the override; for log lines."""
    return 'oracle/_ref (build-time copy of /root/reference/magvit2_pytorch/{magvit2_pytorch,attend,version}.py)' if REFERENCE_ROOT == REF_COPY_ROOT else REFERENCE_ROOT


def place_reference_copy(src_root='/root/reference') -> bool:
    """Build step (``__graft_entry__.build()``): copy the three reference files the loader executes into the git-ignored
    ``oracle/_ref/magvit2_pytorch/`` so that the GPU box can run the reference itself as the checker and the CPU baseline."""
    import shutil
    src = os.path.join(src_root, 'magvit2_pytorch')
    if not all(os.path.isfile(os.path.join(src, f)) for f in REF_FILES):
        return False
    dst = os.path.join(REF_COPY_ROOT, 'magvit2_pytorch')
    os.makedirs(dst, exist_ok=True)
    for f in REF_FILES:
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
    _refresh_root()
    return True


def _refresh_root():
    """re-resolve the reference root (the module constants were fixed at import, so a process that placed the copy itself kept
    pointing at a missing /root/reference)"""
    global REFERENCE_ROOT, _PKG_DIR
    REFERENCE_ROOT = _resolve_root()
    _PKG_DIR = os.path.join(REFERENCE_ROOT, 'magvit2_pytorch')


def _stub(name, **attrs):
    if name in sys.modules:
        mod = sys.modules[name]
    else:
        mod = types.ModuleType(name)
        sys.modules[name] = mod
    for k, v in attrs.items():
        setattr(mod, k, v)
    return mod


def _beartype(obj=None, **_):
    # usable bare (@beartype) on functions / classes -> identity
    if obj is None:
        return lambda o: o
    return obj


def _not_on_path(*_, **__):
    raise RuntimeError('stubbed third-party symbol: not on the scoped hot path')


def install_shims():
    from . import thirdparty

    _stub('beartype', beartype=_beartype)
    _stub('beartype.typing', Tuple=typing.Tuple, List=typing.List, Optional=typing.Optional,
          Union=typing.Union, Type=typing.Type, Literal=typing.Literal)

    class _VGG16_Weights:
        DEFAULT = None

    tv_models = _stub('torchvision.models', VGG16_Weights=_VGG16_Weights, vgg16=_not_on_path)
    _stub('torchvision', models=tv_models)

    kf = _stub('kornia.filters', filter3d=_not_on_path)
    _stub('kornia', filters=kf)

    _stub('gateloop_transformer', SimpleGateLoopLayer=thirdparty.SimpleGateLoopLayer)      # reconstruction, parity unpinned (oracle/thirdparty.py)
    _stub('vector_quantize_pytorch', LFQ=thirdparty.LFQ, FSQ=thirdparty.FSQ)
    _stub('taylor_series_linear_attention', TaylorSeriesLinearAttn=thirdparty.TaylorSeriesLinearAttn)


_loaded = None


def load_reference():
    """Returns the reference's ``magvit2_pytorch.magvit2_pytorch`` module object."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise FileNotFoundError(f'{_PKG_DIR} not present (neither /root/reference nor the build-time copy oracle/_ref)')
    install_shims()

    pkg = types.ModuleType('magvit2_pytorch')
    pkg.__path__ = [_PKG_DIR]        # bypasses __init__.py
    sys.modules['magvit2_pytorch'] = pkg

    def _exec(sub):
        spec = importlib.util.spec_from_file_location(f'magvit2_pytorch.{sub}', os.path.join(_PKG_DIR, f'{sub}.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f'magvit2_pytorch.{sub}'] = mod
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            spec.loader.exec_module(mod)
        setattr(pkg, sub, mod)
        return mod

    _exec('version')
    _exec('attend')
    _loaded = _exec('magvit2_pytorch')
    return _loaded


def make_reference_tokenizer(**kwargs):
    """Reference ``VideoTokenizer`` (magvit2_pytorch.py:1045) in eval mode with VGG / GAN
    construction switched off (perceptual_loss_weight=0 -> VGG never built, :1392)."""
    ref = load_reference()
    kwargs.setdefault('perceptual_loss_weight', 0.)
    kwargs.setdefault('use_gan', False)
    kwargs.setdefault('flash_attn', False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        tok = ref.VideoTokenizer(**kwargs)
    return tok.eval()
