"""Opt-in: diff the reconstructions in oracle/thirdparty.py against the REAL upstream packages whenever they are importable
(they are not in this image, so these tests skip here; a maintainer with `pip install vector-quantize-pytorch
taylor-series-linear-attention` gets the pin the reference itself never had).  Records the upstream versions it ran against."""
import importlib
import importlib.util

import pytest
import torch

from oracle import thirdparty as TP


def _real(name):
    import sys
    mod = sys.modules.get(name)
    if mod is not None and getattr(mod, '__file__', None) is None:       # oracle/ref_loader.py's sys.modules shim, not the package
        pytest.skip(f'{name} is shimmed in this process and not installed: third-party parity stays unpinned')
    try:
        spec = importlib.util.find_spec(name)
    except (ValueError, ImportError):
        spec = None
    if spec is None:
        pytest.skip(f'{name} is not installed (expected in this image): third-party parity stays unpinned')
    return importlib.import_module(name)


def _version(dist):
    try:
        from importlib.metadata import version
        return version(dist)
    except Exception:
        return 'unknown'


def test_lfq_matches_upstream():
    vq = _real('vector_quantize_pytorch')
    print('vector-quantize-pytorch', _version('vector-quantize-pytorch'))
    torch.manual_seed(0)
    kw = dict(dim=32, codebook_size=256, num_codebooks=2, entropy_loss_weight=0.1, commitment_loss_weight=1., diversity_gamma=2.5,
              soft_clamp_input_value=10.)
    up, mine = vq.LFQ(**kw), TP.LFQ(**kw)
    mine.load_state_dict(up.state_dict(), strict=False)
    x = torch.randn(2, 32, 3, 4, 4)
    for train in (False, True):
        up.train(train); mine.train(train)
        with torch.no_grad():
            (q0, i0, a0), b0 = up(x, return_loss_breakdown=True)
            (q1, i1, a1), b1 = mine(x, return_loss_breakdown=True)
        assert torch.equal(i0, i1) and torch.allclose(q0, q1, atol=1e-6)
        assert torch.allclose(torch.as_tensor(a0).float(), torch.as_tensor(a1).float(), atol=1e-5)
        for u, m in zip(b0, b1):
            assert torch.allclose(torch.as_tensor(u).float(), torch.as_tensor(m).float(), atol=1e-5)
    assert torch.allclose(up.indices_to_codes(i0), mine.indices_to_codes(i0), atol=1e-6)


def test_fsq_matches_upstream():
    vq = _real('vector_quantize_pytorch')
    torch.manual_seed(0)
    up, mine = vq.FSQ(levels=[8, 5, 5, 5], dim=32), TP.FSQ(levels=[8, 5, 5, 5], dim=32)
    mine.load_state_dict(up.state_dict(), strict=False)
    x = torch.randn(2, 32, 3, 4, 4)
    with torch.no_grad():
        q0, i0 = up(x)
        q1, i1 = mine(x)
    assert torch.equal(i0, i1) and torch.allclose(q0, q1, atol=1e-6)
    assert torch.allclose(up.indices_to_codes(i0), mine.indices_to_codes(i0), atol=1e-6)


def test_taylor_attention_matches_upstream():
    ta = _real('taylor_series_linear_attention')
    print('taylor-series-linear-attention', _version('taylor-series-linear-attention'))
    torch.manual_seed(0)
    up, mine = ta.TaylorSeriesLinearAttn(dim=64, dim_head=8, heads=16), TP.TaylorSeriesLinearAttn(dim=64, dim_head=8, heads=16)
    mine.load_state_dict(up.state_dict(), strict=False)
    x = torch.randn(3, 40, 64)
    with torch.no_grad():
        assert torch.allclose(up(x), mine(x), atol=1e-5)


def test_lfq_spherical_matches_upstream():
    vq = _real('vector_quantize_pytorch')
    torch.manual_seed(0)
    kw = dict(dim=32, codebook_size=256, num_codebooks=2, entropy_loss_weight=0.1, commitment_loss_weight=1., diversity_gamma=2.5,
              soft_clamp_input_value=10., spherical=True)
    up, mine = vq.LFQ(**kw), TP.LFQ(**kw)
    mine.load_state_dict(up.state_dict(), strict=False)
    x = torch.randn(2, 32, 3, 4, 4)
    for train in (False, True):
        up.train(train); mine.train(train)
        with torch.no_grad():
            (q0, i0, a0), _ = up(x, return_loss_breakdown=True)
            (q1, i1, a1), _ = mine(x, return_loss_breakdown=True)
        assert torch.equal(i0, i1) and torch.allclose(q0, q1, atol=1e-6)
        assert torch.allclose(torch.as_tensor(a0).float(), torch.as_tensor(a1).float(), atol=1e-5)
    assert torch.allclose(up.indices_to_codes(i0), mine.indices_to_codes(i0), atol=1e-6)


def test_simple_gateloop_layer_matches_upstream():
    gl = _real('gateloop_transformer')
    print('gateloop-transformer', _version('gateloop-transformer'))
    torch.manual_seed(0)
    up, mine = gl.SimpleGateLoopLayer(dim=32, use_heinsen=False), TP.SimpleGateLoopLayer(dim=32)
    assert set(up.state_dict()) == set(mine.state_dict()), (set(up.state_dict()), set(mine.state_dict()))      # the parameter names the weight contract relies on
    mine.load_state_dict(up.state_dict())
    x = torch.randn(3, 9, 32)
    with torch.no_grad():
        assert torch.allclose(up(x), mine(x), atol=1e-5)
