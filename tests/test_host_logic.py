"""CPU tests of the product's host side: weight contract, packing / K programs / plan lowering
(executed through the CPU emulation of the kernel semantics, oracle/ndhwc_ops.py), API asserts,
and that the C-ABI library loads and exports every symbol include/mv2.h declares."""
import ctypes
import os
import re

import pytest
import torch

from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer
from magvit2_pytorch_amd import ops_hip
from magvit2_pytorch_amd.spec import init_state_dict, make_cfg, param_specs
from oracle.ndhwc_ops import CpuOps
from oracle.restate import OracleTokenizer
from tests._util import Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emulated_tokenizer(kw, sd, dtype):
    tok = VideoTokenizer(**kw, seed=0)
    tok.load_state_dict(sd)
    tok._set_backend(CpuOps())
    tok.compute_dtype = dtype
    return tok


# ------------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'mv2.h')).read()
    declared = set(re.findall(r'\b(mv2_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no prototypes found'
    assert declared == set(ops_hip.EXPORTS), declared ^ set(ops_hip.EXPORTS)
    assert os.path.exists(ops_hip.LIB_PATH), 'libmv2_hip.so missing: run __graft_entry__.build()'
    lib = ctypes.CDLL(ops_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in mv2.h but not exported'
    lib.mv2_abi_version.restype = ctypes.c_int
    assert lib.mv2_abi_version() == 1
    ops_hip.load_library()                       # declares argtypes for all of them


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    tok = VideoTokenizer(image_size=32, init_dim=8, max_dim=32, codebook_size=256, layers=README_LAYERS, seed=0)
    with pytest.raises(RuntimeError):
        tok.tokenize(torch.randn(1, 3, 5, 32, 32))
    with pytest.raises(ops_hip.Mv2Error):
        ops_hip.HipOps()


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'magvit2_pytorch_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f'{fn} imports the oracle'


# ------------------------------------------------------------------------------- weight contract
def test_state_dict_keys_and_round_trip(tmp_path):
    kw = dict(image_size=32, init_dim=8, max_dim=32, codebook_size=256, layers=README_LAYERS)
    tok = VideoTokenizer(**kw, seed=3)
    names = {p.name for p in param_specs(make_cfg(**kw))}
    assert set(tok.state_dict().keys()) == names
    assert 'encoder_layers.14.1.weight' in names            # the never-executed LayerNorm keeps its keys (reference :1322-1326)
    assert 'decoder_layers.0.0.fn.fn.to_qkv.0.weight' in names and 'encoder_layers.5.0.fn.attn.to_kv.0.weight' in names
    path = tmp_path / 'tok.pt'
    tok.save(str(path))
    tok2 = VideoTokenizer.init_and_load_from(str(path))
    for k, v in tok.state_dict().items():
        assert torch.equal(v, tok2.state_dict()[k])
    # reference checkpoints carry discriminator weights: ignored on load
    sd = dict(tok.state_dict())
    sd['discr.blocks.0.net.0.weight'] = torch.zeros(3)
    tok2.load_state_dict(sd)


def test_ctor_and_call_asserts():
    with pytest.raises(AssertionError):
        VideoTokenizer(image_size=32, layers=README_LAYERS)                          # no codebook_size (reference :1359)
    with pytest.raises(AssertionError):
        VideoTokenizer(image_size=32, layers=README_LAYERS, use_fsq=True, codebook_size=64, fsq_levels=[8, 5])
    with pytest.raises(ValueError):
        VideoTokenizer(image_size=32, layers=('residual', 'bogus'), codebook_size=64)  # reference :1312
    with pytest.raises(NotImplementedError):
        VideoTokenizer(image_size=32, layers=('residual', 'gateloop_time'), codebook_size=64)
    g = Golden('small_lfq')
    tok = emulated_tokenizer(g.kw, g.weights(), torch.float32)
    with pytest.raises(AssertionError):
        tok(torch.randn(1, 3, 9, 16, 16))                                            # wrong image size (reference :1677)
    with pytest.raises(AssertionError):
        tok(torch.randn(1, 3, 8, 32, 32))                                            # (T-1) % 4 != 0 (reference :1691)
    with pytest.raises(AssertionError):
        tok.decode_from_code_indices(torch.zeros(1, 3, 4, 4))                        # float ids (reference :1585)
    with pytest.raises(AssertionError):
        tok.decode_from_code_indices(torch.zeros(1, 17, dtype=torch.long))           # not divisible by fmap^2 (reference :1589)
    with pytest.raises(NotImplementedError):
        tok(torch.randn(1, 3, 9, 32, 32), return_loss=True)


# ------------------------------------------------------------------------------- plan / packing vs golden
@pytest.mark.parametrize('name', ['small_lfq', 'small_fsq', 'small_image', 'mid_lfq'])
def test_plan_fp32_matches_reference_golden(name):
    g = Golden(name)
    tok = emulated_tokenizer(g.kw, g.weights(), torch.float32)
    codes, recon = tok(g.video(), return_codes=True, return_recon=True)
    assert codes.dtype == g.codes.dtype and torch.equal(codes, g.codes)
    assert torch.allclose(g.sample(recon), g.recon_sample, atol=2e-5, rtol=0)
    assert torch.allclose(tok.decode_from_code_indices(g.codes), recon, atol=1e-6, rtol=0)
    flat = g.codes.reshape(g.codes.shape[0], -1)
    assert torch.equal(tok.decode_from_code_indices(flat), tok.decode_from_code_indices(g.codes))
    assert torch.equal(tok.tokenize(g.video()), g.codes)


def test_plan_bf16_storage_within_tolerance():
    """bf16 storage / fp32 accumulate: tolerance yardstick = the reference's own bf16-autocast drift
    (recon mean|d| 4-7e-3, 1.6-2.2 % ids)."""
    g = Golden('mid_lfq')
    tok = emulated_tokenizer(g.kw, g.weights(), torch.bfloat16)
    codes, recon = tok(g.video(), return_codes=True, return_recon=True)
    assert (codes != g.codes).float().mean().item() <= 0.05
    assert (g.sample(recon) - g.recon_sample).abs().mean().item() < 1.5e-2


def test_encode_decode_api_and_multi_codebook():
    kw = dict(image_size=32, init_dim=8, max_dim=32, codebook_size=64, num_codebooks=2,
              layers=('residual', 'compress_space', ('consecutive_residual', 2), 'attend_space', 'compress_time', 'attend_time'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=5, randomize_se=True)
    tok = emulated_tokenizer(kw, sd, torch.float32)
    orc = OracleTokenizer(sd, **cfg)
    v = torch.randn(1, 3, 5, 32, 32, generator=torch.Generator().manual_seed(6))
    lat = tok.encode(v)
    assert torch.allclose(lat, orc.encode(v), atol=1e-5)
    q, ids, aux = tok.encode(v, quantize=True)
    oq, oids = orc.quantize(orc.encode(v))
    assert ids.shape == oids.shape == (1, 3, 16, 16, 2) and torch.equal(ids, oids)
    assert torch.allclose(q, oq, atol=1e-5)
    assert torch.allclose(tok.decode(oq), orc.decode(oq), atol=1e-5)
    assert torch.allclose(tok.decode_from_code_indices(oids), orc.decode_from_code_indices(oids), atol=1e-5)


def test_causality():
    """CausalConv3d / causal time attention: perturbing frame t must leave latent frames that only see
    earlier input unchanged (SURVEY §8c KAT)."""
    g = Golden('small_lfq')
    tok = emulated_tokenizer(g.kw, g.weights(), torch.float32)
    v = g.video()
    a = tok.encode(v)
    v2 = v.clone()
    v2[:, :, 5:] += 1.0                     # frames 5.. feed latent frames >= 2 only (time stride 4, first frame alone)
    b = tok.encode(v2)
    assert torch.equal(a[:, :, :2], b[:, :, :2]) and not torch.equal(a[:, :, 2:], b[:, :, 2:])


# ------------------------------------------------------------------------------- SqueezeExcite lowering variants
def _full_width_case():
    kw = dict(image_size=16, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=7, randomize_se=True)
    v = torch.randn(1, 3, 5, 16, 16, generator=torch.Generator().manual_seed(8))
    return kw, cfg, sd, v


def test_two_launch_squeeze_excite_matches_oracle():
    """Real channel widths 64..512 on a 16x16 frame: every level with C >= 128 lowers SqueezeExcite to
    mv2_se_pool_fc1 + mv2_se_fc2_apply; the emulated plan must still reproduce the oracle (fp32)."""
    kw, cfg, sd, v = _full_width_case()
    tok = emulated_tokenizer(kw, sd, torch.float32)
    names = tok._plan(1, 5)[0].step_names
    assert 'se_pool_fc1' in names and 'se_fc2_apply' in names
    orc = OracleTokenizer(sd, **cfg)
    codes, recon = tok(v, return_codes=True, return_recon=True)
    ocodes, orecon = orc(v, return_codes=True, return_recon=True)
    assert (codes != ocodes).float().mean().item() <= 0.005
    assert (recon - orecon).abs().max().item() < 5e-4


def test_squeeze_excite_forms_agree(monkeypatch):
    """pool/gate/apply (with the pooling partials emitted by the fused conv kernel at C <= 128) against the default
    lowering, bf16 storage: same tensors up to summation order."""
    kw, cfg, sd, v = _full_width_case()
    a_codes, a_recon = emulated_tokenizer(kw, sd, torch.bfloat16)(v, return_codes=True, return_recon=True)
    monkeypatch.setenv('MV2_SE_THREE', '1')
    tok = emulated_tokenizer(kw, sd, torch.bfloat16)
    names = tok._plan(1, 5)[0].step_names
    assert 'se_pool_fc1' not in names and 'se_gate' in names
    b_codes, b_recon = tok(v, return_codes=True, return_recon=True)
    assert (a_codes != b_codes).float().mean().item() <= 0.02
    assert (a_recon - b_recon).abs().mean().item() < 5e-3
    monkeypatch.setenv('MV2_NO_FUSE_POOL', '1')
    c_codes, c_recon = emulated_tokenizer(kw, sd, torch.bfloat16)(v, return_codes=True, return_recon=True)
    assert (b_codes != c_codes).float().mean().item() <= 0.02
    assert (b_recon - c_recon).abs().mean().item() < 5e-3
