"""CPU tests of the product's host side: weight contract, packing / K programs / plan lowering
(executed through the CPU emulation of the kernel semantics, oracle/ndhwc_ops.py), API asserts,
and that the C-ABI library loads and exports every symbol include/mv2.h declares."""
import ctypes
import dataclasses
import os
import re

import pytest
import torch

from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer
from magvit2_pytorch_amd import ops_hip
from magvit2_pytorch_amd import plan as plan_mod
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
    assert lib.mv2_abi_version() == 2
    ops_hip.load_library()                       # declares argtypes for all of them


def test_abi_rejects_malformed_arguments_without_touching_the_gpu():
    """Error behaviour of the boundary (include/mv2.h: negative MV2_E* code, nothing thrown, nothing launched):
    argument validation happens before any HIP call, so it can be exercised on a box without a GPU."""
    lib = ops_hip.load_library()
    EINVAL = -1
    buf = ctypes.create_string_buffer(4096)
    ptr = ctypes.cast(buf, ctypes.c_void_p)
    good = dict(dtype=0, B=1, Ti=4, Hi=16, Wi=16, Ci=64, To=4, Ho=16, Wo=16, co=64, co_pad=128, st=1, sh=1, sw=1, pt=2, ph=1, pw=1,
                kt=3, kh=3, kw=3, nsteps=27, seg_bytes=128, check_bounds=1, act=1, store=0, rt=1, rh=1, rw=1, crop_t=0, out_c=64,
                splitk=0)
    d = ops_hip.ConvDesc(**good)
    # mv2_pad_ndhwc: null pointers, unknown mode, rows that are not whole 8-byte words, reflect padding >= size (as F.pad), misaligned tail
    assert lib.mv2_pad_ndhwc(None, ptr, 1, 4, 8, 8, 16, 2, 1, 1, 1, 0, None) == EINVAL
    for args in ((1, 4, 8, 8, 16, 2, 1, 1, 0, 0), (1, 4, 8, 8, 16, 2, 1, 1, 4, 0), (1, 4, 8, 8, 12, 2, 1, 1, 1, 0), (1, 2, 8, 8, 16, 2, 1, 1, 1, 0),
                 (1, 4, 8, 8, 16, 2, 8, 1, 1, 0), (1, 4, 8, 8, 16, 2, 1, 1, 2, 4), (0, 4, 8, 8, 16, 2, 1, 1, 2, 0)):
        assert lib.mv2_pad_ndhwc(ptr, ptr, *args, None) == EINVAL, args
    # null pointers
    assert lib.mv2_conv(ctypes.byref(d), None, 0, None, None, None, None, None, None, None, 0, None) == EINVAL
    assert lib.mv2_conv3(ctypes.byref(d), None, 0, None, None, None, None, 1, None, 0.0, None, None, None, 0, None, 0, None) == EINVAL
    # malformed descriptors: co_pad not a multiple of 128, unknown dtype, GEGLU with odd widths
    for bad in (dict(co_pad=96), dict(dtype=7), dict(act=3, co=100, out_c=50), dict(seg_bytes=48), dict(kt=9)):
        dd = ops_hip.ConvDesc(**dict(good, **bad))
        assert lib.mv2_conv(ctypes.byref(dd), ptr, 4096, ptr, ptr, None, None, None, ptr, None, 0, None) == EINVAL, bad
    # mv2_conv3 eligibility / tiling is pure host logic: stride-1 causal kt x 3 x 3 over whole 128-byte chunks, C_out >= 64
    assert lib.mv2_conv3_eligible(ctypes.byref(d)) == 1
    th, tw = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.mv2_conv3_fusable(ctypes.byref(d)) == 1
    assert lib.mv2_conv3_tiles(ctypes.byref(d), ctypes.byref(th), ctypes.byref(tw)) == 8 and (th.value, tw.value) == (2, 16)
    d256 = ops_hip.ConvDesc(**dict(good, co=256, co_pad=256, out_c=256, Ci=256, nsteps=108))
    assert lib.mv2_conv3_eligible(ctypes.byref(d256)) == 1
    assert lib.mv2_conv3_fusable(ctypes.byref(d256)) == 0 and lib.mv2_conv3_tiles(ctypes.byref(d256), None, None) == EINVAL
    assert lib.mv2_conv3_counter_bytes(ctypes.byref(d256)) == 4 * 4 * 4 * 2        # one ticket word per (co tile, frame, spatial tile)
    for bad in (dict(st=2), dict(kh=5), dict(Ci=48), dict(co=32, out_c=32), dict(pt=0), dict(store=2), dict(store=1), dict(act=2)):
        dd = ops_hip.ConvDesc(**dict(good, **bad))
        assert lib.mv2_conv3_eligible(ctypes.byref(dd)) == 0, bad
        assert lib.mv2_conv3_fusable(ctypes.byref(dd)) == 0 and lib.mv2_conv3_tiles(ctypes.byref(dd), None, None) == EINVAL
        assert lib.mv2_conv3(ctypes.byref(dd), ptr, 4096, ptr, None, None, None, 0, None, 0.0, None, ptr, None, 0, None, 0, None) == EINVAL, bad
    # the fused second stage needs bf16 and a block that owns every channel (C_out in {64, 128}), no rowscale / residual
    assert lib.mv2_conv3(ctypes.byref(d256), ptr, 4096, ptr, None, ptr, None, 1, None, 0.0, None, ptr, ptr, 1 << 30, ptr, 4096, None) == EINVAL
    d32 = ops_hip.ConvDesc(**dict(good, dtype=1, Ci=32, nsteps=27))
    assert lib.mv2_conv3(ctypes.byref(d32), ptr, 4096, ptr, None, ptr, None, 1, None, 0.0, None, ptr, ptr, 1 << 30, ptr, 4096, None) == EINVAL
    # the other entry points: null pointers / out-of-contract sizes
    assert lib.mv2_se_pool(0, None, None, 0.0, None, 1, 1, 64, 1, None) == EINVAL
    assert lib.mv2_se_pool(0, ptr, ptr, 0.0, ptr, 1, 4, 64, 8, None) == EINVAL                 # more chunks than rows
    assert lib.mv2_se_gate(None, None, None, None, None, None, 1, 64, 32, 1, None) == EINVAL
    assert lib.mv2_se_apply(0, ptr, ptr, None, ptr, 1, 4, 60, None) == EINVAL                   # C % 8 != 0 (bf16)
    assert lib.mv2_se_pool_fc1(0, ptr, ptr, 0.0, ptr, ptr, ptr, 1, 4, 128, 64, 3, None) == EINVAL   # hidden % nslice != 0
    assert lib.mv2_se_fc2_apply(0, ptr, ptr, ptr, ptr, None, ptr, 1, 4, 96, 48, 1, None) == EINVAL  # C/nslice not a power of 2
    assert lib.mv2_rms_rowscale(0, None, None, 1, 64, 0, 1, None) == EINVAL
    assert lib.mv2_attention(0, ptr, ptr, ptr, 1, 4, 8, 48, 4, 0, 1, 4, 1, None) == EINVAL      # dim_head must be 8, 16, 32 or 64
    assert lib.mv2_taylor_state_floats(8) == 657 and lib.mv2_taylor_state_floats(16) == 273 * 17 and lib.mv2_taylor_state_floats(4) == 21 * 5
    assert lib.mv2_taylor_state_floats(32) == 0 and lib.mv2_taylor_state_floats(12) == 0
    assert lib.mv2_taylor_kv_d(0, ptr, ptr, 1, 4, 8, 12, None) == EINVAL and lib.mv2_taylor_kv_d(0, None, ptr, 1, 4, 8, 16, None) == EINVAL    # width / null
    assert lib.mv2_taylor_apply_d(0, ptr, ptr, ptr, 1, 4, 8, 32, None) == EINVAL and lib.mv2_taylor_apply_d(2, ptr, ptr, ptr, 1, 4, 8, 16, None) == EINVAL   # width / dtype
    assert lib.mv2_lfq(0, ptr, ptr, ptr, ptr, ptr, ptr, ptr, None, 1, 64, 17, 2, 10.0, 1.0, None) == EINVAL   # 34 code dims > 32
    assert lib.mv2_fsq_codes(0, None, None, None, None, None, None, None, 1, 64, 4, 1, None) == EINVAL
    assert lib.mv2_taylor_kv(0, ptr, ptr, 1, 4, 16, 8, None) == EINVAL                          # more chunks than tokens
    assert lib.mv2_taylor_kv_fused(0, ptr, ptr, 1, 4, 64, 1, None) == EINVAL                    # more than 32 heads
    assert lib.mv2_taylor_apply_fused(0, ptr, ptr, ptr, None, 1, 4, 16, 2, None) == EINVAL      # chunked state without the reduce workspace
    # mv2_split_bf16: null / misaligned pointers, rows that are not whole 16-byte chunks
    assert lib.mv2_split_bf16(None, ptr, 4, 64, None) == EINVAL
    assert lib.mv2_split_bf16(ptr, ptr, 4, 62, None) == EINVAL
    assert lib.mv2_split_bf16(ctypes.c_void_p(ptr.value + 4), ptr, 4, 64, None) == EINVAL
    # MV2_STORE_ROWS_F32 (fp32 rows out of a bf16 launch) is mv2_conv3's: not with fp32 operands, not with the fused second stage
    d3 = ops_hip.ConvDesc(**dict(good, store=3))
    assert lib.mv2_conv3_eligible(ctypes.byref(d3)) == 1 and lib.mv2_conv3_fusable(ctypes.byref(d3)) == 0
    assert lib.mv2_conv3_eligible(ctypes.byref(ops_hip.ConvDesc(**dict(good, store=3, dtype=1)))) == 0
    assert lib.mv2_conv3(ctypes.byref(d3), ptr, 4096, ptr, None, ptr, None, 1, None, 0.0, None, ptr, None, 0, None, 0, None) == EINVAL
    # mv2_ff: null pointers, widths other than 256 / 512, fp32, a hidden width that is not whole 64-unit blocks, in-place, a token shift whose
    # frames do not tile the rows; the sizing queries are pure host functions (64 KB / 128 KB of fp32 slab per split row tile, one ticket word per tile)
    assert lib.mv2_ff(0, None, ptr, ptr, ptr, ptr, ptr, 128, 256, 704, 0, 1, None, 0, None, 0, None) == EINVAL
    buf2 = ctypes.create_string_buffer(4096); ptr2 = ctypes.cast(buf2, ctypes.c_void_p)
    for args in ((0, 128, 384, 704, 0, 1), (1, 128, 256, 704, 0, 1), (0, 128, 256, 700, 0, 1), (0, 0, 256, 704, 0, 1), (0, 1280, 512, 1408, 256, 3)):
        dt, rows, Cc, hp, sh, Tt = args
        assert lib.mv2_ff(dt, ptr, ptr, ptr, ptr, ptr, ptr2, rows, Cc, hp, sh, Tt, None, 0, None, 0, None) == EINVAL, args
        if sh == 0:
            assert lib.mv2_ff_eligible(dt, rows, Cc, hp) == 0, args
    assert lib.mv2_ff(0, ptr, ptr, ptr, ptr, ptr, ptr, 128, 256, 704, 0, 1, None, 0, None, 0, None) == EINVAL          # x == out
    assert lib.mv2_ff_eligible(0, 20480, 256, 704) == 1 and lib.mv2_ff_eligible(0, 5120, 512, 1408) == 1
    assert lib.mv2_ff_workspace_bytes(20480, 256, 704) == 0                                        # 320 row tiles: whole tiles, no slabs
    assert lib.mv2_ff_workspace_bytes(5120, 512, 1408) == 80 * 3 * 128 * 1024                      # 80 tiles in three parts (256 CUs assumed without a device)
    assert lib.mv2_ff_counter_bytes(5120, 512, 1408) == 80 * 4
    assert lib.mv2_ff(0, ptr, ptr, ptr, ptr, ptr, ptr2, 5120, 512, 1408, 0, 1, None, 0, None, 0, None) == -3            # MV2_EWORKSPACE: split tiles need slabs + tickets


def test_launch_rules_of_the_library_match_their_restatement():
    """The emulated plan must have the launches of the HIP plan, so the two launch rules the plan asks the library about
    (mv2_conv_rms_ok, and through it which launches take the LDS-DMA pointwise kernel: mv2_conv_pointwise_dma) are restated in
    oracle/ndhwc_ops.py (CpuOps.conv_rms_ok / _pw_dma).  Both are pure host functions of the descriptor: swept here without a GPU
    (device_cus() falls back to 256, the value CpuOps assumes) over pointwise and gathered launches, both dtypes, every store / activation."""
    import itertools
    import random
    from oracle.ndhwc_ops import CpuOps
    lib, cpu = ops_hip.load_library(), CpuOps()
    rnd = random.Random(7)
    n = n_dma = n_rms = n_gather = 0
    shapes = [(1, 20, 16, 16), (1, 5, 16, 16), (1, 20, 32, 32), (4, 20, 32, 32), (1, 17, 64, 64), (1, 20, 29, 29), (2, 3, 8, 8), (1, 1, 4, 4), (1, 32, 16, 16)]
    for (B, T, H, W), Ci, co, dtype, act, store in itertools.product(shapes, (64, 128, 256, 704, 1408, 96), (64, 128, 256, 768, 2816), (0, 1), (0, 1, 2, 3), (0, 1)):
        if rnd.random() < 0.6:
            continue
        es = 2 if dtype == 0 else 4
        if (Ci * es) % 128 or (act == 3 and store == 1):
            continue
        d = dict(dtype=dtype, B=B, Ti=T, Hi=H, Wi=W, Ci=Ci, To=T, Ho=H, Wo=W, co=co, co_pad=(co + 127) // 128 * 128, st=1, sh=1, sw=1, pt=0, ph=0, pw=0,
                 kt=1, kh=1, kw=1, nsteps=Ci * es // 128, seg_bytes=128, check_bounds=0, act=act, store=store, rt=1, rh=2 if store == 1 else 1,
                 rw=2 if store == 1 else 1, crop_t=0, out_c=co // 2 if act == 3 else (co // 4 if store == 1 else co), splitk=rnd.choice((0, 0, 0, 1, 2)))
        variants = [d, dict(d, kh=3, kw=3, ph=1, pw=1, check_bounds=1, nsteps=9 * d['nsteps']), dict(d, seg_bytes=64), dict(d, To=(T + 1) // 2, st=2)]
        for v in variants:
            desc = ops_hip.ConvDesc(**v)
            dma, rms = bool(lib.mv2_conv_pointwise_dma(ctypes.byref(desc))), bool(lib.mv2_conv_rms_ok(ctypes.byref(desc)))
            assert cpu._pw_dma(v) == dma, v
            assert cpu.conv_rms_ok(v) == rms, v
            if dma:
                assert v['dtype'] == 0 and v['kh'] == 1 and v['st'] == 1 and rms and lib.mv2_conv_workspace_bytes(ctypes.byref(desc)) == 0, v
            # the gather form of that kernel (round 6; mv2_conv only, so mv2_conv_rms_ok above does not see it): never the pointwise launches,
            # never split over K, whole 128-byte segments of bf16 rows, at least 32 tiles of 128 x 128
            gather = bool(lib.mv2_conv_gather_dma(ctypes.byref(desc)))
            if gather:
                tiles = -(-v['co'] // 128) * -(-(v['B'] * v['To'] * v['Ho'] * v['Wo']) // 128)
                assert not dma and v['dtype'] == 0 and v['seg_bytes'] == 128 and (v['kh'] == 3 or v['st'] == 2) and v['act'] != 3 and v['splitk'] <= 1 and tiles >= 32, v
                assert lib.mv2_conv_workspace_bytes(ctypes.byref(desc)) == 0, v
            n += 1
            n_dma += dma
            n_rms += rms
            n_gather += gather
    assert n > 1000 and n_dma > 50 and n_rms > n_dma and n_rms < n and n_gather > 30, (n, n_dma, n_rms, n_gather)


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
    from magvit2_pytorch_amd.discriminator import discriminator_param_specs
    dnames = {'discr.' + p.name for p in discriminator_param_specs(dim=32, image_size=32, channels=3, max_dim=512)}   # reference :1413-1422
    assert set(tok.state_dict().keys()) == names | dnames
    assert 'encoder_layers.14.1.weight' in names            # the never-executed LayerNorm keeps its keys (reference :1322-1326)
    assert 'decoder_layers.0.0.fn.fn.to_qkv.0.weight' in names and 'encoder_layers.5.0.fn.attn.to_kv.0.weight' in names
    path = tmp_path / 'tok.pt'
    tok.save(str(path))
    tok2 = VideoTokenizer.init_and_load_from(str(path))
    for k, v in tok.state_dict().items():
        assert torch.equal(v, tok2.state_dict()[k])
    # a tokenizer-only state_dict loads too (the discriminator keeps its weights) ...
    before = tok2.state_dict()['discr.to_logits.3.weight'].clone()
    tok2.load_state_dict({k: v for k, v in tok.state_dict().items() if not k.startswith('discr.')})
    assert torch.equal(tok2.state_dict()['discr.to_logits.3.weight'], before)
    # ... and a tokenizer built without the GAN ignores a checkpoint's discriminator / VGG entries
    tok3 = VideoTokenizer(**kw, use_gan=False, seed=4)
    assert set(tok3.state_dict().keys()) == names
    tok3.load_state_dict(dict(tok.state_dict(), **{'vgg.features.0.weight': torch.zeros(3)}))
    assert torch.equal(tok3.state_dict()['conv_in.conv.weight'], tok.state_dict()['conv_in.conv.weight'])


def test_ctor_and_call_asserts():
    with pytest.raises(AssertionError):
        VideoTokenizer(image_size=32, layers=README_LAYERS)                          # no codebook_size (reference :1359)
    with pytest.raises(AssertionError):
        VideoTokenizer(image_size=32, layers=README_LAYERS, use_fsq=True, codebook_size=64, fsq_levels=[8, 5])
    with pytest.raises(ValueError):
        VideoTokenizer(image_size=32, layers=('residual', 'bogus'), codebook_size=64)  # reference :1312
    with pytest.raises(NotImplementedError):
        VideoTokenizer(image_size=32, layers=('residual', 'cond_linear_attend_space'), codebook_size=64, dim_cond=8)     # the reference cannot construct it either
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


def test_token_shifted_width_8_bf16_takes_the_unfused_rms(monkeypatch):
    """at bf16 width 8 a token-shifted layer's K program gathers 8 elements per half-row tap (taps are rounded up to 16-byte
    chunks), 4 of them a neighbouring row's under zero weights.  mv2_conv_rms would sum their squares into 1 / |row|; the lowering must
    take the separate row-scale launch there.  Checked against the RESTATEMENT of the reference (the emulation repeats the kernel's sum,
    so emulation-vs-HIP tests cannot see this): the bf16 drift of the encoder output must be that of the unfused path, and the lowering
    must not contain a fused launch for those layers."""
    kw = dict(image_size=16, init_dim=8, max_dim=16, codebook_size=16, layers=('residual', 'attend_time', 'residual'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=3, randomize_se=True)
    v = torch.randn(2, 3, 5, 16, 16, generator=torch.Generator().manual_seed(4))
    orc = OracleTokenizer(sd, **cfg)
    ref_codes, ref_recon = orc(v, return_codes=True, return_recon=True)
    tok = emulated_tokenizer(kw, sd, torch.bfloat16)
    codes, recon = tok(v, return_codes=True, return_recon=True)
    plan, _ = tok._plan(2, 5)
    names = plan.step_names
    steps = plan.enc_steps + plan.dec_steps
    fused = [getattr(st, 'label', '') for st in steps if getattr(st, 'what', '') == 'mv2_conv_rms']
    assert not any('qkv' in n or 'ff1' in n for n in fused), fused
    err = (recon - ref_recon).abs().mean().item()
    monkeypatch.setattr(plan_mod, 'OPTIONS', dataclasses.replace(plan_mod.OPTIONS, conv_rms=False))
    tok2 = emulated_tokenizer(kw, sd, torch.bfloat16)
    _, recon2 = tok2(v, return_codes=True, return_recon=True)
    err2 = (recon2 - ref_recon).abs().mean().item()
    assert err <= err2 * 1.02 + 1e-6, (err, err2)
    assert names


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
    monkeypatch.setattr(plan_mod, 'OPTIONS', dataclasses.replace(plan_mod.OPTIONS, se_two_launch=False))
    tok = emulated_tokenizer(kw, sd, torch.bfloat16)
    names = tok._plan(1, 5)[0].step_names
    assert 'se_pool_fc1' not in names and 'se_gate' in names
    b_codes, b_recon = tok(v, return_codes=True, return_recon=True)
    assert (a_codes != b_codes).float().mean().item() <= 0.02
    assert (a_recon - b_recon).abs().mean().item() < 5e-3
    monkeypatch.setattr(plan_mod, 'OPTIONS', dataclasses.replace(plan_mod.OPTIONS, se_two_launch=False, fuse_pool=False))
    c_codes, c_recon = emulated_tokenizer(kw, sd, torch.bfloat16)(v, return_codes=True, return_recon=True)
    assert (b_codes != c_codes).float().mean().item() <= 0.02
    assert (b_recon - c_recon).abs().mean().item() < 5e-3


def test_unsupported_widths_are_rejected_at_construction():
    """widths the kernels cannot run must fail in the constructor with a clear message, not as an opaque
    error code at the first forward."""
    for bad in (dict(channels=65), dict(channels=5, use_gan=True), dict(init_dim=12), dict(init_dim=24, layers=('attend_time',)), dict(init_dim=24, layers=('gateloop_time',)),
                dict(attn_dim_head=24), dict(linear_attn_dim_head=32)):
        with pytest.raises(NotImplementedError):
            VideoTokenizer(image_size=32, codebook_size=256, **{**dict(init_dim=8, max_dim=64, layers=README_LAYERS, use_gan=False), **bad})


@pytest.mark.parametrize('channels', [1, 4, 5, 8, 12, 20])
def test_input_channel_counts_match_oracle(channels):
    """1 .. 4 channels travel as 4-channel pixels, 5 .. 64 as pixels of the next multiple of 8 channels (conv_in's (iw, c) K extent); conv_out stores `channels` planes."""
    kw = dict(image_size=16, init_dim=16, max_dim=32, codebook_size=64, channels=channels, input_conv_kernel_size=(3, 5, 5),
              layers=('residual', 'compress_space', 'compress_time', 'residual'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=3, randomize_se=True)
    tok = emulated_tokenizer(dict(kw, use_gan=False), sd, torch.float32)
    v = torch.randn(2, channels, 5, 16, 16, generator=torch.Generator().manual_seed(4))
    ocodes, orecon = OracleTokenizer(sd, **cfg)(v, return_codes=True, return_recon=True)
    codes, recon = tok(v, return_codes=True, return_recon=True)
    assert torch.equal(codes, ocodes) and recon.shape == v.shape and (recon - orecon).abs().max().item() <= 5e-6


def test_plan_cache_is_bounded():
    """one Plan (activation arena + graphs) per input geometry must not accumulate without bound."""
    from oracle.ndhwc_ops import CpuOps
    tok = VideoTokenizer(image_size=32, init_dim=8, max_dim=32, codebook_size=256, layers=README_LAYERS, seed=0)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    tok.max_cached_plans = 2
    for b in (1, 2, 3, 1):
        tok.tokenize(torch.randn(b, 3, 5, 32, 32, generator=torch.Generator().manual_seed(b)))
        assert len(tok._plans) <= 2
    assert {p.B for p in tok._plans.values()} == {3, 1}            # least recently used (B = 2) was dropped


# ------------------------------------------------------------------------------- forward(return_loss=True): host logic on the CPU emulation
LOSS_KW = dict(image_size=32, init_dim=16, max_dim=64, codebook_size=256, layers=README_LAYERS, use_gan=False, perceptual_loss_weight=0.)


@pytest.mark.parametrize('training', [True, False])
def test_forward_return_loss_matches_oracle(training):
    kw = {k: v for k, v in LOSS_KW.items() if k not in ('use_gan', 'perceptual_loss_weight')}
    sd = init_state_dict(make_cfg(**kw), seed=5, randomize_se=True)
    tok = VideoTokenizer(**LOSS_KW, seed=0)
    tok.load_state_dict(sd)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    tok.train(training)
    v = torch.randn(2, 3, 5, 32, 32, generator=torch.Generator().manual_seed(6))
    total, br = tok(v, return_loss=True)
    ototal, d = OracleTokenizer(sd, **make_cfg(**kw)).forward_loss(v, training=training)
    assert br._fields == ('recon_loss', 'lfq_aux_loss', 'quantizer_loss_breakdown', 'perceptual_loss', 'adversarial_gen_loss',
                          'adaptive_adversarial_weight', 'multiscale_gen_losses', 'multiscale_gen_adaptive_weights')
    assert abs(float(total) - float(ototal)) <= 1e-5 * max(1., abs(float(ototal)))
    assert abs(float(br.recon_loss) - float(d['recon_loss'])) <= 1e-5
    assert abs(float(br.lfq_aux_loss) - float(d['aux'])) <= 1e-5
    qb = br.quantizer_loss_breakdown
    for name in ('per_sample_entropy', 'batch_entropy', 'commitment'):
        assert abs(float(getattr(qb, name)) - float(d[name])) <= 1e-5, name
    assert (float(br.lfq_aux_loss) != 0.) == training
    loss, recon = tok(v, return_recon_loss_only=True)
    assert recon.shape == v.shape and abs(float(loss) - float((recon - v).pow(2).mean())) <= 1e-6


def test_forward_return_loss_large_codebook_is_refused_in_train_mode():
    tok = VideoTokenizer(image_size=16, init_dim=16, max_dim=16, codebook_size=2 ** 13, layers=('residual',), use_gan=False, perceptual_loss_weight=0., seed=0)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    v = torch.zeros(1, 3, 1, 16, 16)
    tok.train()
    with pytest.raises(NotImplementedError):
        tok(v, return_loss=True)
    tok.eval()
    total, br = tok(v, return_loss=True)                            # eval mode: no codebook enumeration, aux = 0
    assert float(br.lfq_aux_loss) == 0.


def test_forward_return_loss_refuses_gan_and_perceptual_terms():
    tok = VideoTokenizer(image_size=32, init_dim=16, max_dim=64, codebook_size=256, layers=README_LAYERS, seed=0)   # reference defaults: GAN + VGG
    tok._set_backend(CpuOps())
    with pytest.raises(NotImplementedError):
        tok(torch.zeros(1, 3, 5, 32, 32), return_loss=True)
    with pytest.raises(NotImplementedError):
        tok(torch.zeros(1, 3, 5, 32, 32), return_discr_loss=True)


# ------------------------------------------------------------------------------- conditioned layers: plan lowering on the CPU emulation
@pytest.mark.parametrize('layers,factor', [
    (('residual', 'compress_space', 'cond_residual', 'cond_residual'), 4.),
    (('residual', 'compress_space', 'cond_attend_space', 'compress_time', 'cond_attend_time', 'cond_residual'), 1.)])
def test_conditioned_layers_match_oracle(layers, factor):
    kw = dict(image_size=16, init_dim=16, max_dim=32, codebook_size=64, dim_cond=8, dim_cond_expansion_factor=factor, layers=layers)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=3, randomize_se=True)
    tok = emulated_tokenizer(kw, sd, torch.float32)
    v = torch.randn(2, 3, 5, 16, 16, generator=torch.Generator().manual_seed(1))
    cond = torch.randn(2, 8, generator=torch.Generator().manual_seed(2))
    orc = OracleTokenizer(sd, **cfg)
    ocodes, orecon = orc(v, cond=cond, return_codes=True, return_recon=True)
    codes, recon = tok(v, cond=cond, return_codes=True, return_recon=True)
    assert torch.equal(codes, ocodes) and (recon - orecon).abs().max().item() <= 5e-6
    assert (tok.decode_from_code_indices(ocodes, cond=cond) - orecon).abs().max().item() <= 5e-6
    assert torch.equal(tok.tokenize(v, cond=cond), ocodes)
    with pytest.raises(AssertionError):
        tok(v, return_codes=True)                                   # reference :1542
    with pytest.raises(AssertionError):
        tok(v, cond=torch.zeros(2, 9), return_codes=True)           # reference :1545


def test_conditioned_ctor_rules():
    base = dict(image_size=16, init_dim=16, max_dim=32, codebook_size=64)
    with pytest.raises(AssertionError):
        VideoTokenizer(**base, layers=('residual', 'cond_residual'))                                  # dim_cond missing, reference :1151
    with pytest.raises(ValueError):
        VideoTokenizer(**base, dim_cond=8, layers=('residual', 'cond_attend_space'))                  # expansion factor must be 1
    with pytest.raises(NotImplementedError):
        VideoTokenizer(**base, dim_cond=8, layers=('residual', 'cond_linear_attend_space'))           # the reference cannot construct it
    tok = VideoTokenizer(**base, dim_cond=8, layers=('residual', 'cond_residual'), seed=0)
    assert tok.has_cond and 'encoder_cond_in.0.weight' in tok.state_dict() and 'encoder_layers.1.conv.weights' in tok.state_dict()


# ------------------------------------------------------------------------------- weight images through the C ABI (host-side packer)
@pytest.mark.parametrize('shape,frag', [((64, 64, 3, 3, 3), True), ((128, 64, 3, 3, 3), True), ((96, 128, 1, 3, 3), True), ((512, 512, 1, 1, 1), True),
                                        ((64, 64, 3, 3, 3), False), ((40, 24, 3, 3, 3), False), ((3, 64, 3, 3, 3), False), ((256, 128, 1, 3, 3), False),
                                        ((128, 64, 3, 1, 1), False), ((30, 10, 1, 1, 1), False)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_c_packer_matches_python_packer(shape, frag, dtype):
    """mv2_pack_conv3d (host C++ behind the ABI) produces byte for byte the image pack.py builds — the layout include/mv2.h documents."""
    import ctypes as C
    from magvit2_pytorch_amd import pack as PK
    from magvit2_pytorch_amd.ops_hip import load_library
    lib = load_library()
    w = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape)))
    if frag and not PK.conv3_packable(w, dtype):
        frag_ok = (shape[1] * PK.esize(dtype)) % 128 == 0
        if not frag_ok:
            pytest.skip('not a fragment-major shape')
    pc = PK.pack_conv3d(w, None, dtype)
    want = PK.to_frag_major(pc.wp) if frag else pc.wp
    dt = PK.dtype_code(dtype)
    n = lib.mv2_pack_conv3d_bytes(dt, *shape)
    assert n == want.numel() * want.element_size()
    buf = (C.c_ubyte * n)()
    wc = w.contiguous()
    assert lib.mv2_pack_conv3d(dt, wc.data_ptr(), *shape, int(frag), C.addressof(buf), n) == 0
    got = torch.frombuffer(buf, dtype=torch.uint8)
    assert torch.equal(got, want.contiguous().view(torch.uint8).reshape(-1))
    assert lib.mv2_pack_conv3d(dt, wc.data_ptr(), *shape, int(frag), C.addressof(buf), n - 1) != 0        # short buffer


def test_c_packer_known_bytes():
    """A hand-checkable vector (the one INTEGRATION.md prints): Co = 2, Ci = 8, one tap, bf16, w[n][c] = n * 8 + c + 1."""
    import ctypes as C
    from magvit2_pytorch_amd.ops_hip import load_library
    lib = load_library()
    w = (torch.arange(16, dtype=torch.float32) + 1).reshape(2, 8, 1, 1, 1).contiguous()
    n = lib.mv2_pack_conv3d_bytes(0, 2, 8, 1, 1, 1)
    assert n == 1 * 128 * 128                                        # one K step, 128 padded rows, 128 bytes each
    buf = (C.c_ubyte * n)()
    assert lib.mv2_pack_conv3d(0, w.data_ptr(), 2, 8, 1, 1, 1, 0, C.addressof(buf), n) == 0
    img = torch.frombuffer(buf, dtype=torch.int16).reshape(1, 128, 64)
    # 8 channels = one 16-byte segment: 8 segments share a 128-byte step, this conv has one -> elements 0..7 of each row
    assert img[0, 0, :8].view(torch.bfloat16).float().tolist() == [1., 2., 3., 4., 5., 6., 7., 8.]
    assert img[0, 1, :8].view(torch.bfloat16).float().tolist() == [9., 10., 11., 12., 13., 14., 15., 16.]
    assert int(img[0, 0, 0]) == 0x3F80 and int(img[0, 1, 0]) == 0x4110       # bf16(1.0), bf16(9.0)
    assert not img[0, 2:].any() and not img[0, :2, 8:].any()


def test_separate_first_frame_encoding_matches_oracle():
    kw = dict(image_size=16, init_dim=16, max_dim=32, codebook_size=64, separate_first_frame_encoding=True,
              layers=('residual', 'compress_space', 'compress_time', 'residual', 'compress_time', 'attend_time'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=4, randomize_se=True)
    tok = emulated_tokenizer(kw, sd, torch.float32)
    v = torch.randn(2, 3, 9, 16, 16, generator=torch.Generator().manual_seed(1))
    ocodes, orecon = OracleTokenizer(sd, **cfg)(v, return_codes=True, return_recon=True)
    codes, recon = tok(v, return_codes=True, return_recon=True)
    assert torch.equal(codes, ocodes) and (recon - orecon).abs().max().item() <= 5e-6
    assert (tok.decode_from_code_indices(ocodes) - orecon).abs().max().item() <= 5e-6
    q, _, _ = tok.encode(v, quantize=True)
    assert (tok.decode(q) - orecon).abs().max().item() <= 5e-6
    assert 'conv_in_first_frame.weight' in tok.state_dict() and tok.state_dict()['conv_out_first_frame.weight'].shape == (3, 16, 3, 3)
    with pytest.raises(AssertionError):
        tok(torch.randn(2, 3, 16, 16), return_codes=True)           # a single frame: the reference fails too (conv_in on an empty clip)


# ------------------------------------------------------------------------------- discriminator: lowering on the CPU emulation
def test_discriminator_lowering_matches_oracle():
    from magvit2_pytorch_amd.discriminator import Discriminator
    from oracle.restate import discriminator_forward
    kw = dict(dim=8, image_size=16, channels=3, max_dim=64)
    d = Discriminator(**kw, seed=2)
    d._set_backend(CpuOps())
    d.compute_dtype = torch.float32
    x = torch.randn(3, 3, 16, 16, generator=torch.Generator().manual_seed(3))
    want = discriminator_forward(x, {'discr.' + k: v for k, v in d.state_dict().items()}, 'discr.', **kw)
    got = d(x)
    assert got.shape == (3,) and got.dtype == torch.float32 and (got - want).abs().max().item() <= 2e-6
    with pytest.raises(NotImplementedError):
        Discriminator(**kw, antialiased_downsample=True)


def test_discriminator_non_power_of_two_size_and_default_tokenizer_ctor():
    """the reference discriminator takes any size (num_layers = int(log2 n - 2), to_logits flattens an n // 2**L map, M:607,
    654-657): 24 -> 2 blocks ending on 6 x 6.  A default-constructed tokenizer (use_gan=True) at image_size 96 must construct."""
    from magvit2_pytorch_amd.discriminator import Discriminator
    from oracle.restate import discriminator_forward
    kw = dict(dim=8, image_size=24, channels=3, max_dim=64)
    d = Discriminator(**kw, seed=4)
    assert d.state_dict()['to_logits.3.weight'].shape == (1, 64 * 36)
    d._set_backend(CpuOps())
    d.compute_dtype = torch.float32
    x = torch.randn(2, 3, 24, 24, generator=torch.Generator().manual_seed(5))
    want = discriminator_forward(x, {'discr.' + k: v for k, v in d.state_dict().items()}, 'discr.', **kw)
    assert (d(x) - want).abs().max().item() <= 2e-6
    tok = VideoTokenizer(image_size=96, init_dim=8, max_dim=32, codebook_size=64, layers=('residual', 'compress_space'), seed=0)
    assert tok.discr is not None and tok.state_dict()['discr.to_logits.3.weight'].shape[1] % 36 == 0
    bad = Discriminator(dim=8, image_size=22, channels=3, max_dim=64, seed=0)      # 22 -> 11 -> odd: the reference fails in forward too
    bad._set_backend(CpuOps())
    with pytest.raises(ValueError):
        bad(torch.zeros(1, 3, 22, 22))


def test_forward_return_discr_loss_matches_oracle():
    kw = dict(image_size=16, init_dim=8, max_dim=16, codebook_size=64, layers=('residual', 'compress_space', 'compress_time', 'residual'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=5, randomize_se=True)
    tok = emulated_tokenizer(kw, sd, torch.float32)
    dsd = {k: v.clone() for k, v in tok.state_dict().items() if k.startswith('discr.')}
    v = torch.randn(3, 3, 5, 16, 16, generator=torch.Generator().manual_seed(6))
    torch.manual_seed(77)
    total, br = tok(v, return_discr_loss=True, apply_gradient_penalty=False)
    torch.manual_seed(77)
    want, _, _ = OracleTokenizer(sd, **cfg).forward_discr_loss(v, dsd, dim=16, max_dim=512)
    assert br._fields == ('discr_loss', 'multiscale_discr_losses', 'gradient_penalty')
    assert abs(float(total) - float(want)) <= 2e-6 and float(br.gradient_penalty) == 0.
    with pytest.raises(NotImplementedError):
        tok(v, return_discr_loss=True)                               # gradient penalty (the reference's default) needs autograd


def test_c_consumer_builds_and_runs(tmp_path):
    """include/mv2.h is a plain C99 header and the library is usable from C without Python (the drop-in boundary of the brief):
    tests/abi/abi_smoke.c is compiled with gcc against the header, linked with libmv2_hip.so and run (no kernel launch)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('gcc not available')
    lib_dir = os.path.join(ROOT, 'magvit2_pytorch_amd')
    assert os.path.exists(os.path.join(lib_dir, 'libmv2_hip.so')), 'build the library first (python -m magvit2_pytorch_amd.build)'
    exe = str(tmp_path / 'abi_smoke')
    hdr = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-x', 'c',
                          os.path.join(ROOT, 'include', 'mv2.h')], capture_output=True, text=True)
    assert hdr.returncode == 0, hdr.stderr
    cc = subprocess.run([gcc, '-std=c99', '-Wall', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'abi', 'abi_smoke.c'),
                         '-L', lib_dir, '-l:libmv2_hip.so', '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,/opt/rocm/lib', '-o', exe],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and 'abi smoke ok' in run.stdout, (run.returncode, run.stdout, run.stderr)


def test_c_launch_consumer_host_half(tmp_path):
    """tests/abi/abi_launch.c (the plain-C consumer that launches mv2_conv and mv2_lfq under -m gpu): its host half - test vectors, the C
    arithmetic that produces the expected buffers and their fixture checksums, the packing and sizing calls - builds with gcc as C99 and runs
    without a GPU (-DMV2_ABI_LAUNCH_HOST_ONLY)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('gcc not available')
    lib_dir = os.path.join(ROOT, 'magvit2_pytorch_amd')
    exe = str(tmp_path / 'abi_launch_host')
    cc = subprocess.run([gcc, '-std=c99', '-Wall', '-DMV2_ABI_LAUNCH_HOST_ONLY', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'abi', 'abi_launch.c'),
                         '-L', lib_dir, '-l:libmv2_hip.so', '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,/opt/rocm/lib', '-lm', '-o', exe], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and 'host half) ok' in run.stdout, (run.returncode, run.stdout, run.stderr)


# ------------------------------------------------------------------------------- facade (CPU emulation of the kernels) vs the reference's own outputs
def test_facade_vs_reference_feature_fixtures():
    """tests/golden/feat_*.npz hold outputs of the reference itself for the paths beyond eval encode / decode (oracle/make_golden.py
    FEATURES); the facade's lowering, run on the kernel emulation in fp32, must reproduce them.  The GPU twin of this test is
    tests/test_losses_gpu.py::test_product_vs_reference_feature_fixtures."""
    from magvit2_pytorch_amd.discriminator import Discriminator
    from oracle.make_golden import make_input
    from tests._util import Feature
    for name in ('feat_loss', 'feat_loss_2cb'):
        f = Feature(name)
        tok = VideoTokenizer(**f.kw, use_gan=False, perceptual_loss_weight=0., seed=0)
        tok.load_state_dict(f.weights())
        tok._set_backend(CpuOps())
        tok.compute_dtype = torch.float32
        for mode in ('train', 'eval'):
            tok.train(mode == 'train')
            total, br = tok(f.video(), return_loss=True)
            assert abs(float(total) - f.scalar(f'{mode}_total')) <= 1e-5 * max(1., abs(f.scalar(f'{mode}_total'))), (name, mode)
            assert abs(float(br.lfq_aux_loss) - f.scalar(f'{mode}_aux')) <= 1e-5
    for name in ('feat_cond_residual', 'feat_sep_first_frame', 'feat_pad_reflect', 'feat_pad_replicate', 'feat_attn_dh16', 'feat_attn_dh64', 'feat_linattn_dh16', 'feat_linattn_dh4'):
        f = Feature(name)
        tok = emulated_tokenizer(dict(f.kw, use_gan=False), f.weights(), torch.float32)
        ckw = dict(cond=f.cond()) if f.cond() is not None else {}
        codes, recon = tok(f.video(), return_codes=True, return_recon=True, **ckw)
        assert torch.equal(codes, torch.from_numpy(f.z['codes']))
        assert (recon - torch.from_numpy(f.z['recon'])).abs().max().item() <= 2e-5
    f = Feature('feat_discr')
    d = Discriminator(dim=16, image_size=16, channels=3, max_dim=512, seed=0)
    d.load_state_dict({k[len('discr.'):]: v for k, v in f.discr_weights().items()})
    d._set_backend(CpuOps())
    d.compute_dtype = torch.float32
    logits = d(make_input((3, 3, 16, 16), f.f['dseed'] + 1))
    assert (logits - torch.from_numpy(f.z['logits'])).abs().max().item() <= 5e-6
    tok = VideoTokenizer(**f.kw, seed=0)
    tok.load_state_dict({**f.weights(), **f.discr_weights()})
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    torch.manual_seed(f.f['pick_seed'])
    total, _ = tok(f.video(), return_discr_loss=True, apply_gradient_penalty=False)
    assert abs(float(total) - f.scalar('discr_loss')) <= 1e-5


# ------------------------------------------------------------------------------- headless clips, GAN-side loss terms, multiscale discriminators
SEP_KW = dict(image_size=16, init_dim=16, max_dim=32, codebook_size=64,
              layers=('residual', 'compress_space', 'compress_time', 'residual', 'compress_time', 'attend_time'))
GAN_KW = dict(image_size=16, init_dim=8, max_dim=16, codebook_size=64, layers=('residual', 'compress_space', 'compress_time', 'residual'))


@pytest.mark.parametrize('sep', [False, True])
def test_headless_video_matches_oracle(sep):
    """video_contains_first_frame=False on every method (reference M:1530-1538, 1604, 1646, 1691): a plan without the
    time_padding zero frames and without the output crop; separate_first_frame_encoding does not apply to such clips."""
    kw = dict(SEP_KW, separate_first_frame_encoding=sep)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=21, randomize_se=True)
    tok = emulated_tokenizer(kw, sd, torch.float32)
    orc = OracleTokenizer(sd, **cfg)
    v = torch.randn(2, 3, 8, 16, 16, generator=torch.Generator().manual_seed(22))
    ocodes, orecon = orc(v, return_codes=True, return_recon=True, video_contains_first_frame=False)
    codes, recon = tok(v, return_codes=True, return_recon=True, video_contains_first_frame=False)
    assert codes.shape == (2, 2, 8, 8) and torch.equal(codes, ocodes) and (recon - orecon).abs().max().item() <= 5e-6
    assert (tok.decode_from_code_indices(ocodes, video_contains_first_frame=False) - orecon).abs().max().item() <= 5e-6
    assert (tok.decode_from_code_indices(ocodes.reshape(2, -1), video_contains_first_frame=False) - orecon).abs().max().item() <= 5e-6
    q, ids, _ = tok.encode(v, quantize=True, video_contains_first_frame=False)
    assert torch.equal(ids, ocodes) and (tok.decode(q, video_contains_first_frame=False) - orecon).abs().max().item() <= 5e-6
    # the same tokenizer still serves clips WITH a first frame (separate plans per mode)
    v9 = torch.randn(2, 3, 9, 16, 16, generator=torch.Generator().manual_seed(23))
    c9, r9 = tok(v9, return_codes=True, return_recon=True)
    oc9, or9 = orc(v9, return_codes=True, return_recon=True)
    assert torch.equal(c9, oc9) and (r9 - or9).abs().max().item() <= 5e-6
    with pytest.raises(AssertionError):
        tok(v[:, :, :7], return_codes=True, video_contains_first_frame=False)          # 7 frames: not divisible by 4 (reference :1691)
    with pytest.raises(AssertionError):
        tok(v, return_codes=True)                                                      # 8 frames WITH a first frame: (8 - 1) % 4 != 0
    img = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(24))
    if not sep:      # an image always contains its first frame, whatever the flag says (reference :1681-1685)
        assert torch.equal(tok(img, return_codes=True, video_contains_first_frame=False), orc(img, return_codes=True))


@pytest.mark.parametrize('frames,vcff', [(9, True), (8, False)])
def test_forward_loss_separate_first_frame_matches_oracle(frames, vcff):
    kw = dict(SEP_KW, separate_first_frame_encoding=True)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=4, randomize_se=True)
    tok = VideoTokenizer(**kw, use_gan=False, perceptual_loss_weight=0., seed=0)
    tok.load_state_dict(sd)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    tok.train(True)
    v = torch.randn(2, 3, frames, 16, 16, generator=torch.Generator().manual_seed(1))
    want, d = OracleTokenizer(sd, **cfg).forward_loss(v, training=True, video_contains_first_frame=vcff)
    total, br = tok(v, return_loss=True, video_contains_first_frame=vcff)
    assert abs(float(total) - float(want)) <= 1e-5 * max(1., abs(float(want)))
    assert abs(float(br.recon_loss) - float(d['recon_loss'])) <= 2e-6
    rl, recon = tok(v, return_recon_loss_only=True, video_contains_first_frame=vcff)
    assert abs(float(rl) - float(d['recon_loss'])) <= 2e-6 and recon.shape == v.shape


@pytest.mark.parametrize('training', [True, False])
def test_forward_loss_with_gan_matches_oracle(training):
    """use_gan=True, no VGG: total = recon + aux + adversarial_loss_weight * (-mean D(random reconstructed frame)) (reference M:1826-1832, 1872-1875)"""
    cfg = make_cfg(**GAN_KW)
    sd = init_state_dict(cfg, seed=5, randomize_se=True)
    tok = VideoTokenizer(**GAN_KW, use_gan=True, perceptual_loss_weight=0., adversarial_loss_weight=0.7, seed=0)
    tok.load_state_dict(sd)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    tok.train(training)
    dsd = {k: v.clone() for k, v in tok.state_dict().items() if k.startswith('discr.')}
    v = torch.randn(3, 3, 5, 16, 16, generator=torch.Generator().manual_seed(6))
    torch.manual_seed(78)
    total, br = tok(v, return_loss=True)
    torch.manual_seed(78)
    want, d = OracleTokenizer(sd, **cfg).forward_loss(v, training=training, discr_params=dsd, discr_kw=dict(dim=16, max_dim=512), adversarial_loss_weight=0.7)
    assert abs(float(total) - float(want)) <= 1e-5 * max(1., abs(float(want))), (float(total), float(want))
    assert abs(float(br.adversarial_gen_loss) - float(d['gen_loss'])) <= 5e-6 and float(br.adversarial_gen_loss) != 0. and br.adaptive_adversarial_weight == 1.
    torch.manual_seed(78)
    total2, _ = tok(v, return_loss=True, adversarial_loss_weight=0.)                    # per-call override (reference :1671)
    assert abs(float(total2) - (float(br.recon_loss) + float(br.lfq_aux_loss))) <= 2e-6


def test_multiscale_discriminators_and_vgg_match_oracle():
    from tests._util import ClipDiscr
    cfg = make_cfg(**GAN_KW)
    sd = init_state_dict(cfg, seed=5, randomize_se=True)
    ms = (ClipDiscr(1), ClipDiscr(2))
    vgg = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, stride=2), torch.nn.ReLU(), torch.nn.Flatten())
    tok = VideoTokenizer(**GAN_KW, use_gan=True, vgg=vgg, perceptual_loss_weight=0.3, multiscale_discrs=ms, multiscale_adversarial_loss_weight=0.5, seed=0)
    keys = set(tok.state_dict())
    assert {'multiscale_discrs.0.w', 'multiscale_discrs.1.w'} <= keys and not any(k.startswith('vgg.') for k in keys)     # reference: @remove_vgg
    tok.load_state_dict(sd)                                                                                # tokenizer-only checkpoint: the caller's modules keep their weights
    assert torch.equal(tok.multiscale_discrs[1].w, ms[1].w)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    tok.eval()
    dsd = {k: v.clone() for k, v in tok.state_dict().items() if k.startswith('discr.')}
    v = torch.randn(3, 3, 5, 16, 16, generator=torch.Generator().manual_seed(6))
    orc = OracleTokenizer(sd, **cfg)
    torch.manual_seed(79)
    total, br = tok(v, return_discr_loss=True, apply_gradient_penalty=False)
    torch.manual_seed(79)
    want, _, _ = orc.forward_discr_loss(v, dsd, dim=16, max_dim=512, multiscale_discrs=ms, multiscale_adversarial_loss_weight=0.5)
    assert len(br.multiscale_discr_losses) == 2 and abs(float(total) - float(want)) <= 5e-6
    torch.manual_seed(80)
    total, br = tok(v, return_loss=True)
    torch.manual_seed(80)
    want, d = orc.forward_loss(v, training=False, discr_params=dsd, discr_kw=dict(dim=16, max_dim=512), multiscale_discrs=ms,
                               multiscale_adversarial_loss_weight=0.5, vgg=vgg, perceptual_loss_weight=0.3)
    assert abs(float(total) - float(want)) <= 1e-5 * max(1., abs(float(want))), (float(total), float(want))
    assert abs(float(br.perceptual_loss) - float(d['perceptual'])) <= 5e-6 and float(br.perceptual_loss) > 0.
    assert len(br.multiscale_gen_losses) == 2 and br.multiscale_gen_adaptive_weights == [1., 1.]
    tok.train(True)
    with pytest.raises(NotImplementedError):
        tok(v, return_loss=True)                 # training + VGG + GAN: gradient-norm adaptive weights (autograd)


def _widened_feature_checks(make_tok, to_dev, tol_total, tol_recon, ids_frac):
    """shared by the CPU-emulation test below and its GPU twin (tests/test_zz_widen_gpu.py): the facade against the reference's own
    outputs for headless clips, the GAN / VGG / multiscale loss terms and the losses under separate_first_frame_encoding."""
    from oracle.make_golden import ClipDiscr, tiny_vgg
    from tests._util import Feature
    for name in ('feat_headless', 'feat_headless_sep'):
        f = Feature(name)
        tok = make_tok(dict(f.kw, use_gan=False), f.weights())
        codes, recon = tok(to_dev(f.video()), return_codes=True, return_recon=True, video_contains_first_frame=False)
        want = torch.from_numpy(f.z['codes'])
        assert (codes.cpu() != want).float().mean().item() <= ids_frac, name
        if torch.equal(codes.cpu(), want):
            assert (recon.cpu() - torch.from_numpy(f.z['recon'])).abs().max().item() <= tol_recon, name
        dec = tok.decode_from_code_indices(to_dev(want), video_contains_first_frame=False)
        assert (dec.cpu() - torch.from_numpy(f.z['recon'])).abs().max().item() <= tol_recon, name
    f = Feature('feat_gan_loss')
    g = f.f
    rel = lambda got, key: abs(float(got) - f.scalar(key)) <= tol_total * max(1., abs(f.scalar(key)))
    tok = make_tok(dict(f.kw, use_gan=True, perceptual_loss_weight=0., adversarial_loss_weight=g['adversarial_loss_weight']), {**f.weights(), **f.discr_weights()})
    for mode in ('train', 'eval'):
        tok.train(mode == 'train')
        torch.manual_seed(g['pick_seed'])
        total, br = tok(to_dev(f.video()), return_loss=True)
        assert rel(total, f'gan_{mode}_total') and rel(br.adversarial_gen_loss, f'gan_{mode}_gen') and rel(br.recon_loss, f'gan_{mode}_recon'), mode
    ms = tuple(ClipDiscr(s) for s in g['ms_seeds'])
    tok = make_tok(dict(f.kw, use_gan=True, vgg=tiny_vgg(g['vgg_seed']), perceptual_loss_weight=g['perceptual_loss_weight'], multiscale_discrs=ms,
                        adversarial_loss_weight=g['adversarial_loss_weight'], multiscale_adversarial_loss_weight=g['multiscale_adversarial_loss_weight']),
                   {**f.weights(), **f.discr_weights()})
    tok.eval()
    torch.manual_seed(g['pick_seed'] + 1)
    total, br = tok(to_dev(f.video()), return_loss=True)
    assert rel(total, 'full_total') and rel(br.perceptual_loss, 'full_perceptual') and rel(br.adversarial_gen_loss, 'full_gen')
    assert all(abs(float(a) - float(b)) <= tol_total for a, b in zip(br.multiscale_gen_losses, f.z['full_ms_gen']))
    torch.manual_seed(g['pick_seed'] + 2)
    total, br = tok(to_dev(f.video()), return_discr_loss=True, apply_gradient_penalty=False)
    assert rel(total, 'full_discr_total') and rel(br.discr_loss, 'full_discr')
    assert all(abs(float(a) - float(b)) <= tol_total for a, b in zip(br.multiscale_discr_losses, f.z['full_ms_discr']))
    f = Feature('feat_sep_loss')
    tok = make_tok(dict(f.kw, use_gan=False, perceptual_loss_weight=0.), f.weights())
    for mode in ('train', 'eval'):
        tok.train(mode == 'train')
        total, br = tok(to_dev(f.video()), return_loss=True)
        assert abs(float(total) - f.scalar(f'{mode}_total')) <= tol_total * max(1., abs(f.scalar(f'{mode}_total'))), mode
        rl, _ = tok(to_dev(f.video()), return_recon_loss_only=True)
        assert abs(float(rl) - f.scalar(f'{mode}_recon')) <= tol_total * f.scalar(f'{mode}_recon')


def test_facade_vs_reference_widened_feature_fixtures():
    def make_tok(kw, sd):
        tok = VideoTokenizer(**kw, seed=0)
        tok.load_state_dict(sd)
        tok._set_backend(CpuOps())
        tok.compute_dtype = torch.float32
        return tok
    _widened_feature_checks(make_tok, lambda t: t, tol_total=1e-5, tol_recon=2e-5, ids_frac=0.)


def test_checkpoint_surface_parameters_and_copy_for_eval(tmp_path):
    """reference :1459-1488: parameters() are the tokenizer's own, discr_parameters() the discriminator's, the VGG never enters a
    state_dict, copy_for_eval() drops discriminator / VGG / multiscale discriminators and still encodes / decodes."""
    from tests._util import ClipDiscr, tiny_vgg
    cfg = make_cfg(**GAN_KW)
    sd = init_state_dict(cfg, seed=5, randomize_se=True)
    tok = VideoTokenizer(**GAN_KW, use_gan=True, vgg=tiny_vgg(3), perceptual_loss_weight=0.3, multiscale_discrs=(ClipDiscr(1),), seed=0)
    tok.load_state_dict(sd)
    keys = set(tok.state_dict())
    assert not any(k.startswith('vgg.') for k in keys) and 'multiscale_discrs.0.w' in keys and any(k.startswith('discr.') for k in keys)
    names = {n for n, _ in tok.named_parameters()}
    own = tok.parameters()
    assert isinstance(own, list) and len(own) == len([n for n in names if not n.startswith(('discr.', 'vgg.', 'multiscale_discrs.'))])
    assert len(list(tok.discr_parameters())) == len([n for n in names if n.startswith('discr.')]) > 0
    tok.save(str(tmp_path / 'a.pt'))
    pkg = torch.load(str(tmp_path / 'a.pt'), weights_only=False)
    assert not any(k.startswith('vgg.') for k in pkg['model_state_dict'])
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    v = torch.randn(1, 3, 5, 16, 16, generator=torch.Generator().manual_seed(6))
    codes = tok(v, return_codes=True)
    dup = tok.copy_for_eval()
    assert dup.discr is None and dup.vgg is None and len(dup.multiscale_discrs) == 0 and not dup.training
    assert not any(k.startswith(('discr.', 'vgg.', 'multiscale_discrs.')) for k in dup.state_dict())
    dup._set_backend(CpuOps())
    dup.compute_dtype = torch.float32
    assert torch.equal(dup(v, return_codes=True), codes)
    assert tok.discr is not None and tok.vgg is not None and tok._ops is not None        # the original is untouched


# ------------------------------------------------------------------------------- randomized layer specs: lowering vs oracle
def _random_case(rng):
    size = rng.choice([8, 16])
    layers, fm, ntime = [], size, 0
    for _ in range(rng.randint(1, 5)):
        k = rng.choice(['residual', 'residual', 'compress_space', 'compress_time', 'attend_space', 'linear_attend_space', 'attend_time', 'consecutive_residual'])
        if k == 'compress_space':
            if fm <= 2:
                continue
            fm //= 2
        if k == 'compress_time':
            if ntime >= 2:
                continue
            ntime += 1
        if k == 'consecutive_residual':
            layers.append((k, rng.randint(1, 2)))
        elif k in ('compress_space', 'compress_time') and rng.random() < 0.3:
            layers.append((k, rng.choice([16, 32])))          # explicit output width (reference :1186, 1203)
        else:
            layers.append(k)
    kw = dict(image_size=size, init_dim=rng.choice([8, 16, 32]), max_dim=rng.choice([32, 64]), layers=tuple(layers or ['residual']),
              channels=rng.choice([1, 3, 3, 4]), residual_conv_kernel_size=rng.choice([3, 3, 5]),
              input_conv_kernel_size=rng.choice([(7, 7, 7), (3, 3, 3), (5, 5, 5), (1, 3, 3), (3, 5, 5)]),
              output_conv_kernel_size=rng.choice([(3, 3, 3), (1, 1, 1), (5, 5, 5), (3, 1, 1)]))
    if rng.random() < 0.3:
        kw.update(use_fsq=True, fsq_levels=rng.choice([[8, 5, 5, 5], [4, 4], [7, 5, 5, 5, 5], [3]]))
    else:
        kw.update(codebook_size=rng.choice([4, 64, 256]), num_codebooks=rng.choice([1, 1, 2, 3]))
    if rng.random() < 0.25:
        kw['separate_first_frame_encoding'] = True
    tdf = 2 ** ntime
    vcff = rng.random() < 0.7
    T = (rng.randint(0, 2) * tdf + 1) if vcff else rng.randint(1, 2) * tdf
    if kw.get('separate_first_frame_encoding') and vcff and T < 2:
        T = tdf + 1
    return kw, rng.randint(1, 2), T, vcff


@pytest.mark.parametrize('seed', range(24))
def test_random_specs(seed):
    """Randomized layer specs / kernel sizes / channel counts / quantizers / clip modes: the plan lowering on the kernel emulation must
    reproduce the oracle (ids equal, reconstruction to fp32 round-off).  A sweep of 150 such cases found one bug — in the ORACLE
    (identity projections when the latent width equals the code width) — and none in the lowering."""
    import random
    kw, B, T, vcff = _random_case(random.Random(1000 + seed))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=seed, randomize_se=True)
    tok = VideoTokenizer(**kw, use_gan=False, seed=0)
    tok.load_state_dict(sd)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    v = torch.randn(B, kw['channels'], T, kw['image_size'], kw['image_size'], generator=torch.Generator().manual_seed(100 + seed))
    orc = OracleTokenizer(sd, **cfg)
    orc.taps = {}
    ocodes, orecon = orc(v, return_codes=True, return_recon=True, video_contains_first_frame=vcff)
    codes, recon = tok(v, return_codes=True, return_recon=True, video_contains_first_frame=vcff)
    # an id may differ only where the oracle's own pre-quantisation activation sits inside the fp32 summation-order noise band
    # around a decision boundary (seen once in 150 cases: -1.5e-8 against +6e-8)
    pre = orc.taps['prequant']
    if kw.get('use_fsq'):
        margin = (0.5 - (pre - pre.round()).abs()).amin(dim=-1)
    else:
        ncb = kw['num_codebooks']
        margin = pre.abs().reshape(*pre.shape[:-1], ncb, -1).amin(dim=-1)
        margin = margin.squeeze(-1) if ncb == 1 else margin
    diff = codes != ocodes
    assert not (diff & (margin > 1e-5)).any(), kw
    if not diff.any():
        assert (recon - orecon).abs().max().item() <= 2e-5, kw
    assert (tok.decode_from_code_indices(ocodes, video_contains_first_frame=vcff) - orecon).abs().max().item() <= 2e-5, kw


# ------------------------------------------------------------------------------- pad_mode
@pytest.mark.parametrize('pad_mode', ['reflect', 'replicate', 'circular'])
@pytest.mark.parametrize('sep', [False, True])
def test_pad_mode_matches_oracle(pad_mode, sep):
    """pad_mode reaches conv_in and conv_out only (reference :1109, 1127): mv2_pad_ndhwc gathers the padded image, the convolution runs
    'valid' on it; constant padding when the clip is not longer than the time padding (:925), which the T = 1 image case exercises."""
    kw = dict(image_size=16, init_dim=16, max_dim=32, codebook_size=64, pad_mode=pad_mode, separate_first_frame_encoding=sep,
              input_conv_kernel_size=(3, 5, 5), output_conv_kernel_size=(3, 3, 3),
              layers=('residual', 'compress_space', 'compress_time', 'residual', 'attend_time'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=51, randomize_se=True)
    orc = OracleTokenizer(sd, **cfg)
    for dtype, tol in ((torch.float32, 5e-6), (torch.bfloat16, None)):
        tok = emulated_tokenizer(kw, sd, dtype)
        for frames, vcff in ((5, True), (4, False)) + (((1, True),) if not sep else ()):
            v = torch.randn(2, 3, frames, 16, 16, generator=torch.Generator().manual_seed(52 + frames))
            ocodes, orecon = orc(v, return_codes=True, return_recon=True, video_contains_first_frame=vcff)
            codes, recon = tok(v, return_codes=True, return_recon=True, video_contains_first_frame=vcff)
            if tol is not None:
                assert torch.equal(codes, ocodes), (pad_mode, frames)
                assert (recon - orecon).abs().max().item() <= tol
                assert (tok.decode_from_code_indices(ocodes, video_contains_first_frame=vcff) - orecon).abs().max().item() <= tol
            else:
                assert (codes != ocodes).float().mean().item() <= 0.05 and (recon - orecon).abs().mean().item() <= 2e-2
    names = tok._plan(2, 5)[0].step_names
    assert names.count('pad_ndhwc') == 2                                   # conv_in and conv_out, nothing else
    with pytest.raises(ValueError):
        VideoTokenizer(image_size=16, codebook_size=64, layers=('residual',), pad_mode='mirror')


def test_harness_tools_compile(tmp_path):
    """tools/k3_bench.cpp, tools/se_bench.cpp (host harnesses over the C ABI) and tools/mfma_cadence.cpp (gfx950 micro-benchmark) are the
    cheap way to iterate kernels on a GPU box (~15 s of budget per run): keep them compiling.  se_bench's CPU reference is also run
    (GPU-free --dump mode) and checked against the Python oracle."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which('hipcc') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else None)
    if hipcc is None:
        pytest.skip('hipcc not available')
    for src, extra in (('k3_bench.cpp', ['-ldl']), ('se_bench.cpp', ['-ldl']), ('mfma_cadence.cpp', ['--offload-arch=gfx950'])):
        out = str(tmp_path / src.replace('.cpp', ''))
        r = subprocess.run([hipcc, '-w', '-O1', '-std=c++17', os.path.join(ROOT, 'tools', src), '-o', out, *extra], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, 'magvit2_pytorch_amd', 'libmv2_hip.so')
    d = tmp_path / 'dump'
    d.mkdir()
    r = subprocess.run([str(tmp_path / 'se_bench'), lib, '--dump', str(d)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_se_bench_reference.py'), str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize('ncb', [1, 2])
def test_lfq_spherical_matches_oracle(ncb):
    """eval-mode lfq_spherical = the plain kernels with the quantized value scaled to 1/sqrt(code dims) (the l2 normalisation of a sign
    vector; the ids do not change); the training-mode statistics run on the activations normalised between the launches."""
    kw = dict(image_size=16, init_dim=16, max_dim=32, codebook_size=64, num_codebooks=ncb, lfq_spherical=True,
              layers=('residual', 'compress_space', 'compress_time', 'residual', 'attend_time'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=9, randomize_se=True)
    orc = OracleTokenizer(sd, **cfg)
    v = torch.randn(2, 3, 5, 16, 16, generator=torch.Generator().manual_seed(10))
    ocodes, orecon = orc(v, return_codes=True, return_recon=True)
    tok = VideoTokenizer(**kw, use_gan=False, perceptual_loss_weight=0., seed=0)
    tok.load_state_dict(sd)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    codes, recon = tok(v, return_codes=True, return_recon=True)
    assert torch.equal(codes, ocodes) and (recon - orecon).abs().max().item() <= 5e-6
    assert (tok.decode_from_code_indices(ocodes) - orecon).abs().max().item() <= 5e-6
    tok.eval()
    total, br = tok(v, return_loss=True)                              # eval mode: aux loss 0, as the reference
    want, _ = orc.forward_loss(v, training=False)
    assert abs(float(total) - float(want)) <= 1e-5 and float(br.lfq_aux_loss) == 0.
    tok.train()
    total, br = tok(v, return_loss=True)                              # training mode: statistics on the normalised activations
    want, d = orc.forward_loss(v, training=True)
    assert abs(float(total) - float(want)) <= 1e-5 * max(1., abs(float(want))) and abs(float(br.lfq_aux_loss) - float(d['aux'])) <= 1e-5
    qb = br.quantizer_loss_breakdown
    assert abs(float(qb.commitment) - float(d['commitment'])) <= 1e-5 and abs(float(qb.per_sample_entropy) - float(d['per_sample_entropy'])) <= 1e-5


def test_attention_dropout_is_refused_in_training_mode_only():
    """attn_dropout is inactive in eval mode (the hot path); a training-mode loss with dropout would need random masks"""
    tok = VideoTokenizer(image_size=16, init_dim=16, max_dim=16, codebook_size=64, layers=('residual', 'attend_space'), attn_dropout=0.1,
                         use_gan=False, perceptual_loss_weight=0., seed=0)
    tok._set_backend(CpuOps())
    tok.compute_dtype = torch.float32
    v = torch.randn(1, 3, 1, 16, 16, generator=torch.Generator().manual_seed(0))
    tok.eval()
    tok(v, return_codes=True)
    tok(v, return_loss=True)
    tok.train()
    with pytest.raises(NotImplementedError):
        tok(v, return_loss=True)


def test_tokenize_stream_matches_plain_loop():
    """streaming.tokenize_stream (uploads / downloads on a side stream on the GPU) yields, in order, what tokenize() gives clip by clip;
    on the CPU backend it is the plain loop, which is what this checks — the stream / event choreography has a GPU twin in
    tests/test_zz_widen_gpu.py."""
    from magvit2_pytorch_amd.streaming import tokenize_stream
    g = Golden('small_lfq')
    tok = emulated_tokenizer(g.kw, g.weights(), torch.float32)
    clips = [torch.randn(1, 3, 5, 32, 32, generator=torch.Generator().manual_seed(70 + i)) for i in range(4)]
    got = list(tokenize_stream(tok, clips, depth=2))
    assert len(got) == 4 and all(torch.equal(a, tok.tokenize(v)) for a, v in zip(got, clips))
    assert list(tokenize_stream(tok, [], depth=3)) == []


@pytest.mark.parametrize('vcff,frames', [(True, 9), (False, 8)])
def test_gateloop_time_matches_oracle(vcff, frames):
    """'gateloop_time' (reference :1216-1222; the third-party SimpleGateLoopLayer is a reconstruction, parity unpinned): RMSNorm folded
    into one pointwise GEMM [q | kv | a], then mv2_gateloop_scan along time with the residual add."""
    kw = dict(image_size=16, init_dim=16, max_dim=32, codebook_size=64,
              layers=('residual', 'compress_space', 'gateloop_time', 'compress_time', 'gateloop_time', 'residual'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=12, randomize_se=True)
    assert 'encoder_layers.2.fn.fn.to_qkva.0.weight' in sd and sd['decoder_layers.1.fn.fn.norm.gamma'].shape == (32,)
    orc = OracleTokenizer(sd, **cfg)
    v = torch.randn(2, 3, frames, 16, 16, generator=torch.Generator().manual_seed(13))
    ocodes, orecon = orc(v, return_codes=True, return_recon=True, video_contains_first_frame=vcff)
    tok = emulated_tokenizer(kw, sd, torch.float32)
    codes, recon = tok(v, return_codes=True, return_recon=True, video_contains_first_frame=vcff)
    assert torch.equal(codes, ocodes) and (recon - orecon).abs().max().item() <= 5e-6
    assert tok._plan(2, frames, headless=not vcff)[0].step_names.count('gateloop_scan') == 4
    # causal along time: a later frame cannot change earlier latent frames
    a = tok.encode(v, video_contains_first_frame=vcff)
    v2 = v.clone(); v2[:, :, 5:] += 1.0
    b = tok.encode(v2, video_contains_first_frame=vcff)
    assert torch.equal(a[:, :, :2], b[:, :, :2]) and not torch.equal(a, b)


def test_lowering_options_are_explicit_not_environment(monkeypatch):
    """the lowering reads no environment variable; its switches are the fields of plan.OPTIONS, set through
    lowering_options(), validated, and restored on exit."""
    src = open(plan_mod.__file__).read()
    assert 'os.environ' not in src and 'getenv' not in src
    for stray in ('MV2_NO_CONV_RMS', 'MV2_NO_CONV_IN', 'MV2_NO_CONV_OUT', 'MV2_SE_THREE', 'MV2_NO_PW_SE', 'MV2_NO_FUSE_PW', 'MV2_TAYLOR_BLOCKS'):
        monkeypatch.setenv(stray, 'garbage')
    kw, cfg, sd, v = _full_width_case()
    names = emulated_tokenizer(kw, sd, torch.bfloat16)._plan(1, 5)[0].step_names
    before = plan_mod.OPTIONS
    with plan_mod.lowering_options(se_two_launch=False, pw_se=False) as o:
        assert not o.se_two_launch and not o.pw_se and o.conv_rms
        other = emulated_tokenizer(kw, sd, torch.bfloat16)._plan(1, 5)[0].step_names
    assert plan_mod.OPTIONS is before
    assert names != other and 'se_gate' in other
    with pytest.raises(TypeError):
        with plan_mod.lowering_options(no_such_switch=1):
            pass
    with pytest.raises(ValueError):
        with plan_mod.lowering_options(convin_cpad=5):
            pass


def _run_bench(*argv):
    """bench.py as the driver starts it (a fresh interpreter), on the CPU emulation: the control path end to end"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), *argv, '--dry-run-backend', 'emulation'],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, f'exactly one JSON line expected from rank 0, got {len(lines)}:\n{r.stdout[-1000:]}'
    return json.loads(lines[0])


def test_bench_two_rank_control_path_dry_run():
    """bench.py's N > 1 path (respawn under torch.distributed.run, 127.0.0.1 rendezvous, barriers around the timed
    region, per-rank times, gather_codes, destroy_process_group, JSON assembly) must have executed before the driver's first SCALE run.
    Here it runs with gloo instead of RCCL and the kernel emulation instead of the HIP library; the numbers mean nothing."""
    js = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '1')
    assert js['n_gpus'] == 2 and js['rccl_ranks'] == 2 and len(js['per_rank_ms_per_step']) == 2
    assert js['steps'] == 1 and js['scaling'] == 'weak' and js['higher_is_better'] is True and js['unit'] == 'frames/s'
    assert js['dry_run_backend'] == 'emulation' and 'DRY RUN' in js['config']['workload']
    assert 'configs[2]' in js['config']['workload']                           # the N > 1 default shape
    B, T = js['weak_scaling_base']['per_gpu_batch'], js['weak_scaling_base']['frames_per_clip']
    assert js['config']['global_batch'] == 2 * B
    # value = whole-job frames / max-over-ranks time; the same-shape base comes from rank 0's own time
    assert abs(js['value'] - 2 * B * T / (js['ms_per_step'] * 1e-3)) <= 0.02 * js['value']
    base = js['weak_scaling_base']['n1_frames_per_s_same_shape']
    assert abs(base - B * T / (js['per_rank_ms_per_step'][0] * 1e-3)) <= 0.02 * base
    assert abs(js['weak_scaling_base']['efficiency_vs_same_shape'] - js['value'] / (2 * base)) < 0.02     # `base` is printed rounded to 0.1 frames/s
    assert js['ms_per_step'] >= max(js['per_rank_ms_per_step']) * 0.999      # the barrier-bracketed time is the max over ranks
    assert 'scaling_curve' not in js


def test_bench_single_rank_dry_run_states_no_curve():
    js = _run_bench('--steps', '1', '--warmup', '1')
    assert js['n_gpus'] == 1 and 'rccl_ranks' not in js and 'weak_scaling_base' not in js
    assert js['scaling_curve']['measured'] is False and 'batch4' in js['scaling_curve']['note']
    assert 'configs[1]' in js['config']['workload'] and js['config']['global_batch'] == 1


def test_bench_workloads_match_baseline_configs():
    """the shapes bench.py times are the ones BASELINE.json names (configs[1..4])"""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('mv2_bench', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    shapes = {c: bench.workload(c)[1:5] for c in (1, 2, 3, 4)}
    assert shapes == {1: (1, 17, 128, False), 2: (4, 17, 128, False), 3: (4, 33, 256, False), 4: (2, 65, 128, True)}
    kw4 = bench.workload(4)[0]
    assert kw4['use_fsq'] and kw4['fsq_levels'] == [8, 5, 5, 5]
    assert bench.workload(3)[0]['image_size'] == 256 and bench.workload(3)[0]['max_dim'] == 512
    assert bench.parse(['--config', '3']).config == 3


def test_bf16_gelu_polynomial_matches_exact_gelu():
    """csrc/common.h Act<BF16>::gelu = max(v, 0) + P(t), t = 2 min(|v|, 4.5) / 4.5 - 1: the coefficients in the header, evaluated in float32
    Horner form as the kernel does, stay within 2e-5 of the exact erf-GELU (reference F.gelu, magvit2_pytorch.py:468) on [-12, 12]."""
    import os, re
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'magvit2_pytorch_amd', 'csrc', 'common.h')).read()
    body = src[src.index('template <> struct Act<BF16>'):]
    body = body[body.index('static __device__ __forceinline__ float gelu'):body.index('return fmaxf(v, 0.f) + p;')]
    lead = float(re.search(r'float p = ([-+0-9.e]+)f;', body).group(1))
    rest = [float(x) for x in re.findall(r'p = p \* t \+ ([-+0-9.e]+)f;', body)]
    assert len(rest) == 10 and '4.5f' in body and '0.44444444444444444f' in body
    v = np.linspace(-12, 12, 480001).astype(np.float32)
    a = np.minimum(np.abs(v), np.float32(4.5))
    t = (a * np.float32(0.44444444444444444) - np.float32(1)).astype(np.float32)
    p = np.full_like(v, np.float32(lead))
    for c in rest:
        p = (p * t + np.float32(c)).astype(np.float32)
    g = np.maximum(v, np.float32(0)) + p
    exact = torch.nn.functional.gelu(torch.from_numpy(v).double()).numpy()
    assert np.abs(g - exact).max() <= 2e-5


def test_fused_feed_forward_lowering_rule():
    """plan.Plan._feed_forward takes mv2_ff (one launch) from LoweringOptions.fuse_ff_min_rows rows on, where the packed model carries the
    fragment images (bf16, C = 256 / 512, unconditioned), and the two-launch form below it / with fuse_ff off; both lowerings agree on the
    emulation."""
    kw, cfg, sd, v = _full_width_case()
    outs = {}
    for key, opts in (('rule', {}), ('forced', dict(fuse_ff_min_rows=0)), ('off', dict(fuse_ff=False))):
        with plan_mod.lowering_options(**opts):
            tok = emulated_tokenizer(kw, sd, torch.bfloat16)
            plan = tok._plan(1, 5)[0]
            outs[key] = (list(plan.step_names), tok(v, return_codes=True, return_recon=True))
    small_rows = all(not n.endswith('.ff') for n in outs['rule'][0])          # a tiny test clip stays below 2560 rows per layer
    assert small_rows and sum(n.endswith('.ff') for n in outs['forced'][0]) == 6 and sum(n.endswith('.ff1') for n in outs['off'][0]) == 6
    assert len(outs['forced'][0]) == len(outs['off'][0]) - 6
    assert torch.equal(outs['forced'][1][0], outs['off'][1][0]) and (outs['forced'][1][1] - outs['off'][1][1]).abs().max().item() <= 2e-2
    assert plan_mod.LoweringOptions().fuse_ff_min_rows == 2560


def test_clip_lanes_match_plain_calls_on_the_emulation():
    """streaming.ClipLanes (replicas of the tokenizer on their own HIP streams; on the CPU emulation a plain loop over the replicas) yields, in
    order, what the plain calls give - forward() tuples and tokenize() ids - for more clips than lanes and for an empty list."""
    from magvit2_pytorch_amd.streaming import ClipLanes
    kw = dict(image_size=16, init_dim=8, max_dim=32, codebook_size=64, layers=README_LAYERS)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=3, randomize_se=True)
    tok = emulated_tokenizer(kw, sd, torch.float32)
    clips = [torch.randn(1, 3, 5, 16, 16, generator=torch.Generator().manual_seed(20 + i)) for i in range(5)]
    lanes = ClipLanes(tok, lanes=2)
    assert len(lanes.toks) == 2 and lanes.toks[1] is not tok
    want = [tok(v, return_codes=True, return_recon=True) for v in clips]
    got = list(lanes.map(clips, return_codes=True, return_recon=True))
    assert len(got) == 5 and all(torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) for g, w in zip(got, want))
    ids = list(lanes.map(clips[:3], method='tokenize'))
    assert all(torch.equal(i, w[0]) for i, w in zip(ids, want))
    assert list(lanes.map([])) == []


def test_split_mode_on_the_emulation_reproduces_the_oracle_ids():
    """compute_dtype = 'split' (the id-exact mode that is not 16 x slower: tokenizer.py docstring) end to end on the kernel emulation:
    the encoder plan keeps fp32 rows and runs its ResidualUnit convolutions as mv2_split_bf16 + mv2_conv3(MV2_STORE_ROWS_F32), the
    picture comes from the bf16 plan's ids -> codes -> decoder half.  Ids equal the oracle's, the pre-quantizer activations sit at fp32
    round-off (not at bf16's 1e-3), decode_from_code_indices(tokenize(v)) == forward(v) bit for bit."""
    from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
    from oracle.restate import OracleTokenizer
    kw = dict(image_size=8, init_dim=64, max_dim=64, codebook_size=256, layers=('residual', 'compress_time', 'residual'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=3, randomize_se=True)
    v = torch.randn(1, 3, 5, 8, 8, generator=torch.Generator().manual_seed(5))
    orc = OracleTokenizer(sd, **cfg)
    orc.taps = {}
    oc, orecon = orc(v, return_codes=True, return_recon=True)
    pre = orc.taps['prequant']
    tok = emulated_tokenizer(kw, sd, 'split')
    c, r, pq = tok(v, return_codes=True, return_recon=True, return_prequant=True)
    assert torch.equal(c, oc)
    assert (pq.reshape(pre.shape) - pre).abs().max().item() <= 5e-6
    assert (r - orecon).abs().mean().item() <= 1e-2                   # the bf16 decoder's picture
    assert torch.equal(tok.decode_from_code_indices(c), r) and torch.equal(tok.tokenize(v), c)
    kinds = {(pl.dtype, pl.pm.split_k3) for pl in tok._plans.values()}
    assert kinds == {(torch.float32, True), (torch.bfloat16, False)}
    enc = next(pl for pl in tok._plans.values() if pl.pm.split_k3)
    assert sum(n == 'split_bf16' for n in enc.step_names) == 4        # 2 encoder + 2 decoder units of the (complete) fp32 plan
    with pytest.raises(AssertionError):
        bad = emulated_tokenizer(kw, sd, 'fp16')
        bad(v, return_codes=True)
