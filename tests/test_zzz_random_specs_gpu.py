"""GPU twin of tests/test_host_logic.py::test_random_specs, plus the lockstep tests of the two kernels that were first written without GPU
access (mv2_pad_ndhwc, mv2_gateloop_scan).  Collected by default since round 3 (first passed on an MI355X in the round's first GPU call).

Per seed: a random layer spec / kernel sizes / channel count / quantizer / clip mode (the generator of the CPU test, wider channel
ranges so that the conv3 / pw_se kernels take part), the HIP plan against the kernel emulation launch by launch (fp32 and bf16),
then the facade in fp32 against the oracle with the margin rule of tests/test_parity_gpu.py."""
import random

import pytest
import torch

from magvit2_pytorch_amd import VideoTokenizer
from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
from oracle.restate import OracleTokenizer
from tests.test_host_logic import _random_case
from tests.test_parity_gpu import hip_tokenizer, run_lockstep

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = random.Random(2000 + seed)
    kw, B, T, vcff = _random_case(rng)
    if seed % 2:                                   # every other case at the real kernel widths
        kw['init_dim'], kw['max_dim'] = rng.choice([32, 64]), rng.choice([128, 256])
    return kw, B, T, vcff


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('seed', range(16))
def test_random_specs_lockstep(seed, dtype):
    kw, B, T, vcff = _case(seed)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=seed, randomize_se=True)
    v = torch.randn(B, kw['channels'], T, kw['image_size'], kw['image_size'], generator=torch.Generator().manual_seed(300 + seed))
    try:
        run_lockstep(cfg, sd, v, dtype, headless=not vcff)
    except AssertionError as e:
        if 'ids differ' in str(e) and B * T * kw['image_size'] ** 2 <= 4096:
            pytest.xfail(f'id flip on a tiny token grid (teacher-forced lockstep allows 1 %): {e}')
        raise


@pytest.mark.parametrize('seed', range(16))
def test_random_specs_fp32_vs_oracle(seed):
    kw, B, T, vcff = _case(seed)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=seed, randomize_se=True)
    v = torch.randn(B, kw['channels'], T, kw['image_size'], kw['image_size'], generator=torch.Generator().manual_seed(300 + seed))
    orc = OracleTokenizer(sd, **cfg)
    orc.taps = {}
    ocodes, orecon = orc(v, return_codes=True, return_recon=True, video_contains_first_frame=vcff)
    tok = hip_tokenizer(dict(kw, use_gan=False), sd, torch.float32)
    codes, recon = tok(v.cuda(), return_codes=True, return_recon=True, video_contains_first_frame=vcff)
    pre = orc.taps['prequant']
    if kw.get('use_fsq'):
        margin = (0.5 - (pre - pre.round()).abs()).amin(dim=-1)
    else:
        ncb = kw['num_codebooks']
        margin = pre.abs().reshape(*pre.shape[:-1], ncb, -1).amin(dim=-1)
        margin = margin.squeeze(-1) if ncb == 1 else margin
    diff = codes.cpu() != ocodes
    assert not (diff & (margin > 1e-4)).any(), kw
    if not diff.any():
        assert (recon.cpu() - orecon).abs().max().item() <= 5e-4, kw
    assert (tok.decode_from_code_indices(ocodes.cuda(), video_contains_first_frame=vcff).cpu() - orecon).abs().max().item() <= 5e-4, kw


# ------------------------------------------------------------------------------- pad_mode (mv2_pad_ndhwc: a kernel written without GPU access)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('pad_mode', ['reflect', 'replicate', 'circular'])
def test_pad_mode_lockstep_and_fixture(pad_mode, dtype):
    """mv2_pad_ndhwc against the emulation launch by launch (conv_in with 8-byte pixels and a tail, conv_out with 32-byte rows), then the
    facade in fp32 against the reference's own output (tests/golden/feat_pad_*.npz) where a fixture exists."""
    import numpy as np
    from tests._util import Feature
    kw = dict(image_size=32, init_dim=16, max_dim=64, codebook_size=256, pad_mode=pad_mode, input_conv_kernel_size=(3, 5, 5),
              layers=('residual', 'compress_space', 'compress_time', 'residual', 'attend_time'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=61, randomize_se=True)
    v = torch.randn(2, 3, 5, 32, 32, generator=torch.Generator().manual_seed(62))
    run_lockstep(cfg, sd, v, dtype)
    run_lockstep(cfg, sd, v[:, :, :4], dtype, headless=True)
    if dtype == torch.float32 and pad_mode in ('reflect', 'replicate'):
        f = Feature(f'feat_pad_{pad_mode}')
        tok = hip_tokenizer(dict(f.kw, use_gan=False), f.weights(), torch.float32)
        codes, recon = tok(f.video().cuda(), return_codes=True, return_recon=True)
        want = torch.from_numpy(np.asarray(f.z['codes']))
        assert (codes.cpu() != want).float().mean().item() <= 0.005
        if torch.equal(codes.cpu(), want):
            assert (recon.cpu() - torch.from_numpy(np.asarray(f.z['recon']))).abs().max().item() <= 5e-4




@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gateloop_time_lockstep(dtype):
    """mv2_gateloop_scan (written without GPU access) against the emulation, at a width where the qkva GEMM takes the MFMA path"""
    kw = dict(image_size=32, init_dim=64, max_dim=128, codebook_size=256,
              layers=('residual', 'compress_space', 'gateloop_time', 'compress_time', 'gateloop_time', 'residual'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=67, randomize_se=True)
    run_lockstep(cfg, sd, torch.randn(2, 3, 9, 32, 32, generator=torch.Generator().manual_seed(68)), dtype)
