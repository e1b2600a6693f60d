"""GPU parity of the widened host-level surface (run last: the file name sorts after the kernel parity suites).  Nothing here adds a
kernel: headless clips (video_contains_first_frame=False, reference M:1530-1538, 1604, 1646, 1691), the generator-side GAN /
perceptual / multiscale loss terms and the discriminator branch with multiscale discriminators (M:1730-1896, forward values), and
the losses under separate_first_frame_encoding are new LOWERINGS onto the kernels the suites before this file validate.
Checkers: the reference's own outputs (tests/golden/feat_*.npz), the CPU oracle, and the kernel emulation in lockstep.

Tolerances (fp32 mode): as tests/test_losses_gpu.py — loss scalars 3e-4 relative (summation order, amplified by the entropy
logits), reconstructions 5e-4 absolute, ids equal wherever the oracle's pre-sign margin exceeds 1e-4 and <= 0.5 % different overall."""
import pytest
import torch

from magvit2_pytorch_amd import VideoTokenizer
from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
from oracle.restate import OracleTokenizer
from tests.test_host_logic import _widened_feature_checks
from tests.test_parity_gpu import hip_tokenizer, margin_check, run_lockstep

pytestmark = pytest.mark.gpu

HEADLESS_KW = dict(image_size=32, init_dim=64, max_dim=128, codebook_size=256,
                   layers=('residual', 'compress_space', 'compress_time', 'residual', 'compress_time', 'attend_time'))


def test_product_vs_reference_widened_feature_fixtures():
    def make_tok(kw, sd):
        tok = VideoTokenizer(**kw, seed=0)
        tok.load_state_dict(sd)
        tok = tok.cuda()
        tok.compute_dtype = torch.float32
        return tok
    _widened_feature_checks(make_tok, lambda t: t.cuda(), tol_total=3e-4, tol_recon=5e-4, ids_frac=0.005)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('sep', [False, True])
def test_headless_video(sep, dtype):
    """lockstep against the kernel emulation (every launch of the headless plan), then fp32 parity against the oracle incl. the
    round trip and HIP-graph replay (three calls: eager, capture, replay)."""
    kw = dict(HEADLESS_KW, separate_first_frame_encoding=sep)
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=43, randomize_se=True)
    v = torch.randn(2, 3, 8, 32, 32, generator=torch.Generator().manual_seed(44))
    run_lockstep(cfg, sd, v, dtype, headless=True)
    tok = hip_tokenizer(kw, sd, dtype)
    orc = OracleTokenizer(sd, **cfg)
    orc.taps = {}
    ocodes, orecon = orc(v, return_codes=True, return_recon=True, video_contains_first_frame=False)
    for rep in range(3):
        codes, recon = tok(v.cuda(), return_codes=True, return_recon=True, video_contains_first_frame=False)
        assert codes.shape == (2, 2, 16, 16) and recon.shape == v.shape
        if dtype == torch.float32:
            margin_check(codes, ocodes, orc.taps['prequant'], 1e-4, 0.005)
            if torch.equal(codes.cpu(), ocodes):
                assert (recon.cpu() - orecon).abs().max().item() <= 5e-4
        else:
            assert (codes.cpu() != ocodes).float().mean().item() <= 0.03            # the reference's own bf16-autocast drift is 1.6-2.2 %
            assert (recon.cpu() - orecon).abs().mean().item() <= 1e-2
        assert torch.equal(tok.decode_from_code_indices(codes, video_contains_first_frame=False), recon)      # round trip, bit exact
    if dtype == torch.float32:
        assert (tok.decode_from_code_indices(ocodes.cuda(), video_contains_first_frame=False).cpu() - orecon).abs().max().item() <= 5e-4
        # clips WITH a first frame keep working on the same module (separate plans and graphs per mode)
        v9 = torch.randn(2, 3, 9, 32, 32, generator=torch.Generator().manual_seed(45))
        oc9, or9 = orc(v9, return_codes=True, return_recon=True)
        c9, r9 = tok(v9.cuda(), return_codes=True, return_recon=True)
        margin_check(c9, oc9, orc.taps['prequant'], 1e-4, 0.005)
        if torch.equal(c9.cpu(), oc9):
            assert (r9.cpu() - or9).abs().max().item() <= 5e-4


@pytest.mark.parametrize('training', [True, False])
def test_forward_loss_with_gan_fp32_matches_oracle(training):
    kw = dict(image_size=32, init_dim=32, max_dim=64, codebook_size=256, layers=('residual', 'compress_space', 'compress_time', 'residual', 'attend_space'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=46, randomize_se=True)
    tok = VideoTokenizer(**kw, use_gan=True, perceptual_loss_weight=0., adversarial_loss_weight=0.7, seed=3)
    tok.load_state_dict(sd)
    tok = tok.cuda()
    tok.compute_dtype = torch.float32
    tok.train(training)
    dsd = {k: v.detach().cpu().clone() for k, v in tok.state_dict().items() if k.startswith('discr.')}
    v = torch.randn(3, 3, 5, 32, 32, generator=torch.Generator().manual_seed(47))
    torch.manual_seed(90)
    total, br = tok(v.cuda(), return_loss=True)
    torch.manual_seed(90)
    want, d = OracleTokenizer(sd, **cfg).forward_loss(v, training=training, discr_params=dsd, discr_kw=dict(dim=64, max_dim=512), adversarial_loss_weight=0.7)
    assert abs(float(br.adversarial_gen_loss) - float(d['gen_loss'])) <= 3e-4 * max(1., abs(float(d['gen_loss'])))
    assert abs(float(br.recon_loss) - float(d['recon_loss'])) <= 1e-4 * float(d['recon_loss'])
    assert abs(float(total) - float(want)) <= 3e-4 * max(1., abs(float(want)))


# ------------------------------------------------------------------------------- further lowerings onto validated kernel forms (no new kernel):
# lfq_spherical = the LFQ kernels with another `scale`; tokenize_stream = torch streams / events around tokenize(); 5 .. 8 input
# channels = the 8-channel pixel layout of conv_in (round 1's default)
def test_lfq_spherical_fp32_vs_oracle():
    """eval-mode lfq_spherical: the LFQ kernels with scale = 1/sqrt(code dims) (a value of `scale` the default suite never passes)"""
    kw = dict(image_size=32, init_dim=16, max_dim=64, codebook_size=256, lfq_spherical=True, layers=('residual', 'compress_space', 'compress_time', 'residual', 'attend_time'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=65, randomize_se=True)
    v = torch.randn(2, 3, 5, 32, 32, generator=torch.Generator().manual_seed(66))
    run_lockstep(cfg, sd, v, torch.float32)
    run_lockstep(cfg, sd, v, torch.bfloat16)
    ocodes, orecon = OracleTokenizer(sd, **cfg)(v, return_codes=True, return_recon=True)
    tok = hip_tokenizer(dict(kw, use_gan=False), sd, torch.float32)
    codes, recon = tok(v.cuda(), return_codes=True, return_recon=True)
    assert (codes.cpu() != ocodes).float().mean().item() <= 0.005
    if torch.equal(codes.cpu(), ocodes):
        assert (recon.cpu() - orecon).abs().max().item() <= 5e-4
    assert (tok.decode_from_code_indices(ocodes.cuda()).cpu() - orecon).abs().max().item() <= 5e-4
    tok2 = VideoTokenizer(**kw, use_gan=False, perceptual_loss_weight=0., seed=0)
    tok2.load_state_dict(sd)
    tok2 = tok2.cuda()
    tok2.compute_dtype = torch.float32
    tok2.train()
    total, br = tok2(v.cuda(), return_loss=True)
    want, d = OracleTokenizer(sd, **cfg).forward_loss(v, training=True)
    assert abs(float(total) - float(want)) <= 3e-4 * max(1., abs(float(want))) and abs(float(br.lfq_aux_loss) - float(d['aux'])) <= 3e-4


@pytest.mark.parametrize('depth', [1, 2, 3])
def test_tokenize_stream_on_gpu(depth):
    """uploads and id downloads on a side stream, compute on the current stream: same ids, in order, as clip-by-clip tokenize()"""
    from magvit2_pytorch_amd.streaming import tokenize_stream
    from tests._util import Golden
    g = Golden('small_lfq')
    tok = hip_tokenizer(g.kw, g.weights(), torch.bfloat16)
    clips = [torch.randn(2, 3, 9, 32, 32, generator=torch.Generator().manual_seed(80 + i)).pin_memory() for i in range(7)]
    want = [tok.tokenize(v.cuda()).cpu() for v in clips]
    got = list(tokenize_stream(tok, clips, depth=depth))
    assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('channels', [1, 5, 8, 12, 20])
def test_input_channel_counts_lockstep(channels, dtype):
    """8-channel input pixels (5 .. 8 channels): the round-1 conv_in layout, not exercised by the default suite of round 2"""
    kw = dict(image_size=32, init_dim=16, max_dim=64, codebook_size=256, channels=channels, layers=('residual', 'compress_space', 'compress_time', 'residual'))
    cfg = make_cfg(**kw)
    sd = init_state_dict(cfg, seed=63, randomize_se=True)
    run_lockstep(cfg, sd, torch.randn(2, channels, 5, 32, 32, generator=torch.Generator().manual_seed(64)), dtype)


@pytest.mark.gpu
def test_clip_lanes_two_streams_match_plain_calls():
    """streaming.ClipLanes on the GPU: two replicas on two HIP streams, five clips dealt round-robin (graphs captured per lane), results
    identical to the plain calls on the original tokenizer, in order."""
    from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer
    from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
    from magvit2_pytorch_amd.streaming import ClipLanes
    kw = dict(image_size=32, init_dim=16, max_dim=64, codebook_size=256, layers=README_LAYERS)
    cfg = make_cfg(**kw)
    tok = VideoTokenizer(**kw, seed=0)
    tok.load_state_dict(init_state_dict(cfg, seed=0, randomize_se=True))
    tok = tok.cuda()
    tok.compute_dtype = torch.bfloat16
    clips = [torch.randn(1, 3, 9, 32, 32, generator=torch.Generator().manual_seed(40 + i)).cuda() for i in range(5)]
    want = [tuple(t.clone() for t in tok(v, return_codes=True, return_recon=True)) for v in clips]
    lanes = ClipLanes(tok, lanes=2)
    for _ in range(3):                                                    # eager, capture, replay
        got = list(lanes.map(clips, return_codes=True, return_recon=True))
    torch.cuda.synchronize()
    assert len(got) == 5
    for g, w in zip(got, want):
        assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1])
