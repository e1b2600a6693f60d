import time, torch, sys
sys.path.insert(0, '.')
from magvit2_pytorch_amd import README_LAYERS, VideoTokenizer
from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
def run(name, kw, shape, mode, steps=5):
    cfg = make_cfg(**kw); sd = init_state_dict(cfg, seed=0, randomize_se=True)
    tok = VideoTokenizer(**kw, seed=0); tok.load_state_dict(sd); tok = tok.cuda(); tok.compute_dtype = torch.bfloat16
    v = torch.randn(*shape, generator=torch.Generator().manual_seed(1)).cuda()
    f = (lambda: tok(v, return_codes=True, return_recon=True)) if mode == 'forward' else (lambda: tok.tokenize(v))
    for _ in range(4): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print(f'{name}: {shape} {mode} bf16: {dt*1e3:.2f} ms/step -> {shape[0]*shape[2]/dt:.0f} frames/s', flush=True)
run('configs[3]', dict(image_size=256, init_dim=64, max_dim=512, codebook_size=1024, layers=README_LAYERS), (4, 3, 33, 256, 256), 'forward')
run('configs[4]', dict(image_size=128, init_dim=64, max_dim=512, use_fsq=True, fsq_levels=[8, 5, 5, 5], layers=README_LAYERS), (2, 3, 65, 128, 128), 'tokenize')
