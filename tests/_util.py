"""Shared helpers for the test-suite (golden fixtures, seeded weights / inputs)."""
import json
import os

import numpy as np
import torch

from magvit2_pytorch_amd.spec import init_state_dict, make_cfg
from oracle.make_golden import ClipDiscr, make_input, sha_tensors, tiny_vgg, weights_sha   # noqa: F401

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, f'{name}.npz'), allow_pickle=False)
        self.name = name
        self.kw = json.loads(str(z['cfg']))
        if 'layers' in self.kw:
            self.kw['layers'] = tuple(tuple(l) if isinstance(l, list) else l for l in self.kw['layers'])
        self.cfg = make_cfg(**self.kw)
        self.shape = tuple(int(v) for v in z['shape'])
        self.weight_seed, self.data_seed, self.stride = int(z['weight_seed']), int(z['data_seed']), int(z['stride'])
        self.codes = torch.from_numpy(z['codes'])
        self.recon_sample = torch.from_numpy(z['recon_sample'])
        self.recon_mean, self.recon_std, self.recon_abs_sum = float(z['recon_mean']), float(z['recon_std']), float(z['recon_abs_sum'])
        self._wsha, self._isha = str(z['weights_sha256']), str(z['input_sha256'])
        # full-size cases carry the oracle's pre-sign margin per token (so a test need not re-run the CPU oracle)
        self.margin = torch.from_numpy(z['margin']) if 'margin' in z.files else None

    def weights(self):
        sd = init_state_dict(self.cfg, seed=self.weight_seed, randomize_se=True)
        assert weights_sha(sd) == self._wsha, 'seeded weights are not reproducible on this box (torch RNG drift?)'
        return sd

    def video(self):
        v = make_input(self.shape, self.data_seed)
        assert sha_tensors([v]) == self._isha, 'seeded input is not reproducible on this box'
        return v

    def sample(self, recon):
        return recon[..., ::self.stride, ::self.stride]


def live_cfg(**kw):
    return make_cfg(**kw)


class Feature:
    """tests/golden/feat_*.npz (oracle/make_golden.py FEATURES): outputs of the reference itself on the paths beyond eval
    encode / decode; weights and inputs are regenerated from the seeds recorded in oracle.make_golden.FEATURES."""

    def __init__(self, name):
        from oracle.make_golden import FEATURES
        self.name, self.f = name, FEATURES[name]
        self.z = np.load(os.path.join(GOLDEN_DIR, f'{name}.npz'), allow_pickle=False)
        self.kw = dict(self.f['kw'])
        self.cfg = make_cfg(**self.kw)
        self.shape = tuple(self.f['shape'])

    def weights(self):
        sd = init_state_dict(self.cfg, seed=self.f['wseed'], randomize_se=True)
        assert weights_sha(sd) == str(self.z['weights_sha256']), 'seeded weights are not reproducible on this box'
        return sd

    def video(self):
        v = make_input(self.shape, self.f['dseed'])
        assert sha_tensors([v]) == str(self.z['input_sha256'])
        return v

    def cond(self):
        return make_input((self.shape[0], self.kw['dim_cond']), self.f['cseed']) if 'cseed' in self.f else None

    def discr_weights(self):
        from oracle.make_golden import discr_state_dict
        dsd = discr_state_dict(16, self.kw['image_size'], self.f['discr_seed'])
        assert weights_sha(dsd) == str(self.z['discr_sha256'])
        return dsd

    def scalar(self, key):
        return float(self.z[key])

    def discr_dim(self):
        return self.f.get('discr_dim', 16)

