"""Loader for the reference-generated fixtures in tests/golden/*.npz (see tests/golden/make_golden.py)."""
import glob
import json
import os

import numpy as np
import torch

from ir_sgmcmc_amd.data_loader import synthetic_pair
from oracle import OracleConfig

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_names(prefix='n'):
    """transition fixtures are named n<size>_...; the VI fixtures vi_... have their own loader (tests/test_vi.py)"""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + '*.npz')))


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        meta = json.loads(bytes(self.z['config']).decode())
        self.N = meta.pop('N')
        self.sigma = meta.pop('sigma')
        for k in ('cps', 'student'):
            if meta.get(k) is not None:
                meta[k] = tuple(meta[k])
        self.cfg = OracleConfig(dims=(self.N,) * 3, **meta)
        self.subsampled = 'fixed' not in self.z.files
        self.T = len([k for k in self.z.files if k.endswith('_seed')])

    def t(self, it, key):
        return torch.from_numpy(np.asarray(self.z[f't{it}_{key}']))

    def has(self, it, key):
        return f't{it}_{key}' in self.z.files

    def inputs(self):
        """fixed/moving dicts expanded to C chains, v0, sigma tensor."""
        C, N = self.cfg.no_chains, self.N
        if self.subsampled:
            fixed1, moving1 = synthetic_pair((N,) * 3, seed=0)
            v0 = self._regen_v0()
            chk = self.z['input_checksum']
            got = np.array([fixed1['im'].double().sum(), moving1['im'].double().sum(), v0.double().sum(),
                            (v0.double() ** 2).sum()])
            assert np.allclose(chk, got, rtol=1e-9, atol=1e-9), 'regenerated inputs differ from the fixture checksum'
            im_f, im_m, mask = fixed1['im'], moving1['im'], fixed1['mask']
        else:
            im_f, im_m = torch.from_numpy(self.z['fixed']), torch.from_numpy(self.z['moving'])
            mask, v0 = torch.from_numpy(self.z['mask']), torch.from_numpy(self.z['v0'])
        fixed = {'im': im_f.unsqueeze(0).expand(C, 1, N, N, N).contiguous(),
                 'mask': mask.unsqueeze(0).expand(C, 1, N, N, N).contiguous()}
        moving = {'im': im_m.unsqueeze(0).expand(C, 1, N, N, N).contiguous()}
        return fixed, moving, v0, torch.full_like(v0, self.sigma)

    def _regen_v0(self):
        # mirrors make_golden.initial_state('smooth_noise') -- only the subsampled 64^3 variant uses it
        from oracle import ops
        g = torch.Generator().manual_seed(1000 + sum(map(ord, self.name)))
        shape = (self.cfg.no_chains, 3, *self.cfg.dims_v)
        k = ops.sobolev_kernel_1d(3, 0.5)
        return ops.separable_conv3d_replicate(3.0 * torch.randn(shape, generator=g), k).float().contiguous()

    def noise(self, it):
        C, N = self.cfg.no_chains, self.N
        if self.subsampled:
            torch.manual_seed(int(self.z[f't{it}_seed']))
            eps = torch.randn(C, 3, *self.cfg.dims_v)
            unif = torch.rand(C, 3, N, N, N) if self.cfg.uniform_noise is not None else None
            chk = self.z[f't{it}_noise_checksum']
            assert np.isclose(chk[0], float(eps.double().sum()), rtol=1e-9, atol=1e-9), 'regenerated noise differs'
            return eps, unif
        eps = self.t(it, 'eps')
        unif = self.t(it, 'unif') if self.has(it, 'unif') else None
        return eps, unif

    def gmm_init(self):
        z = self.z
        st = {'log_std': torch.from_numpy(z['gmm_log_std_init']), 'logits': torch.from_numpy(z['gmm_logits_init']),
              'adam': [(int(z[f'gmm_adam{i}_step']), torch.from_numpy(z[f'gmm_adam{i}_m']), torch.from_numpy(z[f'gmm_adam{i}_v']))
                       for i in range(2)]}
        return st

    def sub(self, x):
        """apply the fixture's spatial subsampling to a dense (C,ch,N,N,N) tensor."""
        return x[:, :, ::4, ::4, ::4] if self.subsampled else x
