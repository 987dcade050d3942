"""Free-running trajectories against the reference: ten consecutive transitions of the reference's loop
body (trainer/trainer.py:371-379) on CPU, recorded by tests/golden/make_golden.py -- loss terms, alpha, energy, ||v|| and the
hyper-parameters after every transition, the final velocity.  Nothing is re-synchronised in between, so what is measured is
the accumulated drift of a chain: of the CPU oracle here, of the HIP engine in the `gpu` test."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import OracleChain, OracleConfig
from tests._golden import GOLDEN_DIR

NAMES = ['traj_n16_svf_l2_vd_c1', 'traj_n32_svf_lognormal_learn_c2']


class Trajectory:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        meta = json.loads(bytes(self.z['config']).decode())
        self.N, self.sigma = meta.pop('N'), meta.pop('sigma')
        self.cfg = OracleConfig(dims=(self.N,) * 3, **meta)
        self.T = self.z['trajectory'].shape[0]
        self.rows = self.z['trajectory']  # [T, 5 C]: data, reg, alpha, energy, ||v||

    def inputs(self):
        C, N = self.cfg.no_chains, self.N
        t = lambda k: torch.from_numpy(self.z[k])
        fixed = {'im': t('fixed').unsqueeze(0).expand(C, 1, N, N, N).contiguous(), 'mask': t('mask').unsqueeze(0).expand(C, 1, N, N, N).contiguous()}
        moving = {'im': t('moving').unsqueeze(0).expand(C, 1, N, N, N).contiguous()}
        v0 = t('v0')
        return fixed, moving, v0, torch.full_like(v0, self.sigma)

    def noise(self, it):
        C, N = self.cfg.no_chains, self.N
        torch.manual_seed(int(self.z[f't{it}_seed']))  # the draw order of the reference: randn_like(sigma), then rand(shape)
        eps = torch.randn(C, 3, N, N, N)
        unif = torch.rand(C, 3, N, N, N) if self.cfg.uniform_noise is not None else None
        chk = self.z[f't{it}_noise_checksum']
        assert np.isclose(chk[0], float(eps.double().sum()), rtol=1e-9, atol=1e-9), 'regenerated noise differs from the fixture'
        return eps, unif

    def sub(self, v):
        return v if self.N <= 16 else v[:, :, ::4, ::4, ::4]

    def compare(self, it, data, reg, alpha, energy, vnorm, tol):
        """relative deviation of the five per-chain quantities of transition `it`; returns the worst"""
        C = self.cfg.no_chains
        got = np.concatenate([np.asarray(x, dtype=np.float64).reshape(C) for x in (data, reg, alpha, energy, vnorm)])
        ref = self.rows[it]
        dev = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
        assert dev.max() <= tol, f'transition {it}: relative deviation {dev.max():.2e} (data, reg, alpha, energy, ||v|| x C: {dev})'
        return float(dev.max())


@pytest.mark.parametrize('name', NAMES)
def test_oracle_free_running_trajectory(name):
    tr = Trajectory(name)
    fixed, moving, v0, sigma = tr.inputs()
    orc = OracleChain(tr.cfg, v0=v0, sigma=sigma)
    orc.init_gmm(fixed, moving)
    for it in range(tr.T):
        eps, unif = tr.noise(it)
        o = orc.transition(fixed, moving, eps, unif)
        vn = [float(o['v_new'][c].double().norm()) for c in range(tr.cfg.no_chains)]
        tr.compare(it, o['data'], o['reg'], o['alpha'], o['reg_energy'], vn, 1e-5)
        assert torch.allclose(orc.log_std.detach(), torch.from_numpy(tr.z[f't{it}_gmm_log_std']), atol=2e-5)
    vf = torch.from_numpy(tr.z['v_final'])
    assert float((tr.sub(o['v_new']) - vf).abs().max()) <= 1e-4 * float(vf.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_hip_free_running_trajectory(name):
    """the HIP chain, started from the reference's state and fed the reference's noise, is left alone for ten transitions"""
    from ir_sgmcmc_amd.engine import TransitionEngine
    from tests._report import check
    from tests.test_gpu_transition import engine_config
    DEV = 'cuda:0'
    tr = Trajectory(name)
    fixed, moving, v0, sigma = tr.inputs()
    cfg = engine_config(tr.cfg)
    eng = TransitionEngine(cfg, DEV)
    fd, md = eng.prepare({k: v.to(DEV) for k, v in fixed.items()}, {k: v.to(DEV) for k, v in moving.items()})
    eng.gmm_init(fd, md)
    # start from the reference's exact hyper-parameter state (its warm-up is compared elsewhere): what follows is the drift
    # of the transitions alone
    st, K = eng.state(), cfg.gmm_components
    for k in range(K):
        st.gmm_log_std[k], st.gmm_logits[k] = float(tr.z['gmm_log_std_init'][k]), float(tr.z['gmm_logits_init'][k])
        for i in range(2):
            st.gmm_adam_m[i][k], st.gmm_adam_v[i][k] = float(tr.z[f'gmm_adam{i}_m'][k]), float(tr.z[f'gmm_adam{i}_v'][k])
    if 'reg_loc_init' in tr.z.files:
        st.reg_param[0], st.reg_param[1] = float(tr.z['reg_loc_init']), float(tr.z['reg_log_scale_init'])
    eng.set_state(st)
    v, sig = v0.to(DEV).contiguous(), sigma.to(DEV).contiguous()
    worst = 0.0
    for it in range(tr.T):
        eps, unif = tr.noise(it)
        eng.transition(fd, md, v, sig, eps.to(DEV), unif.to(DEV) if unif is not None else None)
        sc = eng.scalars()
        vn = [float(v[c].double().norm()) for c in range(cfg.no_chains)]
        # north star: loss within 1e-5 relative -- held over the whole free-running trajectory
        worst = max(worst, tr.compare(it, sc['data_term'], sc['reg_term'], sc['alpha'], sc['reg_energy'], vn, 1e-5))
        s2 = eng.state()
        check('trajectory/' + name, 'gmm_log_std', list(s2.gmm_log_std)[:K], tr.z[f't{it}_gmm_log_std'], 2e-5)
        if f't{it}_reg_params' in tr.z.files:
            check('trajectory/' + name, 'reg hyper-parameters', [s2.reg_param[0], s2.reg_param[1]], tr.z[f't{it}_reg_params'], 1e-5)
    check('trajectory/' + name, 'loss terms, alpha, energy, ||v|| over 10 free-running transitions (rel)', worst, 0.0, 1e-5)
    vf = torch.from_numpy(tr.z['v_final'])
    check('trajectory/' + name, 'v after 10 transitions (rel to max)', tr.sub(v.cpu()) / float(vf.abs().max()), vf / float(vf.abs().max()), 1e-4)
