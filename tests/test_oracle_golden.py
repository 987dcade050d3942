"""The oracle against the fixtures generated from the real reference (tests/golden/make_golden.py).

This is what pins the oracle: every transition of every variant is replayed through `oracle.OracleChain`
from the stored inputs and compared with what the reference's `Trainer._SGLD_transition` returned.
"""
import numpy as np
import pytest
import torch

from oracle import OracleChain
from tests._golden import Golden, golden_names

FAST = [n for n in golden_names() if n.startswith('n16') or n.startswith('n32_svf_l2')]
SLOW = [n for n in golden_names() if n not in FAST]


def _replay(name):
    g = Golden(name)
    fixed, moving, v0, sigma = g.inputs()
    orc = OracleChain(g.cfg, v0=v0, sigma=sigma)
    orc.init_gmm(fixed, moving)  # Trainer.__GMM_init restated; compared with the reference's result
    gi = g.gmm_init()
    assert torch.allclose(orc.log_std.detach(), gi['log_std'], atol=1e-5)
    assert torch.allclose(orc.logits.detach(), gi['logits'], atol=1e-5)
    for i, (step, m, v) in enumerate(gi['adam']):
        st = orc.adam_gmm.state[i]
        assert st['step'] == step == 25
        assert torch.allclose(st['m'], m, rtol=1e-3, atol=1e-3 * float(m.abs().max()))
    for it in range(g.T):
        eps, unif = g.noise(it)
        o = orc.transition(fixed, moving, eps, unif)
        np.testing.assert_allclose(o['alpha'], g.t(it, 'alpha').numpy(), atol=5e-6)
        np.testing.assert_allclose(o['data'], g.t(it, 'data').numpy(), rtol=1e-5)
        np.testing.assert_allclose(o['reg'], g.t(it, 'reg').numpy(), rtol=1e-5)
        np.testing.assert_allclose(o['reg_energy'], g.t(it, 'reg_energy').numpy(), rtol=1e-5)
        assert torch.allclose(orc.log_std.detach(), g.t(it, 'gmm_log_std'), atol=1e-5)
        assert torch.allclose(orc.logits.detach(), g.t(it, 'gmm_logits'), atol=1e-5)
        assert torch.allclose(g.sub(o['curr_state']), g.t(it, 'curr_state'), atol=1e-5)
        assert torch.allclose(g.sub(o['displacement']), g.t(it, 'displacement'), atol=1e-4)  # north-star tolerance
        assert torch.allclose(g.sub(o['transformation']), g.t(it, 'transformation'), atol=1e-5)
        assert torch.allclose(g.sub(o['im_moving_warped']), g.t(it, 'im_moving_warped'), atol=1e-5)
        gv = g.t(it, 'grad_v')
        assert float((g.sub(o['grad_v']) - gv).abs().max()) <= 1e-4 * float(gv.abs().max())
        assert torch.allclose(g.sub(o['v_new']), g.t(it, 'v_new'), atol=1e-4)
        if g.has(it, 'reg_loc'):
            assert torch.allclose(orc.loc.detach().double(), g.t(it, 'reg_loc').double(), atol=1e-5)
            assert torch.allclose(orc.log_scale.detach().double(), g.t(it, 'reg_log_scale').double(), atol=1e-5)
        elif g.has(it, 'reg_log_w'):
            assert torch.allclose(orc.log_w_reg.detach(), g.t(it, 'reg_log_w'), atol=1e-6)


@pytest.mark.parametrize('name', FAST)
def test_oracle_matches_reference_fixture(name):
    _replay(name)


@pytest.mark.parametrize('name', SLOW)
def test_oracle_matches_reference_fixture_large(name):
    _replay(name)
