"""Host-side mirror of the reference's plugin surface (config factory, hyper-priors, Adam, loss formulas) -- CPU only."""
import copy
import json
import math
import os

import numpy as np
import pytest
import torch

from ir_sgmcmc_amd.model import distributions as D
from ir_sgmcmc_amd.model import loss as ML
from ir_sgmcmc_amd.optimizers import Adam
from ir_sgmcmc_amd.parse_config import ConfigParser
from ir_sgmcmc_amd.utils.functions import Sobolev_kernel_1D
from oracle import ops as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_cfg(name):
    return json.load(open(os.path.join(ROOT, 'configs', name)))


def test_priors_match_oracle_formulas():
    x = torch.tensor([-1.3, 0.2, 0.9, 2.0])
    assert torch.allclose(D.LogScaleNormalPrior(0.0, 2.3)(x), O.normal_log_pdf(x, 0.0, 2.3), atol=1e-6)
    lp = torch.log_softmax(x, 0)
    assert torch.allclose(D.DirichletPrior(4, 0.5)(lp), O.dirichlet_log_pdf(lp, torch.full((4,), 0.5)), atol=1e-6)
    ly = torch.tensor([9.5, 10.5])
    dof = 3.0 * 16 ** 3
    assert torch.allclose(D.LogEnergyExpGammaPrior(1.4, dof)(ly).float(), O.expgamma_log_pdf(ly, 0.5 * dof, 0.7), rtol=1e-6)
    shape = 0.5 * dof
    lw = torch.tensor(0.3)
    assert torch.allclose(D.LogPrecisionExpGammaPrior(shape=shape, rate=1.0 / shape)(lw), O.expgamma_log_pdf(lw, shape, 1.0 / shape), rtol=1e-6)


def test_reg_losses_match_oracle_formulas():
    dims = [16, 16, 16]
    dof = 3.0 * 16 ** 3
    y = torch.tensor([1234.5, 98765.4])
    l2 = ML.RegLoss_L2(1.4, diff_op='GradientOperator', dims=dims)
    a, b = l2._loss(y)
    ra, rb = O.reg_l2(y, torch.tensor(math.log(1.4)), dof)
    assert torch.allclose(a, ra, rtol=1e-6) and torch.allclose(b, rb)
    ln = ML.RegLoss_LogNormal(1.4, diff_op='GradientOperator', dims=dims, learnable=True)
    loc, log_scale = O.reg_lognormal_init(1.4, dof)
    assert abs(float(ln.loc) - float(loc)) < 1e-5 and abs(float(ln.log_scale) - float(log_scale)) < 1e-5
    a, b = ln._loss(y)
    ra, rb = O.reg_lognormal(y, loc, log_scale, dof)
    assert torch.allclose(a.double(), ra.double(), rtol=1e-6)
    assert ln.learnable and isinstance(ln.diff_op, ML.GradientOperator)


def test_gmm_log_pdf_and_properties_match_oracle():
    g = ML.GMM(4, 1)
    g.init_parameters(0.37)
    assert torch.allclose(g.log_std.detach(), O.gmm_init_log_std(0.37, 4))
    with torch.no_grad():
        g.logits.copy_(torch.tensor([0.1, -0.4, 0.3, 0.0]))
    z = torch.randn(50) * 0.3
    assert torch.allclose(g.log_pdf(z), O.gmm_log_pdf(z, g.log_std, g.logits), atol=1e-6)
    assert torch.allclose(g(z), O.gmm_nll(z, g.log_std, g.logits), rtol=1e-6)
    assert torch.allclose(g.proportions.sum(), torch.tensor(1.0), atol=1e-6)
    assert torch.allclose(g.log_scales, g.log_std) and torch.allclose(g.scales, g.log_std.exp())


def test_adam_rate_decay_matches_oracle_over_many_steps():
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.tensor([0.3, -1.2, 2.0]))
    q = p.detach().clone()
    opt = Adam([{'params': [p], 'lr': 0.2}], lr_decay=0.001)
    ref = O.AdamRateDecay([{'params': [q], 'lr': 0.2}], lr_decay=0.001)
    for _ in range(30):
        g = torch.randn(3)
        p.grad = g.clone()
        opt.step()
        ref.step([g])
    assert torch.allclose(p.detach(), q, atol=1e-6)
    with pytest.raises(ValueError):
        Adam([p], lr=-1.0)


def test_sobolev_kernel_both_outputs():
    k, ks = Sobolev_kernel_1D(3, 0.5)
    np.testing.assert_allclose(k, O.sobolev_kernel_1d(3, 0.5), atol=1e-12)
    assert abs(ks.sum() - 1.0) < 1e-12 and ks[3] > k[3]  # the square-root kernel is sharper


def test_config_factory_resolves_reference_schema(tmp_path):
    cfg = load_cfg('synthetic_gmm_lognormal.json')
    cfg['trainer']['save_dir'] = str(tmp_path)
    config = ConfigParser.from_dict(copy.deepcopy(cfg), timestamp='t0')
    losses = config.init_losses()
    assert type(losses['data']['loss']).__name__ == 'GMM' and losses['data']['loss'].no_components == 4
    assert type(losses['reg']['loss']).__name__ == 'RegLoss_LogNormal' and losses['reg']['loss'].learnable
    assert set(losses['reg']) == {'loss', 'loc_prior', 'scale_prior'}
    assert float(losses['reg']['loc_prior'].dof) == 3.0 * 64 ** 3          # injected by init_losses (parse_config.py:128)
    tm, rm = config.init_transformation_and_registration_modules()
    assert type(tm).__name__ == 'SVF_3D' and tm.no_steps == 12 and type(rm).__name__ == 'RegistrationModule'
    dl = config.init_data_loader()
    fixed, moving, vp = next(iter(dl))
    assert fixed['im'].shape == (1, 1, 64, 64, 64) and fixed['mask'].dtype == torch.bool and moving['seg'].dtype == torch.int16
    assert vp['mu'].shape == (1, 3, 64, 64, 64) and abs(float(vp['log_var'][0, 0, 0, 0, 0]) - math.log(0.25)) < 1e-6
    opt = config.init_optimizer_GMM(losses['data']['loss'])
    assert [g['lr'] for g in opt.param_groups] == [0.2, 0.2] and opt.param_groups[0]['lr_decay'] == 0.001
    assert (tmp_path / 'synthetic_gmm_lognormal' / 't0' / 'config.json').is_file()

    from ir_sgmcmc_amd.trainer import Trainer
    t = Trainer(config, dl, losses, tm, rm, config.init_metrics())
    ec = t._engine_config()
    assert ec.dims == (64, 64, 64) and ec.no_chains == 2 and ec.lr == 0.4 and ec.sobolev_s == 3 and ec.uniform_noise == 0.1
    assert ec.data_loss == 'GMM' and ec.reg_loss == 'RegLoss_LogNormal' and ec.reg_learnable and ec.virtual_decimation
    assert ec.scale_prior == (0.0, pytest.approx(2.3)) and ec.reg_scale_prior == (pytest.approx(2.8), pytest.approx(5.0))
    assert ec.reg_lr == (0.01, 0.01) and ec.gmm_lr_decay == 0.001 and list(ec.dirichlet_alpha) == [0.5] * 4
    with pytest.raises(RuntimeError):   # no engine yet, and never a CPU fallback
        t._SGLD_transition(fixed, moving)


def test_svffd_config_and_control_grid(tmp_path):
    cfg = load_cfg('synthetic_ssd_l2_128.json')
    cfg['trainer']['save_dir'] = str(tmp_path)
    cfg['transformation_module'] = {'type': 'SVFFD_3D', 'args': {'cps': [4, 4, 4]}}
    config = ConfigParser.from_dict(cfg, timestamp='t1')
    tm, _ = config.init_transformation_and_registration_modules()
    dl = config.init_data_loader()
    assert dl.dims_v == (35, 35, 35) and tm.cps == (4, 4, 4)   # ceil(127/4)+3, SURVEY.md section 8(a) row a4
    losses = config.init_losses()
    assert type(losses['data']['loss']).__name__ == 'SSD' and 'scale_prior' not in losses['data']


def test_exact_division_by_markstein_correction(tmp_path):
    """The kernels divide by (n - 1) with a reciprocal + one FMA correction instead of the full IEEE sequence (common.h
    div_exact); results must be bit-identical to the division the reference performs (utils/util.py:418-429)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    src = os.path.join(os.path.dirname(__file__), 'csrc', 'div_exact_check.c')
    exe = str(tmp_path / 'div_exact_check')
    subprocess.run(['gcc', '-O2', '-ffp-contract=off', '-o', exe, src, '-lm'], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert 'mismatches 0' in out.stdout
