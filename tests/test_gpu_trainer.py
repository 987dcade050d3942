"""The drop-in surface on the GPU: config -> Trainer -> `_SGLD_transition` / `step` / `_run_MCMC`, and the stand-alone
module classes composed with torch autograd exactly as the reference composes its own (trainer.py:291-356)."""
import copy
import json
import math
import os

import numpy as np
import pytest
import torch

from ir_sgmcmc_amd.parse_config import ConfigParser
from ir_sgmcmc_amd.trainer import Trainer
from oracle import OracleChain, OracleConfig
from oracle import ops as O
from tests._report import GRAD_RTOL, check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_trainer(tmp_path, name, dims, **trainer_over):
    cfg = json.load(open(os.path.join(ROOT, 'configs', name)))
    cfg['trainer']['save_dir'] = str(tmp_path)
    cfg['data_loader']['args']['dims'] = list(dims)
    cfg['trainer'].update(trainer_over)
    config = ConfigParser.from_dict(copy.deepcopy(cfg), timestamp='t')
    dl = config.init_data_loader()
    losses = config.init_losses()
    tm, rm = config.init_transformation_and_registration_modules()
    return Trainer(config, dl, losses, tm, rm, config.init_metrics(), device=DEV), dl


def test_trainer_transition_matches_oracle(tmp_path):
    N = 20
    t, dl = make_trainer(tmp_path, 'synthetic_gmm_lognormal.json', (N, N, N), MCMC_init='identity')
    fixed, moving, vp = next(iter(dl))
    C = t.no_chains
    oc = OracleConfig(dims=(N, N, N), no_chains=C, reg_loss='RegLoss_LogNormal', reg_learnable=True)
    fx = {k: v.expand(C, *v.shape[1:]).contiguous() for k, v in fixed.items() if k != 'seg'}
    mv = {k: v.expand(C, *v.shape[1:]).contiguous() for k, v in moving.items() if k != 'seg'}
    orc = OracleChain(oc)
    orc.init_gmm(fx, mv)

    fd = {k: v.to(DEV) for k, v in fixed.items()}
    md = {k: v.to(DEV) for k, v in moving.items()}
    # the reference hands expanded views to the transition (trainer.py:361-362)
    fde = {k: v.expand(C, *v.shape[1:]) for k, v in fd.items()}
    mde = {k: v.expand(C, *v.shape[1:]) for k, v in md.items()}
    t._engine_init(fde, mde)
    t._GMM_init(fde, mde, None)  # the zero velocity sample the oracle's init uses (with `vp` the trainer draws sample_q_v, as the reference)
    assert torch.allclose(t.losses['data']['loss'].log_std.detach(), orc.log_std.detach(), atol=2e-4)
    t._SGLD_init(vp)
    assert t.v_curr_state.shape == (C, 3, N, N, N) and float(t.v_curr_state.abs().max()) == 0.0

    gen = torch.Generator().manual_seed(5)
    for it in range(2):
        eps = torch.randn(C, 3, N, N, N, generator=gen)
        unif = torch.rand(C, 3, N, N, N, generator=gen)
        o = orc.transition(fx, mv, eps, unif)
        loss_terms, output, aux = t.step(fde, mde, t.losses['data']['loss'], t.losses['reg']['loss'], eps=eps.to(DEV), unif=unif.to(DEV))
        assert set(loss_terms) == {'data', 'reg'} and set(output) == {'im_moving_warped', 'displacement', 'transformation', 'curr_state'}
        assert set(aux) == {'residuals', 'alpha', 'reg_energy'} and len(loss_terms['data']) == C
        T = 'trainer/gmm_lognormal'
        for c in range(C):
            check(T, 'data_term (rel)', loss_terms['data'][c].item() / abs(o['data'][c]), math.copysign(1.0, o['data'][c]), 1e-5)
            check(T, 'reg_term (rel)', loss_terms['reg'][c].item() / abs(o['reg'][c]), math.copysign(1.0, o['reg'][c]), 1e-5)
            check(T, 'alpha', aux['alpha'][c].item(), o['alpha'][c], 2e-5)
        check(T, 'displacement [voxels]', output['displacement'], o['displacement'], 1e-4)
        check(T, 'curr_state', output['curr_state'], o['curr_state'], 5e-6)
        # the reference's objects: the masked residual view (trainer.py:308) and CLONES of the four volumes (trainer.py:302-305)
        ref_masked = o['residuals'][fx['mask']].view(C, -1)
        assert aux['residuals'].shape == ref_masked.shape
        check(T, 'aux residuals (masked view)', aux['residuals'], ref_masked, 2e-4)
        assert output['displacement'].data_ptr() != t._outputs['displacement'].data_ptr()
        check(T, 'v_new', t.v_curr_state, o['v_new'], 0.4 * GRAD_RTOL * float(o['grad_v'].abs().max()) + 1e-5)
        t.v_curr_state.copy_(o['v_new'].to(DEV))
    st = t.sync_parameters()
    assert abs(float(t.losses['reg']['loss'].loc) - float(orc.loc)) < 1e-5
    assert torch.allclose(t.losses['data']['loss'].log_std.detach().cpu(), orc.log_std.detach(), atol=5e-5)
    assert st.iteration == 2


def test_run_mcmc_end_to_end(tmp_path):
    t, dl = make_trainer(tmp_path, 'synthetic_gmm_lognormal.json', (24, 24, 24), no_iters_burn_in=4, no_samples_MCMC=8,
                         log_period_MCMC=4)
    t.run()
    assert t.MCMC_sampling_speed > 0 and bool(torch.isfinite(t.v_curr_state).all())
    assert t.displacement_mean.shape == (3, 24, 24, 24) and bool(torch.isfinite(t.displacement_std).all())
    res = t.metrics.result()
    assert res['MCMC/chain_1/VD/alpha'] > 0 and 'MCMC/GMM/scale_3' in res and res['MCMC/chain_0/no_non_diffeomorphic_voxels'] == 0
    assert t.engine.state().iteration == 4 + 8 + 100   # burn-in + samples + the reference's 100-sample speed test


def test_checkpoint_resume_continues_the_chain_bit_for_bit(tmp_path):
    """A run that is stopped at a checkpoint and resumed ends exactly where the uninterrupted run ends: velocity field,
    hyper-parameter state (incl. Adam moments and the Philox counter) and the running posterior moments are all restored.
    It also leaves the reference's output files behind (posterior mean / std as .vtk, samples as .nii.gz / .vtk)."""
    kw = dict(no_iters_burn_in=4, no_samples_MCMC=8, log_period_MCMC=2, checkpoint_period=6, save_samples=True)
    a, _ = make_trainer(tmp_path / 'a', 'synthetic_gmm_lognormal.json', (16, 16, 16), **kw)
    a.run()
    ck = a.config.save_dirs['checkpoints'] / 'checkpoint_0000006.pt'
    assert ck.is_file() and (a.config.save_dirs['checkpoints'] / 'checkpoint_0000012.pt').is_file()
    b, _ = make_trainer(tmp_path / 'b', 'synthetic_gmm_lognormal.json', (16, 16, 16), resume=str(ck), **kw)
    b.run()
    assert torch.equal(a.v_curr_state, b.v_curr_state)
    assert torch.equal(a.displacement_mean, b.displacement_mean) and torch.equal(a.displacement_std, b.displacement_std)
    sa, sb = a.engine.state(), b.engine.state()
    assert sa.iteration == sb.iteration == 112
    assert list(sa.gmm_log_std) == list(sb.gmm_log_std) and list(sa.reg_param) == list(sb.reg_param)
    # files
    from ir_sgmcmc_amd.utils.imageio import read_nifti, read_vtk_vectors
    samples = a.config.save_dirs['samples']
    kind, dims, mean = read_vtk_vectors(str(samples / 'MCMC_sample_mean.vtk'))
    assert kind == 'STRUCTURED_POINTS' and dims == (16, 16, 16)
    assert np.allclose(mean, a.displacement_mean.cpu().numpy(), atol=1e-6)
    assert (samples / 'MCMC_sample_std_dev_masked.vtk').is_file()
    im, _ = read_nifti(str(samples / 'MCMC' / 'chain_1_sample_0000012_im_moving_warped.nii.gz'))
    assert im.shape == (16, 16, 16) and np.isfinite(im).all()


def test_module_classes_compose_under_autograd_like_the_reference():
    """SGLD.apply -> SobolevGrad.apply -> SVF_3D -> add_noise -> RegistrationModule -> GMM.map -> losses, then
    loss.backward() and SGD: the reference's own composition, every volume op a HIP kernel."""
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.model.loss import GMM, RegLoss_L2
    from ir_sgmcmc_amd.utils import SGLD, SVF_3D, RegistrationModule, SobolevGrad, transform_coordinates
    N = 16
    f1, m1 = synthetic_pair((N, N, N), seed=2)
    fixed = {k: v.unsqueeze(0) for k, v in f1.items()}
    moving = {k: v.unsqueeze(0) for k, v in m1.items()}
    gen = torch.Generator().manual_seed(9)
    v0 = O.separable_conv3d_replicate(4.0 * torch.randn(1, 3, N, N, N, generator=gen), O.sobolev_kernel_1d(3, 0.5)).contiguous()
    eps = torch.randn(1, 3, N, N, N, generator=gen)
    unif = torch.rand(1, 3, N, N, N, generator=gen)
    oc = OracleConfig(dims=(N, N, N), virtual_decimation=False)
    orc = OracleChain(oc, v0=v0)
    with torch.no_grad():
        orc.log_std.copy_(torch.tensor([-4.0, -2.5, -1.0, 0.5]))
    o = orc.transition(fixed, moving, eps, unif)
    log_std_after = orc.log_std.detach().clone()
    logits_after = orc.logits.detach().clone()

    v = v0.to(DEV).requires_grad_(True)
    sigma = torch.ones_like(v)
    svf, reg, gmm, rl = SVF_3D((N, N, N)), RegistrationModule(), GMM(4, 1).to(DEV), RegLoss_L2(1.4, 'GradientOperator', [N, N, N]).to(DEV)
    with torch.no_grad():   # the data term is evaluated with the mixture AFTER its step (trainer.py:316-327)
        gmm.log_std.copy_(log_std_after)
        gmm.logits.copy_(logits_after)
    k = torch.from_numpy(O.sobolev_kernel_1d(3, 0.5)).float()
    S = torch.stack((k, k, k)).unsqueeze(1)
    Sd = {'x': S.unsqueeze(2).unsqueeze(2), 'y': S.unsqueeze(2).unsqueeze(4), 'z': S.unsqueeze(3).unsqueeze(4)}
    curr = SGLD.apply(v, sigma, 0.4, eps.to(DEV))
    curr_s = SobolevGrad.apply(curr, Sd, (3,) * 6)
    transformation, displacement = svf(curr_s)
    t_noise = transformation + transform_coordinates(-2.0 * 0.1 * unif.to(DEV) + 0.1)
    warped = reg(moving['im'].to(DEV), t_noise)
    z = gmm.map(fixed['im'].to(DEV), warped)
    data_term = gmm(z[fixed['mask'].to(DEV)])
    reg_term, log_y = rl(curr_s)
    (data_term + reg_term.sum()).backward()
    T = 'modules/autograd_composition'
    check(T, 'displacement [voxels]', displacement, o['displacement'], 1e-4)
    check(T, 'im_moving_warped', warped, o['im_moving_warped'], 1e-5)
    check(T, 'data_term (rel)', float(data_term) / abs(o['data'][0]), 1.0, 1e-5)
    check(T, 'reg_term (rel)', float(reg_term[0]) / abs(o['reg'][0]), math.copysign(1.0, o['reg'][0]), 1e-5)
    gmax = float(o['grad_v'].abs().max())
    check(T, 'grad_v (rel to max)', v.grad / gmax, o['grad_v'] / gmax, GRAD_RTOL)
    # nearest-neighbour warps keep dtype (utils/registration.py:20-27)
    assert reg(moving['seg'].to(DEV), transformation.detach()).dtype == torch.int16
    assert reg(moving['mask'].to(DEV), transformation.detach()).dtype == torch.bool
