#!/usr/bin/env python3
"""Generate golden fixtures from the real reference and validate `oracle/` against it.

Runs ONLY in the build container (needs /root/reference, read-only).  It drives the reference's
unmodified `Trainer._SGLD_transition` (trainer/trainer.py:291-356) on CPU through
`Trainer.__new__` + the attributes the method reads, with seeded noise, and
stores inputs + the reference's outputs as `.npz` under tests/golden/.  For every variant the
oracle is run on the same inputs and the max deviations are printed and asserted.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--check-only]

Fixtures are data (inputs and expected outputs); no reference source is stored.
"""
import argparse
import json
import math
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')

from _ref_import import import_reference  # noqa: E402

from ir_sgmcmc_amd.data_loader import synthetic_pair  # noqa: E402
from oracle import OracleChain, OracleConfig, ops  # noqa: E402

torch.set_num_threads(8)

VARIANTS = {
    # name: (N, transitions, cfg overrides, init, store)
    'n16_svf_l2_vd_c1': dict(N=16, T=3, cfg=dict(), init='smooth_noise'),
    'n16_svf_lognormal_learn_c2_psgld': dict(N=16, T=3, cfg=dict(no_chains=2, reg_loss='RegLoss_LogNormal',
                                                                  reg_learnable=True), init='vi', sigma=0.5),
    'n16_svf_l2_learn_novd_nonoise_s2': dict(N=16, T=3, cfg=dict(reg_learnable=True, virtual_decimation=False,
                                                                  uniform_noise=None, lcc_s=2, sobolev_s=2),
                                             init='noise'),
    'n16_svffd4_l2_c1': dict(N=16, T=3, cfg=dict(transformation='SVFFD_3D', cps=(4, 4, 4), lr=0.01),
                             init='noise'),
    'n16_svffd2_lognormal_learn_c2': dict(N=16, T=2, cfg=dict(transformation='SVFFD_3D', cps=(2, 2, 2), lr=0.01,
                                                               no_chains=2, reg_loss='RegLoss_LogNormal',
                                                               reg_learnable=True, sobolev_s=1), init='noise'),
    'n16_svf_big_displacement': dict(N=16, T=2, cfg=dict(w_reg=0.2), init='big'),
    'n16_svf_student_c1': dict(N=16, T=2, cfg=dict(reg_loss='RegLoss_Student', student=(2e-6, 0.7, 1e-6, 1e-6)), init='smooth_noise'),
    'n16_svf_lognormal_l2_c1': dict(N=16, T=2, cfg=dict(reg_loss='RegLoss_LogNormal_L2', w_reg=0.9), init='smooth_noise'),
    # non-learnable RegLoss_LogNormal (model/loss.py:273-312): the coefficient of SURVEY.md row a13 without the hyper-prior terms
    'n16_svf_lognormal_fixed_c1': dict(N=16, T=2, cfg=dict(reg_loss='RegLoss_LogNormal', reg_learnable=False, w_reg=1.1),
                                       init='smooth_noise'),
    'n32_svf_l2_vd_c1': dict(N=32, T=2, cfg=dict(), init='smooth_noise'),
    'n32_svf_lognormal_learn_c2': dict(N=32, T=1, cfg=dict(no_chains=2, reg_loss='RegLoss_LogNormal',
                                                            reg_learnable=True), init='vi', sigma=0.5),
    'n64_svf_l2_vd_c1': dict(N=64, T=2, cfg=dict(), init='smooth_noise', subsample=True),
}


def initial_state(name, spec, cfg):
    """v0 and sigma for a variant (seeded; stored in the fixture so tests never depend on RNG replay)."""
    g = torch.Generator().manual_seed(1000 + sum(map(ord, name)))
    shape = (cfg.no_chains, 3, *cfg.dims_v)
    sigma = float(spec.get('sigma', 1.0))
    kind = spec['init']
    if kind == 'identity':
        v0 = torch.zeros(shape)
    elif kind == 'noise':
        v0 = torch.randn(shape, generator=g)
    elif kind == 'smooth_noise':
        k = ops.sobolev_kernel_1d(3, 0.5)
        v0 = ops.separable_conv3d_replicate(3.0 * torch.randn(shape, generator=g), k)
    elif kind == 'vi':  # mu + eps sigma + x u, data_loader/datasets.py:57-68 + utils/sampler.py:4-21
        v0 = torch.randn(shape, generator=g) * sigma + torch.randn(cfg.no_chains, 1, 1, 1, 1, generator=g) * 0.1
    elif kind == 'big':  # several voxels of displacement, exercises border clamping and far gathers
        k = ops.sobolev_kernel_1d(3, 0.5)
        v0 = ops.separable_conv3d_replicate(12.0 * torch.randn(shape, generator=g), k)
        v0[:, 0] += 2.5
    else:
        raise ValueError(kind)
    return v0.float().contiguous(), sigma


def build_reference(ref, cfg, fixed, moving, v0, sigma):
    """A reference Trainer wired by hand (no BaseTrainer.__init__: hard-coded cuda:0 + tensorboard)."""
    t = ref.Trainer.__new__(ref.Trainer)
    t.device = 'cpu'
    t.no_chains = cfg.no_chains
    if cfg.transformation == 'SVF_3D':
        t.transformation_module = ref.utils.SVF_3D(cfg.dims, no_steps=cfg.no_steps)
    else:
        t.transformation_module = ref.utils.SVFFD_3D(cfg.dims, cfg.cps)
    t.registration_module = ref.utils.RegistrationModule()

    gmm = ref.loss.GMM(cfg.gmm_components, cfg.lcc_s)
    losses = {'data': {'loss': gmm,
                       'scale_prior': ref.distr.LogScaleNormalPrior(*cfg.scale_prior),
                       'proportion_prior': ref.distr.DirichletPrior(cfg.gmm_components, cfg.dirichlet_alpha)},
              'reg': {}}
    reg_cls = getattr(ref.loss, cfg.reg_loss)
    if cfg.reg_loss == 'RegLoss_Student':
        nu0, lambda0, a0, b0 = cfg.student
        reg = reg_cls(diff_op='GradientOperator', dims=list(cfg.dims), nu0=nu0, lambda0=lambda0, a0=a0, b0=b0)
    elif cfg.reg_loss == 'RegLoss_LogNormal_L2':
        reg = reg_cls(w_reg=cfg.w_reg, diff_op='GradientOperator', dims=list(cfg.dims))
    else:
        reg = reg_cls(w_reg=cfg.w_reg, diff_op='GradientOperator', dims=list(cfg.dims), learnable=cfg.reg_learnable)
    losses['reg']['loss'] = reg
    t.optimizer_reg = None
    if cfg.reg_learnable:
        if cfg.reg_loss == 'RegLoss_LogNormal':
            losses['reg']['loc_prior'] = ref.distr.LogEnergyExpGammaPrior(cfg.w_reg, cfg.dof, nu=cfg.reg_loc_prior_nu)
            losses['reg']['scale_prior'] = ref.distr.LogScaleNormalPrior(*cfg.reg_scale_prior)
            t.optimizer_reg = ref.optim.Adam([{'params': [reg.loc], 'lr': cfg.reg_lr[0]},
                                              {'params': [reg.log_scale], 'lr': cfg.reg_lr[1]}],
                                             lr_decay=cfg.reg_lr_decay)
        else:
            shape = 0.5 * cfg.dof
            losses['reg']['w_reg_prior'] = ref.distr.LogPrecisionExpGammaPrior(shape=shape, rate=1.0 / shape)
            t.optimizer_reg = ref.optim.Adam(reg.parameters(), lr=cfg.reg_lr[0], lr_decay=cfg.reg_lr_decay)
    t.losses = losses
    t.optimizer_GMM = ref.optim.Adam([{'params': [gmm.log_std], 'lr': cfg.gmm_lr_log_std},
                                      {'params': [gmm.logits], 'lr': cfg.gmm_lr_logits}],
                                     lr_decay=cfg.gmm_lr_decay)
    t.add_noise_uniform = cfg.uniform_noise is not None
    t.alpha = cfg.uniform_noise
    t.virutal_decimation = cfg.virtual_decimation  # (sic) trainer/trainer.py:42

    S, _ = ref.utils.Sobolev_kernel_1D(cfg.sobolev_s, cfg.sobolev_lambda)  # trainer/trainer.py:568-583
    S = torch.from_numpy(S).float().unsqueeze(0)
    S = torch.stack((S, S, S), 0)
    t.S = {'x': S.unsqueeze(2).unsqueeze(2), 'y': S.unsqueeze(2).unsqueeze(4), 'z': S.unsqueeze(3).unsqueeze(4)}
    t.padding = (cfg.sobolev_s,) * 6

    t.v_curr_state = v0.clone().requires_grad_(True)
    t.SGLD_params = {'sigma': torch.full_like(v0, sigma), 'tau': cfg.lr}
    t.optimizer_SG_MCMC = torch.optim.SGD([t.v_curr_state], lr=cfg.lr)
    return t, gmm, reg


def reference_gmm_init(ref, t, gmm, cfg, fixed, moving):
    """`Trainer.__GMM_init` (trainer/trainer.py:529-547) with the zero velocity sample."""
    v = torch.zeros(1, 3, *cfg.dims_v)
    v_s = ref.utils.SobolevGrad.apply(v, t.S, t.padding)
    transformation, _ = t.transformation_module(v_s)
    warped = t.registration_module(moving['im'][:1], transformation)
    res = gmm.map(fixed['im'][:1], warped)
    res_masked = res[fixed['mask'][:1]]
    gmm.init_parameters(torch.std(res_masked))
    alpha = t._Trainer__get_VD_factor(res, fixed['mask'][:1], gmm)
    for _ in range(25):
        t._step_GMM(res_masked, alpha)


def adam_state(opt):
    out = []
    for g in opt.param_groups:
        for p in g['params']:
            st = opt.state[p]
            out.append((int(st['step']), st['exp_avg'].detach().clone(), st['exp_avg_sq'].detach().clone()))
    return out


def maxdiff(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max())


def run_variant(ref, name, spec, write):
    N = spec['N']
    cfg = OracleConfig(dims=(N, N, N), **spec['cfg'])
    C = cfg.no_chains
    fixed1, moving1 = synthetic_pair(cfg.dims, seed=0)
    fixed = {k: v.unsqueeze(0).expand(C, *v.shape).contiguous() for k, v in fixed1.items()}
    moving = {k: v.unsqueeze(0).expand(C, *v.shape).contiguous() for k, v in moving1.items()}
    v0, sigma = initial_state(name, spec, cfg)

    t, gmm, reg = build_reference(ref, cfg, fixed, moving, v0, sigma)
    reference_gmm_init(ref, t, gmm, cfg, fixed, moving)

    orc = OracleChain(cfg, v0=v0, sigma=torch.full_like(v0, sigma))
    orc.init_gmm(fixed, moving)
    d_init = max(maxdiff(gmm.log_std, orc.log_std), maxdiff(gmm.logits, orc.logits))

    store = {'config': np.frombuffer(json.dumps({**spec['cfg'], 'N': N, 'sigma': sigma}).encode(), dtype=np.uint8),
             'gmm_log_std_init': gmm.log_std.detach().numpy().copy(), 'gmm_logits_init': gmm.logits.detach().numpy().copy()}
    for i, (step, m, v) in enumerate(adam_state(t.optimizer_GMM)):
        store[f'gmm_adam{i}_step'] = np.int64(step)
        store[f'gmm_adam{i}_m'] = m.numpy()
        store[f'gmm_adam{i}_v'] = v.numpy()
    sub = spec.get('subsample', False)
    if not sub:
        store.update(fixed=fixed1['im'].numpy(), moving=moving1['im'].numpy(), mask=fixed1['mask'].numpy(), v0=v0.numpy())
    else:
        store.update(input_checksum=np.array([fixed1['im'].double().sum(), moving1['im'].double().sum(),
                                              v0.double().sum(), (v0.double() ** 2).sum()]))
    if cfg.reg_loss == 'RegLoss_LogNormal':
        store.update(reg_loc_init=reg.loc.detach().numpy().copy(), reg_log_scale_init=reg.log_scale.detach().numpy().copy())

    worst = {}
    for it in range(spec['T']):
        seed = 7000 + 13 * it
        torch.manual_seed(seed)
        eps = torch.randn_like(t.SGLD_params['sigma'])
        unif = torch.rand(C, 3, N, N, N) if cfg.uniform_noise is not None else None
        torch.manual_seed(seed)  # the reference draws randn_like(sigma) then rand(shape): same stream
        loss_terms, output, aux = t._SGLD_transition(fixed, moving, gmm, reg)
        grad_v = t.v_curr_state.grad.detach().clone()

        o = orc.transition(fixed, moving, eps, unif)

        ref_out = {
            'alpha': np.array([float(a) for a in aux['alpha']]),
            'data': np.array([float(x) for x in loss_terms['data']]),
            'reg': np.array([float(x) for x in loss_terms['reg']]),
            'reg_energy': np.array([float(x) for x in aux['reg_energy']]),
            'gmm_log_std': gmm.log_std.detach().numpy().copy(), 'gmm_logits': gmm.logits.detach().numpy().copy(),
            'curr_state': output['curr_state'].numpy(), 'displacement': output['displacement'].numpy(),
            'transformation': output['transformation'].numpy(), 'im_moving_warped': output['im_moving_warped'].numpy(),
            'grad_v': grad_v.numpy(), 'v_new': t.v_curr_state.detach().numpy().copy(),
        }
        # residual field: the reference only returns the masked view; scatter it back for a dense fixture
        z_dense = torch.zeros(C, 1, N, N, N)
        z_dense[fixed['mask']] = aux['residuals'].detach().reshape(-1)
        ref_out['residuals'] = z_dense.numpy()
        if cfg.reg_loss == 'RegLoss_LogNormal':
            ref_out['reg_loc'] = reg.loc.detach().numpy().copy()
            ref_out['reg_log_scale'] = reg.log_scale.detach().numpy().copy()
        elif cfg.reg_loss == 'RegLoss_L2':
            ref_out['reg_log_w'] = reg.log_w_reg.detach().numpy().copy()

        # ---- oracle vs reference
        cmp = {
            'alpha': maxdiff(ref_out['alpha'], o['alpha']),
            'data_rel': maxdiff(ref_out['data'], o['data']) / max(1.0, float(np.abs(ref_out['data']).max())),
            'reg_rel': maxdiff(ref_out['reg'], o['reg']) / max(1.0, float(np.abs(ref_out['reg']).max())),
            'gmm': max(maxdiff(ref_out['gmm_log_std'], orc.log_std), maxdiff(ref_out['gmm_logits'], orc.logits)),
            'curr_state': maxdiff(ref_out['curr_state'], o['curr_state']),
            'displacement': maxdiff(ref_out['displacement'], o['displacement']),
            'warped': maxdiff(ref_out['im_moving_warped'], o['im_moving_warped']),
            'residuals': maxdiff(ref_out['residuals'], torch.where(fixed['mask'], o['residuals'], torch.zeros(()))),
            'grad_rel': maxdiff(ref_out['grad_v'], o['grad_v']) / max(1e-30, float(np.abs(ref_out['grad_v']).max())),
            'v_new': maxdiff(ref_out['v_new'], o['v_new']),
        }
        if cfg.reg_loss == 'RegLoss_LogNormal':
            cmp['reg_params'] = max(maxdiff(ref_out['reg_loc'], orc.loc), maxdiff(ref_out['reg_log_scale'], orc.log_scale))
        elif cfg.reg_loss == 'RegLoss_L2':
            cmp['reg_params'] = maxdiff(ref_out['reg_log_w'], orc.log_w_reg)
        for k, v in cmp.items():
            worst[k] = max(worst.get(k, 0.0), v)

        store[f't{it}_seed'] = np.int64(seed)
        if not sub:
            store[f't{it}_eps'] = eps.numpy()
            if unif is not None:
                store[f't{it}_unif'] = unif.numpy()
        else:
            store[f't{it}_noise_checksum'] = np.array([eps.double().sum(), unif.double().sum() if unif is not None else 0.0])
        for k, v in ref_out.items():
            if sub and isinstance(v, np.ndarray) and v.ndim == 5:
                if k == 'v_new' and it + 1 < spec['T']:
                    # the FULL state the reference carries into the next transition: the tests re-synchronise on it, so that every
                    # transition of a sub-sampled fixture is compared on equal inputs too (3 MB)
                    store[f't{it}_v_new_full'] = v
                v = v[:, :, ::4, ::4, ::4].copy()
            store[f't{it}_{k}'] = v

    print(f'{name:42s} gmm_init {d_init:.1e} | ' + ' '.join(f'{k} {v:.1e}' for k, v in worst.items()))
    tol = dict(alpha=2e-5, data_rel=2e-5, reg_rel=2e-6, gmm=2e-5, curr_state=1e-5, displacement=2e-5, warped=1e-5,
               residuals=2e-3, grad_rel=2e-3, v_new=2e-3, reg_params=1e-5)
    bad = {k: v for k, v in worst.items() if not v <= tol[k]}
    assert d_init < 1e-4 and not bad, f'oracle deviates from the reference in {name}: {bad}'
    if write:
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **store)
        print(f'    wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)')


# Free-running trajectories: T transitions of the reference loop body trainer/trainer.py:371-379 with NO
# re-synchronisation of the state in between -- what the fixtures above cannot show is drift.  Stored: inputs, the seed of every
# transition's noise (regenerated by the tests with the same torch CPU generator; checksums guard that), and per transition the
# loss terms, alpha, the energy, ||v||, the mixture / regulariser parameters; the final v (sub-sampled above 16^3).
TRAJECTORIES = {
    'traj_n16_svf_l2_vd_c1': dict(N=16, T=10, cfg=dict(), init='smooth_noise'),
    'traj_n32_svf_lognormal_learn_c2': dict(N=32, T=10, cfg=dict(no_chains=2, reg_loss='RegLoss_LogNormal', reg_learnable=True),
                                            init='vi', sigma=0.5),
}


def run_trajectory(ref, name, spec, write):
    N, T = spec['N'], spec['T']
    cfg = OracleConfig(dims=(N, N, N), **spec['cfg'])
    C = cfg.no_chains
    fixed1, moving1 = synthetic_pair(cfg.dims, seed=0)
    fixed = {k: v.unsqueeze(0).expand(C, *v.shape).contiguous() for k, v in fixed1.items()}
    moving = {k: v.unsqueeze(0).expand(C, *v.shape).contiguous() for k, v in moving1.items()}
    v0, sigma = initial_state(name, spec, cfg)
    t, gmm, reg = build_reference(ref, cfg, fixed, moving, v0, sigma)
    reference_gmm_init(ref, t, gmm, cfg, fixed, moving)
    orc = OracleChain(cfg, v0=v0, sigma=torch.full_like(v0, sigma))
    orc.init_gmm(fixed, moving)
    store = {'config': np.frombuffer(json.dumps({**spec['cfg'], 'N': N, 'sigma': sigma}).encode(), dtype=np.uint8),
             'gmm_log_std_init': gmm.log_std.detach().numpy().copy(), 'gmm_logits_init': gmm.logits.detach().numpy().copy(),
             'fixed': fixed1['im'].numpy(), 'moving': moving1['im'].numpy(), 'mask': fixed1['mask'].numpy(), 'v0': v0.numpy()}
    for i, (step, m, v) in enumerate(adam_state(t.optimizer_GMM)):
        store[f'gmm_adam{i}_step'] = np.int64(step)
        store[f'gmm_adam{i}_m'] = m.numpy()
        store[f'gmm_adam{i}_v'] = v.numpy()
    if cfg.reg_loss == 'RegLoss_LogNormal':
        store.update(reg_loc_init=reg.loc.detach().numpy().copy(), reg_log_scale_init=reg.log_scale.detach().numpy().copy())
    rows, worst = [], {}
    for it in range(T):
        seed = 9100 + 17 * it
        torch.manual_seed(seed)
        eps = torch.randn_like(t.SGLD_params['sigma'])
        unif = torch.rand(C, 3, N, N, N) if cfg.uniform_noise is not None else None
        torch.manual_seed(seed)
        loss_terms, output, aux = t._SGLD_transition(fixed, moving, gmm, reg)
        o = orc.transition(fixed, moving, eps, unif)
        v_now = t.v_curr_state.detach()
        row = [float(x) for x in loss_terms['data']] + [float(x) for x in loss_terms['reg']] + [float(a) for a in aux['alpha']] + \
              [float(x) for x in aux['reg_energy']] + [float(v_now[c].double().norm()) for c in range(C)]
        rows.append(row)
        store[f't{it}_seed'] = np.int64(seed)
        store[f't{it}_noise_checksum'] = np.array([eps.double().sum(), unif.double().sum() if unif is not None else 0.0])
        store[f't{it}_gmm_log_std'] = gmm.log_std.detach().numpy().copy()
        store[f't{it}_gmm_logits'] = gmm.logits.detach().numpy().copy()
        if cfg.reg_loss == 'RegLoss_LogNormal':
            store[f't{it}_reg_params'] = np.array([float(reg.loc), float(reg.log_scale)])
        cmp = {'data_rel': maxdiff(row[:C], o['data']) / max(1.0, max(abs(x) for x in row[:C])),
               'reg_rel': maxdiff(row[C:2 * C], o['reg']) / max(1.0, max(abs(x) for x in row[C:2 * C])),
               'alpha': maxdiff(row[2 * C:3 * C], o['alpha']),
               'v_rel': maxdiff(v_now, o['v_new']) / float(v_now.abs().max())}
        for k, v in cmp.items():
            worst[k] = max(worst.get(k, 0.0), v)
    store['trajectory'] = np.array(rows)  # [T, 5 C]: data (C), reg (C), alpha (C), energy (C), ||v|| (C)
    v_fin = t.v_curr_state.detach().numpy()
    store['v_final'] = v_fin if N <= 16 else v_fin[:, :, ::4, ::4, ::4].copy()
    print(f'{name:42s} oracle drift over {T} free-running transitions: ' + ' '.join(f'{k} {v:.1e}' for k, v in worst.items()))
    assert worst['data_rel'] < 1e-4 and worst['reg_rel'] < 1e-5 and worst['alpha'] < 1e-4 and worst['v_rel'] < 1e-3, worst
    if write:
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **store)
        print(f'    wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--check-only', action='store_true', help='validate the oracle, write nothing')
    ap.add_argument('--only', default=None)
    args = ap.parse_args()
    ref = import_reference()
    for name, spec in VARIANTS.items():
        if args.only and args.only not in name:
            continue
        run_variant(ref, name, spec, write=not args.check_only)
    for name, spec in TRAJECTORIES.items():
        if args.only and args.only not in name:
            continue
        run_trajectory(ref, name, spec, write=not args.check_only)


if __name__ == '__main__':
    main()
