"""Trainer: the MCMC half of the reference's trainer/trainer.py, driving the fused HIP transition.

Kept surface (same names, arguments and return values as the reference):
  `Trainer(config, data_loader, losses, transformation_module, registration_module, metrics)`, `.run()`,
  `_SGLD_transition(fixed, moving, data_loss, reg_loss) -> (loss_terms, output, aux)`  (trainer.py:291-356),
  `step(...)` (alias asked for by BASELINE.json), `_run_MCMC` (:358-476), `_step_GMM` (:68-77).
The VI stage (`_run_VI`, `_test_VI`, trainer/vi.py) is composed in torch from the HIP-backed modules, as the reference
composes it; it hands its hyper-parameters and optimiser moments to the fused engine before the MCMC stage.

Inside `_SGLD_transition` nothing runs in torch: one call into the C ABI launches the whole transition on the current
HIP stream (ir_sgmcmc_amd/csrc/api.hip: transition_impl).  The hyper-parameters of the loss objects are mirrored into
the device state once (`_engine_init`) and read back lazily (`sync_parameters`) when something logs them.
"""
import time

import numpy as np
import torch

from ..base import BaseTrainer
from ..engine import EngineConfig, TransitionEngine
from ..logger import save_displacement_mean_and_std_dev, save_sample
from ..utils import calc_norm, calc_no_non_diffeomorphic_voxels, calc_DSC_GPU, sample_q_v
from .vi import VIMixin


class LazyScalar:
    """a per-chain scalar that lives on the device until somebody calls .item() / float() on it (no sync otherwise)"""

    def __init__(self, fetch, key, idx):
        self._fetch, self._key, self._idx = fetch, key, idx

    def item(self):
        return float(self._fetch()[self._key][self._idx])

    __float__ = item

    def __add__(self, other):
        return self.item() + float(other)

    __radd__ = __add__

    def __repr__(self):
        return f'{self.item():.6g}'


class Trainer(VIMixin, BaseTrainer):
    def __init__(self, config, data_loader, losses, transformation_module, registration_module, metrics, device='cuda:0'):
        super().__init__(config, data_loader, losses, transformation_module, registration_module, metrics, device)
        self.Sobolev_grad = config['Sobolev_grad']['enabled']
        self.Sobolev_s = int(config['Sobolev_grad']['s']) if self.Sobolev_grad else 0
        self.Sobolev_lambda = float(config['Sobolev_grad']['lambda']) if self.Sobolev_grad else 0.0
        cfg_trainer = config['trainer']
        self.add_noise_uniform = cfg_trainer['uniform_noise']['enabled']
        self.alpha = cfg_trainer['uniform_noise']['magnitude'] if self.add_noise_uniform else 0.0
        self.virutal_decimation = config['virtual_decimation']  # (sic) reference attribute name, trainer.py:42
        self.engine = None
        self.v_curr_state, self.SGLD_params = None, None
        self._scalars_cache, self._outputs = None, None

    # ---------------------------------------------------------------- engine plumbing
    def _engine_config(self):
        cfg = self.config
        data_loss, reg_loss = self.losses['data']['loss'], self.losses['reg']['loss']
        dims = tuple(cfg['data_loader']['args']['dims'])
        t_args = dict(cfg['transformation_module'].get('args', {}))
        kind = type(data_loss).__name__
        ec = dict(dims=dims, no_chains=self.no_chains, cps=tuple(t_args['cps']) if cfg['transformation_module']['type'] == 'SVFFD_3D' else None,
                  no_steps=getattr(self.transformation_module, 'no_steps', 12), sobolev_s=self.Sobolev_s,
                  sobolev_lambda=self.Sobolev_lambda, lr=float(cfg['optimizer_SG_MCMC']['args']['lr']),
                  uniform_noise=float(self.alpha), virtual_decimation=bool(self.virutal_decimation),
                  data_loss='GMM' if kind == 'GMM' else 'SSD', seed=int(cfg['trainer'].get('seed', 0)))
        if cfg['optimizer_SG_MCMC']['type'] != 'SGD':
            raise NotImplementedError('the SG-MCMC field update is plain SGD (reference configs), got ' + cfg['optimizer_SG_MCMC']['type'])
        if kind == 'GMM':
            o = cfg['optimizer_GMM']['args']
            sp = self.losses['data']['scale_prior'].normal
            ec.update(gmm_components=data_loss.no_components, lcc_s=data_loss.s, gmm_lr_log_std=o['lr_log_std'],
                      gmm_lr_logits=o['lr_logits'], gmm_lr_decay=o['lr_decay'],
                      scale_prior=(float(sp.loc), float(sp.log_scale.exp())),
                      dirichlet_alpha=[float(x) for x in self.losses['data']['proportion_prior'].concentration])
        else:
            ec.update(ssd_sigma=data_loss.sigma)
        rname = type(reg_loss).__name__
        if rname not in ('RegLoss_L2', 'RegLoss_LogNormal', 'RegLoss_Student', 'RegLoss_LogNormal_L2'):
            raise NotImplementedError(rname + ' is not wired into the fused transition')
        ec.update(reg_loss=rname, reg_learnable=bool(reg_loss.learnable))
        if rname == 'RegLoss_L2':
            ec.update(w_reg=float(reg_loss.log_w_reg.exp()))
            if reg_loss.learnable:
                o = cfg['optimizer_reg']['args']
                ec.update(reg_lr=(o['lr_log_w_reg'], 0.0), reg_lr_decay=o['lr_decay'])
        elif rname == 'RegLoss_Student':
            ec.update(student=(float(reg_loss.a0), float(reg_loss.b0_twice)))
        elif rname == 'RegLoss_LogNormal_L2':
            ec.update(w_reg=float(reg_loss.gamma_distr.rate) * 2.0)
        else:
            ec.update(w_reg=float(reg_loss.w_reg))
            if reg_loss.learnable:
                o = cfg['optimizer_reg']['args']
                lp, sp = self.losses['reg']['loc_prior'], self.losses['reg']['scale_prior'].normal
                ec.update(reg_lr=(o['lr_loc'], o['lr_log_scale']), reg_lr_decay=o['lr_decay'], loc_prior_nu=float(lp.nu),
                          loc_prior_w_reg=float(lp.w_reg), reg_scale_prior=(float(sp.loc), float(sp.log_scale.exp())))
        return EngineConfig(**ec)

    def _engine_init(self, fixed, moving):
        self.engine = TransitionEngine(self._engine_config(), self.device)
        self._fixed, self._moving = self.engine.prepare(fixed, moving)
        self._mask_idx = None
        # hyper-parameters of the loss objects -> device state
        st = self.engine.state()
        data_loss, reg_loss = self.losses['data']['loss'], self.losses['reg']['loss']
        if type(data_loss).__name__ == 'GMM':
            for k in range(data_loss.no_components):
                st.gmm_log_std[k], st.gmm_logits[k] = float(data_loss.log_std[k].detach()), float(data_loss.logits[k].detach())
        if type(reg_loss).__name__ == 'RegLoss_L2':
            st.reg_param[0] = float(reg_loss.log_w_reg.detach())
        elif type(reg_loss).__name__ == 'RegLoss_LogNormal':
            st.reg_param[0], st.reg_param[1] = float(reg_loss.loc.detach()), float(reg_loss.log_scale.detach())
        self.engine.set_state(st)

    def sync_parameters(self):
        """device state -> the nn.Parameters of the loss objects (what the reference's logging reads, trainer.py:391-402)"""
        st = self.engine.state()
        data_loss, reg_loss = self.losses['data']['loss'], self.losses['reg']['loss']
        with torch.no_grad():
            if type(data_loss).__name__ == 'GMM':
                K = data_loss.no_components
                data_loss.log_std.copy_(torch.tensor(list(st.gmm_log_std)[:K]))
                data_loss.logits.copy_(torch.tensor(list(st.gmm_logits)[:K]))
            if type(reg_loss).__name__ == 'RegLoss_L2':
                reg_loss.log_w_reg.fill_(st.reg_param[0])
            elif type(reg_loss).__name__ == 'RegLoss_LogNormal':
                reg_loss.loc.fill_(st.reg_param[0])
                reg_loss.log_scale.fill_(st.reg_param[1])
        return st

    def _scalars(self):
        if self._scalars_cache is None:
            self._scalars_cache = self.engine.scalars()
        return self._scalars_cache

    # ---------------------------------------------------------------- checkpoint / resume (absent in the reference)
    def state_dict(self):
        """Everything a chain needs to continue bit-for-bit: the velocity field, the SGLD pre-conditioner, the device-side
        hyper-parameter state (GMM / regulariser parameters, their Adam moments and step counts, the Philox iteration
        counter) and the running posterior moments."""
        import ctypes
        st = self.engine.state()
        sigma = getattr(self, '_sigma', None)
        return {'v_curr_state': self.v_curr_state.detach().cpu(), 'sigma': None if sigma is None else sigma.detach().cpu(),
                'tau': self.SGLD_params['tau'], 'engine_state': bytes(ctypes.string_at(ctypes.byref(st), ctypes.sizeof(st))),
                'sample_no': getattr(self, '_sample_no', 0),
                'moments': {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in getattr(self, '_moments', {}).items()},
                'config_name': self.config['name']}

    def load_state_dict(self, sd):
        import ctypes
        from .. import _lib as L
        if self.engine is None:
            raise RuntimeError('load_state_dict: call _engine_init(fixed, moving) first')
        st = L.IrsState()
        raw = sd['engine_state']
        if len(raw) != ctypes.sizeof(st):
            raise ValueError('checkpoint was written by an incompatible build (state struct size differs)')
        ctypes.memmove(ctypes.byref(st), raw, len(raw))
        self.engine.set_state(st)
        self.v_curr_state = sd['v_curr_state'].to(self.device).contiguous()
        self._sigma = None if sd['sigma'] is None else sd['sigma'].to(self.device).contiguous()
        self.SGLD_params = {'tau': sd['tau'], 'sigma': self._sigma if self._sigma is not None else torch.ones_like(self.v_curr_state)}
        self._sample_no = int(sd['sample_no'])
        self._moments = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in sd.get('moments', {}).items()}
        self.sync_parameters()

    def save_checkpoint(self, file_path):
        torch.save(self.state_dict(), file_path)

    def load_checkpoint(self, file_path):
        # plain tensors, numbers and bytes only: a checkpoint is data, never code
        self.load_state_dict(torch.load(file_path, map_location='cpu', weights_only=True))

    # ---------------------------------------------------------------- reference-named pieces
    def _SGLD_init(self, var_params_q_v):
        """Trainer.__SGLD_init (trainer.py:585-611)"""
        shape = [self.no_chains, 3, *var_params_q_v['mu'].shape[-3:]]
        if self.MCMC_init == 'VI':
            vp = {k: v.to(self.device).reshape(1, 3, *shape[2:]) for k, v in var_params_q_v.items()}
            v = torch.empty(shape, device=self.device)
            for idx in range(self.no_chains):
                v[idx] = sample_q_v(vp, no_samples=1)[0]
            sigma = torch.exp(0.5 * vp['log_var']).expand(shape).contiguous()
        elif self.MCMC_init == 'identity':
            v, sigma = torch.zeros(shape, device=self.device), None
        elif self.MCMC_init == 'noise':
            v, sigma = torch.randn(shape, device=self.device), None
        else:
            raise ValueError(self.MCMC_init)
        tau = self.config['optimizer_SG_MCMC']['args']['lr']
        self.SGLD_params = {'sigma': sigma if sigma is not None else torch.ones(shape, device=self.device), 'tau': tau}
        self._sigma = sigma  # None = identity preconditioner (the kernels skip the load)
        self.v_curr_state = v.contiguous()
        C, dv, d = self.no_chains, tuple(shape[2:]), tuple(self.config['data_loader']['args']['dims'])
        new = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        self._outputs = {'curr_state': new(C, 3, *dv), 'im_moving_warped': new(C, 1, *d), 'residuals': new(C, 1, *d),
                         'displacement': new(C, 3, *d), 'transformation': new(C, 3, *d)}

    def _GMM_init(self, fixed, moving, var_params_q_v=None):
        """Trainer.__GMM_init (trainer.py:529-547): one velocity sample, std of the masked residual, 25 warm-up steps"""
        # the reference draws sample_q_v(var_params_q_v) here whatever MCMC_init is (mu = 0, sigma_v_init, u_v_init at the
        # start of a run): the residual std -- and with it the mixture's initial log_std and warm-up trajectory -- comes
        # from THAT warp, not from the identity
        v_sample = None
        if var_params_q_v is not None:
            vp = {k: v.to(self.device).reshape(1, 3, *v.shape[-3:]) for k, v in var_params_q_v.items()}
            v_sample = sample_q_v(vp).contiguous()
        self.engine.gmm_init(self._fixed, self._moving, v_sample)
        self.sync_parameters()

    def _SGLD_transition(self, fixed, moving, data_loss=None, reg_loss=None, eps=None, unif=None, with_outputs=True,
                         like_reference=True):
        """One SG-MCMC transition (trainer.py:291-356).  Returns (loss_terms, output, aux) with the reference's keys.

        `eps` / `unif` inject the two noise draws (parity runs); by default they come from in-kernel Philox.
        `like_reference` (default): `output[...]` are CLONES (trainer.py:302-305) and `aux['residuals']` is the masked view
        `residuals[fixed['mask']].view(no_chains, -1)` (trainer.py:308) -- a caller that keeps samples or feeds the residuals
        to its own code sees exactly the reference's objects.  With False the tensors are the engine-owned buffers the NEXT
        transition overwrites and the residuals stay dense (C,1,D,H,W): what this package's own `_run_MCMC` loop uses, since
        it consumes them at once (saves four volume copies and a gather per transition).
        """
        if self.engine is None:
            raise RuntimeError('call _engine_init / _run_MCMC first')
        out = self._outputs if with_outputs else {k: self._outputs[k] for k in ('curr_state', 'im_moving_warped', 'residuals')}
        self.engine.transition(self._fixed, self._moving, self.v_curr_state, self._sigma, eps, unif, out)
        self._scalars_cache = None
        C = self.no_chains
        lazy = lambda key: [LazyScalar(self._scalars, key, i) for i in range(C)]
        loss_terms = {'data': lazy('data_term'), 'reg': lazy('reg_term')}
        output = {'im_moving_warped': self._outputs['im_moving_warped'], 'displacement': self._outputs['displacement'],
                  'transformation': self._outputs['transformation'], 'curr_state': self._outputs['curr_state']}
        residuals = self._outputs['residuals']
        if like_reference:
            output = {k: v.clone() for k, v in output.items()}
            residuals = residuals.reshape(-1).index_select(0, self._masked_index()).view(C, -1)
        aux = {'residuals': residuals, 'alpha': lazy('alpha'), 'reg_energy': lazy('reg_energy')}
        return loss_terms, output, aux

    def _masked_index(self):
        """flat indices of the masked voxels of all chains, computed once (a boolean index would synchronise every transition)"""
        if getattr(self, '_mask_idx', None) is None:
            m = self._fixed['mask']
            m = m.expand(self.no_chains, *m.shape[1:]) if m.shape[0] == 1 else m
            self._mask_idx = m.reshape(-1).nonzero(as_tuple=False).squeeze(1)
        return self._mask_idx

    step = _SGLD_transition  # BASELINE.json calls the iteration `step()`

    # ---------------------------------------------------------------- MCMC driver (trainer.py:358-476)
    def _run_MCMC(self, fixed, moving, var_params_q_v):
        data_loss, reg_loss = self.losses['data']['loss'], self.losses['reg']['loss']
        self._SGLD_init(var_params_q_v)
        log = self.logger.info
        log(f'\nNO. CHAINS: {self.no_chains}, BURNING IN...')
        n_total = self.no_iters_burn_in + self.no_samples_MCMC
        # the reference logs hyper-parameters and loss terms on every iteration of a short run (trainer.py:389), each a host
        # read-back; `trainer.metrics_period` > 1 thins them out (the device then runs ahead of the host between reads)
        every = int(self.config['trainer'].get('metrics_period', 1 if self.no_samples_MCMC < 1e4 else 100))
        # running posterior mean / M2 of the displacement on the device instead of a
        # host array of every logged sample (trainer.py:365-366)
        mean = torch.zeros_like(self._outputs['displacement'][0])
        m2 = torch.zeros_like(mean)
        n_rec = 0
        cfg_trainer = self.config['trainer']
        checkpoint_period, save_samples = int(cfg_trainer.get('checkpoint_period', 0)), bool(cfg_trainer.get('save_samples', False))
        spacing = self.data_loader.im_spacing if getattr(self.data_loader, 'im_spacing', None) is not None else torch.ones(3)
        first = 1
        if cfg_trainer.get('resume'):
            self.load_checkpoint(cfg_trainer['resume'])
            first = self._sample_no + 1
            if self._moments:
                mean, m2, n_rec = self._moments['mean'], self._moments['m2'], int(self._moments['n'])
            log(f'resumed from {cfg_trainer["resume"]} at sample {self._sample_no}')
        for sample_no in range(first, n_total + 1):
            if sample_no < self.no_iters_burn_in and sample_no % self.log_period_MCMC == 0:
                log(f'burn-in sample no. {sample_no}/{self.no_iters_burn_in}')
            loss_terms, output, aux = self._SGLD_transition(fixed, moving, data_loss, reg_loss, like_reference=False)
            if sample_no == self.no_iters_burn_in:
                log('ENDED BURNING IN')
            self.writer.set_step(sample_no)
            if (sample_no - 1) % every == 0:
                st = self.sync_parameters()
                if type(data_loss).__name__ == 'GMM':
                    for idx in range(data_loss.no_components):
                        self.metrics.update(f'MCMC/GMM/scale_{idx}', data_loss.scales[idx].item())
                        self.metrics.update(f'MCMC/GMM/proportion_{idx}', data_loss.proportions[idx].item())
                if getattr(reg_loss, 'learnable', False):  # trainer.py:397-402
                    if type(reg_loss).__name__ == 'RegLoss_LogNormal':
                        self.metrics.update('MCMC/reg/loc', reg_loss.loc.item())
                        self.metrics.update('MCMC/reg/scale', reg_loss.scale.item())
                    elif type(reg_loss).__name__ == 'RegLoss_L2':
                        self.metrics.update('MCMC/reg/w_reg', reg_loss.log_w_reg.exp().item())
                total = sum(t.item() for t in loss_terms['data']) + sum(t.item() for t in loss_terms['reg'])
                self.metrics.update('MCMC/avg_loss', total / self.no_chains)
                for idx in range(self.no_chains):
                    self.metrics.update(f'MCMC/chain_{idx}/data_term', loss_terms['data'][idx].item())
                    self.metrics.update(f'MCMC/chain_{idx}/reg_term', loss_terms['reg'][idx].item())
                    self.metrics.update(f'MCMC/chain_{idx}/VD/alpha', aux['alpha'][idx].item())
                    self.metrics.update(f'MCMC/chain_{idx}/reg/energy', aux['reg_energy'][idx].item())
            if sample_no > self.no_iters_burn_in and (sample_no % self.log_period_MCMC == 0 or sample_no == self.no_samples_MCMC):
                # The outputs of a call are only those of sample `sample_no` once nothing is pending: a transition dropped by a failed
                # kernel-variant prediction is re-run by a LATER call, and until then the output buffers hold an earlier sample
                # (one sync per log_period; normally a no-op)
                self.engine.flush()
                transformation, displacement = output['transformation'], output['displacement']
                no_folds, log_det_J = calc_no_non_diffeomorphic_voxels(transformation, self.diff_op)
                if 'seg' in moving and 'seg' in fixed and self.structures_dict:
                    seg_warped = self.registration_module(moving['seg'], transformation)
                    DSC = calc_DSC_GPU(self.no_chains, fixed['seg'].expand_as(seg_warped), seg_warped, self.structures_dict)
                    for idx in range(self.no_chains):
                        for j, name in enumerate(self.structures_dict):
                            self.metrics.update(f'MCMC/chain_{idx}/DSC/{name}', float(DSC[idx][j]))
                no_voxels = int(np.prod(displacement.shape[2:]))
                if save_samples:
                    for idx in range(self.no_chains):
                        save_sample(self.config.save_dirs, spacing, sample_no, output['im_moving_warped'][idx:idx + 1],
                                    displacement[idx:idx + 1], log_det_J[idx:idx + 1], 'MCMC', chain_no=idx)
                for idx in range(self.no_chains):
                    n_rec += 1
                    delta = displacement[idx] - mean
                    mean += delta / n_rec
                    m2 += delta * (displacement[idx] - mean)
                    self.metrics.update(f'MCMC/chain_{idx}/no_non_diffeomorphic_voxels', int(no_folds[idx]))
                    if no_folds[idx] > 0.001 * no_voxels:  # trainer.py:441-445
                        log(f'chain {idx}, sample {sample_no}: detected {no_folds} voxels where the sampled '
                            f'transformation is not diffeomorphic; exiting..')
                        raise SystemExit(1)
            if checkpoint_period and sample_no % checkpoint_period == 0:
                self._sample_no, self._moments = sample_no, {'mean': mean, 'm2': m2, 'n': n_rec}
                folder = self.config.save_dirs['checkpoints']
                folder.mkdir(parents=True, exist_ok=True)
                self.save_checkpoint(folder / f'checkpoint_{sample_no:07}.pt')
        self.engine.flush()  # a transition dropped by a failed kernel-variant prediction is re-run before anything is saved
        self.displacement_mean = mean
        self.displacement_std = torch.sqrt(m2 / max(n_rec - 1, 1))
        if n_rec > 0 and cfg_trainer.get('save_outputs', True):
            save_displacement_mean_and_std_dev(self.logger, self.config.save_dirs, spacing, self.displacement_mean,
                                               self.displacement_std, moving.get('mask', fixed['mask'])[0].to(mean.dtype), 'MCMC')  # trainer.py:461-462: the MOVING mask

        # speed test (trainer.py:467-476): 100 x [transition + nearest-neighbour warp of the segmentation]
        n_speed = 100
        torch.cuda.synchronize()
        start = time.perf_counter()
        for _ in range(n_speed):
            _, output, _ = self._SGLD_transition(fixed, moving, data_loss, reg_loss, like_reference=False)
            if 'seg' in moving:
                self.registration_module(moving['seg'], output['transformation'])
        self.engine.flush()
        torch.cuda.synchronize()
        self.MCMC_sampling_speed = self.no_chains * n_speed / (time.perf_counter() - start)
        log(f'\nMCMC sampling speed: {self.MCMC_sampling_speed:.2f} samples/sec')

    def _run_model(self):
        for fixed, moving, var_params_q_v in self.data_loader:
            fixed = {k: v.to(self.device) for k, v in fixed.items()}
            moving = {k: v.to(self.device) for k, v in moving.items()}
            self._engine_init(fixed, moving)
            self._GMM_init(fixed, moving, var_params_q_v)
            var_params_q_v = {k: v.to(self.device) for k, v in var_params_q_v.items()}
            self._sobolev_init()
            self._init_optimizers()
            if self.VI:
                start = time.perf_counter()
                self._run_VI(fixed, moving, var_params_q_v)
                torch.cuda.synchronize()
                self.logger.info(f'VI took {time.perf_counter() - start:.2f} seconds')
                self._test_VI(fixed, moving, var_params_q_v)
                self._push_hyperparameters_to_engine()
                var_params_q_v = {k: v.detach() for k, v in var_params_q_v.items()}
            if self.MCMC:
                self._run_MCMC(fixed, moving, var_params_q_v)
