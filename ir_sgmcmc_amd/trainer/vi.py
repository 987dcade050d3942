"""The VI stage of the reference's Trainer (trainer/trainer.py:79-289): fit q(v) = N(mu, diag(exp(log_var)) + u u^T) with an
antithetic sample pair per iteration, then sample from it.

Unlike the SG-MCMC transition this stage is composed in torch exactly as the reference composes it -- SobolevGrad,
transformation module, uniform jitter, registration module, GMM.map, the losses, `loss.backward()`, Adam -- with every
volume operator a HIP kernel behind its autograd Function (utils/functions.py, utils/transformation.py,
utils/registration.py, model/loss.py of this package).  The hyper-parameters (GMM, learnable regulariser) are optimised by
the torch-side Adam here; their values AND optimiser moments are exchanged with the device state of the fused engine before
and after the stage, so that the MCMC stage continues the same optimiser trajectories (the reference shares the optimiser
objects between the two stages)."""
import time

import torch

from ..logger import (save_displacement_mean_and_std_dev, save_field, save_fixed_im, save_fixed_mask, save_im, save_moving_im,
                      save_moving_mask, save_sample)
from ..utils import (SobolevGrad, add_noise_uniform_field, calc_DSC_GPU, calc_no_non_diffeomorphic_voxels, calc_posterior_statistics,
                     calc_VD_factor, max_field_update, rescale_residuals, sample_q_v)
from ..utils.functions import Sobolev_kernel_1D


class VIMixin:
    # ---------------------------------------------------------------- plumbing between the engine and torch optimisers
    def _sobolev_init(self):
        """Trainer.__Sobolev_gradients_init (trainer.py:568-583)"""
        if not self.Sobolev_grad:
            self.S, self.padding = None, None
            return
        S, _ = Sobolev_kernel_1D(self.Sobolev_s, self.Sobolev_lambda)
        S = torch.as_tensor(S, dtype=torch.float32, device=self.device).reshape(1, -1)
        S = torch.stack((S, S, S), 0)
        self.S = {'x': S.unsqueeze(2).unsqueeze(2), 'y': S.unsqueeze(2).unsqueeze(4), 'z': S.unsqueeze(3).unsqueeze(4)}
        self.padding = (self.Sobolev_s,) * 6

    def _smooth(self, v):
        return SobolevGrad.apply(v, self.S, self.padding) if self.Sobolev_grad else v

    def _init_optimizers(self):
        """torch-side optimisers of the hyper-parameters, continuing the moments the device has accumulated so far
        (the 25 warm-up steps of __GMM_init)."""
        st = self.sync_parameters()
        data_loss, reg_loss = self.losses['data']['loss'], self.losses['reg']['loss']
        for group in (self.losses['data'], self.losses['reg'], {'entropy': self.losses.get('entropy')}):
            for m in group.values():   # losses and their hyper-priors live on the trainer's device (BaseTrainer moves them in the reference)
                if isinstance(m, torch.nn.Module):
                    m.to(self.device)
        self.optimizer_GMM, self.optimizer_reg = None, None
        if type(data_loss).__name__ == 'GMM':
            data_loss.to(self.device)
            K = data_loss.no_components
            self.optimizer_GMM = self.config.init_optimizer_GMM(data_loss)
            for i, p in enumerate((data_loss.log_std, data_loss.logits)):
                self.optimizer_GMM.state[p] = {
                    'step': int(st.gmm_adam_step[i]), 'reinit': 0,
                    'exp_avg': torch.tensor(list(st.gmm_adam_m[i])[:K], dtype=p.dtype, device=p.device),
                    'exp_avg_sq': torch.tensor(list(st.gmm_adam_v[i])[:K], dtype=p.dtype, device=p.device)}
        reg_loss.to(self.device)
        if reg_loss.learnable:
            self.optimizer_reg = self.config.init_optimizer_reg(reg_loss)
            params = [p for g in self.optimizer_reg.param_groups for p in g['params']]
            for i, p in enumerate(params):
                self.optimizer_reg.state[p] = {
                    'step': int(st.reg_adam_step[i]), 'reinit': 0,
                    'exp_avg': torch.tensor(float(st.reg_adam_m[i]), dtype=p.dtype, device=p.device).reshape(p.shape),
                    'exp_avg_sq': torch.tensor(float(st.reg_adam_v[i]), dtype=p.dtype, device=p.device).reshape(p.shape)}

    def _push_hyperparameters_to_engine(self):
        """loss-module parameters + torch optimiser state -> device state (the inverse of sync_parameters)"""
        st = self.engine.state()
        data_loss, reg_loss = self.losses['data']['loss'], self.losses['reg']['loss']
        if self.optimizer_GMM is not None:
            for i, p in enumerate((data_loss.log_std, data_loss.logits)):
                s = self.optimizer_GMM.state[p]
                st.gmm_adam_step[i] = int(s['step'])
                for k in range(data_loss.no_components):
                    (st.gmm_log_std if i == 0 else st.gmm_logits)[k] = float(p[k])
                    st.gmm_adam_m[i][k], st.gmm_adam_v[i][k] = float(s['exp_avg'][k]), float(s['exp_avg_sq'][k])
        name = type(reg_loss).__name__
        if name == 'RegLoss_L2':
            st.reg_param[0] = float(reg_loss.log_w_reg)
        elif name == 'RegLoss_LogNormal':
            st.reg_param[0], st.reg_param[1] = float(reg_loss.loc), float(reg_loss.log_scale)
        if self.optimizer_reg is not None:
            params = [p for g in self.optimizer_reg.param_groups for p in g['params']]
            for i, p in enumerate(params):
                s = self.optimizer_reg.state[p]
                st.reg_adam_step[i] = int(s['step'])
                st.reg_adam_m[i], st.reg_adam_v[i] = float(s['exp_avg']), float(s['exp_avg_sq'])
        self.engine.set_state(st)

    # ---------------------------------------------------------------- reference-named pieces
    def _step_GMM(self, residuals, alpha=1.0):
        """trainer.py:68-77 (torch side; the MCMC transition carries the same step on the device)"""
        data_loss = self.losses['data']['loss']
        if self.optimizer_GMM is None:
            return
        data_term = data_loss(residuals.detach()).sum() * alpha
        data_term = data_term - self.losses['data']['scale_prior'](data_loss.log_scales).sum()
        data_term = data_term - self.losses['data']['proportion_prior'](data_loss.log_proportions).sum()
        self.optimizer_GMM.zero_grad()
        data_term.backward()
        self.optimizer_GMM.step()

    def _get_VD_factor(self, residuals, mask, data_loss):
        """trainer.py:507-514"""
        if not self.virutal_decimation:
            return 1.0
        return calc_VD_factor(rescale_residuals(residuals.detach(), mask, data_loss), mask)

    def _calc_sample_loss_VI(self, data_loss, reg_loss, entropy_loss, fixed, moving, var_params_q_v, v_sample_unsmoothed):
        """trainer.py:79-117"""
        v_sample = self._smooth(v_sample_unsmoothed)
        transformation, displacement = self.transformation_module(v_sample)
        with torch.no_grad():
            no_folds, log_det_J = calc_no_non_diffeomorphic_voxels(transformation, self.diff_op)
        if self.add_noise_uniform:
            transformation = add_noise_uniform_field(transformation, self.alpha)
        im_moving_warped = self.registration_module(moving['im'], transformation)
        output = {'displacement': displacement, 'transformation': transformation, 'im_moving_warped': im_moving_warped,
                  'log_det_J': log_det_J}
        residuals = data_loss.map(fixed['im'], im_moving_warped)
        alpha = self._get_VD_factor(residuals, fixed['mask'], data_loss)
        residuals_masked = residuals[fixed['mask']]
        self._step_GMM(residuals_masked, alpha)
        data_term = data_loss(residuals_masked).sum() * alpha
        reg_term, log_y = reg_loss(v_sample)
        reg_term = reg_term.sum()
        entropy_term = entropy_loss(sample=v_sample_unsmoothed, mu=var_params_q_v['mu'], log_var=var_params_q_v['log_var'],
                                    u=var_params_q_v['u']).sum()
        aux = {'alpha': alpha, 'reg_energy': log_y.exp(), 'no_non_diffeomorphic_voxels': no_folds, 'residuals': residuals_masked}
        loss_terms = {'data': data_term, 'reg': reg_term, 'entropy': entropy_term}
        if reg_loss.learnable:
            if type(reg_loss).__name__ == 'RegLoss_LogNormal':
                loss_terms['reg_loc_prior'] = self.losses['reg']['loc_prior'](log_y).sum()
            elif type(reg_loss).__name__ == 'RegLoss_L2':
                loss_terms['w_reg_prior'] = self.losses['reg']['w_reg_prior'](reg_loss.log_w_reg)
        return loss_terms, output, aux

    def _VI_iteration(self, fixed, moving, var_params_q_v, samples=None):
        """the body of the reference's _run_VI loop (trainer.py:131-170); `samples`: an explicit antithetic pair (tests)"""
        data_loss, reg_loss, entropy_loss = self.losses['data']['loss'], self.losses['reg']['loss'], self.losses['entropy']
        v1, v2 = samples if samples is not None else sample_q_v(var_params_q_v, no_samples=2)
        lt1, output, aux = self._calc_sample_loss_VI(data_loss, reg_loss, entropy_loss, fixed, moving, var_params_q_v, v1)
        lt2, _, _ = self._calc_sample_loss_VI(data_loss, reg_loss, entropy_loss, fixed, moving, var_params_q_v, v2)
        data_term = (lt1['data'] + lt2['data']) / 2.0
        if 'scale_prior' in self.losses['data']:
            data_term = data_term - self.losses['data']['scale_prior'](data_loss.log_scales).sum()
            data_term = data_term - self.losses['data']['proportion_prior'](data_loss.log_proportions).sum()
        reg_term = (lt1['reg'] + lt2['reg']) / 2.0
        if reg_loss.learnable:
            if type(reg_loss).__name__ == 'RegLoss_LogNormal':
                reg_term = reg_term - (lt1['reg_loc_prior'] + lt2['reg_loc_prior']) / 2.0
                reg_term = reg_term - self.losses['reg']['scale_prior'](reg_loss.log_scale).sum()
            elif type(reg_loss).__name__ == 'RegLoss_L2':
                reg_term = reg_term - (lt1['w_reg_prior'] + lt2['w_reg_prior']) / 2.0
        entropy_term = (lt1['entropy'] + lt2['entropy']) / 2.0
        entropy_term = entropy_term + entropy_loss(log_var=var_params_q_v['log_var'], u=var_params_q_v['u']).sum()
        loss = data_term + reg_term - entropy_term
        if self.optimizer_reg is not None:
            self.optimizer_reg.zero_grad()
        self.optimizer_q_v.zero_grad()
        loss.backward()
        if self.optimizer_reg is not None:
            self.optimizer_reg.step()
        self.optimizer_q_v.step()
        return {'data': data_term.detach(), 'reg': reg_term.detach(), 'entropy': entropy_term.detach(), 'loss': loss.detach()}, output, aux

    def _run_VI(self, fixed, moving, var_params_q_v):
        """trainer.py:119-223 (scalars go to the MetricTracker; no TensorBoard figures)"""
        for p in var_params_q_v.values():
            p.requires_grad_(True)
        self.optimizer_q_v = self.config.init_optimizer_q_v(var_params_q_v)
        data_loss, reg_loss = self.losses['data']['loss'], self.losses['reg']['loss']
        spacing = self._spacing()
        if self.config['trainer'].get('save_outputs', True):
            with torch.no_grad():
                save_fixed_im(self.config.save_dirs, spacing, fixed['im'])
                save_fixed_mask(self.config.save_dirs, spacing, fixed['mask'])
                save_moving_im(self.config.save_dirs, spacing, moving['im'])
                if 'mask' in moving:
                    save_moving_mask(self.config.save_dirs, spacing, moving['mask'])
        for iter_no in range(self.start_iter_VI, self.no_iters_VI + 1):
            prev = {k: v.detach().clone() for k, v in var_params_q_v.items()}
            terms, output, aux = self._VI_iteration(fixed, moving, var_params_q_v)
            with torch.no_grad():
                self.writer.set_step(iter_no)
                if type(data_loss).__name__ == 'GMM':
                    for idx in range(data_loss.no_components):
                        self.metrics.update(f'VI/train/GMM/scale_{idx}', data_loss.scales[idx].item())
                        self.metrics.update(f'VI/train/GMM/proportion_{idx}', data_loss.proportions[idx].item())
                if reg_loss.learnable:
                    if type(reg_loss).__name__ == 'RegLoss_LogNormal':
                        self.metrics.update('VI/train/reg/loc', reg_loss.loc.item())
                        self.metrics.update('VI/train/reg/scale', reg_loss.scale.item())
                    elif type(reg_loss).__name__ == 'RegLoss_L2':
                        self.metrics.update('VI/train/reg/w_reg', reg_loss.log_w_reg.exp().item())
                if self.virutal_decimation:
                    self.metrics.update('VI/train/VD/alpha', float(aux['alpha']))
                for k, name in (('data', 'data_term'), ('reg', 'reg_term'), ('entropy', 'entropy_term'), ('loss', 'total_loss')):
                    self.metrics.update(f'VI/train/{name}', terms[k].item())
                self.metrics.update('VI/train/reg/energy', aux['reg_energy'].sum().item())
                self.metrics.update('VI/train/no_non_diffeomorphic_voxels', int(aux['no_non_diffeomorphic_voxels'].sum()))
                for key in var_params_q_v:
                    self.metrics.update(f'VI/train/max_updates/{key}', max_field_update(prev[key], var_params_q_v[key])[0].item())
                if (iter_no % self.log_period_VI == 0 or iter_no == self.no_iters_VI) and 'seg' in moving and self.structures_dict:
                    seg_warped = self.registration_module(moving['seg'], output['transformation'].detach())
                    DSC = calc_DSC_GPU(1, fixed['seg'], seg_warped, self.structures_dict)
                    for j, structure in enumerate(self.structures_dict):
                        self.metrics.update(f'VI/train/DSC/{structure}', float(DSC[0][j]))

    @torch.no_grad()
    def _test_VI(self, fixed, moving, var_params_q_v):
        """trainer.py:225-289: samples of the fitted posterior, their mean / std, the posterior-mean registration, speed"""
        save = self.config['trainer'].get('save_outputs', True)
        spacing = self._spacing()
        dims = tuple(var_params_q_v['mu'].shape[-3:]) if self.transformation_module.__class__.__name__ == 'SVF_3D' else None
        samples = []
        for n in range(1, self.no_samples_VI_test + 1):
            self.writer.set_step(n)
            v = sample_q_v(var_params_q_v, no_samples=1)
            transformation, displacement = self.transformation_module(self._smooth(v))
            samples.append(displacement[0].clone())
            no_folds, log_det_J = calc_no_non_diffeomorphic_voxels(transformation, self.diff_op)
            self.metrics.update('VI/test/no_non_diffeomorphic_voxels', int(no_folds.sum()))
            warped = self.registration_module(moving['im'], transformation)
            if 'seg' in moving and self.structures_dict:
                DSC = calc_DSC_GPU(1, fixed['seg'], self.registration_module(moving['seg'], transformation), self.structures_dict)
                for j, structure in enumerate(self.structures_dict):
                    self.metrics.update(f'VI/test/DSC/{structure}', float(DSC[0][j]))
            if save:
                save_sample(self.config.save_dirs, spacing, n, warped, displacement, log_det_J, 'VI')
        transformation, displacement = self.transformation_module(self._smooth(var_params_q_v['mu']))
        warped = self.registration_module(moving['im'], transformation)
        if save:   # save_variational_posterior_mean (logger/logger.py:198-208)
            save_im(self.config.save_dirs, spacing, warped[0, 0], 'im_moving_warped_mu')
            save_field(self.config.save_dirs, spacing, displacement[0] * spacing[0], 'displacement_mu')
        if samples:
            mean, std = calc_posterior_statistics(torch.stack(samples), device=self.device)
            self.VI_displacement_mean, self.VI_displacement_std = mean, std
            if save and len(samples) > 1:
                save_displacement_mean_and_std_dev(self.logger, self.config.save_dirs, spacing, mean, std,
                                                   fixed['mask'][0].to(mean.dtype), 'VI')
        n_speed = 100
        torch.cuda.synchronize()
        start = time.perf_counter()
        for _ in range(n_speed):
            transformation, _ = self.transformation_module(self._smooth(sample_q_v(var_params_q_v, no_samples=1)))
            self.registration_module(moving['im'], transformation)
            if 'seg' in moving:
                self.registration_module(moving['seg'], transformation)
        torch.cuda.synchronize()
        self.VI_sampling_speed = n_speed / (time.perf_counter() - start)
        self.logger.info(f'\nVI sampling speed: {self.VI_sampling_speed:.2f} samples/sec')

    def _spacing(self):
        sp = getattr(self.data_loader, 'im_spacing', None)
        return sp if sp is not None else torch.ones(3)
