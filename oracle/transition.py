"""CPU oracle: one full SG-MCMC transition  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Restates `Trainer._SGLD_transition` (reference trainer/trainer.py:291-356) and the pieces of
`Trainer` it touches (`_step_GMM` :68-77, `__get_VD_factor` :507-514, `__SGLD_init` :585-611,
`__GMM_init` :529-547, `__Sobolev_gradients_init` :568-583) on top of `oracle.ops`, with the two
random draws (`randn_like(sigma)`, then `rand(transformation.shape)`; utils/util.py:57,53)
passed in explicitly so that a device implementation can be fed the same noise.

Composition and order of operations follow the reference exactly: torch autograd differentiates
the same op graph, the GMM Adam step happens between the VD factor and the data term, per chain,
serially (trainer.py:316-327), and the hyper-priors enter the loss as in :329-339.

The builder-defined SSD data term (BASELINE.json configs 1, 2 and 4) is
`0.5 * sum_mask((F - M o phi) / sigma)^2`; it has no counterpart in the reference, so for it
"parity" means parity with this oracle only (every other stage is the reference-pinned one).
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch

from . import ops


@dataclass
class OracleConfig:
    dims: Tuple[int, int, int]
    no_chains: int = 1
    # transformation model
    transformation: str = 'SVF_3D'  # or 'SVFFD_3D'
    cps: Optional[Tuple[int, int, int]] = None
    no_steps: int = 12
    # Sobolev gradients (None = disabled)
    sobolev_s: Optional[int] = 3
    sobolev_lambda: float = 0.5
    # SGLD
    lr: float = 0.4  # optimizer_SG_MCMC lr == tau (trainer.py:607)
    uniform_noise: Optional[float] = 0.1
    virtual_decimation: bool = True
    # data loss
    data_loss: str = 'GMM'  # or 'SSD' (builder-defined)
    gmm_components: int = 4
    lcc_s: int = 1
    ssd_sigma: float = 0.1
    gmm_lr_log_std: float = 0.2
    gmm_lr_logits: float = 0.2
    gmm_lr_decay: float = 0.001
    scale_prior: Tuple[float, float] = (0.0, 2.3)  # LogScaleNormalPrior(loc, scale)
    dirichlet_alpha: float = 0.5
    # regulariser
    reg_loss: str = 'RegLoss_L2'  # or 'RegLoss_LogNormal', 'RegLoss_Student', 'RegLoss_LogNormal_L2'
    student: Tuple[float, float, float, float] = (2e-6, 1e-6, 1e-6, 1e-6)  # RegLoss_Student(nu0, lambda0, a0, b0)
    w_reg: float = 1.4
    reg_learnable: bool = False
    reg_lr: Tuple[float, float] = (0.01, 0.01)  # (lr_loc, lr_log_scale) or (lr_log_w_reg, -)
    reg_lr_decay: float = 0.001
    reg_loc_prior_nu: float = 1.0  # LogEnergyExpGammaPrior(w_reg, dof, nu)
    reg_scale_prior: Tuple[float, float] = (2.8, 5.0)  # LogScaleNormalPrior on log_scale

    @property
    def dims_v(self):
        if self.transformation == 'SVFFD_3D':
            return ops.control_grid_size(self.dims, self.cps)
        return tuple(self.dims)

    @property
    def dof(self):
        return float(self.dims[0] * self.dims[1] * self.dims[2]) * 3.0  # parse_config.py:120,128


class OracleChain:
    """Mutable sampler state + `transition`; mirrors what `Trainer` keeps between iterations."""

    def __init__(self, cfg: OracleConfig, v0=None, sigma=None):
        self.cfg = cfg
        C = cfg.no_chains
        shape = (C, 3, *cfg.dims_v)
        # fp32 like the reference; under torch.set_default_dtype(torch.float64) the whole chain runs in fp64 (the error-band
        # measurement of tests/golden/make_golden_fp64.py)
        wd = torch.get_default_dtype()
        self.v = (torch.zeros(shape) if v0 is None else v0.clone().to(wd)).requires_grad_(True)
        self.sigma = torch.ones(shape) if sigma is None else sigma.clone().to(wd).expand(shape).contiguous()
        self.sobolev_kernel = None if cfg.sobolev_s is None else \
            torch.from_numpy(ops.sobolev_kernel_1d(cfg.sobolev_s, cfg.sobolev_lambda)).to(wd)

        K = cfg.gmm_components
        self.log_std = torch.zeros(K, requires_grad=True)
        self.logits = torch.zeros(K, requires_grad=True)
        self.adam_gmm = ops.AdamRateDecay([{'params': [self.log_std], 'lr': cfg.gmm_lr_log_std},
                                           {'params': [self.logits], 'lr': cfg.gmm_lr_logits}],
                                          lr_decay=cfg.gmm_lr_decay)
        self.concentration = torch.full((K,), cfg.dirichlet_alpha)

        self.adam_reg = None
        if cfg.reg_loss == 'RegLoss_L2':
            self.log_w_reg = torch.tensor(math.log(cfg.w_reg), requires_grad=cfg.reg_learnable)
            if cfg.reg_learnable:
                self.adam_reg = ops.AdamRateDecay([{'params': [self.log_w_reg], 'lr': cfg.reg_lr[0]}],
                                                  lr_decay=cfg.reg_lr_decay)
        elif cfg.reg_loss == 'RegLoss_LogNormal':
            loc, log_scale = ops.reg_lognormal_init(cfg.w_reg, cfg.dof)
            self.loc = loc.clone().requires_grad_(cfg.reg_learnable)
            self.log_scale = log_scale.clone().requires_grad_(cfg.reg_learnable)
            if cfg.reg_learnable:
                self.adam_reg = ops.AdamRateDecay([{'params': [self.loc], 'lr': cfg.reg_lr[0]},
                                                   {'params': [self.log_scale], 'lr': cfg.reg_lr[1]}],
                                                  lr_decay=cfg.reg_lr_decay)
        elif cfg.reg_loss in ('RegLoss_Student', 'RegLoss_LogNormal_L2'):
            if cfg.reg_learnable:
                raise ValueError(cfg.reg_loss + ' has no learnable parameters (model/loss.py:206,316)')
        else:
            raise ValueError(cfg.reg_loss)

    # ---------------------------------------------------------------- pieces
    def smooth(self, x):
        if self.sobolev_kernel is None:
            return x
        return ops._SobolevStraightThrough.apply(x, self.sobolev_kernel)

    def transform(self, v_s, keep_steps=False):
        cfg = self.cfg
        if cfg.transformation == 'SVFFD_3D':
            v_s = ops.ffd_upsample(v_s, cfg.dims, cfg.cps)
        return ops.svf_exp(v_s, cfg.no_steps, keep_steps=keep_steps)

    def residual(self, im_fixed, im_warped):
        if self.cfg.data_loss == 'GMM':
            return ops.lcc_map(im_fixed, im_warped, self.cfg.lcc_s)
        return im_fixed - im_warped

    def nll(self, z_masked):
        """-sum log p(z) with the current data-loss parameters."""
        if self.cfg.data_loss == 'GMM':
            return ops.gmm_nll(z_masked, self.log_std, self.logits)
        return 0.5 * torch.sum((z_masked / self.cfg.ssd_sigma) ** 2)

    def vd_alpha(self, z, mask):
        """trainer.py:507-514; z, mask: (1,1,D,H,W)."""
        if not self.cfg.virtual_decimation:
            return 1.0
        with torch.no_grad():
            if self.cfg.data_loss == 'GMM':
                x = ops.vd_rescale(z.detach(), mask, self.log_std.detach(), self.logits.detach())
            else:
                x = torch.where(mask, (z.detach() / self.cfg.ssd_sigma) ** 2, torch.zeros_like(z))
            return ops.vd_factor(x, mask)

    def gmm_prior_terms(self):
        cfg = self.cfg
        lp = ops.gmm_log_proportions(self.logits)
        return ops.normal_log_pdf(self.log_std, *cfg.scale_prior).sum() + \
            ops.dirichlet_log_pdf(lp, self.concentration).sum()

    def step_gmm(self, z_masked, alpha):
        """One Adam step on (log_std, logits) of alpha * NLL - priors (trainer.py:68-77)."""
        if self.cfg.data_loss != 'GMM':
            return
        loss = ops.gmm_nll(z_masked.detach(), self.log_std, self.logits) * alpha - self.gmm_prior_terms()
        grads = torch.autograd.grad(loss, [self.log_std, self.logits])
        self.adam_gmm.step(grads)

    def init_gmm(self, fixed, moving, v_sample=None, warm_up=25):
        """`Trainer.__GMM_init` (trainer.py:529-547) for a given (unsmoothed) velocity sample (default 0)."""
        cfg = self.cfg
        if cfg.data_loss != 'GMM':
            return
        with torch.no_grad():
            v = torch.zeros(1, 3, *cfg.dims_v) if v_sample is None else v_sample
            transformation, _ = self.transform(self.smooth(v))
            z = self.residual(fixed['im'][:1], ops.warp_trilinear(moving['im'][:1], transformation))
            zm = z[fixed['mask'][:1]]
            self.log_std.data.copy_(ops.gmm_init_log_std(torch.std(zm), cfg.gmm_components))
        alpha = self.vd_alpha(z, fixed['mask'][:1])
        for _ in range(warm_up):
            self.step_gmm(zm, alpha)

    def reg_terms(self, v_s):
        cfg = self.cfg
        y = ops.reg_energy(v_s)
        if cfg.reg_loss == 'RegLoss_L2':
            return ops.reg_l2(y, self.log_w_reg, cfg.dof)
        if cfg.reg_loss == 'RegLoss_Student':
            return ops.reg_student(y, cfg.dof, *ops.student_params(*cfg.student))
        if cfg.reg_loss == 'RegLoss_LogNormal_L2':
            return ops.reg_lognormal_l2(y, cfg.w_reg, cfg.dof)
        return ops.reg_lognormal(y, self.loc, self.log_scale, cfg.dof)

    # ---------------------------------------------------------------- the transition
    def transition(self, fixed, moving, eps, unif=None, keep=False):
        """fixed/moving: dicts with 'im' (C,1,D,H,W) float and 'mask' (C,1,D,H,W) bool.

        eps: (C,3,Nv^3) standard normal; unif: (C,3,D,H,W) uniform [0,1) (needed iff uniform_noise).
        Returns a dict of outputs/intermediates; mutates v, GMM and regulariser parameters.
        """
        cfg = self.cfg
        C = cfg.no_chains
        out = {}

        v_noisy = ops._SGLD.apply(self.v, self.sigma, cfg.lr, eps)
        v_s = self.smooth(v_noisy)
        if keep:
            transformation, displacement, steps = self.transform(v_s, keep_steps=True)
            out['exp_steps'] = [s.detach() for s in steps]
        else:
            transformation, displacement = self.transform(v_s)

        grid = transformation if cfg.uniform_noise is None else ops.jitter_grid(transformation, cfg.uniform_noise, unif)
        im_warped = ops.warp_trilinear(moving['im'], grid)
        z = self.residual(fixed['im'], im_warped)
        mask = fixed['mask']
        z_masked = z[mask].view(C, -1)

        reg_term, log_y = self.reg_terms(v_s)

        data_term = 0.0
        out.update(alpha=[], data=[], reg=[], reg_energy=[], gmm_log_std=[], gmm_logits=[])
        for c in range(C):
            alpha = self.vd_alpha(z[c:c + 1], mask[c:c + 1])
            self.step_gmm(z_masked[c:c + 1], alpha)
            chain_term = self.nll(z_masked[c]) * alpha
            data_term = data_term + chain_term
            out['alpha'].append(float(alpha))
            out['data'].append(float(chain_term))
            out['reg'].append(float(reg_term[c]))
            out['reg_energy'].append(float(log_y[c].exp()))
            out['gmm_log_std'].append(self.log_std.detach().clone())
            out['gmm_logits'].append(self.logits.detach().clone())

        if cfg.data_loss == 'GMM':
            data_term = data_term - self.gmm_prior_terms()

        reg_total = reg_term.sum()
        if cfg.reg_learnable:
            if cfg.reg_loss == 'RegLoss_LogNormal':
                reg_total = reg_total - ops.expgamma_log_pdf(log_y, 0.5 * cfg.reg_loc_prior_nu * cfg.dof,
                                                             0.5 * cfg.reg_loc_prior_nu * cfg.w_reg).sum()
                reg_total = reg_total - ops.normal_log_pdf(self.log_scale, *cfg.reg_scale_prior).sum()
            else:
                shape = 0.5 * cfg.dof  # parse_config.py:136-140
                reg_total = reg_total - ops.expgamma_log_pdf(self.log_w_reg, shape, 1.0 / shape)

        loss = data_term + reg_total
        params = [self.v]
        if self.adam_reg is not None:
            params += [p for g in self.adam_reg.groups for p in g['params']]
        grads = torch.autograd.grad(loss, params)

        with torch.no_grad():
            self.v -= cfg.lr * grads[0]  # torch.optim.SGD, trainer.py:351
        if self.adam_reg is not None:
            self.adam_reg.step(list(grads[1:]))

        out.update(loss=float(loss), grad_v=grads[0].detach(), v_new=self.v.detach().clone(),
                   curr_state=v_s.detach(), transformation=transformation.detach(),
                   displacement=displacement.detach(), im_moving_warped=im_warped.detach(),
                   residuals=z.detach())
        return out


# ------------------------------------------------------------------------------------------------
# VI stage (trainer/trainer.py:79-223): q(v) = N(mu, diag(exp(log_var)) + u u^T), an antithetic sample pair per iteration.
# ------------------------------------------------------------------------------------------------
@dataclass
class OracleVIConfig:
    lr_mu: float = 0.01
    lr_log_var: float = 0.01
    lr_u: float = 0.01
    lr_decay: float = 0.001


def entropy_terms(sample=None, mu=None, log_var=None, u=None):
    """EntropyMultivariateNormal.forward (model/loss.py:342-372): the sample-dependent term, or (sample=None) the
    log-determinant term."""
    sigma = torch.exp(0.5 * log_var)
    dims = (1, 2, 3, 4)
    if sample is None:
        return 0.5 * (torch.log1p(torch.sum(torch.pow(u / sigma, 2), dim=dims)) + torch.sum(log_var, dim=dims))
    sample_n, u_n = (sample - mu) / sigma, u / sigma
    t1 = torch.sum(torch.pow(sample_n, 2), dim=dims)
    t2 = torch.pow(torch.sum(sample_n * u_n, dim=dims), 2) / (1.0 + torch.sum(torch.pow(u_n, 2), dim=dims))
    return 0.5 * (t1 - t2)


class OracleVI:
    """`Trainer._run_VI` body on top of an OracleChain (which owns the GMM / regulariser parameters and their optimisers)."""

    def __init__(self, chain: OracleChain, var_params, vi: OracleVIConfig = OracleVIConfig()):
        self.chain, self.vi = chain, vi
        self.vp = {k: v.clone().requires_grad_(True) for k, v in var_params.items()}
        self.adam = ops.AdamRateDecay([{'params': [self.vp['mu']], 'lr': vi.lr_mu}, {'params': [self.vp['log_var']], 'lr': vi.lr_log_var},
                                       {'params': [self.vp['u']], 'lr': vi.lr_u}], lr_decay=vi.lr_decay)

    def sample_loss(self, fixed, moving, v_unsmoothed, unif):
        """__calc_sample_loss_VI (trainer.py:79-117); steps the GMM as a side effect."""
        ch, cfg = self.chain, self.chain.cfg
        v_s = ch.smooth(v_unsmoothed)
        transformation, displacement = ch.transform(v_s)
        grid = transformation if cfg.uniform_noise is None else ops.jitter_grid(transformation, cfg.uniform_noise, unif)
        warped = ops.warp_trilinear(moving['im'], grid)
        z = ch.residual(fixed['im'], warped)
        alpha = ch.vd_alpha(z, fixed['mask'])
        zm = z[fixed['mask']]
        ch.step_gmm(zm, alpha)
        data = ch.nll(zm) * alpha
        reg, log_y = ch.reg_terms(v_s)
        terms = {'data': data, 'reg': reg.sum(),
                 'entropy': entropy_terms(v_unsmoothed, self.vp['mu'], self.vp['log_var'], self.vp['u']).sum()}
        if cfg.reg_learnable:
            if cfg.reg_loss == 'RegLoss_LogNormal':
                terms['reg_loc_prior'] = ops.expgamma_log_pdf(log_y, 0.5 * cfg.reg_loc_prior_nu * cfg.dof,
                                                              0.5 * cfg.reg_loc_prior_nu * cfg.w_reg).sum()
            else:
                shape = 0.5 * cfg.dof
                terms['w_reg_prior'] = ops.expgamma_log_pdf(ch.log_w_reg, shape, 1.0 / shape)
        return terms, {'displacement': displacement.detach(), 'im_moving_warped': warped.detach(), 'alpha': float(alpha)}

    def step(self, fixed, moving, eps, x, unif_pair=(None, None)):
        """One VI iteration with the antithetic pair mu +- (eps sigma + x u) (utils/sampler.py:4-21)."""
        ch, cfg, vp = self.chain, self.chain.cfg, self.vp
        sigma = torch.exp(0.5 * vp['log_var'])
        pert = eps * sigma + x * vp['u']
        t1, out, = self.sample_loss(fixed, moving, vp['mu'] + pert, unif_pair[0])[:2]
        t2, _ = self.sample_loss(fixed, moving, vp['mu'] - pert, unif_pair[1])
        data = (t1['data'] + t2['data']) / 2.0
        if cfg.data_loss == 'GMM':
            data = data - ch.gmm_prior_terms()
        reg = (t1['reg'] + t2['reg']) / 2.0
        if cfg.reg_learnable:
            if cfg.reg_loss == 'RegLoss_LogNormal':
                reg = reg - (t1['reg_loc_prior'] + t2['reg_loc_prior']) / 2.0
                reg = reg - ops.normal_log_pdf(ch.log_scale, *cfg.reg_scale_prior).sum()
            else:
                reg = reg - (t1['w_reg_prior'] + t2['w_reg_prior']) / 2.0
        entropy = (t1['entropy'] + t2['entropy']) / 2.0 + entropy_terms(None, None, vp['log_var'], vp['u']).sum()
        loss = data + reg - entropy
        params = [vp['mu'], vp['log_var'], vp['u']]
        n_q = len(params)
        if ch.adam_reg is not None:
            params += [p for g in ch.adam_reg.groups for p in g['params']]
        grads = torch.autograd.grad(loss, params)
        if ch.adam_reg is not None:
            ch.adam_reg.step(list(grads[n_q:]))
        self.adam.step(list(grads[:n_q]))
        return {'data': float(data), 'reg': float(reg), 'entropy': float(entropy), 'loss': float(loss), 'alpha': out['alpha'],
                'displacement': out['displacement']}
