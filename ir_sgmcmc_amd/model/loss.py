"""Data and regularisation losses with the reference's class names and constructor arguments (model/loss.py).

Volume work (the LCC map and its adjoint, the finite-difference energy) runs in the HIP kernels; what is left in torch
is scalar / K-vector arithmetic.  `Trainer._SGLD_transition` does not call these forward methods at all -- it hands the
objects' parameters to the fused engine -- but they keep the reference's call surface (`map`, `forward`, `log_pdf`,
`log_pdf_VD`, `init_parameters`, the properties, `reg_loss(v) -> (loss, log_y)`) usable on their own.
"""
import math
from abc import ABC, abstractmethod

import numpy as np
import torch
from torch import nn
from torch.nn.functional import log_softmax

from . import distributions as model_distr
from .. import ops as _ops
from ..utils.diff_op import DifferentialOperator, GradientOperator


class DataLoss(nn.Module, ABC):
    @abstractmethod
    def forward(self, z):
        pass

    @abstractmethod
    def map(self, im_fixed, im_moving):
        pass


class _LCCMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im_fixed, im_moving, s):
        fhat = _ops.lcc_normalise(im_fixed.contiguous(), s)
        z, sigma_m = _ops.lcc_map_fwd(fhat, im_moving.contiguous(), s)
        ctx.s = s
        ctx.save_for_backward(fhat, z, sigma_m)
        return z

    @staticmethod
    def backward(ctx, g_z):
        fhat, z, sigma_m = ctx.saved_tensors
        return None, _ops.lcc_map_bwd(fhat, z, sigma_m, g_z.contiguous(), ctx.s), None


class GMM(DataLoss):
    """zero-mean Gaussian mixture over the LCC-normalised residual (model/loss.py:38-114)"""

    def __init__(self, no_components, s):
        super().__init__()
        self.no_components = no_components
        self.logits = nn.Parameter(torch.zeros(no_components))
        self.log_std = nn.Parameter(torch.zeros(no_components))
        self.register_buffer('_log_sqrt_2pi', torch.tensor(0.5 * math.log(2.0 * math.pi)))
        self.s = s
        self.kernel_sz = s * 2 + 1
        self.sz = float(self.kernel_sz ** 3)

    @torch.no_grad()
    def init_parameters(self, sigma):
        sigma = float(sigma)
        self.log_std.data.copy_(torch.linspace(math.log(sigma / 100.0), math.log(sigma * 5.0), steps=self.no_components))

    @property
    def log_proportions(self):
        return log_softmax(self.logits + 1e-2, dim=0)

    @property
    def log_scales(self):
        return self.log_std

    @property
    def proportions(self):
        return torch.exp(self.log_proportions)

    @property
    def scales(self):
        return torch.exp(self.log_scales)

    @property
    def precision(self):
        return torch.exp(-2.0 * self.log_std)

    def log_pdf(self, z):
        E = 0.5 * (z.reshape(1, -1, 1) * torch.exp(-1.0 * self.log_std)) ** 2
        return torch.logsumexp((self.log_proportions - self.log_std - self._log_sqrt_2pi) - E, dim=-1)

    def log_pdf_VD(self, z_scaled):
        E = 0.5 * z_scaled ** 2
        return torch.logsumexp((self.log_proportions - self.log_std - self._log_sqrt_2pi) - E, dim=-1)

    def forward(self, z):
        return self.reduce(z)

    def map(self, im_fixed, im_moving):
        """z = LCC(F) - LCC(M) with (2s+1)^3 replicate-padded box statistics (HIP, LDS-tiled)"""
        if im_fixed.dim() == 5 and im_fixed.shape[0] > 1 and im_fixed.stride(0) == 0:
            im_fixed = im_fixed[:1]
        return _LCCMap.apply(im_fixed, im_moving, self.s)

    def reduce(self, z):
        return -1.0 * self.log_pdf(z).sum()


class SSD(DataLoss):
    """builder-defined sum of squared differences (BASELINE.json configs 1, 2, 4): z = F - M o phi,
    loss = 0.5 sum (z / sigma)^2.  No counterpart in the reference."""

    def __init__(self, sigma=0.1):
        super().__init__()
        self.sigma = float(sigma)
        self.no_components = 1

    def map(self, im_fixed, im_moving):
        return im_fixed - im_moving

    def forward(self, z):
        return 0.5 * torch.sum((z / self.sigma) ** 2)


class _Energy(torch.autograd.Function):
    """y_c = sum (replicate-padded forward differences)^2 (HIP reduction, fp64 accumulators); backward = 2 D^T D v"""

    @staticmethod
    def forward(ctx, v):
        ctx.save_for_backward(v)
        return _ops.reg_energy(v.contiguous()).to(v.dtype)

    @staticmethod
    def backward(ctx, g_y):
        v, = ctx.saved_tensors
        with torch.enable_grad():
            vv = v.detach().requires_grad_(True)
            y = torch.sum(GradientOperator()(vv) ** 2, dim=(1, 2, 3, 4, 5))
            g, = torch.autograd.grad(y, vv, g_y.to(y.dtype))
        return g


class RegLoss(nn.Module, ABC):
    """all regularisers are functions of the energy y = |D v|^2 (model/loss.py:122-169)"""

    def __init__(self, diff_op=None, dims=None, learnable=False):
        super().__init__()
        self.dims = dims
        self.dof = np.prod(dims) * 3.0
        self.learnable = learnable
        if diff_op is None:
            self.diff_op = DifferentialOperator()
        elif isinstance(diff_op, str):
            self.diff_op = DifferentialOperator.from_string(diff_op)
        elif isinstance(diff_op, DifferentialOperator):
            self.diff_op = diff_op
        else:
            self.diff_op = diff_op()

    def forward(self, input, *args, **kwargs):
        if isinstance(self.diff_op, GradientOperator):
            y = _Energy.apply(input)
        else:
            y = torch.sum(self.diff_op(input) ** 2, dim=tuple(range(1, input.dim())))
        return self._loss(y, *args, **kwargs)

    @abstractmethod
    def _loss(self, y, *args, **kwargs):
        pass


class RegLoss_L2(RegLoss):
    """0.5 w y - 0.5 dof log w (model/loss.py:172-198)"""

    def __init__(self, w_reg, diff_op=None, dims=None, learnable=False):
        super().__init__(diff_op=diff_op, dims=dims, learnable=learnable)
        self.log_w_reg = nn.Parameter(torch.tensor(math.log(w_reg)), requires_grad=learnable)

    def _loss(self, y):
        return 0.5 * self.log_w_reg.exp() * y - 0.5 * self.dof * self.log_w_reg, y.log()


class RegLoss_Student(RegLoss):
    """model/loss.py:201-241"""

    def __init__(self, diff_op=None, dims=None, nu0=2e-6, lambda0=1e-6, a0=1e-6, b0=1e-6):
        super().__init__(diff_op=diff_op, dims=dims, learnable=False)
        self.a0 = nu0 / 2.0 if nu0 != 2e-6 else a0
        if lambda0 != 1e-6:
            b0 = self.a0 / lambda0
        self.b0_twice = b0 * 2.0

    def _loss(self, y):
        return torch.log(self.b0_twice + y) * (self.a0 + 0.5 * self.dof), y.log()


class RegLoss_EnergyBased(RegLoss):
    """-log p(y) + (dof/2 - 1) log y (model/loss.py:244-270)"""

    @abstractmethod
    def _mlog_energy_prior(self, y, *args, **kwargs):
        pass

    def _loss(self, y, *args, **kwargs):
        return self._mlog_energy_prior(y, *args, **kwargs) + (0.5 * self.dof - 1.0) * y.log(), y.log()


class RegLoss_LogNormal(RegLoss_EnergyBased):
    """log-normal prior on the energy; loc0 = E[expGamma(dof/2, w/2)], log_scale0 = log 4 + log loc0
    (model/loss.py:273-312)"""

    def __init__(self, w_reg=1.0, diff_op=None, dims=None, learnable=False):
        super().__init__(diff_op=diff_op, dims=dims, learnable=learnable)
        loc_init = model_distr.LogEnergyExpGammaPrior(w_reg, self.dof).expectation().clone().detach()
        self.loc = nn.Parameter(loc_init, requires_grad=learnable)
        self.log_scale = nn.Parameter(math.log(4.0) + loc_init.log(), requires_grad=learnable)
        self.w_reg = w_reg

    @property
    def scale(self):
        return self.log_scale.exp()

    def _mlog_energy_prior(self, y, *args, **kwargs):
        return y.log() + self.log_scale + 0.5 * ((y.log() - self.loc) / self.scale) ** 2


class RegLoss_LogNormal_L2(RegLoss_EnergyBased):
    """model/loss.py:315-321"""

    def __init__(self, w_reg, diff_op=None, dims=None):
        super().__init__(diff_op=diff_op, dims=dims, learnable=False)
        self.gamma_distr = model_distr._GammaDistribution(0.5 * self.dof, 0.5 * w_reg, learnable=False)

    def _mlog_energy_prior(self, y, *args, **kwargs):
        return -1.0 * self.gamma_distr(y.log())


class EntropyMultivariateNormal(nn.Module):
    """entropy terms of the VI stage (model/loss.py:342-372); out of the MCMC hot path, kept so that configs resolve"""

    def forward(self, **kwargs):
        log_var, u = kwargs['log_var'], kwargs['u']
        sigma = torch.exp(0.5 * log_var)
        if len(kwargs) == 2:
            return 0.5 * (torch.log1p(torch.sum(torch.pow(u / sigma, 2), dim=(1, 2, 3, 4))) + torch.sum(log_var, dim=(1, 2, 3, 4)))
        sample_n, u_n = (kwargs['sample'] - kwargs['mu']) / sigma, u / sigma
        t1 = torch.sum(torch.pow(sample_n, 2), dim=(1, 2, 3, 4))
        t2 = torch.pow(torch.sum(sample_n * u_n, dim=(1, 2, 3, 4)), 2) / (1.0 + torch.sum(torch.pow(u_n, 2), dim=(1, 2, 3, 4)))
        return 0.5 * (t1 - t2)
