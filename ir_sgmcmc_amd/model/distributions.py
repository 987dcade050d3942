"""Hyper-priors with the reference's class names (model/distributions.py).  Scalar host-side maths on K-vectors or
single values -- there is no volume work here, so no kernel; the fused transition evaluates the same formulas (and
their gradients) on the device in csrc/scalar_kernels.hip from the parameters held by these objects."""
import math

import torch
from torch import nn

LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def _as_param(value, learnable=False):
    t = torch.as_tensor(value, dtype=torch.get_default_dtype()).clone().detach().squeeze()
    return nn.Parameter(t, requires_grad=learnable)


class NormalDistribution(nn.Module):
    """log N(x; loc, scale) (model/distributions.py:11-59); defaults loc=0, scale=log(10) as in the reference"""

    def __init__(self, loc=None, scale=None, learnable=False):
        super().__init__()
        self.loc = _as_param(0.0 if loc is None else loc, learnable)
        self.log_scale = _as_param(math.log(math.log(10) if scale is None else float(scale)), learnable)

    def forward(self, x):
        return -0.5 * ((x - self.loc) * torch.exp(-self.log_scale)) ** 2 - self.log_scale - LOG_SQRT_2PI


def gamma_log_pdf(log_x, shape, rate):
    """log Gamma(x; shape, rate) as a function of log x (model/distributions.py:112-113)"""
    shape, rate = torch.as_tensor(shape), torch.as_tensor(rate)
    return shape * torch.log(rate) + (shape - 1) * log_x - rate * log_x.exp() - torch.lgamma(shape)


def expgamma_log_pdf(x, shape, rate):
    """density of X = log Z, Z ~ Gamma(shape, rate) (model/distributions.py:165-166)"""
    return gamma_log_pdf(x, shape, rate) + x


def expgamma_expectation(shape, rate):
    """E[log Z] = digamma(shape) - log(rate) (model/distributions.py:169-170)"""
    shape, rate = torch.as_tensor(shape), torch.as_tensor(rate)
    return torch.digamma(shape) - torch.log(rate)


class _GammaDistribution(nn.Module):
    def __init__(self, shape=1e-3, rate=1e-3, shape_learnable=False, rate_learnable=False, learnable=False):
        super().__init__()
        self.shape = _as_param(shape, learnable and shape_learnable)
        self.rate = _as_param(rate, learnable and rate_learnable)

    def expectation(self):
        return self.shape / self.rate

    def forward(self, log_x):
        return gamma_log_pdf(log_x, self.shape, self.rate)


class ExpGammaDistribution(nn.Module):
    def __init__(self, shape=1e-3, rate=1e-3, shape_learnable=False, rate_learnable=False, learnable=False):
        super().__init__()
        self.gamma_distribution = _GammaDistribution(shape, rate, shape_learnable, rate_learnable, learnable)

    def expectation(self):
        return expgamma_expectation(self.gamma_distribution.shape, self.gamma_distribution.rate)

    def forward(self, x):
        return self.gamma_distribution(x) + x


class DirichletPrior(nn.Module):
    """log Dir(pi; alpha) as a function of log pi (model/distributions.py:180-211)"""

    def __init__(self, no_classes, alpha=None):
        super().__init__()
        alpha = 0.5 if alpha is None else alpha
        try:
            conc = torch.full((no_classes,), float(alpha))
        except (TypeError, ValueError):
            conc = torch.as_tensor(alpha, dtype=torch.get_default_dtype()).clone().squeeze()
            if conc.numel() != no_classes:
                raise ValueError('Invalid tensor size. Expected {}, got: {}'.format(no_classes, conc.numel()))
        self.concentration = nn.Parameter(conc, requires_grad=False)

    def forward(self, log_proportions):
        c = self.concentration
        return (log_proportions * (c - 1.0)).sum(-1) + torch.lgamma(c.sum(-1)) - torch.lgamma(c).sum(-1)


class LogPrecisionExpGammaPrior(nn.Module):
    """hyper-prior on log w_reg (model/distributions.py:214-225)"""

    def __init__(self, shape=1e-3, rate=1e-3, shape_learnable=False, rate_learnable=False, learnable=False):
        super().__init__()
        self.expgamma_distribution = ExpGammaDistribution(shape, rate, shape_learnable, rate_learnable, learnable)

    def forward(self, x):
        return self.expgamma_distribution(x)


class LogEnergyExpGammaPrior(nn.Module):
    """prior on the location parameter of the log-normal energy model (model/distributions.py:228-245)"""

    def __init__(self, w_reg, dof, nu=1.0, learnable=False):
        super().__init__()
        self.nu = nn.Parameter(torch.tensor(float(nu)), requires_grad=learnable)
        self.register_buffer('w_reg', torch.tensor(float(w_reg)))
        self.register_buffer('dof', torch.tensor(float(dof), dtype=torch.float64))

    def expectation(self):
        return expgamma_expectation(0.5 * self.nu * self.dof, 0.5 * self.nu * self.w_reg)

    def forward(self, log_energy):
        return expgamma_log_pdf(log_energy, 0.5 * self.nu * self.dof, 0.5 * self.nu * self.w_reg)


class LogScaleNormalPrior(nn.Module):
    """prior on a log standard deviation / log scale (model/distributions.py:248-258)"""

    def __init__(self, loc, scale, learnable=False):
        super().__init__()
        self.normal = NormalDistribution(loc, scale, learnable)

    def forward(self, log_scale):
        return self.normal(log_scale)
