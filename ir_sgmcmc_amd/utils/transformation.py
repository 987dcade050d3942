"""Transformation models (reference utils/transformation.py): SVF_3D, Cubic_B_spline_FFD_3D, SVFFD_3D.

Forward and backward both run in the HIP kernels; torch.autograd only carries the saved scaling-and-squaring steps.
"""
from abc import ABC, abstractmethod

import torch
from torch import nn

from .. import ops as _ops


class TransformationModule(nn.Module, ABC):
    @abstractmethod
    def forward(self, v):
        pass


class _SVFExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, no_steps):
        v = v.contiguous()
        transformation, displacement, steps = _ops.svf_exp_fwd(v, no_steps)
        ctx.save_for_backward(v, steps)
        return transformation, displacement

    @staticmethod
    def backward(ctx, g_t, g_disp):
        v, steps = ctx.saved_tensors
        D, H, W = v.shape[2:]
        g = torch.zeros_like(v) if g_t is None else g_t.clone()
        if g_disp is not None:  # displacement = d * (n - 1) / 2 per channel (x <-> W ...)
            sc = torch.tensor([(W - 1) / 2.0, (H - 1) / 2.0, (D - 1) / 2.0], device=v.device).view(1, 3, 1, 1, 1)
            g = g + g_disp * sc
        return _ops.svf_exp_bwd(v, steps, g.contiguous()), None


class SVF_3D(TransformationModule):
    """stationary velocity field integrated by scaling and squaring (utils/transformation.py:51-76)"""

    def __init__(self, dims, no_steps=12):
        super().__init__()
        self.dims = tuple(dims)
        self.no_steps = no_steps

    def forward(self, v):
        return _SVFExp.apply(v, self.no_steps)


class _FFDUp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, dims, cps):
        ctx.cps = cps
        return _ops.ffd_up(v.contiguous(), dims, cps)

    @staticmethod
    def backward(ctx, g):
        return _ops.ffd_adjoint(g.contiguous(), ctx.cps), None, None


class Cubic_B_spline_FFD_3D(TransformationModule):
    """dense velocity from control-point parameters by separable cubic B-spline interpolation
    (utils/transformation.py:126-153)"""

    def __init__(self, dims, cps):
        super().__init__()
        self.dims = tuple(dims)
        self.stride = tuple(cps)

    def forward(self, v):
        return _FFDUp.apply(v, self.dims, self.stride)


class SVFFD_3D(TransformationModule):
    """utils/transformation.py:156-164"""

    def __init__(self, dims, cps):
        super().__init__()
        self.cubic_B_spline_FFD = Cubic_B_spline_FFD_3D(dims, cps)
        self.SVF_3D = SVF_3D(dims)
        self.dims, self.cps = tuple(dims), tuple(cps)

    def forward(self, v):
        return self.SVF_3D(self.cubic_B_spline_FFD(v))
