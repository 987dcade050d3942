"""Differential operators (reference utils/diff_op.py)."""
import numpy as np
import torch
from torch import nn

from .. import ops as _ops


class DifferentialOperator(nn.Module):
    """identity operator; `from_string` builds a subclass by (sub)name (utils/diff_op.py:17-38)"""

    @staticmethod
    def from_string(s, *args, **kwargs):
        for cls in (GradientOperator,):
            if cls.__name__ in s:
                return cls(*args, **kwargs)
        raise ValueError('Unknown differential operator: {}'.format(s))

    def forward(self, input):
        return input


class _EnergyGrad(torch.autograd.Function):
    """forward: nabla; backward: exact adjoint of the replicate-padded forward differences (autograd glue in torch)."""

    @staticmethod
    def forward(ctx, v, transformation):
        ctx.shape = v.shape
        ctx.transformation = transformation
        return _ops.gradient_operator(v.contiguous(), transformation)

    @staticmethod
    def backward(ctx, g):
        C, _, D, H, W = ctx.shape
        out = torch.zeros(ctx.shape, device=g.device, dtype=g.dtype)
        sp = [2.0 / (W - 1), 2.0 / (H - 1), 2.0 / (D - 1)] if ctx.transformation else [1.0, 1.0, 1.0]
        for a, dim in enumerate((4, 3, 2)):
            ga = g[:, a].permute(0, 4, 1, 2, 3) / sp[a]  # (C, comp, D, H, W)
            n = ctx.shape[dim]
            w = ga.narrow(dim, 0, n - 1).clone()
            # the last plane replicates the previous difference: its gradient lands on difference n-2
            idx = [slice(None)] * 5
            idx[dim] = n - 2
            idx2 = list(idx)
            idx2[dim] = n - 1
            w[tuple(idx)] = w[tuple(idx)] + ga[tuple(idx2)]
            out.narrow(dim, 1, n - 1).add_(w)
            out.narrow(dim, 0, n - 1).sub_(w)
        return out, None


class GradientOperator(DifferentialOperator):
    """forward differences, the difference array replicate-padded; output (C, 3[d/dx,d/dy,d/dz], D, H, W, 3[component])
    (utils/diff_op.py:62-96)"""

    def __init__(self):
        super().__init__()
        self.pixel_spacing = None

    def _set_spacing(self, field):
        self.pixel_spacing = 2.0 / (np.asarray(field.shape[2:]) - 1)

    def forward(self, v, transformation=False):
        if transformation and self.pixel_spacing is None:
            self._set_spacing(v)
        return _EnergyGrad.apply(v, bool(transformation))
