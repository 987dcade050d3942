"""RegistrationModule (reference utils/registration.py:6-41)."""
import torch
from torch import nn

from .. import ops as _ops
from .util import init_identity_grid_3D


class _Warp(torch.autograd.Function):
    """trilinear / border / align_corners warp; the gradient flows to the transformation only (the moving image is data)"""

    @staticmethod
    def forward(ctx, im, transformation):
        transformation = transformation.contiguous()
        ctx.save_for_backward(im, transformation)
        return _ops.warp(im, transformation)

    @staticmethod
    def backward(ctx, g):
        im, transformation = ctx.saved_tensors
        ident = init_identity_grid_3D(transformation.shape[2:], transformation.device).permute(0, 4, 1, 2, 3)
        return None, _ops.warp_displacement_bwd(im, (transformation - ident).contiguous(), g.contiguous())


class RegistrationModule(nn.Module):
    """warps float images (trilinear) and bool masks / int16 label maps (nearest neighbour)"""

    def forward(self, im_or_seg_moving, transformation):
        im = im_or_seg_moving
        if im.dim() == 5 and im.shape[0] > 1 and im.stride(0) == 0:
            im = im[:1]  # .expand()-ed chains share one volume
        im = im.contiguous()
        if im.dtype == torch.float32:
            return _Warp.apply(im, transformation)
        if im.dtype in (torch.bool, torch.int16):
            return _ops.warp(im, transformation.detach().contiguous())
        raise NotImplementedError  # utils/registration.py:32
