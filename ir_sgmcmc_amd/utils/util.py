"""Numerics helpers with the reference's names (utils/util.py), backed by the HIP operators where there is volume work.

Tiny scalar / bookkeeping helpers (json io, MetricTracker, coordinate scaling of a whole tensor by a constant) are
plain host code, as in the reference.
"""
import json
import math
from collections import OrderedDict
from pathlib import Path

import torch

from .. import ops as _ops


def ensure_dir(dirname):
    dirname = Path(dirname)
    if not dirname.is_dir():
        dirname.mkdir(parents=True, exist_ok=False)


def read_json(fname):
    with Path(fname).open('rt') as handle:
        return json.load(handle, object_hook=OrderedDict)


def write_json(content, fname):
    with Path(fname).open('wt') as handle:
        json.dump(content, handle, indent=4, sort_keys=False)


def get_control_grid_size(dims, cps):
    """utils/util.py:61-69"""
    return _ops.control_grid_size(dims, cps)


def _axis_scale(field, inverse=False):
    # channel c <-> its own axis (x <-> W ...); identical to utils/util.py:418-443 for the cubic volumes it supports
    n = [float(s - 1) for s in reversed(field.shape[2:])]
    f = [(x / 2.0) if inverse else (2.0 / x) for x in n]
    return torch.tensor(f, dtype=field.dtype, device=field.device).view(1, -1, *([1] * (field.dim() - 2)))


def transform_coordinates(field):
    """absolute voxel units -> normalised [-1, 1] units (utils/util.py:418-429)"""
    return field * _axis_scale(field)


def transform_coordinates_inv(field):
    """normalised -> absolute voxel units (utils/util.py:432-443)"""
    return field * _axis_scale(field, inverse=True)


def init_identity_grid_3D(dims, device=None):
    """(1, D, H, W, 3) identity grid in [-1, 1], channel 0 = x = last axis (utils/util.py:263-278)"""
    nz, ny, nx = dims[0], dims[1], dims[2]
    x = torch.linspace(-1, 1, steps=nx, device=device).view(1, 1, nx).expand(nz, ny, nx)
    y = torch.linspace(-1, 1, steps=ny, device=device).view(1, ny, 1).expand(nz, ny, nx)
    z = torch.linspace(-1, 1, steps=nz, device=device).view(nz, 1, 1).expand(nz, ny, nx)
    return torch.stack((x, y, z), dim=-1).unsqueeze(0)


def get_noise_uniform(shape, device, alpha):
    return -2.0 * alpha * torch.rand(shape, device=device) + alpha


def get_noise_Langevin(sigma, tau):
    return math.sqrt(2.0 * tau) * sigma * torch.randn_like(sigma)


def add_noise_uniform_field(field, alpha):
    """utils/util.py:44-45"""
    return field + transform_coordinates(get_noise_uniform(field.shape, field.device, alpha))


def add_noise_Langevin(field, sigma, tau):
    """utils/util.py:48-49; eps is drawn with torch's device generator, the add runs in the HIP kernel"""
    return _ops.perturb_smooth(field.contiguous(), None, sigma.contiguous(), torch.randn_like(sigma), tau=tau)


def separable_conv_3D(field, *args):
    """Both branches of utils/util.py:350-406: (kernel (3,1,k), padding_sz) or (S_x, S_y, S_z, padding).

    The 2-argument branch (utils/util.py:362-392) pads the last axis by replicate, flattens the volume, runs a zero-padded
    conv1d and crops -- three times, permuting the axes in between.  The crop removes exactly the positions the zero padding
    and the neighbouring rows reach, so it IS one (2 p + 1)-tap filter per axis with replicate padding, like the 4-argument
    branch (verified against the imported reference to 3e-7 and pinned by tests/golden/utils_ops.npz, including per-channel
    kernels: `groups=3` gives channel c row c of the kernel tensor)."""
    k = args[0].reshape(args[0].shape[0], -1).detach().cpu()
    if len(args) == 4:  # the three axis kernels of every reference call site are the same taps reshaped
        ky, kz = (a.reshape(a.shape[0], -1).detach().cpu() for a in args[1:3])
        if not (torch.equal(k, ky) and torch.equal(k, kz)):
            raise NotImplementedError('separable_conv_3D: different kernels per axis')
    f = field.contiguous()
    if all(torch.equal(k[0], k[c]) for c in range(1, k.shape[0])):
        return _ops.perturb_smooth(f, k[0].tolist())
    out = torch.empty_like(f)
    for c in range(k.shape[0]):  # a kernel of its own per channel: filter with each, keep the matching channel
        out[:, c] = _ops.perturb_smooth(f, k[c].tolist())[:, c]
    return out


def calc_norm(field):
    """voxel-wise L2 norm of a batch of 3-D vector fields (utils/util.py:215-225)"""
    return torch.linalg.vector_norm(field, ord=2, dim=1, keepdim=True)


def calc_det_J(nabla):
    """utils/util.py:72-91 on an explicit nabla tensor (C,3,D,H,W,3)"""
    a, b, c = nabla[..., 0], nabla[..., 1], nabla[..., 2]
    return (a[:, 0] * b[:, 1] * c[:, 2] + b[:, 0] * c[:, 1] * a[:, 2] + c[:, 0] * a[:, 1] * b[:, 2]
            - a[:, 2] * b[:, 1] * c[:, 0] - b[:, 2] * c[:, 1] * a[:, 0] - c[:, 2] * a[:, 1] * b[:, 0])


def calc_no_non_diffeomorphic_voxels(transformation, diff_op=None):
    """(NaN count of log det J per chain as numpy, log det J) -- utils/util.py:209-212, one fused HIP kernel"""
    cnt, log_det = _ops.log_det_jacobian(transformation.contiguous())
    return cnt.cpu().numpy(), log_det


@torch.no_grad()
def calc_posterior_statistics(samples, device='cuda:0'):
    samples = samples.to(device)
    return torch.mean(samples, dim=0), torch.std(samples, dim=0)


@torch.no_grad()
def calc_DSC_GPU(no_samples, seg_fixed, seg_moving, structures_dict):
    """Dice scores on the device (utils/util.py:123-148)"""
    DSC = torch.zeros(no_samples, len(structures_dict))
    for idx in range(no_samples):
        f, m = seg_fixed[idx], seg_moving[idx]
        for j, label in enumerate(structures_dict.values()):
            num = 2.0 * ((f == label) & (m == label)).sum()
            den = (f == label).sum() + (m == label).sum()
            DSC[idx, j] = num / den  # a label absent from both gives 0 / 0 = NaN, as in the reference (its `except` never fires)
    return DSC.numpy()


def rescale_residuals(res, mask, data_loss):
    """VD-rescaled residual x = sum_k r_k (z / sigma_k)^2 (utils/util.py:330-347).  The reference obtains it as
    sum_k s_k * d(-log p)/d(s_k) with a nested backward; the closed form with the responsibilities r_k is the same number."""
    with torch.no_grad():
        z = torch.where(mask, res, torch.zeros_like(res)).reshape(1, -1, 1)
        s = z * torch.exp(-1.0 * data_loss.log_std)
        t = (data_loss.log_proportions - data_loss.log_std) - 0.5 * s ** 2
        r = torch.softmax(t, dim=-1)
        return torch.sum(r * s * s, dim=-1).view(res.shape)


@torch.no_grad()
def calc_VD_factor(residual, mask):
    """virtual decimation factor from the lag-1 correlations of the rescaled residual (utils/util.py:446-485)"""
    var_res = torch.mean(residual[mask] ** 2)
    n = mask.sum()
    rm = torch.where(mask, residual, torch.zeros_like(residual))
    cov = [torch.sum(rm[:, :, :-1] * rm[:, :, 1:]) / n, torch.sum(rm[:, :, :, :-1] * rm[:, :, :, 1:]) / n,
           torch.sum(rm[:, :, :, :, :-1] * rm[:, :, :, :, 1:]) / n]
    sq = [torch.clamp(-2.0 / math.pi * torch.log(c / var_res), max=1.0) for c in cov]
    return torch.sqrt(sq[0] * sq[1] * sq[2])


@torch.no_grad()
def max_field_update(field_old, field_new):
    """largest voxel-wise update of a vector field in the L2 norm and its index (utils/util.py:281-299)"""
    diff = torch.abs(calc_norm(field_new) - calc_norm(field_old))   # difference of the norms, as the reference defines it
    return torch.max(diff), torch.argmax(diff)


class MetricTracker:
    """running means keyed by name (utils/util.py:488-510 without the pandas dependency)"""

    def __init__(self, *keys, writer=None):
        self.writer = writer
        self._total = {k: 0.0 for k in keys}
        self._count = {k: 0 for k in keys}

    def reset(self):
        for k in self._total:
            self._total[k], self._count[k] = 0.0, 0

    def update(self, key, value, n=1):
        if self.writer is not None:
            self.writer.add_scalar(key, value)
        self._total[key] = self._total.get(key, 0.0) + value * n
        self._count[key] = self._count.get(key, 0) + n

    def avg(self, key):
        return self._total[key] / max(self._count[key], 1)

    def result(self):
        return {k: self.avg(k) for k in self._total}
