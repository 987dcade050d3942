"""Deterministic synthetic fixed/moving pair.

The reference ships no image data (`configs/*/config.json:92` point at the author's cluster), so the
bench, the smoke test and the parity tests use this generator.  It yields the same dict contract as
the reference's `BiobankDataset.__getitem__` (data_loader/datasets.py:107-145): `fixed`/`moving` dicts
with `im` float32 (1,D,H,W), `mask` bool, `seg` int16, plus the VI parameters `mu`, `log_var`, `u`.
Host-side, CPU tensors; the trainer moves them to the device.
"""
import math

import torch


def _coords(dims):
    D, H, W = dims
    z = torch.linspace(-1, 1, D).view(D, 1, 1)
    y = torch.linspace(-1, 1, H).view(1, H, 1)
    x = torch.linspace(-1, 1, W).view(1, 1, W)
    return z, y, x


def _blobs(dims, shift_a, shift_b):
    z, y, x = _coords(dims)
    za, ya, xa = shift_a
    r2 = (z - za) ** 2 + (y - ya) ** 2 + (x - xa) ** 2
    zb, yb, xb = 0.3 + shift_b[0], -0.2 + shift_b[1], 0.1 + shift_b[2]
    q2 = (z - zb) ** 2 + (y - yb) ** 2 + (x - xb) ** 2
    return torch.exp(-4.0 * r2) + 0.5 * torch.exp(-30.0 * q2)


def synthetic_pair(dims, seed=0, noise=0.02):
    """Two Gaussian blobs + white noise; the moving image has both blobs shifted.  Returns (fixed, moving)."""
    dims = tuple(int(d) for d in dims)
    g = torch.Generator().manual_seed(seed)
    im_f = _blobs(dims, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)) + noise * torch.randn(dims, generator=g)
    im_m = _blobs(dims, (0.08, -0.04, 0.05), (0.06, 0.05, 0.02)) + noise * torch.randn(dims, generator=g)
    z, y, x = _coords(dims)
    r2 = z ** 2 + y ** 2 + x ** 2
    mask = (r2 < 0.9).expand(dims)
    # three nested label shells so that nearest-neighbour warps / Dice have something to chew on
    seg = (torch.zeros(dims, dtype=torch.int16) + (r2 < 0.5).to(torch.int16) * 10 + (r2 < 0.2).to(torch.int16) * 6
           + (r2 < 0.05).to(torch.int16) * 33)

    def pack(im):
        return {'im': im.float().unsqueeze(0).contiguous(), 'mask': mask.unsqueeze(0).contiguous(),
                'seg': seg.unsqueeze(0).contiguous()}

    return pack(im_f), pack(im_m)


def init_var_params(dims_v, sigma_v_init=0.5, u_v_init=0.1):
    """mu = 0, log_var = log(sigma^2), u = const (data_loader/datasets.py:57-68)."""
    shape = (3, *dims_v)
    return {'mu': torch.zeros(shape), 'log_var': torch.full(shape, math.log(sigma_v_init ** 2)),
            'u': torch.full(shape, float(u_v_init))}
