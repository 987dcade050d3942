"""Known-answer tests of the reference, restated against the oracle (device-free), plus the explicit
trilinear sampler/adjoints against ATen.

Reference tests restated: tests/test_diff.py:9-113 (gradient operator, det J), tests/test_utils.py:12-30 (norm),
:75-99 (B-spline / SVFFD shapes), :101-151 (separable conv of an all-ones kernel = 27).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops

ATOL = 1e-4  # reference tests/test_setup.py:46


def test_diff_uniform_field_has_zero_gradient():
    v = torch.zeros(1, 3, 16, 16, 16)
    v[0, 0], v[0, 1], v[0, 2] = 5.0, 4.0, 2.0
    assert torch.allclose(ops.forward_differences(v), torch.zeros(1, 3, 16, 16, 16, 3), atol=ATOL)


def test_diff_linear_field():
    N = 16
    z, y, x = torch.meshgrid(torch.arange(N), torch.arange(N), torch.arange(N), indexing='ij')
    v = torch.zeros(1, 3, N, N, N)
    v[0, 0] = x.float()
    v[0, 1] = 1.5 * y + 3.0 * z + 1.0
    nabla = ops.forward_differences(v)
    exp_x = torch.zeros(1, 3, N, N, N)
    exp_x[:, 0] = 1.0
    exp_y = torch.zeros(1, 3, N, N, N)
    exp_y[:, 1], exp_y[:, 2] = 1.5, 3.0
    assert torch.allclose(nabla[..., 0], exp_x, atol=ATOL)
    assert torch.allclose(nabla[..., 1], exp_y, atol=ATOL)
    assert torch.allclose(nabla[..., 2], torch.zeros(1, 3, N, N, N), atol=ATOL)


def test_replicated_last_difference():
    # x^2 sampled at 0..3 -> forward differences [1, 3, 5] padded to [1, 3, 5, 5] (utils/diff_op.py:83-85)
    v = torch.zeros(1, 3, 4, 4, 4)
    v[0, 0] = (torch.arange(4.0) ** 2).view(1, 1, 4)
    assert torch.equal(ops.forward_differences(v)[0, 0, 0, 0, :, 0], torch.tensor([1.0, 3.0, 5.0, 5.0]))


def test_log_det_J_identity_and_stretch():
    N = 16
    ident = ops.identity_grid((N, N, N)).permute(0, 4, 1, 2, 3)
    det = ops.det_jacobian(ops.forward_differences(ident, transformation=True))
    assert torch.allclose(torch.log(det + 1e-5), torch.zeros_like(det), atol=ATOL)
    stretch = ident + (ident + 1.0)  # displacement 2 i / (N - 1) per axis -> det J = 8
    det = ops.det_jacobian(ops.forward_differences(stretch, transformation=True))
    assert torch.allclose(torch.log(det), torch.full_like(det, math.log(8.0)), atol=ATOL)


def test_det_J_polynomial():
    N = 4
    z, y, x = [t.float() for t in torch.meshgrid(torch.arange(N), torch.arange(N), torch.arange(N), indexing='ij')]
    nx = torch.stack((x, z ** 2, y)).unsqueeze(0)
    ny = torch.stack((y, x ** 2, z)).unsqueeze(0)
    nz = torch.stack((y ** 2, x, x)).unsqueeze(0)
    det = ops.det_jacobian(torch.stack((nx, ny, nz), dim=-1))
    true = x ** 4 - x ** 2 * y ** 3 - x ** 2 * z + x * y ** 2 - x * y * z ** 2 + y ** 2 * z ** 3
    assert torch.allclose(det[0], true, atol=ATOL)


def test_norm():
    v = torch.cat((torch.ones(1, 3, 8, 8, 8), 2.0 * torch.ones(1, 3, 8, 8, 8)))
    n = ops.field_norm(v)
    assert torch.allclose(n[0], torch.full((1, 8, 8, 8), math.sqrt(3.0)), atol=ATOL)
    assert torch.allclose(n[1], torch.full((1, 8, 8, 8), math.sqrt(12.0)), atol=ATOL)


def test_separable_conv_ones_kernel_gives_27():
    v = torch.zeros(2, 3, 16, 16, 16)
    v[0, 1], v[1, 2] = 1.0, 1.0
    out = ops.separable_conv3d_replicate(v, torch.ones(3))
    assert out.shape == v.shape
    assert torch.allclose(out, 27.0 * v, atol=ATOL)


def test_sobolev_kernel_values():
    k = ops.sobolev_kernel_1d(3, 0.5)
    np.testing.assert_allclose(k, [.0104, .0417, .1563, .5833, .1563, .0417, .0104], atol=6e-5)
    assert abs(k.sum() - 1.0) < 1e-12


@pytest.mark.parametrize('N,cps', [(64, 4), (16, 2), (17, 4)])
def test_ffd_shapes(N, cps):
    dims, c = (N,) * 3, (cps,) * 3
    g = ops.control_grid_size(dims, c)
    assert g == (math.ceil((N - 1) / cps) + 3,) * 3
    v = torch.randn(1, 3, *g)
    dense = ops.ffd_upsample(v, dims, c)
    assert dense.shape == (1, 3, N, N, N)
    t, d = ops.svf_exp(dense)
    assert t.shape == d.shape == (1, 3, N, N, N)
    # partition of unity: constant control points give the same constant dense field
    assert torch.allclose(ops.ffd_upsample(torch.ones(1, 3, *g), dims, c), torch.ones(1, 3, N, N, N), atol=1e-5)


def test_svf_zero_velocity_is_identity_and_translation():
    N = 12
    t, d = ops.svf_exp(torch.zeros(1, 3, N, N, N))
    assert torch.equal(d, torch.zeros_like(d))
    v = torch.zeros(1, 3, N, N, N)
    v[:, 0] = 0.5
    t, d = ops.svf_exp(v)
    assert torch.allclose(d[:, 0, :, :, 2:-2], torch.full((1, N, N, N - 4), 0.5), atol=1e-5)


@pytest.mark.parametrize('scale', [0.3, 3.0])
def test_explicit_trilinear_matches_aten(scale):
    torch.manual_seed(0)
    B, Cn, D, H, W = 2, 3, 7, 9, 11
    inp = torch.randn(B, Cn, D, H, W, requires_grad=True)
    grid = (ops.identity_grid((D, H, W)) + scale * 0.2 * torch.randn(B, D, H, W, 3)).requires_grad_(True)
    # exercise exact-border and out-of-range coordinates
    with torch.no_grad():
        grid[0, 0, 0, :, 0] = 1.0
        grid[0, 1, 0, :, 1] = -1.0
        grid[1, 2, 3, :, 2] = 1.5
    out = F.grid_sample(inp, grid, mode='bilinear', padding_mode='border', align_corners=True)
    mine = ops.trilinear_sample_explicit(inp.detach(), grid.detach())
    assert torch.allclose(out, mine, atol=1e-6)
    gout = torch.randn_like(out)
    gi, gg = torch.autograd.grad(out, [inp, grid], gout)
    gi2, gg2 = ops.trilinear_backward_explicit(inp.detach(), grid.detach(), gout)
    assert torch.allclose(gi, gi2, atol=1e-5)
    assert torch.allclose(gg, gg2, atol=1e-4)


def test_vd_closed_form_matches_nested_autograd():
    torch.manual_seed(1)
    z = torch.randn(1, 1, 8, 8, 8) * 0.3
    mask = torch.rand(1, 1, 8, 8, 8) > 0.2
    log_std = torch.linspace(-3.0, 0.5, 4)
    logits = torch.tensor([0.1, -0.2, 0.3, 0.0])
    a = ops.vd_rescale_autograd(z, mask, log_std, logits)
    b = ops.vd_rescale(z, mask, log_std, logits)
    assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
    alpha = ops.vd_factor(b, mask)
    assert 0.0 < float(alpha) <= 1.0


def test_lcc_map_is_zero_for_identical_images_and_affine_invariant():
    torch.manual_seed(2)
    im = torch.rand(1, 1, 10, 10, 10)
    assert torch.allclose(ops.lcc_map(im, im, 1), torch.zeros_like(im), atol=1e-6)
    assert torch.allclose(ops.lcc_map(im, 3.0 * im + 2.0, 1), torch.zeros_like(im), atol=2e-3)


def test_adam_rate_decay_matches_closed_form_first_step():
    p = torch.tensor([1.0, -2.0])
    opt = ops.AdamRateDecay([{'params': [p], 'lr': 0.2}], lr_decay=0.001)
    opt.step([torch.tensor([0.5, -0.25])])
    # first step of Adam moves every coordinate by lr * sign(grad) (up to eps)
    assert torch.allclose(p, torch.tensor([0.8, -1.8]), atol=1e-6)
    opt.step([torch.tensor([0.5, -0.25])])
    clr = 0.2 / (1 + 1 * 0.001)
    assert torch.allclose(p, torch.tensor([0.8 - clr, -1.8 + clr]), atol=1e-5)
