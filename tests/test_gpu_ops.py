"""Operator-level parity of the HIP kernels (through the C ABI) against the CPU oracle.  GPU only."""
import math

import numpy as np

import pytest
import torch

from ir_sgmcmc_amd import ops as G
from oracle import ops as O
from tests._report import GRAD_RTOL, check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(t):
    return t.to(DEV).contiguous()


def maxdiff(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def smooth_field(C, dims, amp, seed):
    g = torch.Generator().manual_seed(seed)
    return O.separable_conv3d_replicate(amp * torch.randn(C, 3, *dims, generator=g), O.sobolev_kernel_1d(3, 0.5)).contiguous()


# (5, 13, 3): narrower than the kernel's 2 s + 1 taps on two axes -- replicate padding folds a tap back more than once; the
# reference's F.pad(mode='replicate') takes any size, and so does a kernel that reads through clamped coordinates
@pytest.mark.parametrize('dims', [(16, 16, 16), (12, 20, 28), (5, 13, 3)])
@pytest.mark.parametrize('s', [1, 3, 4])
def test_perturb_smooth(dims, s):
    g = torch.Generator().manual_seed(0)
    v = torch.randn(2, 3, *dims, generator=g)
    sigma = torch.rand(2, 3, *dims, generator=g) + 0.5
    eps = torch.randn(2, 3, *dims, generator=g)
    k = G.sobolev_kernel_1d(s, 0.5)
    ref = O.separable_conv3d_replicate(O.langevin_perturb(v, sigma, 0.4, eps), k)
    out = G.perturb_smooth(dev(v), k, dev(sigma), dev(eps), tau=0.4)
    assert maxdiff(out, ref) < 2e-6
    # smoothing only, all-ones 3-tap kernel -> 27 (reference tests/test_utils.py:101-151)
    ones = torch.zeros(2, 3, 16, 16, 16)
    ones[0, 1], ones[1, 2] = 1.0, 1.0
    out = G.perturb_smooth(dev(ones), [1.0, 1.0, 1.0])
    assert maxdiff(out, 27.0 * ones) < 1e-4


@pytest.mark.parametrize('s', [1, 2, 3])
def test_sobolev_tile_shapes(s):
    """The Sobolev kernel picks a 64x32 or a 32x16 column tile from the volume size; both must agree with the oracle (and
    with each other bit for bit -- same tap order) on a ragged volume that leaves every tile edge partially filled."""
    dims = (37, 45, 70)
    g = torch.Generator().manual_seed(5)
    v = torch.randn(1, 3, *dims, generator=g)
    k = G.sobolev_kernel_1d(s, 0.5)
    ref = O.separable_conv3d_replicate(v, k)
    outs = {}
    from ir_sgmcmc_amd._lib import option_set
    try:
        for shape, code in (('small', 1), ('big', 2)):
            option_set('sobolev_tile', code)   # process-wide switch of the stateless operator (irs_option_set)
            outs[shape] = G.perturb_smooth(dev(v), k)
            assert maxdiff(outs[shape], ref) < 2e-6
    finally:
        option_set('sobolev_tile', 0)
    assert torch.equal(outs['small'], outs['big'])


@pytest.mark.parametrize('dims', [(37, 45, 70), (16, 16, 16), (64, 64, 64)])
@pytest.mark.parametrize('s', [1, 3])
def test_noise_generated_while_staging_is_the_same_noise(dims, s):
    """SGLD.forward fused into the smoothing kernel (the perturbed velocity is never written) against the two-kernel form: the
    same Philox counters -> the same normals -> bit-identical smoothed fields, for in-kernel noise, injected eps, and a sigma
    field; ragged volumes leave every tile edge partially filled and make the replicate padding (noise included) matter."""
    from ir_sgmcmc_amd._lib import option_set
    g = torch.Generator().manual_seed(9)
    v = dev(torch.randn(2, 3, *dims, generator=g))
    sigma = dev(torch.rand(2, 3, *dims, generator=g) + 0.5)
    eps = dev(torch.randn(2, 3, *dims, generator=g))
    k = G.sobolev_kernel_1d(s, 0.5)
    cases = [dict(tau=0.4, seed=5, iteration=7), dict(tau=0.4, seed=5, iteration=7, sigma=sigma), dict(tau=0.4, eps=eps),
             dict(tau=0.4, eps=eps, sigma=sigma)]
    outs = {}
    try:
        # ps_rows: the fused kernel's tile, 32 x 32 (three columns per thread) or 32 x 16 (two: what a sigma field with generated noise
        # takes by default -- the 32-row form of that variant needs 149 VGPRs); 0 = that default
        for fuse, rows in ((0, 0), (1, 0), (1, 16), (1, 32)):
            option_set('fuse_noise', fuse)
            option_set('ps_rows', rows)
            outs[fuse, rows] = [G.perturb_smooth(v, k, **kw) for kw in cases]
    finally:
        option_set('fuse_noise', 1)
        option_set('ps_rows', 0)
    for key in ((1, 0), (1, 16), (1, 32)):
        for a, b in zip(outs[0, 0], outs[key]):
            assert torch.equal(a, b), key
    assert not torch.equal(outs[1, 0][0], outs[1, 0][2])   # (different noise sources do differ)


def test_philox_noise_statistics():
    v = torch.zeros(1, 3, 64, 64, 64, device=DEV)
    a = G.perturb_smooth(v, None, tau=0.5, seed=7, iteration=3)   # sqrt(2 tau) = 1 -> N(0,1)
    b = G.perturb_smooth(v, None, tau=0.5, seed=7, iteration=3)
    c = G.perturb_smooth(v, None, tau=0.5, seed=7, iteration=4)
    assert torch.equal(a, b) and not torch.equal(a, c)
    x = a.double().flatten()
    n = x.numel()
    assert abs(float(x.mean())) < 5.0 / math.sqrt(n)
    assert abs(float(x.var()) - 1.0) < 5.0 * math.sqrt(2.0 / n)
    assert abs(float((x ** 4).mean()) - 3.0) < 0.05
    # channels / neighbours uncorrelated
    assert abs(float((a[0, 0] * a[0, 1]).double().mean())) < 5.0 / math.sqrt(n / 3)
    assert abs(float((a[0, 0, :, :, 1:] * a[0, 0, :, :, :-1]).double().mean())) < 5.0 / math.sqrt(n / 3)


@pytest.mark.parametrize('dims,amp', [((16, 16, 16), 2.0), ((16, 16, 16), 25.0), ((10, 14, 22), 6.0), ((32, 32, 32), 5.0)])
def test_svf_exp_forward(dims, amp):
    v = smooth_field(2, dims, amp, 1)
    t_ref, d_ref, steps_ref = O.svf_exp(v, 12, keep_steps=True)
    t, d, steps = G.svf_exp_fwd(dev(v), 12)
    assert maxdiff(steps[0], steps_ref[1]) < 1e-9 + 1e-6 * float(steps_ref[1].abs().max())
    if amp <= 10.0:
        assert maxdiff(steps[-1], steps_ref[-1]) < 5e-6 * max(1.0, float(steps_ref[-1].abs().max()))
        assert maxdiff(d, d_ref) < 1e-4   # north-star tolerance on the displacement (voxels)
        assert maxdiff(t, t_ref) < 1e-5
    # amp = 25 folds the grid (tens of voxels, |grad d| >> 1): rounding differences are amplified at every step
    assert maxdiff(d, d_ref) < 2e-5 * max(1.0, float(d_ref.abs().max()))


@pytest.mark.parametrize('dims,amp', [((16, 16, 16), 2.0), ((16, 16, 16), 25.0), ((10, 14, 22), 6.0)])
@pytest.mark.parametrize('upstream', ['smooth', 'white'])
def test_svf_exp_backward(dims, amp, upstream):
    v = smooth_field(1, dims, amp, 2).requires_grad_(True)
    g = torch.Generator().manual_seed(3)
    # a white-noise upstream gradient turns every 1e-6-voxel difference in a sampling position into an O(1e-6)
    # relative difference per step (neighbouring gradient values are unrelated), hence the looser bound
    g_last = torch.randn(1, 3, *dims, generator=g) if upstream == 'white' else smooth_field(1, dims, 1.0, 33)
    _, _, steps_ref = O.svf_exp(v, 12, keep_steps=True)
    gv_ref, = torch.autograd.grad(steps_ref[-1], v, g_last)
    _, _, steps = G.svf_exp_fwd(dev(v.detach()), 12, want_outputs=False)
    gv = G.svf_exp_bwd(dev(v.detach()), steps, dev(g_last))
    tol = GRAD_RTOL
    assert maxdiff(gv, gv_ref) < tol * float(gv_ref.abs().max())


def test_svf_exp_backward_large_smooth_displacement():
    """8-voxel smooth field on several tiles: the last adjoint steps run in the any-radius kernel, whose source boxes are
    shrunk with the coarse displacement extrema.  Same result with the plain boxes (fixed-point accumulation: only the
    per-tile scale differs), and -- except at the few voxels whose sampling position sits within rounding of a cell boundary,
    where the CPU and the GPU forward pass pick different cells -- with autograd through the oracle."""
    dims, amp = (40, 36, 44), 9.0
    v = smooth_field(1, dims, amp, 2).requires_grad_(True)
    g_last = smooth_field(1, dims, 1.0, 33)
    _, _, steps_ref = O.svf_exp(v, 12, keep_steps=True)
    gv_ref, = torch.autograd.grad(steps_ref[-1], v, g_last)
    assert float(steps_ref[-2].abs().max()) * 0.5 * (min(dims) - 1) > 2.0   # d_11 beyond the radius-2 gather
    _, _, steps = G.svf_exp_fwd(dev(v.detach()), 12, want_outputs=False)
    gv = G.svf_exp_bwd(dev(v.detach()), steps, dev(g_last))
    assert torch.equal(gv, G.svf_exp_bwd(dev(v.detach()), steps, dev(g_last)))   # integer accumulation: order-independent
    scale = float(gv_ref.abs().max())
    from ir_sgmcmc_amd._lib import option_set
    try:
        option_set('coarse_box', 0)
        assert maxdiff(gv, G.svf_exp_bwd(dev(v.detach()), steps, dev(g_last))) < 2e-6 * scale
    finally:
        option_set('coarse_box', 1)
    bad = ((gv.cpu() - gv_ref).abs() > GRAD_RTOL * scale).sum()
    assert int(bad) <= 1e-3 * gv_ref.numel()


def test_any_radius_adjoint_walks_huge_source_boxes_with_plain_division():
    """The any-radius adjoint walks the source box of a tile in one flat order and splits the flat index with a multiply-high
    division that is exact while (sources in the box) x (box extent in x, y) < 2^32; beyond that it divides.  A folding
    100-voxel field at 128^3 without the coarse-grid refinement gives every tile (nearly) the whole volume as its box
    (2e6 sources x 16384), with it the boxes are small: the two must agree (only the per-tile fixed-point scale differs)."""
    dims, amp = (128, 128, 128), 100.0
    v = smooth_field(1, dims, amp, 2)
    g_last = smooth_field(1, dims, 1.0, 33)
    _, _, steps = G.svf_exp_fwd(dev(v), 12, want_outputs=False)
    assert float(steps[-2].abs().max()) * 0.5 * (min(dims) - 1) > 46.0   # B0 = tile +- 47 or more: >= 126 x 102 x 102 sources
    gv = G.svf_exp_bwd(dev(v), steps, dev(g_last))
    from ir_sgmcmc_amd._lib import option_set
    try:
        option_set('coarse_box', 0)
        gv_plain = G.svf_exp_bwd(dev(v), steps, dev(g_last))
    finally:
        option_set('coarse_box', 1)
    assert bool(torch.isfinite(gv).all())
    assert maxdiff(gv, gv_plain) < 2e-6 * float(gv_plain.abs().max())


@pytest.mark.parametrize('N,cps', [(16, 4), (16, 2), (17, 4), (20, 3)])
def test_ffd_up_and_adjoint(N, cps):
    dims, c = (N,) * 3, (cps,) * 3
    Gd = O.control_grid_size(dims, c)
    g = torch.Generator().manual_seed(4)
    v = torch.randn(2, 3, *Gd, generator=g).requires_grad_(True)
    ref = O.ffd_upsample(v, dims, c)
    out = G.ffd_up(dev(v.detach()), dims, c)
    assert maxdiff(out, ref) < 2e-6
    gd = torch.randn(2, 3, *dims, generator=g)
    gref, = torch.autograd.grad(ref, v, gd)
    gout = G.ffd_adjoint(dev(gd), c)
    assert maxdiff(gout, gref) < 1e-5


@pytest.mark.parametrize('alpha', [0.0, 0.1])
def test_warp_forward_backward(alpha):
    dims = (14, 18, 22)
    g = torch.Generator().manual_seed(5)
    im = torch.rand(1, 1, *dims, generator=g)
    v = smooth_field(2, dims, 8.0, 6)
    t_ref, _, steps = O.svf_exp(v, 12, keep_steps=True)
    d_last = steps[-1].clone().requires_grad_(True)
    unif = torch.rand(2, 3, *dims, generator=g) if alpha > 0 else None
    grid = O.identity_grid(dims).permute(0, 4, 1, 2, 3) + d_last
    if alpha > 0:
        grid = O.jitter_grid(grid, alpha, unif)
    ref = O.warp_trilinear(im.expand(2, -1, -1, -1, -1), grid)
    out = G.warp_displacement(dev(im), dev(d_last.detach()), dev(unif) if alpha > 0 else None, alpha)
    assert maxdiff(out, ref) < 2e-6
    gw = torch.randn(2, 1, *dims, generator=g)
    gref, = torch.autograd.grad(ref, d_last, gw)
    gout = G.warp_displacement_bwd(dev(im), dev(d_last.detach()), dev(gw), dev(unif) if alpha > 0 else None, alpha)
    assert maxdiff(gout, gref) < 2e-5 * float(gref.abs().max())
    # public RegistrationModule path on an explicit transformation, float + nearest
    out2 = G.warp(dev(im), dev(t_ref))
    assert maxdiff(out2, O.warp_trilinear(im.expand(2, -1, -1, -1, -1), t_ref)) < 2e-6
    seg = (torch.rand(1, 1, *dims, generator=g) * 50).to(torch.int16)
    assert torch.equal(G.warp(dev(seg), dev(t_ref)).cpu(), O.warp_nearest(seg.expand(2, -1, -1, -1, -1), t_ref))
    msk = torch.rand(1, 1, *dims, generator=g) > 0.5
    assert torch.equal(G.warp(dev(msk), dev(t_ref)).cpu(), O.warp_nearest(msk.expand(2, -1, -1, -1, -1), t_ref))
    with pytest.raises(NotImplementedError):
        G.warp(dev(im).double(), dev(t_ref))


@pytest.mark.parametrize('dims', [(16, 16, 16), (12, 20, 40), (9, 8, 33)])
@pytest.mark.parametrize('s', [1, 2])
def test_lcc_forward_backward(dims, s):
    g = torch.Generator().manual_seed(7)
    F = torch.rand(1, 1, *dims, generator=g)
    M = (torch.rand(2, 1, *dims, generator=g)).requires_grad_(True)
    fhat_ref, _, _ = O.lcc_normalise(F, s)
    fhat, sig = G.lcc_normalise(dev(F), s, want_sigma=True)
    assert maxdiff(fhat, fhat_ref) < 2e-5
    z_ref = O.lcc_map(F.expand(2, -1, -1, -1, -1), M, s)
    z, sigm = G.lcc_map_fwd(fhat, dev(M.detach()), s)
    assert maxdiff(z, z_ref) < 5e-5
    gz = torch.randn(2, 1, *dims, generator=g)
    gref, = torch.autograd.grad(z_ref, M, gz)
    gout = G.lcc_map_bwd(fhat, z, sigm, dev(gz), s)
    assert maxdiff(gout, gref) < 1e-4 * float(gref.abs().max())


def test_reg_energy_gradient_operator_and_det_j():
    dims = (10, 12, 14)
    v = smooth_field(2, dims, 3.0, 8)
    assert torch.allclose(G.reg_energy(dev(v)).cpu(), O.reg_energy(v).double(), rtol=1e-6)
    assert maxdiff(G.gradient_operator(dev(v)), O.forward_differences(v)) < 1e-6
    # reference tests/test_diff.py: uniform field -> 0, identity -> log det J = 0, 2x stretch -> log 8
    u = torch.zeros(1, 3, 16, 16, 16)
    u[0, 0], u[0, 1], u[0, 2] = 5.0, 4.0, 2.0
    assert float(G.gradient_operator(dev(u)).abs().max()) < 1e-4
    ident = O.identity_grid((16, 16, 16)).permute(0, 4, 1, 2, 3).contiguous()
    cnt, ld = G.log_det_jacobian(dev(ident))
    assert int(cnt[0]) == 0 and float(ld.abs().max()) < 1e-4
    cnt, ld = G.log_det_jacobian(dev(ident + (ident + 1.0)))
    assert int(cnt[0]) == 0 and float((ld - math.log(8.0)).abs().max()) < 1e-4
    t, _ = O.svf_exp(smooth_field(1, (16, 16, 16), 60.0, 9))  # folds somewhere
    cnt, ld = G.log_det_jacobian(dev(t))
    ld_ref = O.det_jacobian(O.forward_differences(t, transformation=True)).log()
    assert int(cnt[0]) == int(torch.isnan(ld_ref).sum())


def test_utility_operators_match_reference_fixture():
    """separable_conv_3D (2-argument branch, utils/util.py:362-392; reference tests/test_utils.py:117-133 restated on a
    non-constant field and asymmetric / per-channel kernels), calc_norm (utils/util.py:215-225) and calc_DSC_GPU
    (utils/util.py:123-148) against outputs of the imported reference (tests/golden/make_golden_utils.py)."""
    import os
    from ir_sgmcmc_amd.utils import util as U
    from tests._golden import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'utils_ops.npz'))
    v = torch.from_numpy(z['field']).to(DEV)
    for tag in ('k3', 'k5', 'sobolev', 'kc'):
        k = torch.from_numpy(z[f'{tag}_kernel'])
        S = (k if k.dim() == 2 else torch.stack((k, k, k), 0)).unsqueeze(1)
        p = (S.shape[-1] - 1) // 2
        check('utils/separable_conv_3D', f'2-arg {tag}', U.separable_conv_3D(v, S.to(DEV), p), z[f'{tag}_out_2arg'], 1e-6)
        if k.dim() == 1:  # the 4-argument call of the same taps gives the same field
            S4 = S.to(DEV)
            out4 = U.separable_conv_3D(v, S4.unsqueeze(2).unsqueeze(2), S4.unsqueeze(2).unsqueeze(4), S4.unsqueeze(3).unsqueeze(4), (p,) * 6)
            check('utils/separable_conv_3D', f'4-arg {tag}', out4, z[f'{tag}_out_2arg'], 1e-6)
    # the reference's own known answer: all-ones 3-tap kernels on a constant field give 27 (tests/test_utils.py:117-133)
    ones = torch.zeros(2, 3, 16, 16, 16, device=DEV)
    ones[0, 1], ones[1, 2] = 1.0, 1.0
    out = U.separable_conv_3D(ones, torch.ones(3, 1, 3, device=DEV), 1)
    assert torch.allclose(out[0, 1], torch.full_like(out[0, 1], 27.0)) and float(out[0, 0].abs().max()) == 0.0
    check('utils/calc_norm', 'norm', U.calc_norm(v), z['norm'], 1e-6)
    labels = {str(i): int(l) for i, l in enumerate(z['dsc_labels'])}
    dsc = U.calc_DSC_GPU(3, torch.from_numpy(z['seg_fixed']).to(DEV), torch.from_numpy(z['seg_moving']).to(DEV), labels)
    assert np.array_equal(np.isnan(dsc), np.isnan(z['dsc'])) and np.allclose(np.nan_to_num(dsc), np.nan_to_num(z['dsc']), atol=1e-6)
