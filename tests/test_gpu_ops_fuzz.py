"""Seeded random shapes / parameters for the STATELESS operators of the C ABI (the entry points a maintainer binds one by one:
include/irsgmcmc.h) against the oracle -- their launch paths differ from the fused transition's (planar layouts, variants chosen
from a bound the operator measures itself, staged any-radius adjoint).  Gradient fields: all but the cell-face elements within
tolerance (tests/test_gpu_fuzz.py: check_but_flips)."""
import os
import random

import pytest
import torch

from ir_sgmcmc_amd import ops as G
from oracle import ops as O
from tests._report import GRAD_RTOL
from tests.conftest import fuzz_seeds
from tests.test_gpu_fuzz import check_but_flips
from tests.test_gpu_ops import dev, maxdiff, smooth_field

pytestmark = pytest.mark.gpu
SEEDS = fuzz_seeds('IRS_OPS_FUZZ_SEEDS', range(5), range(10))   # (IRS_LONG=1: ten draws; IRS_OPS_FUZZ_SEEDS=100: a longer hunt)


def _dims(r, lo=7, hi=46):
    return tuple(r.randint(lo, hi) for _ in range(3))


@pytest.mark.parametrize('seed', SEEDS)
def test_random_svf_exp_forward_and_backward(seed):
    r = random.Random(300 + seed)
    dims, C, steps_n = _dims(r), r.choice([1, 2, 3]), r.choice([1, 2, 5, 8, 12])
    amp = r.choice([0.0, 0.5, 2.0, 6.0, 12.0]) * min(1.0, 0.15 * min(dims))   # (no fold of a tiny volume)
    v = smooth_field(C, dims, amp, seed).requires_grad_(True)
    g_last = smooth_field(C, dims, 1.0, 1000 + seed)
    t_ref, d_ref, steps_ref = O.svf_exp(v, steps_n, keep_steps=True)
    t, d, steps = G.svf_exp_fwd(dev(v.detach()), steps_n)
    name = f'ops_fuzz/exp_{seed}_{"x".join(map(str, dims))}_C{C}_n{steps_n}_amp{amp:.2g}'
    dmax = max(1.0, float(d_ref.abs().max()))
    tol_d = 1e-4 * max(1.0, dmax / 4.0)   # north star 1e-4 voxels; a field of many voxels carries more fp32 rounding than that
    assert maxdiff(d, d_ref) < tol_d, name
    assert maxdiff(t, t_ref) < tol_d * 2.0 / (min(dims) - 1) + 2e-7, name   # (the same deviation in the [-1, 1] units of the grid)
    gv_ref, = torch.autograd.grad(steps_ref[-1], v, g_last)
    gv = G.svf_exp_bwd(dev(v.detach()), steps, dev(g_last))
    assert torch.equal(gv, G.svf_exp_bwd(dev(v.detach()), steps, dev(g_last))), name   # every variant is deterministic
    check_but_flips(name, 'grad_v (rel to max)', gv, gv_ref, GRAD_RTOL, float(gv_ref.abs().max()))


@pytest.mark.parametrize('seed', SEEDS)
def test_random_ffd_warp_lcc_energy(seed):
    r = random.Random(900 + seed)
    g = torch.Generator().manual_seed(seed)
    # cubic B-spline FFD: up-sampling and its adjoint
    dims, cps = _dims(r, 8, 40), tuple(r.choice([2, 3, 4, 5]) for _ in range(3))
    C = r.choice([1, 2])
    Gd = O.control_grid_size(dims, cps)
    v = torch.randn(C, 3, *Gd, generator=g).requires_grad_(True)
    ref = O.ffd_upsample(v, dims, cps)
    name = f'ops_fuzz/misc_{seed}_{"x".join(map(str, dims))}_cps{"x".join(map(str, cps))}'
    assert maxdiff(G.ffd_up(dev(v.detach()), dims, cps), ref) < 2e-6 * max(1.0, float(ref.abs().max())), name
    gd = torch.randn(C, 3, *dims, generator=g)
    gref, = torch.autograd.grad(ref, v, gd)
    assert maxdiff(G.ffd_adjoint(dev(gd), cps), gref) < 2e-5 * max(1.0, float(gref.abs().max())), name
    # warp of an image by id + d (+ jitter) and its adjoint
    alpha = r.choice([0.0, 0.1, 0.4])
    im = torch.rand(1, 1, *dims, generator=g)
    d_last = (smooth_field(C, dims, r.choice([0.5, 4.0, 10.0]), seed + 5) * (2.0 / (min(dims) - 1))).requires_grad_(True)
    unif = torch.rand(C, 3, *dims, generator=g) if alpha > 0 else None
    grid = O.identity_grid(dims).permute(0, 4, 1, 2, 3) + d_last
    if alpha > 0:
        grid = O.jitter_grid(grid, alpha, unif)
    wref = O.warp_trilinear(im.expand(C, -1, -1, -1, -1), grid)
    out = G.warp_displacement(dev(im), dev(d_last.detach()), dev(unif) if alpha > 0 else None, alpha)
    assert maxdiff(out, wref) < 6e-6, name   # (values in [0, 1]; eight products of three weights each)
    gw = torch.randn(C, 1, *dims, generator=g)
    gref, = torch.autograd.grad(wref, d_last, gw)
    gout = G.warp_displacement_bwd(dev(im), dev(d_last.detach()), dev(gw), dev(unif) if alpha > 0 else None, alpha)
    check_but_flips(name, 'warp adjoint (rel to max)', gout, gref, 2e-5, float(gref.abs().max()))
    # LCC map forward / adjoint (the window must fit: s < half the volume)
    s = r.choice([1, 2])
    if min(dims) > 4 * s:
        Fi = torch.rand(1, 1, *dims, generator=g)
        M = torch.rand(C, 1, *dims, generator=g).requires_grad_(True)
        fhat_ref, _, _ = O.lcc_normalise(Fi, s)
        fhat, _ = G.lcc_normalise(dev(Fi), s, want_sigma=True)
        assert maxdiff(fhat, fhat_ref) < 5e-5, name
        z_ref = O.lcc_map(Fi.expand(C, -1, -1, -1, -1), M, s)
        z, sigm = G.lcc_map_fwd(fhat, dev(M.detach()), s)
        assert maxdiff(z, z_ref) < 1e-4, name
        gz = torch.randn(C, 1, *dims, generator=g)
        gref, = torch.autograd.grad(z_ref, M, gz)
        assert maxdiff(G.lcc_map_bwd(fhat, z, sigm, dev(gz), s), gref) < 2e-4 * float(gref.abs().max()), name
    # regulariser energy and the difference operator
    vv = smooth_field(C, dims, 3.0, seed + 9)
    assert torch.allclose(G.reg_energy(dev(vv)).cpu(), O.reg_energy(vv).double(), rtol=2e-6), name
    assert maxdiff(G.gradient_operator(dev(vv)), O.forward_differences(vv)) < 2e-6, name
