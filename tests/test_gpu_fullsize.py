"""Parity at the benchmark's FULL size (256^3, BASELINE.json configs[1]) through size-independent properties.

The CPU oracle needs minutes per operator at this size, so these tests do not compare against it; they check properties
that hold for any size and that a wrong index, halo, segment boundary or tile edge at 256^3 would break:
constant / translation / identity invariants, linearity, the adjoint identity <J u, w> = <u, J^T w>, affine invariance of
LCC, closed-form regulariser energy, bit-reproducibility of the whole transition."""
import math

import pytest
import torch

from ir_sgmcmc_amd import ops as G
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
N = 256


def rnd(*shape, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, generator=g, device=DEV)


def smooth(C, amp, seed):
    v = G.perturb_smooth(rnd(C, 3, N, N, N, seed=seed), G.sobolev_kernel_1d(3, 0.5))
    return v * (amp / float(v.abs().max()))


def test_sobolev_constant_linearity_and_segment_seams():
    k = G.sobolev_kernel_1d(3, 0.5)
    c = torch.full((1, 3, N, N, N), 0.37, device=DEV)
    assert float((G.perturb_smooth(c, k) - 0.37).abs().max()) < 1e-6          # normalised kernel, replicate padding
    a, b = rnd(1, 3, N, N, N, seed=1), rnd(1, 3, N, N, N, seed=2)
    lhs = G.perturb_smooth(2.5 * a - b, k)
    rhs = 2.5 * G.perturb_smooth(a, k) - G.perturb_smooth(b, k)
    assert float((lhs - rhs).abs().max()) < 2e-5
    # a field that depends on z only is filtered identically in every column: tile and segment seams leave no trace
    z = torch.sin(torch.arange(N, device=DEV, dtype=torch.float32) * 0.21).view(1, 1, N, 1, 1).expand(1, 3, N, N, N).contiguous()
    out = G.perturb_smooth(z, k)
    assert float((out - out[:, :, :, :1, :1]).abs().max()) == 0.0


def test_exp_identity_translation_and_dmax_variants():
    zero = torch.zeros(1, 3, N, N, N, device=DEV)
    t, d, _ = G.svf_exp_fwd(zero, 12)
    assert float(d.abs().max()) == 0.0
    # constant velocity = pure translation: the displacement equals the velocity away from the border (exactly representable
    # steps: 1.5 voxels / 4096 doubles twelve times); large enough for the last steps to leave the radius-1 kernels
    v = torch.zeros(1, 3, N, N, N, device=DEV)
    v[:, 0], v[:, 1], v[:, 2] = 1.5, -0.75, 2.25
    _, d, _ = G.svf_exp_fwd(v, 12)
    m = 8
    inner = d[:, :, m:-m, m:-m, m:-m]
    assert float((inner[:, 0] - 1.5).abs().max()) < 2e-4
    assert float((inner[:, 1] + 0.75).abs().max()) < 2e-4
    assert float((inner[:, 2] - 2.25).abs().max()) < 2e-4


def test_exp_adjoint_identity():
    """<d exp(v)[u], w> == <u, exp_bwd(v; w)> with the directional derivative by central differences.  exp is only
    piecewise smooth (trilinear kinks) and evaluated in fp32, so the difference quotient itself wanders by +-0.5 % with the
    step (tools/adjoint_probe.py, same at 64^3 where the gradient is pinned against autograd): the bound is 1 %."""
    v = smooth(1, 3.0, 11)
    u, w = smooth(1, 1.0, 12), smooth(1, 1.0, 13)
    _, _, steps = G.svf_exp_fwd(v, 12, want_outputs=False)
    gv = G.svf_exp_bwd(v, steps, w)
    rhs = float((u.double() * gv.double()).sum())
    eps = 0.0125
    _, _, sp = G.svf_exp_fwd(v + eps * u, 12, want_outputs=False)
    _, _, sm = G.svf_exp_fwd(v - eps * u, 12, want_outputs=False)
    lhs = float((((sp[-1] - sm[-1]).double() / (2 * eps)) * w.double()).sum())
    assert abs(lhs - rhs) < 1e-2 * max(abs(lhs), abs(rhs))


def test_exp_bwd_is_linear_in_the_upstream_gradient():
    v = smooth(1, 2.0, 21)
    _, _, steps = G.svf_exp_fwd(v, 12, want_outputs=False)
    a, b = rnd(1, 3, N, N, N, seed=22), rnd(1, 3, N, N, N, seed=23)
    lhs = G.svf_exp_bwd(v, steps, 0.5 * a + 2.0 * b)
    rhs = 0.5 * G.svf_exp_bwd(v, steps, a) + 2.0 * G.svf_exp_bwd(v, steps, b)
    assert float((lhs - rhs).abs().max()) < 1e-4 * float(rhs.abs().max())


def test_warp_identity_and_lcc_affine_invariance():
    im = rnd(1, 1, N, N, N, seed=31)
    zero = torch.zeros(1, 3, N, N, N, device=DEV)
    # identity grid: linspace(-1, 1, N) mapped back to voxels is an integer only up to ulp(N) = 1.5e-5 (as in ATen), and the
    # white-noise neighbours differ by O(1)
    assert float((G.warp_displacement(im, zero) - im).abs().max()) < 3e-4
    for s in (1, 2):
        z1 = G.lcc_normalise(im, s)
        z2 = G.lcc_normalise(3.0 * im + 0.25, s)
        assert float((z1 - z2).abs().max()) < 2e-4
        flat = G.lcc_normalise(torch.full_like(im, 0.7), s)
        assert float(flat.abs().max()) < 1e-2                               # (I - u) = 0 up to rounding, sigma = 1e-5
    # the adjoint of the LCC map against a central difference of the map itself
    fhat = G.lcc_normalise(rnd(1, 1, N, N, N, seed=32), 1)
    u, w = rnd(1, 1, N, N, N, seed=33), rnd(1, 1, N, N, N, seed=34)
    z, sig = G.lcc_map_fwd(fhat, im, 1)
    gm = G.lcc_map_bwd(fhat, z, sig, w, 1)
    eps = 1e-2
    zp, _ = G.lcc_map_fwd(fhat, im + eps * u, 1)
    zm, _ = G.lcc_map_fwd(fhat, im - eps * u, 1)
    lhs = float((((zp - zm).double() / (2 * eps)) * w.double()).sum())
    rhs = float((u.double() * gm.double()).sum())
    assert abs(lhs - rhs) < 5e-3 * max(abs(lhs), abs(rhs))


def test_reg_energy_closed_form():
    # v_x = a x, v_y = b y, v_z = c z: every forward difference is a constant and the replicated last one counts twice,
    # so each axis contributes n differences per line: y = (a^2 + b^2 + c^2) N^3
    ar = torch.arange(N, device=DEV, dtype=torch.float32)
    v = torch.zeros(2, 3, N, N, N, device=DEV)
    a, b, c = 0.5, -0.25, 0.125
    v[:, 0] = a * ar.view(1, 1, 1, N)
    v[:, 1] = b * ar.view(1, 1, N, 1)
    v[:, 2] = c * ar.view(1, N, 1, 1)
    y = G.reg_energy(v).cpu()
    want = (a * a + b * b + c * c) * N ** 3
    assert torch.allclose(y, torch.full_like(y, want), rtol=1e-9)


def _pair():
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    f, m = synthetic_pair((N, N, N), seed=0)
    to = lambda d: {k: v.unsqueeze(0).to(DEV).contiguous() for k, v in d.items() if k != 'seg'}
    return to(f), to(m)


def test_full_transition_is_reproducible_and_sane():
    """Two engines, same seed: bit-identical chains (no atomics anywhere on the default path); another seed: another chain."""
    fixed, moving = _pair()
    res = []
    for seed in (3, 3, 4):
        eng = TransitionEngine(EngineConfig(dims=(N, N, N), seed=seed), DEV)
        fd, md = eng.prepare(fixed, moving)
        eng.gmm_init(fd, md)
        v = torch.zeros(1, 3, N, N, N, device=DEV)
        for _ in range(3):
            eng.transition(fd, md, v)
        sc = eng.scalars()
        assert eng.state().iteration == 3 and bool(torch.isfinite(v).all())
        assert all(math.isfinite(float(x)) for x in (sc['data_term'][0], sc['reg_term'][0], sc['alpha'][0]))
        assert 0.0 < float(sc['alpha'][0]) <= 1.0
        res.append((v.clone(), float(sc['data_term'][0])))
        del eng
    assert torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    assert not torch.equal(res[0][0], res[2][0])


def test_ssd_transition_reproducible_and_gradient_is_the_adjoint():
    """BASELINE.json config 4's loss at its own size (256^3 SSD + RegLoss_L2, one GPU; the z-slab run of the same
    configuration is compared with this engine in tests/test_gpu_slab.py).
    (1) Two engines, same seed: bit-identical chains.
    (2) The gradient the transition applies is the adjoint of its own forward map: with the regulariser switched off
        (w_reg -> 0) and no noise, <dL/dv, u> equals the central difference of the data term along u, L(v) = the SSD
        data term the engine reports."""
    fixed, moving = _pair()
    res = []
    for seed in (3, 3):
        eng = TransitionEngine(EngineConfig(dims=(N, N, N), data_loss='SSD', virtual_decimation=False, reg_loss='RegLoss_L2', w_reg=1.4,
                                            seed=seed), DEV)
        fd, md = eng.prepare(fixed, moving)
        eng.gmm_init(fd, md)
        v = smooth(1, 2.0, 31)
        for _ in range(3):
            eng.transition(fd, md, v)
        sc = eng.scalars()
        assert eng.state().iteration == 3 and bool(torch.isfinite(v).all()) and float(sc['alpha'][0]) == 1.0
        res.append((v.clone(), float(sc['data_term'][0]), float(sc['reg_term'][0])))
        del eng
    assert torch.equal(res[0][0], res[1][0]) and res[0][1:] == res[1][1:]

    # Noise-free images for the difference quotient: with the pair's white noise (0.02 per voxel) the SSD gradient at this
    # size is dominated by the noise's own piecewise-constant slopes, <g, u> becomes a random walk over 5e7 terms, and the
    # ~9 % of samples that cross a cell face between v - eps u and v + eps u perturb the quotient by more than the coherent part
    # (tools/adjoint_probe_ssd.py: 64^3 and 128^3 agree to 1 % either way, 256^3 to 0.5 % without the noise, 20 % with it).
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    f0, m0 = synthetic_pair((N, N, N), seed=0, noise=0.0)
    fixed = {k: x.unsqueeze(0).to(DEV).contiguous() for k, x in f0.items() if k != 'seg'}
    moving = {k: x.unsqueeze(0).to(DEV).contiguous() for k, x in m0.items() if k != 'seg'}

    def data_term(v_in, want_grad=False):
        # no Langevin noise (lr -> 0 keeps the perturbation and the update negligible), no jitter, no Sobolev smoothing:
        # v_s = v, and grad_v is the data gradient plus a vanishing regulariser term
        eng = TransitionEngine(EngineConfig(dims=(N, N, N), data_loss='SSD', virtual_decimation=False, reg_loss='RegLoss_L2', w_reg=1e-12,
                                            sobolev_s=0, uniform_noise=0.0, lr=1e-30, seed=1), DEV)
        fd, md = eng.prepare(fixed, moving)
        eng.gmm_init(fd, md)
        vv = v_in.clone()
        g = torch.empty_like(vv) if want_grad else None
        eng.transition(fd, md, vv, outputs={'grad_v': g} if want_grad else None)
        d = float(eng.scalars()['data_term'][0])
        del eng
        return d, g

    v = smooth(1, 2.0, 41)
    u = smooth(1, 1.0, 42)
    _, g = data_term(v, True)
    rhs = float((g.double() * u.double()).sum())
    eps = 0.05
    lp, _ = data_term(v + eps * u)
    lm, _ = data_term(v - eps * u)
    lhs = (lp - lm) / (2 * eps)
    assert abs(lhs - rhs) < 1e-2 * max(abs(lhs), abs(rhs)), (lhs, rhs)


def test_config5_192_cubed_psgld_two_chains():
    """BASELINE.json config 5 at its own size: 192^3, configs/experiment1 semantics -- GMM / LCC s = 1 with virtual decimation,
    learnable RegLoss_LogNormal, Sobolev s = 3, jitter 0.1, TWO chains, and the pre-conditioned ("pSGLD") update with a sigma
    FIELD as the VI stage leaves it (sigma = exp(log_var / 2), here 0.5 with a smooth modulation).  Size-independent properties:
    bit-identical repeat runs; the two chains, started apart, share the mixture (updated serially, trainer.py:316-327) but
    stay distinct; with sigma = s everywhere, one transition from the same state moves v by s^2 times the sigma = 1 step's
    gradient part (SGLD.backward: grad * sigma^2, utils/functions.py:83-84)."""
    M = 192
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    f, m = synthetic_pair((M, M, M), seed=0)
    to = lambda d: {k: v.unsqueeze(0).to(DEV).contiguous() for k, v in d.items() if k != 'seg'}
    fixed, moving = to(f), to(m)
    cfg = lambda: EngineConfig(dims=(M, M, M), no_chains=2, reg_loss='RegLoss_LogNormal', reg_learnable=True, seed=9)
    g = torch.Generator(device=DEV).manual_seed(5)
    v0 = G.perturb_smooth(torch.randn(2, 3, M, M, M, generator=g, device=DEV), G.sobolev_kernel_1d(3, 0.5))
    t = torch.linspace(0.0, math.pi, M, device=DEV)
    sigma = (0.5 * (1.0 + 0.2 * torch.sin(t).view(M, 1, 1) * torch.sin(t).view(1, M, 1))).expand(2, 3, M, M, M).contiguous()
    res = []
    for _ in range(2):
        eng = TransitionEngine(cfg(), DEV)
        fd, md = eng.prepare(fixed, moving)
        eng.gmm_init(fd, md)
        v = v0.clone()
        for _ in range(3):
            eng.transition(fd, md, v, sigma)
        sc, st = eng.scalars(), eng.state()
        assert st.iteration == 3 and st.gmm_adam_step[0] == 25 + 3 * 2  # one mixture step per chain and transition
        assert bool(torch.isfinite(v).all()) and all(0.0 < a <= 1.0 for a in sc['alpha'])
        res.append((v.clone(), sc['data_term'], list(st.gmm_log_std)))
        del eng
    assert torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert float((res[0][0][0] - res[0][0][1]).abs().max()) > 1e-3  # two chains, two states
    # the pre-conditioner: no noise (eps = 0 injected), no jitter; v_new - v = -lr sigma^2 grad
    zero = torch.zeros(2, 3, M, M, M, device=DEV)
    steps = {}
    for s in (1.0, 0.5):
        eng = TransitionEngine(EngineConfig(dims=(M, M, M), no_chains=2, reg_loss='RegLoss_LogNormal', reg_learnable=True,
                                            uniform_noise=0.0, seed=9), DEV)
        fd, md = eng.prepare(fixed, moving)
        eng.gmm_init(fd, md)
        v = v0.clone()
        eng.transition(fd, md, v, torch.full_like(v0, s), zero)
        steps[s] = v - v0
        del eng
    ref = steps[1.0]
    assert float((steps[0.5] - 0.25 * ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-9
