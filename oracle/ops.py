"""CPU oracle for the SG-MCMC registration inner loop  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A torch-CPU (fp32) restatement of every function on the hot path of dgrzech/ir-sgmcmc
(`Trainer._SGLD_transition`, reference `trainer/trainer.py:291-356`).  Each function cites the
reference file:line it follows.  All line numbers are relative to /root/reference.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package -- and there only as the checker / CPU baseline.  The product package
(`ir_sgmcmc_amd`) never imports it and has no CPU fallback.

Pinning: validated stage by stage against the *imported* reference by
`tests/golden/make_golden.py` (run in the build container, where /root/reference exists);
the outputs of the reference are committed as fixtures under `tests/golden/*.npz` and
`tests/test_oracle_golden.py` re-checks this oracle against them everywhere (no reference needed).

The third-party arithmetic the reference leans on (`F.grid_sample`, `F.conv3d`,
`F.conv_transpose1d`, `torch.logsumexp`) lives in PyTorch; the oracle calls the same ATen ops so
that its numerics are op-for-op those of the reference's CPU path.  Explicit (loop-free but
autograd-free) restatements of the trilinear sampler and of its two adjoints are given too
(`trilinear_sample_explicit`, `trilinear_backward_explicit`): they document the exact arithmetic
the HIP kernels implement and are checked against ATen in `tests/test_oracle_ops.py`.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)

# --------------------------------------------------------------------------------------------
# a1  SGLD noise                                      utils/functions.py:76-84, utils/util.py:48-58
# --------------------------------------------------------------------------------------------


def langevin_perturb(v, sigma, tau, eps):
    """v + sqrt(2 tau) * sigma * eps  (reference draws eps = randn_like(sigma), utils/util.py:56-58)."""
    return v + math.sqrt(2.0 * tau) * sigma * eps


class _SGLD(torch.autograd.Function):
    """forward: Langevin perturbation; backward: grad * sigma^2 (utils/functions.py:76-84)."""

    @staticmethod
    def forward(ctx, v, sigma, tau, eps):
        ctx.save_for_backward(sigma)
        return langevin_perturb(v, sigma, tau, eps)

    @staticmethod
    def backward(ctx, g):
        sigma, = ctx.saved_tensors
        return g * sigma ** 2, None, None, None


# --------------------------------------------------------------------------------------------
# a2  Sobolev smoothing                     utils/functions.py:24-49,98-109; utils/util.py:394-404
# --------------------------------------------------------------------------------------------


def sobolev_kernel_1d(s, lam):
    """Normalised middle column of (I - lam L)^-1 for the (2s+1)-point 1-D Laplacian L.

    utils/functions.py:24-49 obtains it from an eigendecomposition (half.dot(half[s])); the
    middle column of the inverse is the same vector, computed here by a direct solve.
    """
    n = 2 * s + 1
    lap = -2.0 * np.eye(n) + np.eye(n, k=1) + np.eye(n, k=-1)
    e = np.zeros(n)
    e[s] = 1.0
    col = np.linalg.solve(np.eye(n) - lam * lap, e)
    return col / col.sum()


def separable_conv3d_replicate(field, k1d):
    """Replicate-pad by s on all six faces, then 1-D cross-correlation along z, y, x (that order).

    The 4-argument branch of utils/util.py:394-404 as driven by trainer/trainer.py:568-583
    (same (2s+1)-tap kernel for the three channels, groups=3).
    """
    k1d = torch.as_tensor(k1d, dtype=field.dtype)
    s = (k1d.numel() - 1) // 2
    w = k1d.view(1, 1, -1).expand(3, 1, -1)
    x = F.pad(field, (s,) * 6, mode='replicate')
    x = F.conv3d(x, w.reshape(3, 1, -1, 1, 1), groups=3)
    x = F.conv3d(x, w.reshape(3, 1, 1, -1, 1), groups=3)
    x = F.conv3d(x, w.reshape(3, 1, 1, 1, -1), groups=3)
    return x


class _SobolevStraightThrough(torch.autograd.Function):
    """forward = smoothing, backward = identity (utils/functions.py:98-109)."""

    @staticmethod
    def forward(ctx, x, k1d):
        return separable_conv3d_replicate(x, k1d)

    @staticmethod
    def backward(ctx, g):
        return g, None


# --------------------------------------------------------------------------------------------
# a3  SVF scaling and squaring      utils/transformation.py:51-76; utils/util.py:263-278,418-443
# --------------------------------------------------------------------------------------------


def identity_grid(dims):
    """(1, D, H, W, 3) grid in [-1, 1] for dims = (D, H, W); channel 0 = x runs along the LAST tensor axis.

    utils/util.py:263-278.  The reference reads `nx, ny, nz = dims[0], dims[1], dims[2]` and therefore only works
    for cubic volumes (all its configs and tests are cubic; anything else raises a shape error at
    utils/transformation.py:71).  For cubic volumes this function is identical; for non-cubic ones it is the
    consistent generalisation (x <-> W, y <-> H, z <-> D) -- builder-defined, no reference behaviour exists.
    """
    nz, ny, nx = dims[0], dims[1], dims[2]
    x = torch.linspace(-1, 1, steps=nx).view(1, 1, nx).expand(nz, ny, nx)
    y = torch.linspace(-1, 1, steps=ny).view(1, ny, 1).expand(nz, ny, nx)
    z = torch.linspace(-1, 1, steps=nz).view(nz, 1, 1).expand(nz, ny, nx)
    return torch.stack((x, y, z), dim=-1).unsqueeze(0)


def to_normalised(field):
    """voxel units -> [-1, 1] units (utils/util.py:418-429).  The reference scales channel i by 2 / (shape[2 + i] - 1),
    which pairs x with D: harmless for the cubic volumes it supports; here channel c is paired with its own axis
    (x <-> W, y <-> H, z <-> D), identical for cubic volumes."""
    scale = torch.tensor([2.0 / float(n - 1) for n in reversed(field.shape[2:])], dtype=field.dtype)
    return field * scale.view(1, -1, 1, 1, 1)


def to_voxels(field):
    """[-1, 1] units -> voxel units (utils/util.py:432-443); same axis pairing as `to_normalised`."""
    scale = torch.tensor([float(n - 1) / 2.0 for n in reversed(field.shape[2:])], dtype=field.dtype)
    return field * scale.view(1, -1, 1, 1, 1)


def svf_exp(v, no_steps=12, keep_steps=False):
    """Scaling and squaring: d0 = normalised(v) / 2^steps; d <- d + d o (id + d), `no_steps` times.

    utils/transformation.py:63-76.  Returns (transformation in [-1,1], displacement in voxels
    [, list of d_0..d_steps]).
    """
    idg = identity_grid(v.shape[2:])
    d = to_normalised(v) / float(2 ** no_steps)
    steps = [d]
    for _ in range(no_steps):
        grid = idg + d.permute(0, 2, 3, 4, 1)
        d = d + F.grid_sample(d, grid, mode='bilinear', padding_mode='border', align_corners=True)
        steps.append(d)
    transformation = idg.permute(0, 4, 1, 2, 3) + d
    if keep_steps:
        return transformation, to_voxels(d), steps
    return transformation, to_voxels(d)


# --------------------------------------------------------------------------------------------
# a4  cubic B-spline FFD                       utils/transformation.py:79-164; utils/util.py:61-69
# --------------------------------------------------------------------------------------------


def control_grid_size(dims, cps):
    """ceil((N - 1) / cps) + 3 control points per axis (utils/util.py:61-69)."""
    return tuple(int(math.ceil((n - 1) / c) + 1 + 2) for n, c in zip(dims, cps))


def bspline_kernel_1d(stride):
    """(4*stride - 1)-tap sampled cubic B-spline (utils/transformation.py:79-102)."""
    n = 4 * stride - 1
    r = n // 2
    t = torch.abs((torch.arange(n, dtype=torch.float64) - r) / stride)
    inner = 2.0 / 3.0 + (0.5 * t - 1.0) * t ** 2
    outer = -((t - 2.0) ** 3) / 6.0
    k = torch.where(t < 1, inner, torch.where(t < 2, outer, torch.zeros_like(t)))
    return k.float()


def ffd_upsample(v_cp, dims, cps):
    """Control-point velocities -> dense velocities: per axis a transposed 1-D convolution of
    stride cps with the sampled B-spline, padding (len-1)//2, then crop [cps : cps + N].

    utils/transformation.py:105-153.  Axis i of the loop is tensor axis i+2 and uses cps[i].
    """
    v = v_cp
    for i, c in enumerate(cps):
        k = bspline_kernel_1d(c).to(v.dtype)  # (fp32 like the reference; the fp64 error-band runs carry their dtype through)
        p = (k.numel() - 1) // 2
        ax = i + 2
        x = v.transpose(ax, -1)
        shp = x.shape
        groups = int(np.prod(shp[1:-1]))
        x = x.reshape(shp[0], groups, shp[-1])
        x = F.conv_transpose1d(x, k.expand(groups, 1, -1), stride=c, padding=p, groups=groups)
        x = x.reshape(*shp[:-1], x.shape[-1])
        v = x.transpose(-1, ax)
    sl = (slice(None), slice(None)) + tuple(slice(c, c + n) for c, n in zip(cps, dims))
    return v[sl]


# --------------------------------------------------------------------------------------------
# a5 / a6  uniform jitter and warps                    utils/util.py:44-53; utils/registration.py:17-30
# --------------------------------------------------------------------------------------------


def jitter_grid(transformation, alpha, unif):
    """transformation + normalised(alpha - 2 alpha u), u ~ U[0,1)  (utils/util.py:44-53)."""
    return transformation + to_normalised(-2.0 * alpha * unif + alpha)


def warp_trilinear(im, transformation):
    """Trilinear / border / align_corners resample (utils/registration.py:29-30)."""
    return F.grid_sample(im, transformation.permute(0, 2, 3, 4, 1), mode='bilinear', padding_mode='border',
                         align_corners=True)


def warp_nearest(seg, transformation):
    """Nearest-neighbour resample of masks / label maps, cast back (utils/registration.py:20-27)."""
    out = F.grid_sample(seg.float(), transformation.permute(0, 2, 3, 4, 1), mode='nearest', padding_mode='border',
                        align_corners=True)
    return out.to(seg.dtype)


# --------------------------------------------------------------------------------------------
# explicit trilinear sampler + adjoints (what the HIP kernels implement)
# --------------------------------------------------------------------------------------------


def _unnormalise_clip(g, size):
    """ATen grid_sampler coordinate pipeline for align_corners=True, padding_mode='border':
    i = ((g + 1) / 2) * (size - 1), clipped to [0, size-1]; d(i)/d(g) = (size-1)/2 strictly inside, else 0.
    """
    i = ((g + 1.0) / 2.0) * (size - 1)
    inside = (i > 0) & (i < size - 1)
    return i.clamp(0, size - 1), inside.to(g.dtype) * ((size - 1) / 2.0)


def trilinear_sample_explicit(inp, grid, want_taps=False):
    """inp (B,Cn,D,H,W), grid (B,D,H,W,3) -> (B,Cn,D,H,W); same maths as F.grid_sample(bilinear,border,True)."""
    B, Cn, D, H, W = inp.shape
    ix, mx = _unnormalise_clip(grid[..., 0], W)
    iy, my = _unnormalise_clip(grid[..., 1], H)
    iz, mz = _unnormalise_clip(grid[..., 2], D)
    x0, y0, z0 = ix.floor(), iy.floor(), iz.floor()
    tx, ty, tz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    flat = inp.reshape(B, Cn, -1)
    out = torch.zeros(B, Cn, *grid.shape[1:4], dtype=inp.dtype)
    taps = []
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                ok = ((xi <= W - 1) & (yi <= H - 1) & (zi <= D - 1)).to(inp.dtype)
                w = (tx if dx else 1 - tx) * (ty if dy else 1 - ty) * (tz if dz else 1 - tz) * ok
                lin = (zi.clamp(max=D - 1) * H + yi.clamp(max=H - 1)) * W + xi.clamp(max=W - 1)
                val = torch.gather(flat, 2, lin.reshape(B, 1, -1).expand(B, Cn, -1)).reshape(out.shape)
                out = out + w.unsqueeze(1) * val
                taps.append((dx, dy, dz, lin, val, ok))
    if want_taps:
        return out, (tx, ty, tz, mx, my, mz, taps)
    return out


def trilinear_backward_explicit(inp, grid, gout, need_input_grad=True):
    """Adjoint of `trilinear_sample_explicit`: returns (grad_input or None, grad_grid (B,D,H,W,3))."""
    B, Cn, D, H, W = inp.shape
    _, (tx, ty, tz, mx, my, mz, taps) = trilinear_sample_explicit(inp, grid, want_taps=True)
    ggx = torch.zeros_like(tx)
    ggy = torch.zeros_like(tx)
    ggz = torch.zeros_like(tx)
    ginp = torch.zeros(B, Cn, D * H * W, dtype=inp.dtype) if need_input_grad else None
    for dx, dy, dz, lin, val, ok in taps:
        wx, wy, wz = (tx if dx else 1 - tx), (ty if dy else 1 - ty), (tz if dz else 1 - tz)
        sx, sy, sz = (1.0 if dx else -1.0), (1.0 if dy else -1.0), (1.0 if dz else -1.0)
        dot = (gout * val).sum(1) * ok
        ggx = ggx + sx * wy * wz * dot
        ggy = ggy + sy * wx * wz * dot
        ggz = ggz + sz * wx * wy * dot
        if need_input_grad:
            w = (wx * wy * wz * ok).unsqueeze(1)
            ginp.scatter_add_(2, lin.reshape(B, 1, -1).expand(B, Cn, -1), (w * gout).reshape(B, Cn, -1))
    ggrid = torch.stack((ggx * mx, ggy * my, ggz * mz), dim=-1)
    if need_input_grad:
        ginp = ginp.reshape(inp.shape)
    return ginp, ggrid


# --------------------------------------------------------------------------------------------
# a7  LCC map                                                         model/loss.py:53-59,102-111
# --------------------------------------------------------------------------------------------


def box_sum_replicate(x, s):
    """(2s+1)^3 all-ones cross-correlation with replicate padding (the frozen Conv3d of model/loss.py:56-58)."""
    k = 2 * s + 1
    w = torch.ones(1, 1, k, k, k, dtype=x.dtype)
    return F.conv3d(F.pad(x, (s,) * 6, mode='replicate'), w)


def lcc_normalise(im, s):
    """(I - u) / sqrt(var + 1e-10), u = box(I)/k^3, var = box((I-u)^2)/k^3  (model/loss.py:103-109)."""
    n = float((2 * s + 1) ** 3)
    u = box_sum_replicate(im, s) / n
    var = box_sum_replicate(torch.pow(im - u, 2), s) / n
    sigma = torch.sqrt(var + 1e-10)
    return (im - u) / sigma, u, sigma


def lcc_map(im_fixed, im_moving, s):
    """z = LCC(F) - LCC(M)  (model/loss.py:102-111)."""
    return lcc_normalise(im_fixed, s)[0] - lcc_normalise(im_moving, s)[0]


# --------------------------------------------------------------------------------------------
# a8  Gaussian mixture                                                   model/loss.py:61-100
# --------------------------------------------------------------------------------------------


def gmm_log_proportions(logits):
    """log_softmax(logits + 1e-2) (model/loss.py:67-69)."""
    return torch.log_softmax(logits + 1e-2, dim=0)


def gmm_init_log_std(sigma, no_components):
    """linspace(log(sigma/100), log(5 sigma), K) (model/loss.py:61-65)."""
    return torch.linspace(math.log(float(sigma) / 100.0), math.log(float(sigma) * 5.0), steps=no_components)


def gmm_log_pdf(z, log_std, logits):
    """log sum_k pi_k N(z; 0, sigma_k) per element of flattened z -> (1, n)  (model/loss.py:87-93)."""
    E = 0.5 * (z.reshape(1, -1, 1) * torch.exp(-1.0 * log_std)) ** 2
    return torch.logsumexp((gmm_log_proportions(logits) - log_std - LOG_SQRT_2PI) - E, dim=-1)


def gmm_nll(z, log_std, logits):
    """-sum log p(z) (model/loss.py:99-100,113-114)."""
    return -1.0 * gmm_log_pdf(z, log_std, logits).sum()


# --------------------------------------------------------------------------------------------
# a9  virtual decimation       utils/util.py:330-347,446-485; model/loss.py:95-97; trainer.py:507-514
# --------------------------------------------------------------------------------------------


def vd_rescale_autograd(res, mask, log_std, logits):
    """The reference's route: nested autograd through log_pdf_VD (utils/util.py:330-347)."""
    res_masked = torch.where(mask, res, torch.zeros_like(res)).detach()
    scaled = (res_masked.reshape(1, -1, 1) * torch.exp(-1.0 * log_std.detach())).requires_grad_(True)
    E = 0.5 * scaled ** 2
    lp = torch.logsumexp((gmm_log_proportions(logits.detach()) - log_std.detach() - LOG_SQRT_2PI) - E, dim=-1)
    g, = torch.autograd.grad(-1.0 * lp.sum(), scaled)
    return torch.sum(scaled.detach() * g, dim=-1).reshape(res.shape)


def vd_rescale(res, mask, log_std, logits):
    """Closed form of the above: x = sum_k r_k (z / sigma_k)^2, r = softmax_k(log pi_k - log sigma_k - E_k); 0 off-mask."""
    zm = torch.where(mask, res, torch.zeros_like(res)).reshape(-1, 1)
    q = (zm * torch.exp(-1.0 * log_std)) ** 2
    r = torch.softmax((gmm_log_proportions(logits) - log_std - LOG_SQRT_2PI) - 0.5 * q, dim=-1)
    return (r * q).sum(-1).reshape(res.shape)


def vd_factor(x, mask):
    """alpha = sqrt(prod_axes min(1, -(2/pi) log(corr_axis)))  (utils/util.py:446-485); x, mask: (1,1,D,H,W)."""
    n = mask.sum()
    var = torch.mean(x[mask] ** 2)
    xm = torch.where(mask, x, torch.zeros_like(x))
    cov = [torch.sum(xm[:, :, :-1] * xm[:, :, 1:]) / n,
           torch.sum(xm[:, :, :, :-1] * xm[:, :, :, 1:]) / n,
           torch.sum(xm[:, :, :, :, :-1] * xm[:, :, :, :, 1:]) / n]
    sq = [torch.clamp(-2.0 / math.pi * torch.log(c / var), max=1.0) for c in cov]
    return torch.sqrt(sq[0] * sq[1] * sq[2])


# --------------------------------------------------------------------------------------------
# a11  regulariser energy                               utils/diff_op.py:62-96; model/loss.py:152-161
# --------------------------------------------------------------------------------------------


def forward_differences(v, transformation=False):
    """(C,3,D,H,W) -> (C, 3[d/dx,d/dy,d/dz], D,H,W, 3[component]); the DIFFERENCE array is replicate-padded,
    so the last plane repeats the previous difference (utils/diff_op.py:78-96)."""
    dx = F.pad(v[:, :, :, :, 1:] - v[:, :, :, :, :-1], (0, 1, 0, 0, 0, 0), mode='replicate')
    dy = F.pad(v[:, :, :, 1:] - v[:, :, :, :-1], (0, 0, 0, 1, 0, 0), mode='replicate')
    dz = F.pad(v[:, :, 1:] - v[:, :, :-1], (0, 0, 0, 0, 0, 1), mode='replicate')
    if transformation:
        D, H, W = v.shape[2:]
        dx = dx / (2.0 / (W - 1))
        dy = dy / (2.0 / (H - 1))
        dz = dz / (2.0 / (D - 1))
    return torch.stack([torch.stack((dx[:, c], dy[:, c], dz[:, c]), 1) for c in range(3)], dim=-1)


def reg_energy(v):
    """y_c = sum (forward differences)^2 per chain (model/loss.py:158-159)."""
    return torch.sum(forward_differences(v) ** 2, dim=(1, 2, 3, 4, 5))


def det_jacobian(nabla):
    """3x3 determinant of utils/util.py:72-91; nabla[..., j] = gradient of component j."""
    a, b, c = nabla[..., 0], nabla[..., 1], nabla[..., 2]
    return (a[:, 0] * b[:, 1] * c[:, 2] + b[:, 0] * c[:, 1] * a[:, 2] + c[:, 0] * a[:, 1] * b[:, 2]
            - a[:, 2] * b[:, 1] * c[:, 0] - b[:, 2] * c[:, 1] * a[:, 0] - c[:, 2] * a[:, 1] * b[:, 0])


def field_norm(field):
    """voxel-wise L2 norm, (C,3,...) -> (C,1,...) (utils/util.py:215-225)."""
    return torch.sqrt((field ** 2).sum(1, keepdim=True))


# --------------------------------------------------------------------------------------------
# a12  regulariser losses and hyper-priors                model/loss.py:172-321; model/distributions.py
# --------------------------------------------------------------------------------------------


def normal_log_pdf(x, loc, scale):
    """model/distributions.py:57-59 (parameterised by log scale there)."""
    log_scale = math.log(scale)
    return -0.5 * ((x - loc) * math.exp(-log_scale)) ** 2 - log_scale - LOG_SQRT_2PI


def gamma_log_pdf(log_x, shape, rate):
    """model/distributions.py:112-113."""
    shape = torch.as_tensor(shape, dtype=torch.float32)
    rate = torch.as_tensor(rate, dtype=torch.float32)
    return shape * torch.log(rate) + (shape - 1) * log_x - rate * log_x.exp() - torch.lgamma(shape)


def expgamma_log_pdf(x, shape, rate):
    """model/distributions.py:165-166."""
    return gamma_log_pdf(x, shape, rate) + x


def expgamma_expectation(shape, rate):
    """model/distributions.py:169-170."""
    shape = torch.as_tensor(shape, dtype=torch.float32)
    rate = torch.as_tensor(rate, dtype=torch.float32)
    return torch.digamma(shape) - torch.log(rate)


def dirichlet_log_pdf(log_proportions, concentration):
    """model/distributions.py:209-211."""
    return (log_proportions * (concentration - 1.0)).sum(-1) + torch.lgamma(concentration.sum(-1)) \
        - torch.lgamma(concentration).sum(-1)


def reg_l2(y, log_w_reg, dof):
    """(0.5 w y - 0.5 dof log w, log y)  (model/loss.py:197-198)."""
    return 0.5 * log_w_reg.exp() * y - 0.5 * dof * log_w_reg, y.log()


def reg_lognormal_init(w_reg, dof, nu=1.0):
    """loc0 = E[expGamma(nu dof/2, nu w/2)], log_scale0 = log 4 + log loc0 (model/loss.py:298-303)."""
    # the reference builds `dof` from numpy (float64) and `nu`, `w_reg` as float32 tensors, so loc/log_scale
    # come out as float64 0-dim parameters (model/distributions.py:234-242); mirrored here
    shape = torch.tensor(0.5 * nu * dof, dtype=torch.float64)
    rate = torch.tensor(0.5, dtype=torch.float32) * torch.tensor(nu, dtype=torch.float32) * torch.tensor(w_reg, dtype=torch.float32)
    loc = (torch.digamma(shape) - torch.log(rate)).clone().detach()
    return loc, math.log(4.0) + loc.log()


def reg_lognormal(y, loc, log_scale, dof):
    """model/loss.py:266-312: -log LogNormal(y) + (dof/2 - 1) log y, and log y."""
    ly = y.log()
    return ly + log_scale + 0.5 * ((ly - loc) / log_scale.exp()) ** 2 + (0.5 * dof - 1.0) * ly, ly


def reg_student(y, dof, a0=1e-6, b0_twice=2e-6):
    """model/loss.py:234-241."""
    return torch.log(b0_twice + y) * (a0 + 0.5 * dof), y.log()


def student_params(nu0=2e-6, lambda0=1e-6, a0=1e-6, b0=1e-6):
    """RegLoss_Student.__init__ (model/loss.py:206-232): nu0 overrides a0, lambda0 overrides b0 -> (a0, 2 b0)"""
    a = nu0 / 2.0 if nu0 != 2e-6 else a0
    if lambda0 != 1e-6:
        b0 = a / lambda0
    return a, b0 * 2.0


def reg_lognormal_l2(y, w_reg, dof):
    """RegLoss_LogNormal_L2 (model/loss.py:315-321 through RegLoss_EnergyBased._loss :262-270):
    -log Gamma(y; dof/2, w/2) + (dof/2 - 1) log y, with the reference's fp32 parameters and operation order."""
    shape, rate = torch.tensor(0.5 * dof), torch.tensor(0.5 * w_reg)
    ly = y.log()
    gamma_log_pdf = shape * torch.log(rate) + (shape - 1) * ly - rate * ly.exp() - torch.lgamma(shape)
    return -1.0 * gamma_log_pdf + (0.5 * dof - 1.0) * ly, ly


# --------------------------------------------------------------------------------------------
# a10  Adam with rate decay (scalars only)                       optimizers/adam_rate_decay.py:32-99
# --------------------------------------------------------------------------------------------


class AdamRateDecay:
    """Adam whose lr is lr / (1 + step * lr_decay); bias correction counts from the last re-init.

    Functional restatement working on (tensor, grad) pairs; `groups` = list of dicts with
    'params' (list of tensors updated in place) and 'lr'.
    """

    def __init__(self, groups, lr_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.groups = groups
        self.lr_decay, self.betas, self.eps = lr_decay, betas, eps
        self.state = {}

    def step(self, grads):
        """grads: list (same flat order as the params) of gradient tensors."""
        b1, b2 = self.betas
        idx = 0
        for g in self.groups:
            for p in g['params']:
                grad = grads[idx]
                st = self.state.setdefault(idx, None)
                if st is None:
                    st = self.state[idx] = {'step': 0, 'reinit': 0, 'm': torch.zeros_like(p), 'v': torch.zeros_like(p)}
                clr = g['lr'] / (1 + st['step'] * self.lr_decay)
                st['step'] += 1
                bc1 = 1 - b1 ** (st['step'] - st['reinit'])
                bc2 = 1 - b2 ** (st['step'] - st['reinit'])
                st['m'].mul_(b1).add_(grad, alpha=1 - b1)
                st['v'].mul_(b2).addcmul_(grad, grad, value=1 - b2)
                denom = (st['v'].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.data.addcdiv_(st['m'], denom, value=-clr / bc1)
                idx += 1
