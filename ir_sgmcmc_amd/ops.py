"""Functional wrappers over the stateless C-ABI operators (torch tensors in, torch tensors out, GPU only).

Each function names the reference call it stands in for; the classes in `ir_sgmcmc_amd.utils` / `.model` that mirror
the reference's modules are thin shells over these.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L


def _dims5(t, ch=None):
    if t.dim() != 5 or (ch is not None and t.shape[1] != ch):
        raise L.IrsError(f'expected a (C,{ch if ch else "ch"},D,H,W) tensor, got {tuple(t.shape)}')
    return t.shape[0], t.shape[2], t.shape[3], t.shape[4]


def sobolev_kernel_1d(s, lam):
    """Normalised middle column of (I - lambda L)^-1, L the (2s+1)-point 1-D Laplacian
    (reference utils/functions.py:24-49, where it is read off an eigendecomposition)."""
    n = 2 * s + 1
    lap = -2.0 * np.eye(n) + np.eye(n, k=1) + np.eye(n, k=-1)
    e = np.zeros(n)
    e[s] = 1.0
    col = np.linalg.solve(np.eye(n) - lam * lap, e)
    return (col / col.sum()).astype(np.float32)


def control_grid_size(dims, cps):
    """utils/util.py:61-69"""
    return tuple(int(math.ceil((n - 1) / c) + 1 + 2) for n, c in zip(dims, cps))


def perturb_smooth(v, kernel, sigma=None, eps=None, tau=None, seed=0, iteration=0):
    """SGLD.forward (+ injected or Philox noise) followed by SobolevGrad.forward
    (utils/functions.py:76-109).  tau=None -> smoothing only; kernel=None -> perturbation only."""
    lib = L.load()
    Cn, D, H, W = _dims5(v, 3)
    out = torch.empty_like(v)
    s = 0 if kernel is None else (len(kernel) - 1) // 2
    tmp = torch.empty_like(v) if s > 0 else None
    k = (C.c_float * (2 * s + 1))(*[float(x) for x in kernel]) if s > 0 else None
    L.check(lib.irs_perturb_smooth(L.dev_ptr(v, torch.float32), L.dev_ptr(sigma, torch.float32, True),
                                   L.dev_ptr(eps, torch.float32, True), -1.0 if tau is None else float(tau),
                                   C.cast(k, C.c_void_p) if k is not None else None, s, Cn, D, H, W,
                                   L.dev_ptr(tmp, None, True), L.dev_ptr(out), seed, iteration, L.stream_ptr()))
    return out


def svf_exp_fwd(v, no_steps=12, want_outputs=True):
    """SVF_3D.forward (utils/transformation.py:63-76).  Returns (transformation, displacement, steps)."""
    lib = L.load()
    Cn, D, H, W = _dims5(v, 3)
    steps = torch.empty((no_steps,) + tuple(v.shape), device=v.device, dtype=torch.float32)
    t = torch.empty_like(v) if want_outputs else None
    d = torch.empty_like(v) if want_outputs else None
    L.check(lib.irs_svf_exp_fwd(L.dev_ptr(v, torch.float32), L.dev_ptr(steps), L.dev_ptr(t, None, True),
                                L.dev_ptr(d, None, True), no_steps, Cn, D, H, W, L.stream_ptr()))
    return t, d, steps


def svf_exp_bwd(v, steps, g_last):
    """Gradient w.r.t. v of the scaling-and-squaring chain given dL/d(d_last) in normalised units."""
    lib = L.load()
    Cn, D, H, W = _dims5(v, 3)
    scratch = torch.empty((2,) + tuple(v.shape), device=v.device, dtype=torch.float32)
    g_v = torch.empty_like(v)
    L.check(lib.irs_svf_exp_bwd(L.dev_ptr(v, torch.float32), L.dev_ptr(steps, torch.float32),
                                L.dev_ptr(g_last.contiguous(), torch.float32), L.dev_ptr(scratch), L.dev_ptr(g_v),
                                steps.shape[0], Cn, D, H, W, L.stream_ptr()))
    return g_v


def ffd_up(v_cp, dims, cps):
    """Cubic_B_spline_FFD_3D.forward (utils/transformation.py:146-153)."""
    lib = L.load()
    Cn = v_cp.shape[0]
    D, H, W = dims
    if tuple(v_cp.shape[2:]) != control_grid_size(dims, cps):
        raise L.IrsError(f'control grid {tuple(v_cp.shape[2:])} does not match dims {dims} / cps {cps}')
    dense = torch.empty((Cn, 3, D, H, W), device=v_cp.device, dtype=torch.float32)
    tmp = torch.empty((2, Cn, 3, D, H, W), device=v_cp.device, dtype=torch.float32)
    L.check(lib.irs_ffd_up(L.dev_ptr(v_cp, torch.float32), L.dev_ptr(dense), L.dev_ptr(tmp), Cn, D, H, W, *cps,
                           L.stream_ptr()))
    return dense


def ffd_adjoint(g_dense, cps):
    lib = L.load()
    Cn, D, H, W = _dims5(g_dense, 3)
    G = control_grid_size((D, H, W), cps)
    g_cp = torch.empty((Cn, 3, *G), device=g_dense.device, dtype=torch.float32)
    tmp = torch.empty((2, Cn, 3, D, H, W), device=g_dense.device, dtype=torch.float32)
    L.check(lib.irs_ffd_adjoint(L.dev_ptr(g_dense.contiguous(), torch.float32), L.dev_ptr(g_cp), L.dev_ptr(tmp), Cn, D, H,
                                W, *cps, L.stream_ptr()))
    return g_cp


def warp_displacement(im, d_last, unif=None, alpha=0.0, seed=0, iteration=0):
    """Trilinear warp at id + d_last (+ uniform jitter): registration_module(im, transformation_with_noise)."""
    lib = L.load()
    Cn, D, H, W = _dims5(d_last, 3)
    out = torch.empty((Cn, 1, D, H, W), device=d_last.device, dtype=torch.float32)
    L.check(lib.irs_warp_fwd(L.dev_ptr(im, torch.float32), im.shape[0], L.dev_ptr(d_last, torch.float32),
                             L.dev_ptr(unif, torch.float32, True), float(alpha), L.dev_ptr(out), Cn, D, H, W, seed,
                             iteration, L.stream_ptr()))
    return out


def warp_displacement_bwd(im, d_last, g_warped, unif=None, alpha=0.0, seed=0, iteration=0):
    lib = L.load()
    Cn, D, H, W = _dims5(d_last, 3)
    g_d = torch.empty_like(d_last)
    L.check(lib.irs_warp_bwd(L.dev_ptr(im, torch.float32), im.shape[0], L.dev_ptr(d_last, torch.float32),
                             L.dev_ptr(unif, torch.float32, True), float(alpha),
                             L.dev_ptr(g_warped.contiguous(), torch.float32), L.dev_ptr(g_d), Cn, D, H, W, seed, iteration,
                             L.stream_ptr()))
    return g_d


def warp(im, transformation):
    """RegistrationModule.forward (utils/registration.py:17-32): float -> trilinear; bool / int16 -> nearest."""
    lib = L.load()
    Cn, D, H, W = _dims5(transformation, 3)
    if im.dim() != 5 or im.shape[1] != 1 or im.shape[0] not in (1, Cn) or tuple(im.shape[2:]) != (D, H, W):
        raise L.IrsError(f'image shape {tuple(im.shape)} does not match transformation {tuple(transformation.shape)}')
    t = L.dev_ptr(transformation, torch.float32)
    if im.dtype == torch.float32:
        out = torch.empty((Cn, 1, D, H, W), device=im.device, dtype=torch.float32)
        L.check(lib.irs_warp_transformation(L.dev_ptr(im), im.shape[0], t, L.dev_ptr(out), Cn, D, H, W, L.stream_ptr()))
        return out
    if im.dtype == torch.bool:
        out = torch.empty((Cn, 1, D, H, W), device=im.device, dtype=torch.bool)
        L.check(lib.irs_warp_nearest_u8(L.dev_ptr(im), im.shape[0], t, L.dev_ptr(out), Cn, D, H, W, L.stream_ptr()))
        return out
    if im.dtype == torch.int16:
        out = torch.empty((Cn, 1, D, H, W), device=im.device, dtype=torch.int16)
        L.check(lib.irs_warp_nearest_i16(L.dev_ptr(im), im.shape[0], t, L.dev_ptr(out), Cn, D, H, W, L.stream_ptr()))
        return out
    raise NotImplementedError  # same error behaviour as utils/registration.py:32


def lcc_normalise(im, s, want_sigma=False):
    """(I - u) / sqrt(var + 1e-10) with (2s+1)^3 replicate-padded box statistics (model/loss.py:103-109)."""
    lib = L.load()
    Cn, D, H, W = _dims5(im, 1)
    out = torch.empty_like(im)
    sig = torch.empty_like(im) if want_sigma else None
    L.check(lib.irs_lcc_normalise(L.dev_ptr(im, torch.float32), L.dev_ptr(out), L.dev_ptr(sig, None, True), s, Cn, D, H, W,
                                  L.stream_ptr()))
    return (out, sig) if want_sigma else out


def lcc_map_fwd(fhat, warped, s):
    lib = L.load()
    Cn, D, H, W = _dims5(warped, 1)
    z = torch.empty_like(warped)
    sig = torch.empty_like(warped)
    L.check(lib.irs_lcc_map_fwd(L.dev_ptr(fhat, torch.float32), fhat.shape[0], L.dev_ptr(warped, torch.float32),
                                L.dev_ptr(z), L.dev_ptr(sig), s, Cn, D, H, W, L.stream_ptr()))
    return z, sig


def lcc_map_bwd(fhat, z, sigma_m, g_z, s):
    lib = L.load()
    Cn, D, H, W = _dims5(z, 1)
    g = torch.empty_like(z)
    L.check(lib.irs_lcc_map_bwd(L.dev_ptr(fhat, torch.float32), fhat.shape[0], L.dev_ptr(z, torch.float32),
                                L.dev_ptr(sigma_m, torch.float32), L.dev_ptr(g_z.contiguous(), torch.float32), L.dev_ptr(g),
                                s, Cn, D, H, W, L.stream_ptr()))
    return g


def reg_energy(v):
    """sum of squared replicate-padded forward differences per chain (model/loss.py:158-159) -> (C,) float64"""
    lib = L.load()
    Cn, D, H, W = _dims5(v, 3)
    y = torch.empty(Cn, device=v.device, dtype=torch.float64)
    scratch = torch.empty(lib.irs_reduce_scratch_doubles(), device=v.device, dtype=torch.float64)
    L.check(lib.irs_reg_energy(L.dev_ptr(v, torch.float32), L.dev_ptr(y), L.dev_ptr(scratch), Cn, D, H, W, L.stream_ptr()))
    return y


def gradient_operator(v, transformation=False):
    """GradientOperator.forward (utils/diff_op.py:78-96) -> (C,3,D,H,W,3)"""
    lib = L.load()
    Cn, D, H, W = _dims5(v, 3)
    nabla = torch.empty((Cn, 3, D, H, W, 3), device=v.device, dtype=torch.float32)
    L.check(lib.irs_gradient_operator(L.dev_ptr(v, torch.float32), L.dev_ptr(nabla), int(bool(transformation)), Cn, D, H, W,
                                      L.stream_ptr()))
    return nabla


def log_det_jacobian(transformation):
    """calc_no_non_diffeomorphic_voxels (utils/util.py:209-212): (NaN count per chain as int64 tensor, log det J)"""
    lib = L.load()
    Cn, D, H, W = _dims5(transformation, 3)
    ld = torch.empty((Cn, D, H, W), device=transformation.device, dtype=torch.float32)
    cnt = torch.empty(Cn, device=transformation.device, dtype=torch.int64)
    L.check(lib.irs_log_det_jacobian(L.dev_ptr(transformation, torch.float32), L.dev_ptr(ld), L.dev_ptr(cnt), Cn, D, H, W,
                                     L.stream_ptr()))
    return cnt, ld
