"""SGLD / SobolevGrad autograd functions and the Sobolev kernel (reference utils/functions.py)."""
import numpy as np
import torch

from .. import ops as _ops


def Laplacian_1D(N):
    return -2.0 * np.eye(N) + np.eye(N, k=1) + np.eye(N, k=-1)


def Sobolev_kernel_1D(_s, _lambda):
    """(smoothing kernel, its 'square root'), both normalised to unit sum (utils/functions.py:24-49).
    kernel = middle column of (I - lambda L)^-1; sqrt kernel = middle column of (I - lambda L)^-1/2."""
    n = 2 * _s + 1
    w, v = np.linalg.eigh(Laplacian_1D(n))
    w = 1.0 - _lambda * w
    kernel = (v / w) @ v[_s]
    kernel_sqrt = (v / np.sqrt(w)) @ v[_s]
    return kernel / kernel.sum(), kernel_sqrt / kernel_sqrt.sum()


class SGLD(torch.autograd.Function):
    """forward: v + sqrt(2 tau) sigma eps; backward: sigma^2 * grad (utils/functions.py:76-84)"""

    @staticmethod
    def forward(ctx, v_curr_state, sigma, tau, eps=None):
        ctx.save_for_backward(sigma)
        eps = torch.randn_like(sigma) if eps is None else eps
        return _ops.perturb_smooth(v_curr_state.contiguous(), None, sigma.contiguous(), eps.contiguous(), tau=tau)

    @staticmethod
    def backward(ctx, grad_output):
        sigma, = ctx.saved_tensors
        return sigma ** 2 * grad_output, None, None, None


class SobolevGrad(torch.autograd.Function):
    """forward: separable smoothing with replicate padding; backward: identity (utils/functions.py:98-109).
    S: dict with 'x','y','z' kernels of shape (3,1,1,1,k) / (3,1,1,k,1) / (3,1,k,1,1) as the reference builds them
    (trainer/trainer.py:568-583), or a plain 1-D sequence of taps."""

    @staticmethod
    def forward(ctx, input, S, padding=None):
        k = S['x'] if isinstance(S, dict) else S
        k = torch.as_tensor(k).reshape(-1)
        taps = k[:k.numel() // 3].tolist() if isinstance(S, dict) else k.tolist()
        return _ops.perturb_smooth(input.contiguous(), taps)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None, None
