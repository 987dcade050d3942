"""Adam with lr / (1 + step * lr_decay) and bias corrections counted from the last re-initialisation
(reference optimizers/adam_rate_decay.py:10-99).  Host-side, for the handful of scalar hyper-parameters; the fused
transition carries the same update rule on the device (csrc/scalar_kernels.hip: adam_step_decay)."""
import math

import torch
from torch.optim import Optimizer


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, lr_decay=0.0, amsgrad=False):
        for name, val in (('learning rate', lr), ('lr_decay value', lr_decay), ('epsilon value', eps)):
            if not 0.0 <= val:
                raise ValueError('Invalid {}: {}'.format(name, val))
        for i, b in enumerate(betas):
            if not 0.0 <= b < 1.0:
                raise ValueError('Invalid beta parameter at index {}: {}'.format(i, b))
        super().__init__(params, dict(lr=lr, lr_decay=lr_decay, betas=betas, eps=eps, weight_decay=weight_decay,
                                      amsgrad=amsgrad))

    @torch.no_grad()
    def step(self, closure=None, reinit=False):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                grad = p.grad
                if grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                st = self.state[p]
                fresh = len(st) == 0
                if fresh:
                    st['step'] = 0
                if fresh or reinit:
                    st['reinit'] = st['step']
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                    if group['amsgrad']:
                        st['max_exp_avg_sq'] = torch.zeros_like(p)
                clr = group['lr'] / (1 + st['step'] * group['lr_decay'])
                st['step'] += 1
                n = st['step'] - st['reinit']
                bc1, bc2 = 1 - b1 ** n, 1 - b2 ** n
                if group['weight_decay'] != 0:
                    grad = grad.add(p, alpha=group['weight_decay'])
                st['exp_avg'].mul_(b1).add_(grad, alpha=1 - b1)
                st['exp_avg_sq'].mul_(b2).addcmul_(grad, grad, value=1 - b2)
                if group['amsgrad']:
                    torch.max(st['max_exp_avg_sq'], st['exp_avg_sq'], out=st['max_exp_avg_sq'])
                    denom = (st['max_exp_avg_sq'].sqrt() / math.sqrt(bc2)).add_(group['eps'])
                else:
                    denom = (st['exp_avg_sq'].sqrt() / math.sqrt(bc2)).add_(group['eps'])
                p.addcdiv_(st['exp_avg'], denom, value=-clr / bc1)
        return loss
