"""sample_q_v (reference utils/sampler.py:4-21): reparameterised sample of the VI posterior, used to seed the chains."""
import torch


def sample_q_v(var_params_q_v, no_samples=1):
    mu_v, log_var_v, u_v = var_params_q_v['mu'], var_params_q_v['log_var'], var_params_q_v['u']
    sigma_v = torch.exp(0.5 * log_var_v)
    eps = torch.randn_like(sigma_v)
    x = torch.randn(1, device=u_v.device)
    if no_samples == 1:
        return mu_v + eps * sigma_v + x * u_v
    return mu_v + (eps * sigma_v + x * u_v), mu_v - (eps * sigma_v + x * u_v)
