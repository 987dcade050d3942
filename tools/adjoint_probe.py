import torch, sys
sys.path.insert(0,'/root/repo')
from ir_sgmcmc_amd import ops as G
DEV='cuda:0'
for N in (64, 256):
    def rnd(*shape, seed=0):
        g = torch.Generator(device=DEV).manual_seed(seed)
        return torch.randn(*shape, generator=g, device=DEV)
    def smooth(C, amp, seed):
        v = G.perturb_smooth(rnd(C, 3, N, N, N, seed=seed), G.sobolev_kernel_1d(3, 0.5))
        return v * (amp / float(v.abs().max()))
    for amp in (0.5, 3.0):
        v = smooth(1, amp, 11); u, w = smooth(1, 1.0, 12), smooth(1, 1.0, 13)
        _, _, steps = G.svf_exp_fwd(v, 12, want_outputs=False)
        gv = G.svf_exp_bwd(v, steps, w)
        rhs = float((u.double() * gv.double()).sum())
        for eps in (0.2, 0.05, 0.0125, 0.003):
            _, _, sp = G.svf_exp_fwd(v + eps * u, 12, want_outputs=False)
            _, _, sm = G.svf_exp_fwd(v - eps * u, 12, want_outputs=False)
            lhs = float((((sp[-1].double() - sm[-1].double()) / (2 * eps)) * w.double()).sum())
            print(N, amp, eps, lhs, rhs, (lhs-rhs)/abs(rhs))
