import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine
from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.ops import perturb_smooth, sobolev_kernel_1d
DEV = 'cuda:0'
N, T = 24, 6
f1, m1 = synthetic_pair((N, N, N), seed=0)
fixed = {k: v.unsqueeze(0).to(DEV) for k, v in f1.items() if k != 'seg'}
moving = {k: v.unsqueeze(0).to(DEV) for k, v in m1.items() if k != 'seg'}
g = torch.Generator().manual_seed(3)
v0 = perturb_smooth(torch.randn(1, 3, N, N, N, generator=g).to(DEV), sobolev_kernel_1d(3, 0.5))
v0 = v0 * (3.5 / float(v0.abs().max()))
for mode in (0, 3):
    eng = TransitionEngine(EngineConfig(dims=(N, N, N), seed=11), DEV)
    eng.option('predict_variants', mode)
    fd, md = eng.prepare(fixed, moving)
    eng.gmm_init(fd, md)
    v = v0.clone()
    disp = torch.zeros(1, 3, N, N, N, device=DEV)
    for t in range(T):
        eng.transition(fd, md, v, outputs={'displacement': disp})
        torch.cuda.synchronize()
        print(mode, t, 'max disp', float(disp.abs().max()), 'recovered so far', eng.recovered_transitions, 'vnorm', float(v.norm()))
    eng.flush()
    print(mode, 'final iteration', eng.state().iteration, 'recovered', eng.recovered_transitions, 'vnorm', float(v.norm()))
