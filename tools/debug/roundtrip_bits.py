"""Is irs_set_state(irs_get_state()) invisible to the chain?  Same chain with and without a state round trip between transitions."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine
from ir_sgmcmc_amd.ops import perturb_smooth, sobolev_kernel_1d

DEV = 'cuda:0'
dims = (21, 17, 23)
f1, m1 = synthetic_pair(dims, seed=25)
fixed = {k: v.unsqueeze(0).to(DEV).contiguous() for k, v in f1.items() if k != 'seg'}
moving = {k: v.unsqueeze(0).to(DEV).contiguous() for k, v in m1.items() if k != 'seg'}
g = torch.Generator().manual_seed(25)
v0 = perturb_smooth(torch.randn(1, 3, *dims, generator=g).to(DEV), sobolev_kernel_1d(3, 0.5))
v0 = v0 * (0.5 / float(v0.abs().max()))
for loss in ('GMM', 'SSD'):
    for reg, learn in (('RegLoss_L2', False), ('RegLoss_LogNormal', True)):
        res = []
        for trip_after in (None, 1, 2):
            eng = TransitionEngine(EngineConfig(dims=dims, data_loss=loss, reg_loss=reg, reg_learnable=learn, seed=25), DEV)
            eng.option('predict_variants', 0)
            fd, md = eng.prepare(fixed, moving)
            eng.gmm_init(fd, md)
            v = v0.clone()
            for t in range(4):
                eng.transition(fd, md, v)
                if trip_after is not None and t + 1 == trip_after:
                    eng.set_state(eng.state())
            eng.flush()
            st = eng.state()
            res.append((v.clone(), list(st.gmm_log_std)[:4], st.reg_param[0]))
        for i in (1, 2):
            print(loss, reg, 'round trip after transition', i, ': v bit-equal', torch.equal(res[0][0], res[i][0]),
                  'max diff %.3e' % float((res[0][0] - res[i][0]).abs().max()), 'log_std equal', res[0][1] == res[i][1], 'reg equal', res[0][2] == res[i][2])
