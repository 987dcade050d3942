"""Which (predict_variants mode, displacement amplitude) pairs of tests/test_gpu_transition.py::test_variant_prediction_never_
changes_the_result are BIT-identical to mode 0 (every variant launched)?  Prints one line per pair."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ir_sgmcmc_amd.data_loader import synthetic_pair  # noqa: E402
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine  # noqa: E402
from ir_sgmcmc_amd.ops import perturb_smooth, sobolev_kernel_1d  # noqa: E402

DEV = 'cuda:0'
N = 24
f1, m1 = synthetic_pair((N, N, N), seed=0)
fixed = {k: v.unsqueeze(0).to(DEV) for k, v in f1.items() if k != 'seg'}
moving = {k: v.unsqueeze(0).to(DEV) for k, v in m1.items() if k != 'seg'}
for amp in (0.3, 1.6, 3.5, 7.0):
    g = torch.Generator().manual_seed(3)
    v0 = perturb_smooth(torch.randn(1, 3, N, N, N, generator=g).to(DEV), sobolev_kernel_1d(3, 0.5))
    v0 = v0 * (amp / float(v0.abs().max()))
    eps = torch.randn(1, 3, N, N, N, generator=g).to(DEV)
    res = {}
    for mode in (0, 1, 2):
        eng = TransitionEngine(EngineConfig(dims=(N, N, N), seed=1), DEV)
        eng.option('predict_variants', mode)
        fd, md = eng.prepare(fixed, moving)
        eng.gmm_init(fd, md)
        v = v0.clone()
        out = {k: torch.empty(1, 3, N, N, N, device=DEV) for k in ('grad_v', 'displacement')}
        for _ in range(3):
            eng.transition(fd, md, v, eps=eps, outputs=out)
        eng.flush()
        torch.cuda.synchronize()
        res[mode] = (v.clone(), out['grad_v'].clone(), out['displacement'].clone())
    for mode in (1, 2):
        print(f'amp {amp}: mode {mode} vs 0: v {torch.equal(res[mode][0], res[0][0])} grad {torch.equal(res[mode][1], res[0][1])} '
              f'disp {torch.equal(res[mode][2], res[0][2])} | max dev grad {float((res[mode][1] - res[0][1]).abs().max() / res[0][1].abs().max()):.2e}')
