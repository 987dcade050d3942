"""One draw of tests/test_gpu_recovery_fuzz.py with variations: which ingredient makes the chain differ from the plain one?
    python tools/debug/recovery_case.py SEED"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine
from ir_sgmcmc_amd.ops import perturb_smooth, sobolev_kernel_1d
from tests.test_gpu_recovery_fuzz import _draw, _state_tuple
from tests.test_gpu_transition import DEV, outputs_for, to_dev


def chain(kw, v0, fixed, moving, hook, run_ahead, seq, knobs=()):
    eng = TransitionEngine(EngineConfig(**kw), DEV)
    eng.option('predict_variants', hook)
    eng.option('run_ahead', run_ahead)
    for k, val in knobs:
        eng.option(k, val)
    fd, md = eng.prepare(fixed, moving)
    eng.gmm_init(fd, md)
    v = v0.clone()
    out = outputs_for(eng.cfg)
    for op in seq:
        if op == 't':
            eng.transition(fd, md, v)
        elif op == 't_out':
            eng.transition(fd, md, v, outputs=out)
        elif op == 'flush':
            eng.flush()
        elif op == 'state':
            eng.state()
        elif op == 'scalars':
            eng.scalars()
        else:
            eng.set_state(eng.state())
    eng.flush()
    return v, _state_tuple(eng.state()), eng.recovered_transitions


seed = int(sys.argv[1])
kw, amp, hook, run_ahead, seq = _draw(seed)
dims, C = kw['dims'], kw['no_chains']
print(kw, amp, hook, run_ahead, seq)
f1, m1 = synthetic_pair(dims, seed=seed)
fixed = to_dev({k: v.unsqueeze(0) for k, v in f1.items() if k != 'seg'})
moving = to_dev({k: v.unsqueeze(0) for k, v in m1.items() if k != 'seg'})
g = torch.Generator().manual_seed(seed)
v0 = perturb_smooth(torch.randn(C, 3, *dims, generator=g).to(DEV), sobolev_kernel_1d(3, 0.5))
v0 = v0 * (amp / float(v0.abs().max()))
n_t = sum(op in ('t', 't_out') for op in seq)
plain = chain(kw, v0, fixed, moving, 0, 0, ['t'] * n_t)
only_t = ['t' if op in ('t', 't_out') else op for op in seq]
for name, args in (('as drawn', (hook, run_ahead, seq)), ('hook 0', (0, run_ahead, seq)), ('run_ahead 0', (hook, 0, seq)),
                   ('transitions only', (hook, run_ahead, ['t'] * n_t)), ('no outputs', (hook, run_ahead, only_t)),
                   ('hook 0, outputs where drawn, nothing else', (0, 0, [op for op in seq if op in ('t', 't_out')])),
                   ('hook 1, outputs where drawn, nothing else', (1, run_ahead, [op for op in seq if op in ('t', 't_out')]))):
    r = chain(kw, v0, fixed, moving, *args)
    print(f'{name:45s}: v bit-equal {torch.equal(r[0], plain[0])}  max diff {float((r[0] - plain[0]).abs().max()):.2e}  state equal {r[1] == plain[1]}  recovered {r[2]}')
