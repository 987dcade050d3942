"""Random CALL SEQUENCES against the recovery machinery of the fused engine: transitions whose kernel-variant assumptions fail on the
device are no-ops there and are re-run by a later call (include/irsgmcmc.h: irs_transition / irs_flush).  Whatever the host does in
between -- flush, read the state or the scalars, write the state back, ask for outputs, run further ahead or not at all -- the chain
must be, bit for bit, the chain of an engine that launched every variant all along.  In-kernel Philox noise: a dropped transition
must not have consumed its counter, a re-run must draw what the dropped one would have drawn."""
import os
import random

import pytest
import torch

from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine
from tests.conftest import fuzz_seeds
from tests.test_gpu_transition import DEV, outputs_for, to_dev

pytestmark = pytest.mark.gpu
OPS = ['t', 't', 't', 't', 't_out', 'flush', 'state', 'scalars', 'roundtrip']


def _draw(seed):
    r = random.Random(5000 + seed)
    dims = tuple(r.randint(16, 30) for _ in range(3))
    cfg = dict(dims=dims, no_chains=r.choice([1, 1, 2]), data_loss=r.choice(['GMM', 'GMM', 'SSD']), seed=seed,
               reg_loss=r.choice(['RegLoss_L2', 'RegLoss_LogNormal']), lr=r.choice([0.05, 0.4]))
    cfg['reg_learnable'] = cfg['reg_loss'] == 'RegLoss_LogNormal' and r.random() < 0.5
    return cfg, r.choice([0.5, 4.0, 9.0]), r.choice([1, 3, 3]), r.choice([0, 1, 2, 3]), [r.choice(OPS) for _ in range(r.randint(6, 14))] + ['t']


def _state_tuple(st, K=8):
    return (list(st.gmm_log_std)[:K], list(st.gmm_logits)[:K], list(st.reg_param), int(st.iteration), list(st.gmm_adam_step), list(st.reg_adam_step))


# default draws: 3 (GMM, 9-voxel start, every step mispredicted, outputs asked for), 7 (two chains, SSD, learnable LogNormal, run-ahead 2,
# flushes), 8 (two chains, state round trips before the first transition, run-ahead 3), 10 (production guesses, run-ahead 2, flush / state
# reads between transitions).  IRS_LONG=1: the twelve of round 4; IRS_RECOVERY_FUZZ_SEEDS=100: a longer hunt
@pytest.mark.parametrize('seed', fuzz_seeds('IRS_RECOVERY_FUZZ_SEEDS', (3, 7, 8, 10), range(12)))
def test_random_call_sequence_equals_the_plain_chain(seed):
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.ops import perturb_smooth, sobolev_kernel_1d
    kw, amp, hook, run_ahead, seq = _draw(seed)
    dims, C = kw['dims'], kw['no_chains']
    f1, m1 = synthetic_pair(dims, seed=seed)
    fixed = to_dev({k: v.unsqueeze(0) for k, v in f1.items() if k != 'seg'})
    moving = to_dev({k: v.unsqueeze(0) for k, v in m1.items() if k != 'seg'})
    g = torch.Generator().manual_seed(seed)
    v0 = perturb_smooth(torch.randn(C, 3, *dims, generator=g).to(DEV), sobolev_kernel_1d(3, 0.5))
    v0 = v0 * (amp / float(v0.abs().max()))
    n_t = sum(op in ('t', 't_out') for op in seq)

    # the plain chain: every variant launched, nothing assumed, nothing to recover
    ref = TransitionEngine(EngineConfig(**kw), DEV)
    ref.option('predict_variants', 0)
    fd, md = ref.prepare(fixed, moving)
    ref.gmm_init(fd, md)
    v_ref = v0.clone()
    for _ in range(n_t):
        ref.transition(fd, md, v_ref)
    ref.flush()
    st_ref, sc_ref = _state_tuple(ref.state()), ref.scalars()
    assert ref.recovered_transitions == 0

    eng = TransitionEngine(EngineConfig(**kw), DEV)
    eng.option('predict_variants', hook)   # 1: the production guesses; 3: "always tiny" -- every step beyond one voxel is mispredicted
    eng.option('run_ahead', run_ahead)
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
        else:   # the state read (which flushes) and written back: the chain must not notice
            eng.set_state(eng.state())
    eng.flush()
    what = f'seed {seed}: {dims} C={C} {kw["data_loss"]} amp {amp} hook {hook} run_ahead {run_ahead} {"".join(o[0] if o != "t_out" else "T" for o in seq)}'
    assert _state_tuple(eng.state()) == st_ref, what
    assert torch.equal(v, v_ref), (what, float((v - v_ref).abs().max()))
    assert eng.scalars() == sc_ref, what
