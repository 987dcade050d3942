#!/usr/bin/env python3
"""How far is the REFERENCE'S OWN fp32 arithmetic from an fp64 evaluation of the same transition?

The GPU tests hold the HIP path to the north-star tolerances (loss 1e-5 relative, displacement 1e-4 voxels) against the fp32
oracle / the reference-generated fixtures.  Two comparisons at the benchmark sizes use wider numbers -- the displacement at
256^3 (fp32 positions in [-1, 1] have an ulp of 7.6e-6 voxels there) and the gradient (trilinear interpolation's derivative
jumps across cell faces).  This script measures, with data instead of prose, the error band those wider numbers have to sit
in:  max |fp32 - fp64| of the displacement, of v.grad (relative to its maximum) and the fraction of voxels whose gradient
differs by more than 1e-3 of the maximum, for

  (a) the unmodified reference `Trainer._SGLD_transition` (trainer/trainer.py:291-356) at 32^3 and 64^3, evaluated in fp32 and
      again with every module and tensor in fp64 (`torch.set_default_dtype(torch.float64)`, modules `.double()`, the SAME noise
      values: `torch.randn_like` / `torch.rand` hand out the fp32 draws cast to the working precision while the reference runs).
      One stand-in is unavoidable: `RegistrationModule.forward` (utils/registration.py:13-32) dispatches on the tensor TYPE STRING
      and refuses a DoubleTensor, so the fp64 run calls the one ATen op behind its float branch (utils/registration.py:30)
      directly.  The reference tree is not modified.
  (b) the oracle at 128^3 and 256^3 on the very inputs `tests/test_gpu_transition.py::test_transition_matches_oracle_at_full_size`
      uses (same seeds), fp32 against fp64.

Output: tests/golden/fp64_bands.json (a few numbers per case; data, no reference source).  Runs only in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fp64.py [--sizes 128,256] [--skip-reference]
"""
import argparse
import json
import os
import sys
import time
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')

from ir_sgmcmc_amd.data_loader import synthetic_pair  # noqa: E402
from oracle import OracleChain, OracleConfig  # noqa: E402

torch.set_num_threads(8)
GRAD_RTOL = 1e-3  # tests/_report.py


def band(out32, out64):
    """deviation of an fp32 evaluation from the fp64 one"""
    d32, d64 = out32['displacement'].double(), out64['displacement'].double()
    g32, g64 = out32['grad_v'].double(), out64['grad_v'].double()
    gmax = float(g64.abs().max())
    dev = (g32 - g64).abs() / gmax
    return {
        'displacement_max_abs_dev_voxels': float((d32 - d64).abs().max()),
        'displacement_max_voxels': float(d64.abs().max()),
        'grad_max_rel_dev': float(dev.max()),
        'grad_frac_beyond_1e-3': float((dev > GRAD_RTOL).double().mean()),
        'grad_9999_permille_rel_dev': float(dev.flatten().kthvalue(max(1, int(0.9999 * dev.numel()))).values),
        'data_term_rel_dev': float(abs(float(out32['data'][0]) - float(out64['data'][0])) / abs(float(out64['data'][0]))),
        'reg_term_rel_dev': float(abs(float(out32['reg'][0]) - float(out64['reg'][0])) / abs(float(out64['reg'][0]))),
    }


def full_size_inputs(N):
    """exactly the inputs of test_transition_matches_oracle_at_full_size (gmm)"""
    f1, m1 = synthetic_pair((N, N, N), seed=0)
    fixed = {k: v.unsqueeze(0).contiguous() for k, v in f1.items() if k != 'seg'}
    moving = {k: v.unsqueeze(0).contiguous() for k, v in m1.items() if k != 'seg'}
    gen = torch.Generator().manual_seed(21)
    lo = torch.randn(1, 3, N // 8, N // 8, N // 8, generator=gen)
    v0 = torch.nn.functional.interpolate(lo, size=(N, N, N), mode='trilinear', align_corners=True)
    v0 = (v0 * (3.0 / float(v0.abs().max()))).contiguous()
    eps = torch.randn(1, 3, N, N, N, generator=gen)
    unif = torch.rand(1, 3, N, N, N, generator=gen)
    return fixed, moving, v0, eps, unif


def oracle_run(N, dtype, inputs):
    fixed, moving, v0, eps, unif = inputs
    torch.set_default_dtype(dtype)
    try:
        cast = lambda d: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()}
        orc = OracleChain(OracleConfig(dims=(N, N, N)), v0=v0.to(dtype))
        orc.init_gmm(cast(fixed), cast(moving))
        o = orc.transition(cast(fixed), cast(moving), eps.to(dtype), unif.to(dtype))
        return {k: (o[k].detach() if torch.is_tensor(o[k]) else o[k]) for k in ('displacement', 'grad_v', 'data', 'reg', 'alpha')}
    finally:
        torch.set_default_dtype(torch.float32)


def reference_run(ref, N, dtype, amp):
    """the reference's own transition on a smooth `amp`-voxel field, in `dtype`"""
    import make_golden as MG
    cfg = OracleConfig(dims=(N, N, N))
    f1, m1 = synthetic_pair(cfg.dims, seed=0)
    g = torch.Generator().manual_seed(77)
    lo = torch.randn(1, 3, max(N // 8, 2), max(N // 8, 2), max(N // 8, 2), generator=g)
    v0 = torch.nn.functional.interpolate(lo, size=(N, N, N), mode='trilinear', align_corners=True)
    v0 = (v0 * (amp / float(v0.abs().max()))).contiguous()
    eps32 = torch.randn(1, 3, N, N, N, generator=g)
    unif32 = torch.rand(1, 3, N, N, N, generator=g)
    torch.set_default_dtype(dtype)
    real_randn_like, real_rand = torch.randn_like, torch.rand
    try:
        fixed = {k: (v.unsqueeze(0).to(dtype) if v.is_floating_point() else v.unsqueeze(0)) for k, v in f1.items()}
        moving = {k: (v.unsqueeze(0).to(dtype) if v.is_floating_point() else v.unsqueeze(0)) for k, v in m1.items()}
        t, gmm, reg = MG.build_reference(ref, cfg, fixed, moving, v0.to(dtype), 1.0)
        if dtype == torch.float64:
            for m in (gmm, reg, t.transformation_module):
                m.double()
            t.S = {k: v.double() for k, v in t.S.items()}
            # utils/registration.py:13-32 refuses a DoubleTensor (type-string dispatch): the one ATen op of its float branch
            t.registration_module = lambda im, transformation: torch.nn.functional.grid_sample(
                im, transformation.permute(0, 2, 3, 4, 1), mode='bilinear', padding_mode='border', align_corners=True)
            # the optimiser was built on the fp32 parameters: rebuild it on the converted ones
            t.optimizer_GMM = ref.optim.Adam([{'params': [gmm.log_std], 'lr': cfg.gmm_lr_log_std},
                                              {'params': [gmm.logits], 'lr': cfg.gmm_lr_logits}], lr_decay=cfg.gmm_lr_decay)
        MG.reference_gmm_init(ref, t, gmm, cfg, fixed, moving)
        torch.randn_like = lambda x, *a, **k: eps32.to(x.dtype)       # the reference draws randn_like(sigma), then rand(shape)
        torch.rand = lambda *a, **k: unif32.to(dtype)
        loss_terms, output, aux = t._SGLD_transition(fixed, moving, gmm, reg)
        return {'displacement': output['displacement'].detach(), 'grad_v': t.v_curr_state.grad.detach().clone(),
                'data': [float(x) for x in loss_terms['data']], 'reg': [float(x) for x in loss_terms['reg']]}
    finally:
        torch.randn_like, torch.rand = real_randn_like, real_rand
        torch.set_default_dtype(torch.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='128,256', help='oracle fp32-vs-fp64 sizes')
    ap.add_argument('--skip-reference', action='store_true')
    args = ap.parse_args()
    path = os.path.join(HERE, 'fp64_bands.json')
    out = json.load(open(path)) if os.path.isfile(path) else {}
    out['_about'] = ('max |fp32 - fp64| of one _SGLD_transition: the error band of the reference\'s own arithmetic '
                     '(tests/golden/make_golden_fp64.py)')
    if not args.skip_reference:
        from _ref_import import import_reference
        ref = import_reference()
        for N, amp in ((32, 3.0), (64, 3.0), (64, 0.5)):
            t0 = time.time()
            b = band(reference_run(ref, N, torch.float32, amp), reference_run(ref, N, torch.float64, amp))
            out[f'reference_{N}_amp{amp}'] = b
            print(f'reference {N}^3 amp {amp}: {b}  ({time.time() - t0:.0f} s)', flush=True)
    for N in [int(x) for x in args.sizes.split(',') if x]:
        t0 = time.time()
        inp = full_size_inputs(N)
        b = band(oracle_run(N, torch.float32, inp), oracle_run(N, torch.float64, inp))
        out[f'oracle_{N}_gmm_full_size_test_inputs'] = b
        print(f'oracle {N}^3: {b}  ({time.time() - t0:.0f} s)', flush=True)
        json.dump(out, open(path, 'w'), indent=1)
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
