"""One configuration of tests/test_gpu_fuzz.py in detail: HIP engine, fp32 oracle and fp64 oracle side by side -- is a deviation
the engine's or the band of fp32 arithmetic itself?      python tools/debug/fuzz_case.py SEED [SEED ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import TransitionEngine
from oracle import OracleChain
from oracle import ops as O
from tests.test_gpu_fuzz import _draw
from tests.test_gpu_transition import DEV, engine_config, outputs_for, to_dev


def run(seed):
    oc, amp, sigma = _draw(1000 + seed)
    C, dims, dv = oc.no_chains, oc.dims, oc.dims_v
    amp = min(amp, 0.2 * min(dims))
    f1, m1 = synthetic_pair(dims, seed=seed)
    fixed = {k: v.unsqueeze(0).expand(C, *v.shape).contiguous() for k, v in f1.items() if k != 'seg'}
    moving = {k: v.unsqueeze(0).expand(C, *v.shape).contiguous() for k, v in m1.items() if k != 'seg'}
    gen = torch.Generator().manual_seed(seed)
    v0 = O.separable_conv3d_replicate(amp * torch.randn(C, 3, *dv, generator=gen), O.sobolev_kernel_1d(2, 0.5)).contiguous()
    sig = torch.full((C, 3, *dv), sigma) if sigma is not None else None
    eps = torch.randn(C, 3, *dv, generator=gen)
    unif = torch.rand(C, 3, *dims, generator=gen) if oc.uniform_noise is not None else None
    print(f'--- seed {seed}: {dims} C={C} {oc.transformation} steps {oc.no_steps} sobolev {oc.sobolev_s} jitter {oc.uniform_noise} {oc.data_loss} '
          f'K={oc.gmm_components} s={oc.lcc_s} vd={oc.virtual_decimation} {oc.reg_loss} learnable={oc.reg_learnable} amp={amp} sigma={sigma}')
    orc = OracleChain(oc, v0=v0, sigma=sig)
    orc.init_gmm(fixed, moving)
    o32 = orc.transition(fixed, moving, eps, unif)
    torch.set_default_dtype(torch.float64)
    try:
        f64 = lambda d: {k: (v.double() if v.is_floating_point() else v) for k, v in d.items()}
        orc64 = OracleChain(oc, v0=v0.double(), sigma=sig.double() if sig is not None else None)
        orc64.init_gmm(f64(fixed), f64(moving))
        o64 = orc64.transition(f64(fixed), f64(moving), eps.double(), unif.double() if unif is not None else None)
    finally:
        torch.set_default_dtype(torch.float32)
    cfg = engine_config(oc)
    eng = TransitionEngine(cfg, DEV)
    fd, md = eng.prepare(to_dev(fixed), to_dev(moving))
    eng.gmm_init(fd, md)
    v = v0.to(DEV).contiguous()
    out = outputs_for(cfg)
    eng.transition(fd, md, v, sig.to(DEV).contiguous() if sig is not None else None, eps.to(DEV), unif.to(DEV) if unif is not None else None, out)
    eng.flush()
    sc1 = eng.scalars()
    for key, okey in (('reg_term', 'reg'), ('data_term', 'data'), ('reg_energy', 'reg_energy')):
        if okey not in o32:
            continue
        h = [float(x) for x in sc1[key][:C]]
        a32 = [float(x) for x in torch.as_tensor(o32[okey]).flatten()]
        a64 = [float(x) for x in torch.as_tensor(o64[okey]).flatten()]
        rel = lambda p, q: max(abs(x - y) / max(abs(y), 1e-30) for x, y in zip(p, q))
        print(f'transition 1 {key}: HIP vs fp32 oracle {rel(h, a32):.2e}, HIP vs fp64 {rel(h, a64):.2e}, fp32 oracle vs fp64 {rel(a32, a64):.2e}   (fp64: {a64})')
    g64 = o64['grad_v']
    gmax = float(g64.abs().max())
    for name, g in (('oracle fp32', o32['grad_v'].double()), ('HIP engine ', out['grad_v'].cpu().double())):
        dev = (g - g64).abs() / gmax
        i = int(dev.argmax())
        idx = tuple(int(x) for x in torch.unravel_index(torch.tensor(i), dev.shape))
        print(f'{name} vs fp64 oracle: grad_v max deviation {float(dev.max()):.3e} of max|grad| at {idx}, mean {float(dev.mean()):.2e}')
    # second transition, every chain continued from the fp32 oracle's v_new (as the test does)
    eps2 = torch.randn(C, 3, *dv, generator=gen)
    unif2 = torch.rand(C, 3, *dims, generator=gen) if oc.uniform_noise is not None else None
    v.copy_(o32['v_new'].to(DEV))
    o32b = orc.transition(fixed, moving, eps2, unif2)
    out2 = outputs_for(cfg)
    eng.transition(fd, md, v, sig.to(DEV).contiguous() if sig is not None else None, eps2.to(DEV), unif2.to(DEV) if unif2 is not None else None, out2)
    sc = eng.scalars()
    gb = o32b['grad_v'].double()
    db = (out2['grad_v'].cpu().double() - gb).abs() / float(gb.abs().max())
    i = int(db.argmax())
    idx = tuple(int(x) for x in torch.unravel_index(torch.tensor(i), db.shape))
    print(f'transition 2: HIP vs oracle fp32 grad_v {float(db.max()):.3e} at {idx} (mean {float(db.mean()):.2e}); per chain max {[float(db[c].max()) for c in range(C)]}')
    print(f'   alpha {sc["alpha"][:C]} vs {o32b["alpha"]};  data {sc["data_term"][:C]} vs {o32b["data"]}; displacement {float((out2["displacement"].cpu() - o32b["displacement"]).abs().max()):.2e}')
    n = db.numel()
    print(f'   voxels (x channels x chains) above 1e-3: {int((db > 1e-3).sum())}, above 1e-4: {int((db > 1e-4).sum())}, above 1e-5: {int((db > 1e-5).sum())} of {n}; '
          f'v_new deviation {float((v.cpu() - o32b["v_new"]).abs().max()):.2e} (max|v| {float(o32b["v_new"].abs().max()):.2f})')
    # the fp64 oracle through the same second transition (continued from the fp32 oracle's v_new and mixture, like the engine)
    torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            orc64.v.copy_(o32['v_new'].double())
        o64b = orc64.transition(f64(fixed), f64(moving), eps2.double(), unif2.double() if unif2 is not None else None)
    finally:
        torch.set_default_dtype(torch.float32)
    g64b = o64b['grad_v']
    gm = float(g64b.abs().max())
    a32 = float((gb - g64b).abs()[idx]) / gm
    ahip = float((out2['grad_v'].cpu().double() - g64b).abs()[idx]) / gm
    print(f'   at that voxel, against the fp64 oracle: oracle fp32 {a32:.3e}, HIP {ahip:.3e};  fp32-vs-fp64 max anywhere {float((gb - g64b).abs().max()) / gm:.3e}')
    st = eng.state()
    K = oc.gmm_components
    print('   gmm log_std', list(st.gmm_log_std)[:K], 'oracle', orc.log_std.detach().tolist())
    d = (out['grad_v'].cpu().double() - o32['grad_v'].double()).abs() / gmax
    print(f'HIP vs oracle fp32: {float(d.max()):.3e};  displacement HIP vs fp32 {float((out["displacement"].cpu() - o32["displacement"]).abs().max()):.2e}, '
          f'fp32 vs fp64 {float((o32["displacement"].double() - o64["displacement"]).abs().max()):.2e}')


if __name__ == '__main__':
    for s in sys.argv[1:]:
        run(int(s))
