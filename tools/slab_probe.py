"""Single-rank cost of the staged (z-slab) transition against the fused one: what the per-stage host orchestration and the
planar / unfused kernels of the staged path cost before any exchange.  Usage: python tools/slab_probe.py [N ...]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine
from ir_sgmcmc_amd.slab import SlabEngine

dev = torch.device('cuda', 0)
for N in [int(a) for a in sys.argv[1:]] or [128, 256]:
    dims = (N, N, N)
    f, m = synthetic_pair(dims, seed=0)
    fx = {k: v.unsqueeze(0).to(dev) for k, v in f.items() if k != 'seg'}
    mv = {k: v.unsqueeze(0).to(dev) for k, v in m.items() if k != 'seg'}
    res = {}
    for name, cls in (('fused', TransitionEngine), ('staged', SlabEngine)):
        eng = cls(EngineConfig(dims=dims, seed=1), dev)
        fd, md = eng.prepare(fx, mv)
        eng.gmm_init(fd, md)
        v = torch.zeros(1, 3, *dims, device=dev)
        for _ in range(3):
            eng.transition(fd, md, v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.transition(fd, md, v)
        torch.cuda.synchronize()
        res[name] = 1e3 * (time.perf_counter() - t0) / 20
        del eng
    print(f'N {N}: fused {res["fused"]:.3f} ms  staged (1 rank, no exchange) {res["staged"]:.3f} ms', flush=True)
