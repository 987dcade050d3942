"""transition time against the number of chains held by ONE engine (the reference's configs use two chains)"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine
from ir_sgmcmc_amd.data_loader import synthetic_pair
dev = 'cuda:0'
for N in (128, 256):
    f, m = synthetic_pair((N, N, N), seed=0)
    for C in (1, 2, 4):
        fx = {k: v.unsqueeze(0).to(dev) for k, v in f.items() if k != 'seg'}
        mv = {k: v.unsqueeze(0).to(dev) for k, v in m.items() if k != 'seg'}
        eng = TransitionEngine(EngineConfig(dims=(N, N, N), no_chains=C, seed=5), dev)
        fd, md = eng.prepare(fx, mv)
        eng.gmm_init(fd, md)
        v = torch.zeros(C, 3, N, N, N, device=dev)
        for _ in range(3):
            eng.transition(fd, md, v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            eng.transition(fd, md, v)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'N {N} chains {C}: {dt * 1e3:.3f} ms per transition, {dt * 1e3 / C:.3f} ms per chain, {C / dt:.1f} samples/s', flush=True)
        del eng
