"""Does replaying the transition as a HIP graph shorten the gaps between its ~50 dependent launches?
Captures one `irs_transition` (torch.cuda.CUDAGraph on the current stream; the library only launches on that stream) and
compares replay time with plain launches.  Usage: python tools/graph_probe.py [N ...]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine

dev = torch.device('cuda', 0)
for N in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    dims = (N, N, N)
    eng = TransitionEngine(EngineConfig(dims=dims, seed=1), dev)
    f, m = synthetic_pair(dims, seed=0)
    fx = {k: v.unsqueeze(0).to(dev) for k, v in f.items() if k != 'seg'}
    mv = {k: v.unsqueeze(0).to(dev) for k, v in m.items() if k != 'seg'}
    fd, md = eng.prepare(fx, mv)
    eng.gmm_init(fd, md)
    v = torch.zeros(1, 3, *dims, device=dev)
    for _ in range(5):
        eng.transition(fd, md, v)
    torch.cuda.synchronize()

    def timeit(fn, n=30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    plain = timeit(lambda: eng.transition(fd, md, v))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.transition(fd, md, v)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.transition(fd, md, v)
    graph = timeit(g.replay)
    print(f'N {N}: plain {plain:.3f} ms  graph replay {graph:.3f} ms  finite {bool(torch.isfinite(v).all())}', flush=True)
    del eng, g
