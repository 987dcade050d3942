"""Duration of every transition of a short run (events on the launch stream between the calls): is the default bench window (20 steps
after 5 warm-up) in the steady state?  python tools/debug/per_transition_times.py [--size 256] [--steps 60]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import TransitionEngine

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--steps', type=int, default=60)
a = ap.parse_args()
dev = torch.device('cuda', 0)
N = a.size
eng = TransitionEngine(bench.engine_config(N, 'gmm', 1234, 1), dev)
f1, m1 = synthetic_pair((N, N, N), seed=0)
fixed, moving = eng.prepare({k: v.unsqueeze(0).to(dev) for k, v in f1.items() if k != 'seg'},
                            {k: v.unsqueeze(0).to(dev) for k, v in m1.items() if k != 'seg'})
eng.gmm_init(fixed, moving)
v = torch.zeros(1, 3, N, N, N, device=dev)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
ev[0].record()
import time
host = []
for i in range(a.steps):
    t0 = time.perf_counter()
    eng.transition(fixed, moving, v)
    host.append(1e3 * (time.perf_counter() - t0))
    ev[i + 1].record()
    if os.environ.get('IRS_LAUNCH_LOG'):
        print(f'[transition {i} enqueued]', file=sys.stderr, flush=True)
eng.flush()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)]
print('per transition [ms]:', ' '.join(f'{x:.3f}' for x in ms))
print('host time of the call [ms]:', ' '.join(f'{x:.2f}' for x in host))
for lo, hi in ((0, 5), (5, 25), (25, 45), (45, a.steps)):
    if hi <= a.steps:
        print(f'transitions {lo}..{hi - 1}: mean {sum(ms[lo:hi]) / (hi - lo):.4f} ms')
