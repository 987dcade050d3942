"""The SSD side run of bench.py (256^3, 5 warm-up + 25 timed transitions from rest: the chain crosses the one- and two-voxel bounds of its
last squaring steps inside this window) -- ms per transition, transitions re-run after a failed prediction, per-transition times.
python tools/debug/ssd_window.py [--reps 3]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import TransitionEngine

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--size', type=int, default=256)
a = ap.parse_args()
dev = torch.device('cuda', 0)
N = a.size
for rep in range(a.reps):
    eng = TransitionEngine(bench.engine_config(N, 'ssd', 1234, 1), dev)
    f1, m1 = synthetic_pair((N, N, N), seed=0)
    fixed, moving = eng.prepare({k: v.unsqueeze(0).to(dev) for k, v in f1.items() if k != 'seg'},
                                {k: v.unsqueeze(0).to(dev) for k, v in m1.items() if k != 'seg'})
    eng.gmm_init(fixed, moving)
    v = torch.zeros(1, 3, N, N, N, device=dev)
    for _ in range(5):
        eng.transition(fixed, moving, v)
    torch.cuda.synchronize()
    r0 = eng.recovered_transitions
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(26)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(25):
        eng.transition(fixed, moving, v)
        ev[i + 1].record()
    eng.flush()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 25
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(25)]
    print(f'rep {rep}: {ms:.4f} ms per transition, re-run transitions in the window: {eng.recovered_transitions - r0} (before it: {r0});',
          'per transition:', ' '.join(f'{x:.2f}' for x in per), flush=True)
    del eng
    torch.cuda.empty_cache()
