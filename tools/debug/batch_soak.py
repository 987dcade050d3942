"""Long chains in the reference's regime (128^3, two chains in one engine): the one-launch data term (`data_batch` 1) against the
serial form (0) -- the velocity and the hyper-parameter state after T transitions must be bit-identical, at rest and started 6 voxels
away.  python tools/debug/batch_soak.py [--T 1500]"""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import TransitionEngine

ap = argparse.ArgumentParser()
ap.add_argument('--T', type=int, default=1500)
ap.add_argument('--size', type=int, default=128)
a = ap.parse_args()
dev = torch.device('cuda', 0)
N, C = a.size, 2
h = lambda b: hashlib.sha256(b).hexdigest()[:16]
ok = True
for amp in (0.0, 6.0):
    out = {}
    for mode in (1, 0):
        eng = TransitionEngine(bench.engine_config(N, 'gmm', 77, C), dev)
        eng.option('data_batch', mode)
        f1, m1 = synthetic_pair((N, N, N), seed=0)
        fd, md = eng.prepare({k: v.unsqueeze(0).to(dev) for k, v in f1.items() if k != 'seg'},
                             {k: v.unsqueeze(0).to(dev) for k, v in m1.items() if k != 'seg'})
        eng.gmm_init(fd, md)
        v = bench.initial_velocity('wave' if amp else 'identity', amp, N, dev).expand(C, 3, N, N, N).contiguous()
        for _ in range(a.T):
            eng.transition(fd, md, v)
        eng.flush()
        torch.cuda.synchronize()
        st = eng.state()
        out[mode] = (h(v.cpu().numpy().tobytes()), h(bytes(st)), eng.recovered_transitions,
                     bool(torch.isfinite(v).all()))
        del eng
    same = out[0][:2] == out[1][:2]
    ok = ok and same and out[0][3] and out[1][3]
    print(f'{N}^3 C={C} start {amp} voxels away, {a.T} transitions: data_batch 1 {out[1]}  0 {out[0]}  -> {"bit-identical" if same else "DIFFERENT"}', flush=True)
sys.exit(0 if ok else 1)
