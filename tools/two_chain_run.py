"""One engine, 128^3, two chains (the regime of every reference config: configs/experiment1/config.json) -- N transitions, nothing else;
the program to put behind `rocprofv3 --kernel-trace --stats --` or IRS_LAUNCH_LOG=1.

    python tools/two_chain_run.py [--size 128] [--chains 2] [--steps 100]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=128)
ap.add_argument('--chains', type=int, default=2)
ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--init', default='identity', help="'wave': a chain started --amp voxels away (bench.py: initial_velocity)")
ap.add_argument('--amp', type=float, default=0.0)
a = ap.parse_args()
r = bench.side_run(a.size, 'gmm', a.init, a.amp, a.steps, 10, torch.device('cuda', 0), chains=a.chains)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (dict, list))})
