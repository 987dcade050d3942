"""How long does the HOST take to enqueue one transition (no synchronisation), next to what the GPU takes to run it?
    python tools/enqueue_probe.py [--sizes 64 128 256]
A chain is host-bound where the first exceeds the second."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', type=int, nargs='+', default=[64, 128, 256])
    ap.add_argument('--steps', type=int, default=200)
    a = ap.parse_args()
    import torch
    from bench import engine_config, initial_velocity
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.engine import TransitionEngine
    dev = torch.device('cuda', 0)
    for N in a.sizes:
        for chains in (1, 2):
            eng = TransitionEngine(engine_config(N, 'gmm', 1234, chains), dev)
            f1, m1 = synthetic_pair((N, N, N), seed=0)
            fixed, moving = eng.prepare({k: v.unsqueeze(0).to(dev) for k, v in f1.items() if k != 'seg'},
                                        {k: v.unsqueeze(0).to(dev) for k, v in m1.items() if k != 'seg'})
            eng.gmm_init(fixed, moving)
            v = initial_velocity('identity', 0.0, N, dev).expand(chains, 3, N, N, N).contiguous()
            for _ in range(20):
                eng.transition(fixed, moving, v)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                eng.transition(fixed, moving, v)
            t1 = time.perf_counter()
            eng.flush()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            print(f'{N}^3 C={chains}: host enqueue {1e3 * (t1 - t0) / a.steps:.3f} ms per call, run {1e3 * (t2 - t0) / a.steps:.3f} ms per call', flush=True)
            del eng
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
