"""Compute-side cost of ONE rank of the slab schedule, measured on one GPU: a slab context of rank r of `world` with a
transport that moves nothing (the ghost planes keep whatever they hold -- results are meaningless, the launch sequence,
the windows and the kernel work are exactly those of a real rank).  What an N-GPU run can reach at best, before any
communication time: t(1 GPU fused) / t(one rank of N).

    python tools/slab_probe.py [--size 256] [--loss gmm|ssd] [--worlds 1,2,4,8] [--ghost-max 4]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--loss', default='gmm')
    ap.add_argument('--worlds', default='1,2,4,8')
    ap.add_argument('--ghost-max', type=int, default=0)
    ap.add_argument('--steps', type=int, default=30)
    args = ap.parse_args()
    import torch
    from bench import engine_config
    from ir_sgmcmc_amd import _lib as L
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.engine import TransitionEngine
    from ir_sgmcmc_amd.slab import SlabComm, SlabEngine
    dev = torch.device('cuda', 0)
    lib = L.load()
    N = args.size
    f1, m1 = synthetic_pair((N, N, N), seed=0)
    fixed = {k: v.unsqueeze(0) for k, v in f1.items() if k != 'seg'}
    moving = {k: v.unsqueeze(0) for k, v in m1.items() if k != 'seg'}
    out = {}

    def timeit(eng, fd, md, v):
        for _ in range(5):
            eng.transition(fd, md, v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.transition(fd, md, v)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / args.steps

    eng = TransitionEngine(engine_config(N, args.loss, 1), dev)
    fd, md = eng.prepare({k: v.to(dev) for k, v in fixed.items()}, {k: v.to(dev) for k, v in moving.items()})
    eng.gmm_init(fd, md)
    st_ref = eng.state()
    out['fused'] = timeit(eng, fd, md, torch.zeros(1, 3, N, N, N, device=dev))
    del eng
    ex = L.EXCHANGE_FN(lambda user, x, n, stream: 0)
    ar = L.ALLREDUCE_FN(lambda user, buf, count, mx, stream: 0)
    for world in [int(w) for w in args.worlds.split(',')]:
        rank = world // 2
        h = C.c_void_p()
        L.check(lib.irs_comm_create_callbacks(ex, ar, None, rank, world, C.byref(h)))
        comm = SlabComm(h, rank, world, keep=(ex, ar))
        eng = SlabEngine(engine_config(N, args.loss, 1), dev, comm, ghost_max=args.ghost_max)
        fd, md = eng.prepare(fixed, moving)
        eng.set_state(st_ref)  # the mixture of the whole volume (a partial all-reduce would initialise a different one)
        v = eng.new_local(3)
        ms = timeit(eng, fd, md, v)
        st = eng.status()
        out[f'rank_of_{world}'] = {'ms': ms, 'planes': eng.b - eng.a, 'held': eng.hi - eng.lo, 'fwd_rounds': st['last_fwd_rounds'],
                                   'bwd_rounds': st['last_bwd_rounds'], 'best_case_speedup': out['fused'] / ms}
        del eng
        comm.close()
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
