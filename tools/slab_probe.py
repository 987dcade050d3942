"""Compute-side cost of ONE rank of the slab schedule, measured on one GPU: a slab context of rank r of `world` with a
transport that moves nothing (the ghost planes keep whatever they hold -- results are meaningless, the launch sequence,
the windows and the kernel work are exactly those of a real rank).  What an N-GPU run can reach at best, before any
communication time: t(1 GPU fused) / t(one rank of N).

    python tools/slab_probe.py [--size 256] [--loss gmm|ssd] [--worlds 1,2,4,8] [--ghost-max 4]

`--transport ipc --worlds 2,4`: REAL concurrent ranks instead -- `world` processes share the GPU and run one chain in
`world` slabs over the peer-mapped transport (csrc/ipc.hip: asynchronous, nothing synchronises inside an exchange), and then
the same ranks, still concurrent, over the transport that moves nothing.  Both numbers are wall time per transition of the
SLOWEST rank with all ranks competing for the one device (so they are a multiple of a real rank's time, not a speed-up); their
difference is what the hand-overs of a real transport cost on top of the launch sequence: flag round trips, push / drain
kernels, ranks waiting for each other.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _concurrent_worker(rank, world, port, q, size, loss, ghost_max, steps):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from bench import engine_config
        from ir_sgmcmc_amd import _lib as L
        from ir_sgmcmc_amd.data_loader import synthetic_pair
        from ir_sgmcmc_amd.slab import SlabComm, SlabEngine
        torch.cuda.set_device(0)
        dev = torch.device('cuda', 0)
        lib = L.load()
        N = size
        f1, m1 = synthetic_pair((N, N, N), seed=0)
        fixed = {k: v.unsqueeze(0) for k, v in f1.items() if k != 'seg'}
        moving = {k: v.unsqueeze(0) for k, v in m1.items() if k != 'seg'}

        def timeit(eng, fd, md, v):
            # (the velocity goes back to zero after every transition, in both modes: with a transport that moves nothing the ghost
            # planes of the gradient hold garbage, the update carries it into the boundary planes of v, and a chain left to itself
            # drifts until its ghost-width plans outgrow the held margin)
            for i in range(5):
                eng.transition(fd, md, v)
                v.zero_()
            eng.flush()
            v.zero_()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                eng.transition(fd, md, v)
                v.zero_()  # (a memset of the slab-local field, < 0.5 % of a transition, in both modes)
            eng.flush()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return 1e3 * float(t.item()) / steps

        res = {}
        comm = SlabComm.ipc()
        comm.selftest()
        eng = SlabEngine(engine_config(N, loss, 1), dev, comm, ghost_max=ghost_max)
        fd, md = eng.prepare(fixed, moving)
        eng.gmm_init(fd, md)
        state = eng.state()
        res['ipc_ms'] = timeit(eng, fd, md, eng.new_local(3))
        res['transport'] = comm.describe()
        st = eng.status()
        res.update(planes=eng.b - eng.a, held=eng.hi - eng.lo, fwd_rounds=st['last_fwd_rounds'], bwd_rounds=st['last_bwd_rounds'],
                   exchanges_per_transition=st['exchanges'] / max(st['transitions'], 1), mispredictions=st['mispredictions'])
        dist.barrier()
        del eng
        comm.close()
        ex = L.EXCHANGE_FN(lambda user, x, n, stream: 0)
        ar = L.ALLREDUCE_FN(lambda user, buf, count, mx, stream: 0)
        h = C.c_void_p()
        L.check(lib.irs_comm_create_callbacks(ex, ar, None, rank, world, C.byref(h)))
        comm = SlabComm(h, rank, world, keep=(ex, ar))
        eng = SlabEngine(engine_config(N, loss, 1), dev, comm, ghost_max=ghost_max)
        fd, md = eng.prepare(fixed, moving)
        eng.set_state(state)
        res['moves_nothing_ms'] = timeit(eng, fd, md, eng.new_local(3))
        del eng
        comm.close()
        if rank == 0:
            res['handover_cost_ms'] = res['ipc_ms'] - res['moves_nothing_ms']
            q.put(res)
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)
    dist.barrier()
    dist.destroy_process_group()


def concurrent(args):
    """`world` real ranks sharing the GPU (children started before anything here touches it)"""
    import socket

    import torch.multiprocessing as mp
    out = {'mode': 'concurrent ranks sharing one GPU: ms per transition of the slowest rank, all ranks competing for the device',
           'size': args.size, 'loss': args.loss}
    for world in [int(w) for w in args.worlds.split(',')]:
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        procs = [ctx.Process(target=_concurrent_worker, args=(r, world, port, q, args.size, args.loss, args.ghost_max, args.steps)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(400)
        hung = [p for p in procs if p.is_alive()]
        for p in hung:
            p.kill()
        if hung or any(p.exitcode for p in procs):
            out[f'ranks_{world}'] = {'error': f'exit codes {[p.exitcode for p in procs]}'}
            continue
        out[f'ranks_{world}'] = q.get(timeout=10)
    print(json.dumps(out, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--loss', default='gmm')
    ap.add_argument('--worlds', default='1,2,4,8')
    ap.add_argument('--ghost-max', type=int, default=0)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--transport', choices=['none', 'ipc'], default='none')
    args = ap.parse_args()
    if args.transport == 'ipc':
        return concurrent(args)
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
            v.zero_()  # (see the concurrent mode: ghost planes that hold garbage make a chain drift)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            eng.transition(fd, md, v)
            v.zero_()
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
