"""Where does a slab chain leave the fused engine's?  Per-plane deviation of v after T transitions (ranks sharing cuda:0 over the
ipc transport), for a test configuration of tests/test_gpu_slab.py.

    python tools/debug/slab_diff.py [--world 3] [--N 30] [--ghost-max 6] [--amp 12] [--exact 0|1] [--T 3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q, a):
    import faulthandler
    import time
    import torch
    import torch.distributed as dist
    faulthandler.dump_traceback_later(a.watchdog, exit=True)  # a rank that stops making progress says where
    t_start = time.perf_counter()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ir_sgmcmc_amd.slab import SlabComm, SlabEngine
    from tests.test_gpu_slab import _run_fused, _setup
    torch.cuda.set_device(0)
    comm = SlabComm.create(a.transport, 'cuda:0')
    N = tuple(a.dims) if a.dims else a.N
    cfg, fixed, moving, v0, noise = _setup(N, 1, 'GMM', amp=a.amp, reg=a.reg, transitions=a.T)
    eng = SlabEngine(cfg, 'cuda:0', comm, ghost_max=a.ghost_max)
    if a.exact:
        eng.option('slab_exact', 1)
    fd, md = eng.prepare(fixed, moving)
    eng.gmm_init(fd, md)
    v = eng.local_v(v0)
    disp, gv = eng.new_local(3), eng.new_local(3)
    hist = []
    for eps, unif in noise:
        eng.transition(fd, md, v, None, eng.local_v(eps), eng.local(unif), {'displacement': disp, 'grad_v': gv})
        if a.final_only and len(hist) + 1 < len(noise):
            hist.append(None)   # (soak runs: thousands of exchanges back to back, nothing synchronises in between)
            continue
        st = eng.status()
        hist.append((eng.gather(v), eng.gather(disp), eng.gather(gv)))
        if rank == 0:
            print(f'[{time.perf_counter() - t_start:7.2f} s] transition done: rounds fwd', st['last_fwd_rounds'], 'bwd', st['last_bwd_rounds'], 'exact so far', st['exact_transitions'],
                  'mispredictions', st['mispredictions'], flush=True)
    if rank == 0:
        # the fused engine, transition by transition
        from ir_sgmcmc_amd.engine import TransitionEngine
        ref = TransitionEngine(cfg, 'cuda:0')
        f2, m2 = ref.prepare({k: t.to('cuda:0') for k, t in fixed.items()}, {k: t.to('cuda:0') for k, t in moving.items()})
        ref.gmm_init(f2, m2)
        vr = v0.to('cuda:0').contiguous()
        dr, gr = torch.zeros_like(vr), torch.zeros_like(vr)
        D = N if isinstance(N, int) else N[0]
        edge = lambda z: '  <- slab edge' if z % (D // world) in (0, D // world - 1) else ''
        for t, (eps, unif) in enumerate(noise):
            ref.transition(f2, m2, vr, None, eps.to('cuda:0'), unif.to('cuda:0'), {'displacement': dr, 'grad_v': gr})
            ref.flush()
            if hist[t] is None:
                continue
            print(f'--- after transition {t}: per-plane max deviation (v rel. to max | displacement [voxels] | grad_v rel. to max)')
            dv = (hist[t][0] - vr.cpu()).abs().amax(dim=(0, 1, 3, 4)) / float(vr.abs().max())
            dd = (hist[t][1] - dr.cpu()).abs().amax(dim=(0, 1, 3, 4))
            dg = (hist[t][2] - gr.cpu()).abs().amax(dim=(0, 1, 3, 4)) / float(gr.abs().max())
            rel = (hist[t][2] - gr.cpu()).abs() / float(gr.abs().max())
            print(f'    grad_v elements beyond 1e-5 of max: {int((rel > 1e-5).sum())} of {rel.numel()} (beyond 1e-6: {int((rel > 1e-6).sum())})')
            for z in range(D):
                if max(dv[z], dd[z], dg[z]) > 1e-5 or edge(z):
                    print(f'  z {z:3d}  {float(dv[z]):.2e} | {float(dd[z]):.2e} | {float(dg[z]):.2e}' + edge(z))
    dist.barrier()
    del eng
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    import socket

    import torch.multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=3)
    ap.add_argument('--N', type=int, default=30)
    ap.add_argument('--dims', type=int, nargs=3, default=None, help='D H W (instead of --N)')
    ap.add_argument('--reg', default='RegLoss_LogNormal')
    ap.add_argument('--ghost-max', type=int, default=6)
    ap.add_argument('--amp', type=float, default=12.0)
    ap.add_argument('--exact', type=int, default=0)
    ap.add_argument('--T', type=int, default=3)
    ap.add_argument('--transport', default='ipc')
    ap.add_argument('--watchdog', type=int, default=150)
    ap.add_argument('--final-only', type=int, default=0, help='compare after the last transition only (soak run)')
    a = ap.parse_args()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, a.world, port, q, a)) for r in range(a.world)]
    [p.start() for p in ps]
    [p.join(200) for p in ps]
    [p.kill() for p in ps if p.is_alive()]
