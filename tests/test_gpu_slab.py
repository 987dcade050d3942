"""z-slab decomposition (ir_sgmcmc_amd/slab.py): the staged, windowed transition must reproduce the fused one.

1 rank: the stage sequence over the full window == irs_transition.
2 ranks: two processes share cuda:0 and exchange ghost planes over gloo (host staging) -- the same code path that uses
RCCL point-to-point on a multi-GPU node; the assembled result must match the single-engine transition."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(N, C, data_loss, seed=0):
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.engine import EngineConfig
    from oracle import ops as O
    cfg = EngineConfig(dims=(N, N, N), no_chains=C, data_loss=data_loss, virtual_decimation=True, lcc_s=1,
                       reg_loss='RegLoss_LogNormal', reg_learnable=True, seed=seed)
    f1, m1 = synthetic_pair((N, N, N), seed=3)
    fixed = {k: v.unsqueeze(0).to(DEV) for k, v in f1.items() if k != 'seg'}
    moving = {k: v.unsqueeze(0).to(DEV) for k, v in m1.items() if k != 'seg'}
    g = torch.Generator().manual_seed(17)
    v0 = O.separable_conv3d_replicate(9.0 * torch.randn(C, 3, N, N, N, generator=g), O.sobolev_kernel_1d(3, 0.5)).contiguous()
    noise = [(torch.randn(C, 3, N, N, N, generator=g), torch.rand(C, 3, N, N, N, generator=g)) for _ in range(2)]
    return cfg, fixed, moving, v0, noise


def _run(engine_cls, cfg, fixed, moving, v0, noise):
    eng = engine_cls(cfg, DEV)
    fd, md = eng.prepare(fixed, moving)
    eng.gmm_init(fd, md)
    v = v0.to(DEV).contiguous()
    disp = torch.zeros(cfg.no_chains, 3, *cfg.dims, device=DEV)
    scal = []
    for eps, unif in noise:
        eng.transition(fd, md, v, None, eps.to(DEV), unif.to(DEV), {'displacement': disp})
        scal.append(eng.scalars())
    return eng, v, disp, scal


@pytest.mark.parametrize('data_loss,C', [('GMM', 1), ('SSD', 2)])
def test_slab_single_rank_equals_fused(data_loss, C):
    from ir_sgmcmc_amd.engine import TransitionEngine
    from ir_sgmcmc_amd.slab import SlabEngine
    cfg, fixed, moving, v0, noise = _setup(24, C, data_loss)
    _, v_ref, d_ref, s_ref = _run(TransitionEngine, cfg, fixed, moving, v0, noise)
    eng, v, d, s = _run(SlabEngine, cfg, fixed, moving, v0, noise)
    # identical kernels; only the fp64 summation order of the partial sums differs (reduce -> all-reduce -> scalar kernel)
    assert float((v - v_ref).abs().max()) <= 1e-5 * float(v_ref.abs().max())
    assert float((d - d_ref).abs().max()) <= 1e-5
    for a, b in zip(s, s_ref):
        for key in ('alpha', 'data_term', 'reg_term', 'reg_energy'):
            np.testing.assert_allclose(a[key], b[key], rtol=1e-6)
    assert eng.state().iteration == 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, data_loss, C, N):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from ir_sgmcmc_amd.engine import TransitionEngine
        from ir_sgmcmc_amd.slab import SlabEngine
        cfg, fixed, moving, v0, noise = _setup(N, C, data_loss)
        eng, v, d, s = _run(SlabEngine, cfg, fixed, moving, v0, noise)
        assert (eng.a, eng.b) == ((rank * N) // world, ((rank + 1) * N) // world) and eng.exchanged_planes > 0
        assert getattr(eng, "mispredictions", 0) == 0  # second transition ran in predicted-width mode
        v_full = eng.gather_slabs(v)
        d_full = eng.gather_slabs(d)
        if rank == 0:
            _, v_ref, d_ref, s_ref = _run(TransitionEngine, cfg, fixed, moving, v0, noise)
            dv = float((v_full - v_ref).abs().max()) / float(v_ref.abs().max())
            dd = float((d_full - d_ref).abs().max())
            ds = max(abs(a[k][c] - b[k][c]) / max(abs(b[k][c]), 1e-30) for a, b in zip(s, s_ref)
                     for k in ('alpha', 'data_term', 'reg_term') for c in range(C))
            q.put((dv, dd, ds, eng.exchanged_planes))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('data_loss,C,N,world', [('GMM', 1, 32, 2), ('SSD', 2, 24, 2), ('GMM', 1, 36, 3)])
def test_slab_ranks_exchange_ghost_planes(data_loss, C, N, world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, data_loss, C, N)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    dv, dd, ds, planes = q.get(timeout=10)
    from tests._report import check
    T = f'slab/{data_loss}_C{C}_N{N}_ranks{world}'
    check(T, 'v_new (rel to max)', dv, 0.0, 1e-5)
    check(T, 'displacement [voxels]', dd, 0.0, 1e-5)
    check(T, 'loss terms (rel)', ds, 0.0, 1e-6)
    assert planes > 0
