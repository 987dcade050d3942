#!/usr/bin/env python3
"""bench.py -- SG-MCMC transitions/s of the HIP path on synthetic volumes (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--loss gmm|ssd]

A "step" is one full `_SGLD_transition` (noise + Sobolev, 12-step scaling and squaring, warp, LCC/GMM data term with
virtual decimation, regulariser, backward, SGLD update) over one synthetic fixed/moving pair resident in HBM, with
in-kernel Philox noise.  One process per GPU; for N > 1 the driver launches this file under torch.distributed.run and
every rank samples its own chain of the same pair (no data-path collective: chains only share hyper-parameters in the
reference, trainer.py:316-327) -> weak scaling in chains; value = all ranks' transitions / max-over-ranks wall time.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the adjoint of one squaring step): algorithmic
bytes per launch (36 B/voxel: read dL/dd_{k+1} 12 + read d_k 12 + write dL/dd_k 12) over its average duration measured
with HIP events on the launch stream inside `irs_transition_timed`.  `cpu_baseline` times the CPU oracle (torch, all
host cores) on a bounded sample and is a reported baseline, not a target.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_VOXEL = {'gmm': 878.0, 'ssd': 854.0}  # algorithmic bytes / voxel / chain / transition
BWD_STEP_BYTES_PER_VOXEL = 36.0
FWD_STEP_BYTES_PER_VOXEL = 24.0


def pmc_traffic_bytes(n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate FETCH_SIZE and
    WRITE_SIZE runs of this same command at 256^3; profiles/r01_v10_pmc_traffic.json).  Corrections as the
    MI355X guide prescribes: counters are KiB; FETCH_SIZE reads exactly 1/2 of the bytes of this kernel's dword-per-lane
    loads (calibrated in the same run on perturb_kernel, whose read volume is known); WRITE_SIZE is exact.
    None for sizes that were not profiled."""
    path = os.path.join(ROOT, 'profiles', 'r01_v10_pmc_traffic.json')
    if n != 256 or not os.path.isfile(path):
        return None
    t = json.load(open(path))
    for k, v in t.items():
        if 'exp_bwd_march_kernel<false, 1>' in k:
            return (2.0 * v['FETCH_SIZE_KiB_raw'] + v['WRITE_SIZE_KiB_raw']) * 1024.0
    return None


def cpu_baseline(n_small, reps, loss):
    """the CPU oracle (op-for-op the reference's ATen sequence) on a bounded sample, all host cores"""
    import torch
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from oracle import OracleChain, OracleConfig
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get('IRS_CPU_THREADS', '16'))))  # a 1-GPU box owns a 16-core share
    torch.set_num_threads(cores)
    print(f'[bench] cpu baseline: oracle at {n_small}^3 on {cores} threads ...', file=sys.stderr, flush=True)
    dims = (n_small,) * 3
    cfg = OracleConfig(dims=dims, data_loss='GMM' if loss == 'gmm' else 'SSD', virtual_decimation=(loss == 'gmm'))
    f1, m1 = synthetic_pair(dims, seed=0)
    fixed = {k: v.unsqueeze(0) for k, v in f1.items() if k != 'seg'}
    moving = {k: v.unsqueeze(0) for k, v in m1.items() if k != 'seg'}
    ch = OracleChain(cfg)
    ch.init_gmm(fixed, moving)
    g = torch.Generator().manual_seed(0)
    eps = torch.randn(1, 3, *dims, generator=g)
    unif = torch.rand(1, 3, *dims, generator=g)
    ch.transition(fixed, moving, eps, unif)  # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    for i in range(reps):
        ch.transition(fixed, moving, eps, unif)
        print(f'[bench] cpu baseline: {i + 1}/{reps} after {time.perf_counter() - t0:.1f} s', file=sys.stderr, flush=True)
    dt = (time.perf_counter() - t0) / reps
    return dt, cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=256, help='volume edge N (N^3 voxels)')
    ap.add_argument('--loss', choices=['gmm', 'ssd'], default='gmm')
    ap.add_argument('--init', choices=['identity', 'smooth', 'wave'], default='identity',
                    help="chain start: identity (MCMC_init 'identity', sub-voxel displacements) or a smooth random velocity "
                         "field of --init-amp voxels (exercises the large-displacement kernel variants)")
    ap.add_argument('--init-amp', type=float, default=3.0)
    ap.add_argument('--decomp', choices=['chains', 'slab'], default='chains',
                    help='N > 1: independent chains per rank (weak scaling, default) or ONE chain split into z-slabs with '
                         'ghost-plane exchange (strong scaling, ir_sgmcmc_amd/slab.py)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-size', type=int, default=64)
    ap.add_argument('--cpu-reps', type=int, default=5)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    # IRS_BENCH_DEVICE / IRS_BENCH_BACKEND exist only to rehearse the multi-rank path on a 1-GPU box (gloo, shared device)
    dev_index = int(os.environ.get('IRS_BENCH_DEVICE', local_rank))
    backend = os.environ.get('IRS_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == 'nccl' else torch.device('cpu')

    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.engine import EngineConfig, TransitionEngine

    N = args.size
    dims = (N, N, N)
    V = N ** 3
    cfg = EngineConfig(dims=dims, no_chains=1, data_loss='GMM' if args.loss == 'gmm' else 'SSD',
                       virtual_decimation=(args.loss == 'gmm'), reg_loss='RegLoss_L2', w_reg=1.4, seed=1234 + rank)
    slab = args.decomp == 'slab' and world > 1
    if slab:
        from ir_sgmcmc_amd.slab import SlabEngine
        cfg.seed = 1234  # one chain: every rank must draw the same Philox noise
        eng = SlabEngine(cfg, dev)
    else:
        eng = TransitionEngine(cfg, dev)
    f1, m1 = synthetic_pair(dims, seed=0)
    fixed = {k: v.unsqueeze(0).to(dev) for k, v in f1.items() if k != 'seg'}
    moving = {k: v.unsqueeze(0).to(dev) for k, v in m1.items() if k != 'seg'}
    fixed, moving = eng.prepare(fixed, moving)
    eng.gmm_init(fixed, moving)
    v = torch.zeros(1, 3, *dims, device=dev)  # MCMC_init: identity (trainer.py:596-598)
    if args.init == 'smooth':
        from ir_sgmcmc_amd.ops import perturb_smooth, sobolev_kernel_1d
        g = torch.Generator(device='cpu').manual_seed(7)
        lo = torch.randn(1, 3, N // 8, N // 8, N // 8, generator=g)
        v = torch.nn.functional.interpolate(lo, size=dims, mode='trilinear', align_corners=True).to(dev).contiguous()
        v = perturb_smooth(v, sobolev_kernel_1d(3, 0.5)) * (args.init_amp / float(v.abs().max()))
        v = v.contiguous()
    elif args.init == 'wave':  # one half-wave across the volume: large but smooth, like a converged registration
        t = torch.linspace(0.0, math.pi, N, device=dev)
        sz, sy, sx = torch.sin(t).view(N, 1, 1), torch.sin(t).view(1, N, 1), torch.sin(t).view(1, 1, N)
        v = torch.stack([sz * sy * sx, -sz * sy * sx, 0.7 * sz * sy * sx]).unsqueeze(0).mul(args.init_amp).contiguous()

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        eng.transition(fixed, moving, v)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.transition(fixed, moving, v)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(v).all()), 'chain diverged'

    # ---- side measurements on rank 0 of a single-GPU run (outside the timed region) ------------------------------------
    extras = {}
    if world == 1 and dev.type == 'cuda':
        # (1) the reference's own speed definition (trainer.py:467-476): every sample also warps the segmentation
        #     (nearest neighbour) with the sampled transformation
        from ir_sgmcmc_amd.ops import warp
        seg = f1['seg'].unsqueeze(0).to(dev).contiguous() if 'seg' in f1 else torch.zeros(1, 1, *dims, dtype=torch.int16, device=dev)
        outs = {'transformation': torch.empty(1, 3, *dims, device=dev), 'displacement': torch.empty(1, 3, *dims, device=dev)}
        for _ in range(2):
            eng.transition(fixed, moving, v, outputs=outs)
            warp(seg, outs['transformation'])
        sync()
        n_seg = max(3, args.steps // 2)
        t1 = time.perf_counter()
        for _ in range(n_seg):
            eng.transition(fixed, moving, v, outputs=outs)
            warp(seg, outs['transformation'])
        sync()
        extras['with_transformation_output_and_seg_warp'] = {'value': n_seg / (time.perf_counter() - t1), 'unit': 'transitions/s',
                                                             'definition': 'trainer.py:467-476 (transition + nearest-neighbour warp of the segmentation)'}
        # (2) what a plain device copy of one field reaches on this box (planar fp32, same bytes as one squaring step reads + writes)
        a, b = torch.empty(3 * V, device=dev), torch.empty(3 * V, device=dev)
        for _ in range(3):
            b.copy_(a)
        sync()
        t1 = time.perf_counter()
        for _ in range(20):
            b.copy_(a)
        sync()
        extras['device_copy_GBps'] = 20 * 2 * 3 * V * 4 / (time.perf_counter() - t1) / 1e9
        del a, b, outs

    # per-stage HIP-event timings (outside the timed region), averaged over a few transitions
    reps = max(3, min(10, args.steps))
    acc = None
    timing_eng = eng
    if slab:  # per-stage events come from the fused path: time them on a plain engine (same kernels, full window)
        timing_eng = TransitionEngine(cfg, dev)
        timing_eng.prepare(fixed, moving)
        timing_eng.gmm_init(fixed, moving)
        v = torch.zeros(1, 3, *dims, device=dev)
    for _ in range(reps):
        tm = timing_eng.transition(fixed, moving, v, timed=True)
        acc = tm if acc is None else {k: acc[k] + tm[k] for k in tm}
    tm = {k: x / reps for k, x in acc.items()}
    steps = cfg.no_steps
    bwd_kernel_ms = tm['exp_bwd_primary_avg_ms']  # mean duration of exp_bwd_march_kernel<false,1> alone (HIP events)
    fwd_kernel_ms = tm['exp_fwd_ms'] / steps

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = (1 if slab else world) * args.steps / elapsed
        achieved = BWD_STEP_BYTES_PER_VOXEL * V / (bwd_kernel_ms * 1e-3) / 1e9
        out = {
            'metric': 'SG-MCMC transitions/sec (full _SGLD_transition, 1 chain per GPU)', 'value': value,
            'unit': 'transitions/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if slab else 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{N}^3 synthetic pair, SVF_3D 12 steps, '
                                   + ('GMM(K=4)/LCC(s=1) + virtual decimation' if args.loss == 'gmm' else 'SSD')
                                   + ', RegLoss_L2 w=1.4, Sobolev s=3, uniform noise 0.1, SGLD lr 0.4, Philox noise',
                       'volume': [N, N, N], 'init': args.init + (f' amp {args.init_amp}' if args.init == 'smooth' else ''),
                       'chains_per_gpu': 1,
                       'parallelism': (f'1 chain in {world} z-slabs, ghost-plane exchange' if slab else f'{world} independent chain(s)')},
            'roofline': {'bound': 'hbm', 'kernel': 'exp_bwd_march_kernel<false,1> (adjoint of one squaring step)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': pmc_traffic_bytes(N), 'algorithmic_bytes_per_launch': BWD_STEP_BYTES_PER_VOXEL * V,
                         'avg_launch_ms': bwd_kernel_ms,
                         'note': 'priced against HBM as the path prescribes; the kernel itself is bound by vector-instruction issue: '
                                 '80.4 M wave64 VALU instructions x 4 cycles / 1024 SIMDs = 150 us of its 203 us at 256^3 '
                                 '(profiles/r01_v8_sq_counters.json, DESIGN.md section 4)'},
            'transition_roofline': {'algorithmic_bytes': BYTES_PER_VOXEL[args.loss] * V,
                                    'frac_of_device_copy': (BYTES_PER_VOXEL[args.loss] * V / (ms_per_step * 1e-3) / 1e9 / extras['device_copy_GBps']) if 'device_copy_GBps' in extras else None,
                                    'achieved_GBps': BYTES_PER_VOXEL[args.loss] * V / (ms_per_step * 1e-3) / 1e9,
                                    'frac_of_8TBps': BYTES_PER_VOXEL[args.loss] * V / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            'exp_step_fwd': {'avg_launch_ms': fwd_kernel_ms,
                             'achieved_GBps': FWD_STEP_BYTES_PER_VOXEL * V / (fwd_kernel_ms * 1e-3) / 1e9},
            'stage_ms': tm, 'workspace_GB': eng.workspace_bytes / 1e9, **extras,
        }
        if not args.no_cpu_baseline and world == 1:
            n_small = min(args.cpu_size, N)
            dt, cores = cpu_baseline(n_small, args.cpu_reps, args.loss)
            scale = (N / n_small) ** 3
            out['cpu_baseline'] = {'value': 1.0 / (dt * scale), 'unit': 'transitions/s', 'cores': cores, 'kind': 'port',
                                   'sample': f'{args.cpu_reps} transitions of the torch-CPU oracle at {n_small}^3 '
                                             f'({dt:.2f} s each, {cores} threads), scaled by voxel count x{scale:.0f} to {N}^3'}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
