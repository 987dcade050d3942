#!/usr/bin/env python3
"""bench.py -- SG-MCMC transitions/s of the HIP path on synthetic volumes (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--loss gmm|ssd] [--decomp slab|chains]

A "step" is one full `_SGLD_transition` (noise + Sobolev, 12-step scaling and squaring, warp, LCC/GMM data term with
virtual decimation, regulariser, backward, SGLD update) over one synthetic fixed/moving pair resident in HBM, with
in-kernel Philox noise.  One process per GPU.

N > 1 (default `--decomp slab`, "scaling": "strong"): ONE chain, the volume split into N z-slabs; ghost planes travel
between neighbouring GPUs through RCCL send / recv issued by the library on its (high-priority) communication stream, three small
all-reduces per transition carry the displacement bounds and the partial sums (csrc/slab.hip).  value = transitions of that one
chain / max-over-ranks wall time.  `--decomp chains` is the other, trivial, decomposition: independent chains per rank, no
data-path collective (weak scaling).  Without a launcher (no WORLD_SIZE in the environment) `--gpus N` starts the N ranks
itself, as child processes of torch.distributed.run, before anything touches a GPU.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the adjoint of one squaring step): algorithmic
bytes per launch (36 B/voxel: read dL/dd_{k+1} 12 + read d_k 12 + write dL/dd_k 12) over its average duration measured
with HIP events on the launch stream inside `irs_transition_timed`.  `cpu_baseline` times the CPU oracle (torch, all
host cores) on a bounded sample and is a reported baseline, not a target.  `also` carries what a registration costs away
from the headline's best case (N = 1 only): the SSD loss of config 4, a chain started from a displaced field, a chain with a
sigma-field preconditioner (pSGLD, MCMC_init 'VI'), 128^3 with its own roofline fraction, and a sustained run of >= 500 transitions.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_VOXEL = {'gmm': 878.0, 'ssd': 854.0}  # algorithmic bytes / voxel / chain / transition
BWD_STEP_BYTES_PER_VOXEL = 36.0
FWD_STEP_BYTES_PER_VOXEL = 24.0
TRAFFIC_PROFILE = os.path.join('profiles', 'r05_pmc_traffic.json')  # written by tools/profile_round.sh for THIS library build


def traffic_from_profile(n):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed with this round (separate
    FETCH_SIZE / WRITE_SIZE runs of this command at 256^3).  Corrections as the MI355X guide prescribes: counters are KiB;
    FETCH_SIZE reads exactly 1/2 of the bytes of this kernel's dword-per-lane loads (calibrated in the same run on
    perturb_kernel, whose read volume is known); WRITE_SIZE is exact.  None when the profile is absent, of another size, or
    a tuning build of the library (IRS_LIB) is being measured."""
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    if n != 256 or not os.path.isfile(path) or os.environ.get('IRS_LIB'):
        return None, None
    t = json.load(open(path))
    for k, v in t.items():
        if 'exp_bwd_march_kernel<false, 1>' in k:
            return (2.0 * v['FETCH_SIZE_KiB_raw'] + v['WRITE_SIZE_KiB_raw']) * 1024.0, TRAFFIC_PROFILE
    return None, None


def cpu_baseline(n, reps, loss, warmup=None):
    """the CPU oracle (op-for-op the reference's ATen sequence) on a bounded sample, all host cores: `warmup` untimed transitions
    (default: 3 up to 128^3, where they cost seconds; above that one 64^3 transition warms the thread pool and the allocator),
    then `reps` timed ones.  Returns (seconds per transition, cores, per-repetition seconds)."""
    import torch
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from oracle import OracleChain, OracleConfig
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get('IRS_CPU_THREADS', '16'))))  # a 1-GPU box owns a 16-core share
    torch.set_num_threads(cores)
    print(f'[bench] cpu baseline: oracle at {n}^3 on {cores} threads ...', file=sys.stderr, flush=True)
    def setup(m):
        dims = (m,) * 3
        cfg = OracleConfig(dims=dims, data_loss='GMM' if loss == 'gmm' else 'SSD', virtual_decimation=(loss == 'gmm'))
        f1, m1 = synthetic_pair(dims, seed=0)
        fixed = {k: v.unsqueeze(0) for k, v in f1.items() if k != 'seg'}
        moving = {k: v.unsqueeze(0) for k, v in m1.items() if k != 'seg'}
        ch = OracleChain(cfg)
        ch.init_gmm(fixed, moving)
        g = torch.Generator().manual_seed(0)
        return ch, fixed, moving, torch.randn(1, 3, *dims, generator=g), torch.rand(1, 3, *dims, generator=g)

    if warmup is None:
        warmup = 3 if n <= 128 else 0
    ch, fixed, moving, eps, unif = setup(n if warmup else 64)
    for _ in range(max(warmup, 1)):
        ch.transition(fixed, moving, eps, unif)
    if not warmup:
        ch, fixed, moving, eps, unif = setup(n)
    each = []
    for i in range(reps):
        t0 = time.perf_counter()
        ch.transition(fixed, moving, eps, unif)
        each.append(time.perf_counter() - t0)
        print(f'[bench] cpu baseline: {n}^3 {i + 1}/{reps}: {each[-1]:.1f} s', file=sys.stderr, flush=True)
    return sum(each) / reps, cores, each


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of torch.distributed.run (this process
    has not touched a GPU and never will), pass their output through and leave with their exit code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] no launcher in the environment: starting', ' '.join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def initial_velocity(kind, amp, N, dev):
    import torch
    dims = (N, N, N)
    if kind == 'smooth':
        from ir_sgmcmc_amd.ops import perturb_smooth, sobolev_kernel_1d
        g = torch.Generator(device='cpu').manual_seed(7)
        lo = torch.randn(1, 3, N // 8, N // 8, N // 8, generator=g)
        v = torch.nn.functional.interpolate(lo, size=dims, mode='trilinear', align_corners=True).to(dev).contiguous()
        v = perturb_smooth(v, sobolev_kernel_1d(3, 0.5)) * (amp / float(v.abs().max()))
        return v.contiguous()
    if kind == 'wave':  # one half-wave across the volume: large but smooth, like a converged registration
        t = torch.linspace(0.0, math.pi, N, device=dev)
        sz, sy, sx = torch.sin(t).view(N, 1, 1), torch.sin(t).view(1, N, 1), torch.sin(t).view(1, 1, N)
        return torch.stack([sz * sy * sx, -sz * sy * sx, 0.7 * sz * sy * sx]).unsqueeze(0).mul(amp).contiguous()
    return torch.zeros(1, 3, *dims, device=dev)  # MCMC_init: identity (trainer.py:596-598)


def engine_config(N, loss, seed, chains=1):
    from ir_sgmcmc_amd.engine import EngineConfig
    return EngineConfig(dims=(N, N, N), no_chains=chains, data_loss='GMM' if loss == 'gmm' else 'SSD', virtual_decimation=(loss == 'gmm'),
                        reg_loss='RegLoss_L2', w_reg=1.4, seed=seed)


def workload_name(N, loss):
    return (f'{N}^3 synthetic pair, SVF_3D 12 steps, ' + ('GMM(K=4)/LCC(s=1) + virtual decimation' if loss == 'gmm' else 'SSD')
            + ', RegLoss_L2 w=1.4, Sobolev s=3, uniform noise 0.1, SGLD lr 0.4, Philox noise')


def side_run(N, loss, init, amp, steps, warmup, dev, timed_reps=0, chains=1, sigma=None):
    """one fused single-GPU workload outside the headline: ms per transition (+ the per-stage events when asked); `sigma`: the value
    of a preconditioner FIELD (a chain started from the VI posterior, MCMC_init 'VI' -- 14 of the reference's 16 configs)"""
    import torch
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.engine import TransitionEngine
    eng = TransitionEngine(engine_config(N, loss, 1234, chains), dev)
    f1, m1 = synthetic_pair((N, N, N), seed=0)
    fixed, moving = eng.prepare({k: v.unsqueeze(0).to(dev) for k, v in f1.items() if k != 'seg'},
                                {k: v.unsqueeze(0).to(dev) for k, v in m1.items() if k != 'seg'})
    eng.gmm_init(fixed, moving)
    v = initial_velocity(init, amp, N, dev).expand(chains, 3, N, N, N).contiguous()
    sig = torch.full_like(v, float(sigma)) if sigma is not None else None
    for _ in range(warmup):
        eng.transition(fixed, moving, v, sig)
    # A sub-millisecond transition (128^3) leaves the host a fraction of a millisecond to wake up from the run-ahead wait and enqueue
    # the next one: on a box whose cores are busy with somebody else's work one repetition now and then comes out 20 % long.  Three
    # repetitions of the timed loop, the MEDIAN reported, every repetition listed.
    # (The same at 256^3 for the short side runs: 25 transitions are 0.12 s, and a single host stall of 20 ms -- seen once in round 5 on the SSD
    # line: 5.59 ms where eight runs on another box measured 4.68-4.77 -- is 16 % of such a window.)
    reps = 3 if (N <= 128 or steps <= 100) else 1
    rep_ms = []
    for _ in range(reps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.transition(fixed, moving, v, sig)
        eng.flush()
        torch.cuda.synchronize(dev)
        rep_ms.append(1e3 * (time.perf_counter() - t0) / steps / chains)  # per chain: every chain of the batch makes one transition per call
    ms = sorted(rep_ms)[len(rep_ms) // 2]
    assert bool(torch.isfinite(v).all()), 'chain diverged'
    out = {'ms_per_transition': ms, 'transitions_per_s': 1e3 / ms, 'steps': steps, 'warmup': warmup, 'chains_in_engine': chains,
           'repetitions_ms': [round(x, 4) for x in rep_ms],
           'achieved_GBps': BYTES_PER_VOXEL[loss] * N ** 3 / (ms * 1e-3) / 1e9,
           'frac_of_8TBps': BYTES_PER_VOXEL[loss] * N ** 3 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if timed_reps:
        acc = None
        for _ in range(timed_reps):
            tm = eng.transition(fixed, moving, v, sig, timed=True)
            acc = tm if acc is None else {k: acc[k] + tm[k] for k in tm}
        out['stage_ms'] = {k: x / timed_reps for k, x in acc.items()}
        bk = out['stage_ms']['exp_bwd_primary_avg_ms']
        out['dominant_kernel'] = {'avg_launch_ms': bk, 'achieved_GBps': BWD_STEP_BYTES_PER_VOXEL * N ** 3 / (bk * 1e-3) / 1e9,
                                  'frac': BWD_STEP_BYTES_PER_VOXEL * N ** 3 / (bk * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del eng
    torch.cuda.empty_cache()
    return out


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
    ap.add_argument('--decomp', choices=['slab', 'chains'], default='slab',
                    help='N > 1: ONE chain split into z-slabs with ghost-plane exchange over RCCL (strong scaling, default) or '
                         'independent chains per rank (weak scaling)')
    ap.add_argument('--transport', choices=['auto', 'rccl', 'ipc', 'rehearsal'], default=None,
                    help='slab mode: how ghost planes travel.  rccl: ncclSend / ncclRecv groups (csrc/comm.hip); ipc: peer-mapped landing '
                         'buffers written by the producer, sequence flags (csrc/ipc.hip); auto (default on a node): bring up both, time a '
                         'few transitions with each, keep the faster; rehearsal: Python callbacks with host staging (one-GPU tests only). '
                         'IRS_BENCH_BACKEND=ipc|gloo in the environment selects ipc / rehearsal as well')
    ap.add_argument('--allow-chain-fallback', action='store_true',
                    help='slab mode: when no slab transport comes up, measure independent chains per rank instead ("scaling": "weak") '
                         'and exit 0; without this flag that is an error (exit 3): a weak-scaling number must not pass for a point of the '
                         'strong-scaling curve')
    ap.add_argument('--split', choices=['auto', '0', '1'], default='auto',
                    help='slab mode: interior / boundary split around an exchange (overlap with more launches); auto = time both, keep the faster')
    ap.add_argument('--ghost-max', type=int, default=0, help='slab mode: widest ghost zone of one exchange (0 = library default)')
    ap.add_argument('--watchdog', type=int, default=420, help='N > 1: seconds after which a run that has not finished dumps its stacks and exits (0 = off)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the `also` workloads (SSD, displaced start, 128^3, sustained run)')
    ap.add_argument('--cpu-size', type=int, default=256, help='edge of the CPU-baseline volume (capped at --size); 256: one transition takes ~25 s on 16 cores')
    ap.add_argument('--cpu-reps', type=int, default=1)
    ap.add_argument('--cpu-reps-128', type=int, default=5, help='timed 128^3 transitions of the CPU oracle (after 3 warm-up) reported next to the bench-size one; 0 = skip')
    args = ap.parse_args()

    # no launcher (or a stale single-process WORLD_SIZE = 1 in the environment): start the ranks ourselves
    if args.gpus > 1 and int(os.environ.get('WORLD_SIZE', '1')) == 1 and 'TORCHELASTIC_RUN_ID' not in os.environ:
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
            os.environ.pop(k, None)
        sys.exit(spawn_ranks(args.gpus))
    if args.gpus > 1 and args.watchdog > 0:
        # a multi-rank run that stops making progress (a transport that hangs instead of failing) must end with a diagnosis,
        # not sit there until somebody's outer time limit: dump every thread's stack and leave
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True, file=sys.stderr)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    # IRS_BENCH_DEVICE puts every rank on ONE device (a 1-GPU box): torch's own group then runs over gloo (RCCL refuses two
    # ranks on a device) and the slab transport is ipc (asynchronous, the product transport) or the synchronous rehearsal one
    shared_device = 'IRS_BENCH_DEVICE' in os.environ
    dev_index = int(os.environ.get('IRS_BENCH_DEVICE', local_rank))
    env_backend = os.environ.get('IRS_BENCH_BACKEND', '')
    transport = args.transport or {'ipc': 'ipc', 'gloo': 'rehearsal', 'rehearsal': 'rehearsal', 'nccl': 'rccl'}.get(env_backend, 'ipc' if shared_device else 'auto')
    backend = 'gloo' if (shared_device or transport == 'rehearsal') else 'nccl'
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.engine import TransitionEngine
    from ir_sgmcmc_amd.parallel import ChainParallel

    par = ChainParallel()
    N = args.size
    dims = (N, N, N)
    V = N ** 3
    slab = args.decomp == 'slab' and world > 1
    f1, m1 = synthetic_pair(dims, seed=0)
    fixed = {k: v.unsqueeze(0) for k, v in f1.items() if k != 'seg'}
    moving = {k: v.unsqueeze(0) for k, v in m1.items() if k != 'seg'}
    slab_status, slab_failure, trials = None, None, {}
    if slab:
        from ir_sgmcmc_amd.slab import SlabComm, SlabEngine

        def agree(ok):  # a verdict every rank shares
            t = torch.tensor([1 if ok else 0], device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        def bring_up(name):
            """communicator + the library's self-test (all-reduces of each kind and neighbour exchanges of several sizes, checked
            word by word) + a slab engine with the mixture initialised + a short trial; None when it fails on ANY rank"""
            comm_, eng_, err = None, None, None
            try:
                comm_ = SlabComm.create(name, dev)
                comm_.selftest()
            except Exception as e:  # noqa: BLE001
                err = f'{name}: {type(e).__name__}: {e}'
            if not agree(err is None):
                print(f'[bench] rank {rank}: slab transport {name} failed: {err or "on another rank"}', file=sys.stderr, flush=True)
                return None, err or f'{name}: failed on another rank'
            try:
                eng_ = SlabEngine(engine_config(N, args.loss, 1234), dev, comm_, ghost_max=args.ghost_max)
                f_l, m_l = eng_.prepare(fixed, moving)   # fixed image / mask cut to the held planes; the moving image stays whole
                eng_.gmm_init(f_l, m_l)
                v_l = eng_.local(initial_velocity(args.init, args.init_amp, N, dev))
                for _ in range(3):
                    eng_.transition(f_l, m_l, v_l)
                eng_.flush()
                torch.cuda.synchronize(dev)
                dist.barrier()
                t0_ = time.perf_counter()
                for _ in range(8):
                    eng_.transition(f_l, m_l, v_l)
                eng_.flush()
                torch.cuda.synchronize(dev)
                ms = par.max_over_ranks(time.perf_counter() - t0_) * 1e3 / 8
            except Exception as e:  # noqa: BLE001
                err = f'{name}: {type(e).__name__}: {e}'
            if not agree(err is None):
                print(f'[bench] rank {rank}: slab trial over {name} failed: {err or "on another rank"}', file=sys.stderr, flush=True)
                return None, err or f'{name}: trial failed on another rank'
            return (comm_, eng_, f_l, m_l, ms), None

        # Every transport that comes up is verified and timed on a few transitions; the fastest carries the measurement.  One
        # that cannot be brought up (on ANY rank) does not take the run down with it as long as another one works.
        best, failures, preflight_info = None, [], None
        preflight = transport == 'auto' and (not shared_device or os.environ.get('IRS_BENCH_PREFLIGHT') == '1')
        for name in (['ipc', 'rccl'] if transport == 'auto' else [transport]):
            if name == 'ipc' and preflight:
                # one rank per device: the peer-mapped transport's first cross-device stores happen in a CHILD of every rank
                # (ir_sgmcmc_amd/ipc_preflight.py) -- a mapping this node cannot reach costs the children, not the run
                from ir_sgmcmc_amd import ipc_preflight
                box = [f'irs_pre_{os.getpid()}_{int.from_bytes(os.urandom(4), "little"):08x}'] if rank == 0 else [None]
                dist.broadcast_object_list(box, src=0, **({'device': dev} if backend == 'nccl' else {}))
                ok, pre = ipc_preflight.run(box[0], rank, world, dev_index)
                if not agree(ok):
                    err = f'ipc: {pre if not ok else "pre-flight failed on another rank"}'
                    print(f'[bench] rank {rank}: {err}', file=sys.stderr, flush=True)
                    failures.append(err)
                    continue
                preflight_info = pre
                # sequence flags in device memory (one xGMI store + local polls per hand-over instead of two PCIe round trips) where
                # EVERY rank's child saw them work; host flags otherwise (csrc/ipc.hip; IRS_IPC_FLAGS set by hand is left alone)
                if 'IRS_IPC_FLAGS' not in os.environ and agree(bool(pre.get('device_flags'))):
                    os.environ['IRS_IPC_FLAGS'] = 'device'
            got, err = bring_up(name)
            if got is None:
                failures.append(err)
                continue
            trials[name] = got[4]
            if best is not None and got[4] >= best[1][4]:
                loser, got = got, None
            else:
                loser, best = (best[1] if best is not None else None), (name, got)
            if loser is not None:  # engine first (its streams drain), then its communicator (collective)
                comm_l, eng_l = loser[0], loser[1]
                eng_l.flush()
                del loser, eng_l
                torch.cuda.synchronize(dev)
                comm_l.close()
        if failures:
            slab_failure = '; '.join(failures)
        if best is None:
            if not args.allow_chain_fallback:
                print(f'[bench] rank {rank}: no slab transport came up ({slab_failure}); --allow-chain-fallback would measure '
                      f'independent chains instead', file=sys.stderr, flush=True)
                sys.exit(3)
            slab = False
    split_trials = {}
    if slab:
        transport, (comm, eng, fixed, moving, _) = best
        cfg = eng.cfg
        v = eng.local(initial_velocity(args.init, args.init_amp, N, dev))
        # The interior / boundary split around an exchange buys overlap with more, smaller launches (one rank of eight at 256^3:
        # 1.23 against 1.18 ms of launch sequence, tools/slab_probe.py); whether the overlap repays them depends on what a hand-over
        # costs on THIS node.  Measured, not guessed: eight transitions each way on a scratch field, the faster form carries the run
        # (same exchanges, same arithmetic, same chain: tests/test_gpu_slab.py::test_slab_without_the_interior_boundary_split).
        if args.split == 'auto' and transport != 'rehearsal':
            scratch = v.clone()
            for mode in (1, 0):
                eng.option('slab_split', mode)
                for _ in range(2):
                    eng.transition(fixed, moving, scratch)
                eng.flush()
                torch.cuda.synchronize(dev)
                dist.barrier()
                t0_ = time.perf_counter()
                for _ in range(8):
                    eng.transition(fixed, moving, scratch)
                eng.flush()
                torch.cuda.synchronize(dev)
                split_trials['split' if mode else 'unsplit'] = par.max_over_ranks(time.perf_counter() - t0_) * 1e3 / 8
            del scratch
            # (max_over_ranks: every rank holds the same two numbers, so every rank takes the same decision)
            split_on = split_trials['split'] <= split_trials['unsplit']
            eng.option('slab_split', 1 if split_on else 0)
        elif args.split in ('0', '1'):
            split_on = args.split == '1'
            eng.option('slab_split', int(args.split))
        else:
            split_on = True
    else:
        cfg = engine_config(N, args.loss, par.chain_seed(1234))
        eng = TransitionEngine(cfg, dev)
        fixed, moving = eng.prepare({k: v.to(dev) for k, v in fixed.items()}, {k: v.to(dev) for k, v in moving.items()})
        eng.gmm_init(fixed, moving)
        v = initial_velocity(args.init, args.init_amp, N, dev)

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
    # a transition dropped by a failed kernel-variant / ghost-width prediction is re-run by the library; the last two of a run only
    # when asked: inside the timed region, so that the K steps timed are K VALID transitions whatever happened (normally a no-op)
    eng.flush()
    sync()
    elapsed_here = time.perf_counter() - t0
    elapsed = par.max_over_ranks(elapsed_here)
    assert bool(torch.isfinite(v).all()), 'chain diverged'
    slab_diag = None
    if slab:
        slab_status = eng.status()   # mispredictions are reported in the output line (`slab`), not fatal: the chain was repaired
        # What makes a first run on a node diagnosable (outside the timed region): every rank's own time per transition, and the
        # hand-over timeline of one more transition on a scratch copy of the field -- per exchange round and all-reduce, how long the
        # communication stream took from "data ready" to "delivered" and how long the compute stream STALLED at the launch that
        # needs it (irs_slab_timeline_*: HIP events on the two streams the executor already orders).  A slow transport shows as long
        # hand-overs on every rank, load imbalance as stalls on the fast ranks only, a split that does not overlap as stall ~ hand-over.
        mine = {'rank': rank, 'ms_per_transition': 1e3 * elapsed_here / args.steps}
        try:
            scratch = v.clone()
            eng.timeline_arm(2)
            for _ in range(2):
                eng.transition(fixed, moving, scratch)
            eng.flush()
            entries, total_us = eng.timeline()
            del scratch
            mine.update(transition_us=round(total_us, 1), stall_us=round(sum(e['stall_us'] for e in entries), 1),
                        handover_us=round(sum(e['handover_us'] for e in entries), 1),
                        rounds=[[e['kind'][0] + str(e['stage']), e['k'], e['width'], e['handover_us'], e['stall_us']] for e in entries])
        except Exception as e:  # noqa: BLE001  (a diagnostic must not cost the measurement)
            mine['timeline_error'] = f'{type(e).__name__}: {e}'
        try:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)   # (nccl backend: staged through the current device, set above)
            slab_diag = gathered
        except Exception as e:  # noqa: BLE001  (a diagnostic must not cost the measurement: rank 0 then reports itself only)
            mine['gather_error'] = f'{type(e).__name__}: {e}'
            slab_diag = [mine]

    # ---- side measurements on rank 0 (outside the timed region) ------------------------------------------------------------
    extras = {}
    if world == 1:
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

    # per-stage HIP-event timings of the fused single-GPU launch sequence (rank 0, outside the timed region)
    tm = None
    if rank == 0:
        timing_eng, tf, tmv, tv = eng, fixed, moving, v
        if slab:  # the slab engine holds slab-local arrays: the events come from a fused engine on the whole volume
            timing_eng = TransitionEngine(engine_config(N, args.loss, 1234), dev)
            tf, tmv = timing_eng.prepare({k: x.unsqueeze(0).to(dev) for k, x in f1.items() if k != 'seg'},
                                         {k: x.unsqueeze(0).to(dev) for k, x in m1.items() if k != 'seg'})
            timing_eng.gmm_init(tf, tmv)
            tv = torch.zeros(1, 3, *dims, device=dev)
            for _ in range(3):
                timing_eng.transition(tf, tmv, tv)
        reps = max(3, min(10, args.steps))
        acc = None
        for _ in range(reps):
            t = timing_eng.transition(tf, tmv, tv, timed=True)
            acc = t if acc is None else {k: acc[k] + t[k] for k in t}
        tm = {k: x / reps for k, x in acc.items()}
        if slab:
            del timing_eng
    if world > 1:
        dist.barrier()

    if rank == 0:
        steps = cfg.no_steps
        bwd_kernel_ms = tm['exp_bwd_primary_avg_ms']  # mean duration of exp_bwd_march_kernel<false,1> alone (HIP events)
        fwd_kernel_ms = tm['exp_fwd_ms'] / steps
        ms_per_step = 1e3 * elapsed / args.steps
        value = (1 if slab else world) * args.steps / elapsed
        achieved = BWD_STEP_BYTES_PER_VOXEL * V / (bwd_kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = traffic_from_profile(N)
        if slab:
            via = {'rccl': 'RCCL send/recv', 'ipc': 'peer-mapped landing buffers (producer-side stores, sequence flags)' + (' -- all ranks on ONE device' if shared_device else ''),
                   'rehearsal': 'gloo with host staging (one-GPU REHEARSAL of the schedule, not a measurement)'}[transport]
            para = f'1 chain in {world} z-slabs, ghost planes over {via}, {slab_status["last_fwd_rounds"]}+{slab_status["last_bwd_rounds"]} exchange rounds for 2x12 squaring steps'
        else:
            para = f'{world} independent chain(s)'
        out = {
            'metric': 'SG-MCMC transitions/sec (full _SGLD_transition)', 'value': value,
            'unit': 'transitions/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak' if (world > 1 and not slab) else 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload_name(N, args.loss), 'volume': [N, N, N],
                       'init': args.init + (f' amp {args.init_amp}' if args.init != 'identity' else ''),
                       'chains': 1 if (slab or world == 1) else world, 'parallelism': para},
            'roofline': {'bound': 'hbm', 'kernel': 'exp_bwd_march_kernel<false,1> (adjoint of one squaring step)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_from_committed_profile': traffic_src,
                         'algorithmic_bytes_per_launch': BWD_STEP_BYTES_PER_VOXEL * V, 'avg_launch_ms': bwd_kernel_ms,
                         'measured_on': 'whole volume, one GPU (HIP events around the launch)'},
            'transition_roofline': {'algorithmic_bytes': BYTES_PER_VOXEL[args.loss] * V,
                                    'frac_of_device_copy': (BYTES_PER_VOXEL[args.loss] * V / (ms_per_step * 1e-3) / 1e9 / extras['device_copy_GBps']) if 'device_copy_GBps' in extras else None,
                                    'achieved_GBps': BYTES_PER_VOXEL[args.loss] * V / (ms_per_step * 1e-3) / 1e9,
                                    'frac_of_8TBps': BYTES_PER_VOXEL[args.loss] * V / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / (world if slab else 1)},
            'exp_step_fwd': {'avg_launch_ms': fwd_kernel_ms,
                             'achieved_GBps': FWD_STEP_BYTES_PER_VOXEL * V / (fwd_kernel_ms * 1e-3) / 1e9},
            'stage_ms': tm, 'workspace_GB': eng.workspace_bytes / 1e9, **extras,
        }
        if slab_failure:
            out['slab_transport_failure'] = slab_failure
        if slab:
            st = slab_status
            out['slab'] = {'transport': transport, 'transport_info': comm.describe(), 'transport_trials_ms': trials, 'ipc_preflight': preflight_info, 'interior_boundary_split': split_on, 'split_trials_ms': split_trials, 'planes_owned': eng.b - eng.a, 'planes_held': eng.hi - eng.lo, 'ghost_max': eng.ghost_max,
                           'exchange_rounds_per_transition': st['exchanges'] / max(st['transitions'], 1),
                           'MB_sent_per_transition_rank0': st['exchanged_bytes'] / max(st['transitions'], 1) / 1e6,
                           'exact_transitions': st['exact_transitions'], 'mispredictions': st['mispredictions'],
                           'ms_per_transition_by_rank': {'min': min(d['ms_per_transition'] for d in slab_diag), 'max': max(d['ms_per_transition'] for d in slab_diag),
                                                         'all': [round(d['ms_per_transition'], 4) for d in slab_diag]},
                           # per rank, one sampled transition: [kind + stage ('e' exchange of buffer IRS_SB_*, 'a' all-reduce IRS_AR_*), adjoint step,
                           # ghost planes, hand-over us on the communication stream, stall us of the compute stream]
                           'round_wait_us': [{k: d.get(k) for k in ('rank', 'transition_us', 'handover_us', 'stall_us', 'rounds', 'timeline_error', 'gather_error') if k in d} for d in slab_diag]}
        if world == 1 and not args.no_extras:
            also = {}
            print('[bench] extras: SSD, displaced start, 128^3, sustained run ...', file=sys.stderr, flush=True)
            del eng
            torch.cuda.empty_cache()
            other = 'ssd' if args.loss == 'gmm' else 'gmm'
            also[f'{other}_{N}'] = dict(side_run(N, other, 'identity', 0.0, 25, 5, dev), workload=workload_name(N, other),
                                        note='BASELINE.json config 4 names the SSD loss' if other == 'ssd' else '')
            also['displaced_init'] = dict(side_run(N, args.loss, 'wave', 6.0, 20, 5, dev), init='wave amp 6.0 voxels',
                                          note='last squaring steps leave the radius-1 kernels (DESIGN.md section 4)')
            also[f'sigma_field_{N}'] = dict(side_run(N, args.loss, 'identity', 0.0, 20, 5, dev, timed_reps=3, sigma=0.5), sigma='field, 0.5 everywhere',
                                            note="pSGLD: the Langevin noise preconditioned by a sigma field (MCMC_init 'VI', utils/functions.py:78-84)")
            if N != 128:
                also['size_128'] = dict(side_run(128, args.loss, 'identity', 0.0, 50, 10, dev, timed_reps=5), workload=workload_name(128, args.loss))
            if N != 128:  # the reference runs C = 2 chains of a pair (configs/*/config.json): batched in one engine they fill the GPU at 128^3
                also['size_128_two_chains'] = dict(side_run(128, args.loss, 'identity', 0.0, 50, 10, dev, chains=2), workload=workload_name(128, args.loss),
                                                   note='C = 2 chains in one engine, as every reference config; ms per chain-transition')
                # ... and with the sigma field of an MCMC_init 'VI' start on top: the regime 14 of the reference's 16 configs run in
                also['size_128_two_chains_sigma_field'] = dict(side_run(128, args.loss, 'identity', 0.0, 50, 10, dev, chains=2, sigma=0.5), workload=workload_name(128, args.loss),
                                                               sigma='field, 0.5 everywhere', note='C = 2 chains + pSGLD preconditioner: configs/experiment1 semantics at its own size')
            also['sustained'] = dict(side_run(N, args.loss, 'identity', 0.0, 500, 5, dev), note='500 consecutive transitions from the identity')
            out['also'] = also
        if not args.no_cpu_baseline and world == 1:
            # 3 warm-up + >= 5 timed transitions at 128^3 and >= 1 at 256^3 (the workload itself): both reported,
            # `value` is the one measured at the bench size (or the largest size asked for, scaled by the voxel count)
            n_cpu = min(args.cpu_size, N)
            dt, cores, each = cpu_baseline(n_cpu, args.cpu_reps, args.loss)
            scale = (N / n_cpu) ** 3
            out['cpu_baseline'] = {'value': 1.0 / (dt * scale), 'unit': 'transitions/s', 'cores': cores, 'kind': 'port',
                                   'sample': f'{args.cpu_reps} transition(s) of the torch-CPU oracle at {n_cpu}^3 '
                                             f'({dt:.2f} s each, {cores} threads)' + (f', scaled by voxel count x{scale:.0f} to {N}^3' if scale != 1 else '')}
            if n_cpu > 128 and args.cpu_reps_128 > 0:
                dt128, _, each128 = cpu_baseline(128, args.cpu_reps_128, args.loss, warmup=3)
                out['cpu_baseline']['at_128'] = {'value': 1.0 / dt128, 'unit': 'transitions/s', 'seconds_each': [round(x, 3) for x in each128],
                                                 'sample': f'3 warm-up + {args.cpu_reps_128} timed transitions of the torch-CPU oracle at 128^3 ({dt128:.2f} s each, {cores} threads)'}
        print(json.dumps(out), flush=True)
    if world > 1:
        # tear down in order: engine (its streams drain), the library's RCCL communicator, then torch's process group
        torch.cuda.synchronize(dev)
        dist.barrier()
        if slab:
            eng = None
            comm.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
