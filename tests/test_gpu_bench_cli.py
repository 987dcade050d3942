"""bench.py as the driver runs it: the N = 1 line, and the N = 2 script path (spawn -> torch.distributed.run -> slab engine)
on ONE GPU over the peer-mapped `ipc` transport (asynchronous: what a node runs, minus xGMI) and over the synchronous gloo /
host-staging rehearsal, so that the first real multi-GPU SCALE run is not the script's first execution; and the exit code
when no slab transport comes up."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env_extra, timeout=420, expect_rc=0):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID')}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if expect_rc:
        assert p.returncode != 0, f'expected a failure\n--- stdout\n{p.stdout[-2000:]}\n--- stderr\n{p.stderr[-2000:]}'
        return p
    assert p.returncode == 0, f'rc {p.returncode}\n--- stdout\n{p.stdout[-2000:]}\n--- stderr\n{p.stderr[-4000:]}'
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]   # exactly ONE JSON line
    return json.loads(lines[0])


def test_bench_one_gpu_line_contract():
    d = _run(['--gpus', '1', '--size', '64', '--steps', '10', '--warmup', '2', '--no-cpu-baseline', '--no-extras'], {})
    assert d['n_gpus'] == 1 and d['steps'] == 10 and d['warmup'] == 2 and d['unit'] == 'transitions/s'
    assert d['higher_is_better'] is True and d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert abs(d['value'] - 1e3 / d['ms_per_step']) < 1e-6 * d['value']      # value = transitions / max-over-ranks wall time
    assert d['roofline']['bound'] == 'hbm' and 0.0 < d['roofline']['frac'] < 1.0
    assert d['config']['volume'] == [64, 64, 64] and 'workload' in d['config']


def test_bench_two_ranks_ipc():
    """the N = 2 path over the PRODUCT transport that allows two ranks on one device: asynchronous exchanges, the library's own
    communication stream and events -- what `bench.py --gpus 2` runs on a node, minus xGMI"""
    d = _run(['--gpus', '2', '--size', '64', '--steps', '5', '--warmup', '2', '--no-cpu-baseline'], {'IRS_BENCH_DEVICE': '0'})
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['steps'] == 5
    assert 'slab_transport_failure' not in d
    s = d['slab']
    assert s['transport'] == 'ipc' and 'ipc' in s['transport_trials_ms']
    assert s['mispredictions'] == 0 and s['planes_owned'] == 32 and s['planes_held'] > 32
    assert s['exchange_rounds_per_transition'] > 0 and d['value'] > 0
    assert '2 z-slabs' in d['config']['parallelism'] and 'peer-mapped' in d['config']['parallelism']
    # the overlapped (interior / boundary split) and the unsplit launch sequence were both timed, the faster one carried the run
    assert set(s['split_trials_ms']) == {'split', 'unsplit'}
    assert s['interior_boundary_split'] == (s['split_trials_ms']['split'] <= s['split_trials_ms']['unsplit'])


def test_bench_auto_transport_survives_a_transport_that_fails():
    """`--transport auto` is what `bench.py --gpus N` runs on a node: bring up BOTH product transports, verify and time each, keep
    the faster.  With the ranks on one device RCCL cannot come up (it refuses two ranks per device): the run must go on with ipc,
    say which transport carried it and what failed -- strong scaling, exit 0."""
    d = _run(['--gpus', '2', '--size', '48', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--transport', 'auto'],
             {'IRS_BENCH_DEVICE': '0', 'IRS_BENCH_PREFLIGHT': '1'})
    assert d['scaling'] == 'strong' and d['slab']['transport'] == 'ipc'
    assert list(d['slab']['transport_trials_ms']) == ['ipc'] and 'rccl' in d['slab_transport_failure']
    assert 'ipc:' in d['slab']['transport_info'] and d['slab']['mispredictions'] == 0
    # (one rank per device, a node: the transport is first brought up, self-tested and timed by a child of every rank)
    pre = d['slab']['ipc_preflight']
    assert pre['exchange_us'] > 0 and 'ipc:' in pre['info']
    # ... and the children probed the flag placement: device-memory flags are used exactly where every child saw them work
    assert isinstance(pre['device_flags'], bool) and (pre['device_flags'] or pre['device_flags_error'])
    assert ('sequence flags in the landing areas' in d['slab']['transport_info']) == pre['device_flags']


def test_bench_auto_transport_survives_a_preflight_child_that_dies():
    """the peer-mapped transport's first stores into another rank's memory happen in a child of every rank: a child that dies there
    (a mapping the node cannot reach is a memory fault, not an error code) costs the run the ipc transport, not the run -- here
    RCCL cannot come up either (one device), so the run ends with the no-transport exit code and says what happened.
    (IRS_IPC_BOOT_TIMEOUT_S: the surviving child waits that long for the dead one at the next bootstrap step -- with the pre-flight's
    default of 45 s this test took 2 x 120 s of the round-4 suite's 764.)"""
    p = _run(['--gpus', '2', '--size', '32', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--transport', 'auto'],
             {'IRS_BENCH_DEVICE': '0', 'IRS_BENCH_PREFLIGHT': '1', 'IRS_IPC_PREFLIGHT_SIMULATE_FAULT': '1', 'IRS_IPC_BOOT_TIMEOUT_S': '6'}, expect_rc=1)
    assert 'pre-flight child of rank 1 ended with code' in p.stderr and 'no slab transport came up' in p.stderr
    d = _run(['--gpus', '2', '--size', '32', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--transport', 'auto', '--allow-chain-fallback'],
             {'IRS_BENCH_DEVICE': '0', 'IRS_BENCH_PREFLIGHT': '1', 'IRS_IPC_PREFLIGHT_SIMULATE_FAULT': '1', 'IRS_IPC_BOOT_TIMEOUT_S': '6'})
    assert d['scaling'] == 'weak' and 'pre-flight' in d['slab_transport_failure']


def test_bench_four_ranks_ipc():
    """middle ranks with two neighbours, through the script"""
    d = _run(['--gpus', '4', '--size', '64', '--steps', '4', '--warmup', '2', '--no-cpu-baseline'], {'IRS_BENCH_DEVICE': '0'})
    assert d['n_gpus'] == 4 and d['scaling'] == 'strong' and d['slab']['transport'] == 'ipc'
    assert d['slab']['planes_owned'] == 16 and d['slab']['mispredictions'] == 0 and d['value'] > 0
    # the self-diagnosis of a first run on a node: every rank's own time, and per hand-over of one sampled transition how long the
    # communication stream took and how long the compute stream stalled (include/irsgmcmc.h: irs_slab_timeline_*)
    s = d['slab']
    per = s['ms_per_transition_by_rank']
    assert len(per['all']) == 4 and 0 < per['min'] <= per['max'] and abs(per['max'] - d['ms_per_step']) < 1e-3 * d['ms_per_step'] + 1e-6
    assert [r['rank'] for r in s['round_wait_us']] == [0, 1, 2, 3]
    for r in s['round_wait_us']:
        assert 'timeline_error' not in r, r
        assert r['transition_us'] > 0 and r['handover_us'] > 0 and r['stall_us'] >= 0
        kinds = {e[0][0] for e in r['rounds']}
        assert kinds == {'e', 'a'} and len(r['rounds']) >= 5          # exchanges and all-reduces, in schedule order
        assert all(e[3] >= 0 and e[4] >= 0 for e in r['rounds'])
        assert sum(1 for e in r['rounds'] if e[0][0] == 'a') >= 3      # bounds, statistics, data-term sums


def test_bench_exits_non_zero_without_a_slab_transport():
    """RCCL refuses two ranks on one device: with that the only transport asked for, the run must FAIL (a weak-scaling number must
    not pass for a point of the strong-scaling curve); --allow-chain-fallback turns it into the chain decomposition, which says
    so in its line"""
    args = ['--gpus', '2', '--size', '32', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--transport', 'rccl']
    p = _run(args, {'IRS_BENCH_DEVICE': '0'}, expect_rc=1)
    assert 'no slab transport came up' in p.stderr
    d = _run(args + ['--allow-chain-fallback'], {'IRS_BENCH_DEVICE': '0'})
    assert d['scaling'] == 'weak' and 'slab_transport_failure' in d and 'slab' not in d


def test_bench_two_ranks_rehearsal():
    d = _run(['--gpus', '2', '--size', '64', '--steps', '3', '--warmup', '2', '--no-cpu-baseline'],
             {'IRS_BENCH_BACKEND': 'gloo', 'IRS_BENCH_DEVICE': '0'})
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['steps'] == 3
    assert 'slab_transport_failure' not in d
    s = d['slab']
    assert s['transport'] == 'rehearsal'
    assert s['mispredictions'] == 0 and s['planes_owned'] == 32 and s['planes_held'] > 32
    assert s['exchange_rounds_per_transition'] > 0 and d['value'] > 0
    assert '2 z-slabs' in d['config']['parallelism']
