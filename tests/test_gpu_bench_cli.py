"""bench.py as the driver runs it: the N = 1 line, and the N = 2 script path (spawn -> torch.distributed.run -> slab engine)
rehearsed on ONE GPU over the gloo / host-staging transport, so that the first real multi-GPU SCALE run is not the script's
first execution.  (The rehearsal exercises everything except RCCL itself: rank bring-up, the slab layout, the schedule, ghost
exchanges, all-reduces, status read-back, the JSON contract, tear-down.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env_extra, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID')}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
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


def test_bench_two_ranks_rehearsal():
    d = _run(['--gpus', '2', '--size', '64', '--steps', '3', '--warmup', '2', '--no-cpu-baseline'],
             {'IRS_BENCH_BACKEND': 'gloo', 'IRS_BENCH_DEVICE': '0'})
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['steps'] == 3
    assert 'slab_transport_failure' not in d
    s = d['slab']
    assert s['mispredictions'] == 0 and s['planes_owned'] == 32 and s['planes_held'] > 32
    assert s['exchange_rounds_per_transition'] > 0 and d['value'] > 0
    assert '2 z-slabs' in d['config']['parallelism']
