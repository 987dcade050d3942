import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The default `-m gpu` run has to fit the driver's time limit with room to spare: the soak runs and the extra fuzz draws only run
# with IRS_LONG=1 in the environment (an explicit skip with this reason otherwise -- never a marker `-m gpu` would deselect silently).
LONG = os.environ.get('IRS_LONG') == '1'
long_only = pytest.mark.skipif(not LONG, reason='long-running: set IRS_LONG=1 (soak runs, extra fuzz seeds)')


def fuzz_seeds(env_name, default, long_default):
    """seed list of a fuzz suite: `env_name` overrides; `default` draws normally -- the ones that exercise distinct code paths, picked
    from the 10 .. 12 the suites ran in round 4 --, `long_default` with IRS_LONG=1"""
    v = os.environ.get(env_name)
    if v:
        return list(range(int(v)))
    return list(long_default) if LONG else list(default)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests must not silently pass/skip on the GPU box; on a CPU-only box `-m "not gpu"` deselects them."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


# ---- parity report: every GPU parity test records its worst deviation next to the tolerance it was held to
# (tests/_report.py); written to gpurun_out/parity_report.json at session end and copied into profiles/
def pytest_sessionfinish(session, exitstatus):
    import json
    from tests._report import PARITY
    if not PARITY:
        return
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'parity_report.json'), 'w') as f:
        json.dump(PARITY, f, indent=1, sort_keys=True)
