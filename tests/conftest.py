import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
