"""ir_sgmcmc_amd.ipc_preflight.run: a child that cannot bring the peer-mapped transport up (here: no GPU at all) is an answer, not an
exception -- `bench.py --transport auto` then goes on with RCCL.  (The working path and a child that DIES mid-way run on the GPU:
tests/test_gpu_bench_cli.py.)"""
import pytest
import torch

from ir_sgmcmc_amd import ipc_preflight


def test_a_failing_child_is_reported_not_raised():
    if torch.cuda.is_available():
        ok, info = ipc_preflight.run('irs_pre_test_bad_device', 0, 1, 10 ** 6, timeout=120)   # a device that does not exist
    else:
        ok, info = ipc_preflight.run('irs_pre_test_no_gpu', 0, 1, 0, timeout=120)
    assert ok is False
    assert 'pre-flight child of rank 0 ended with code' in info


@pytest.mark.gpu
def test_flag_placement_probe_and_its_simulated_failure(monkeypatch):
    """the child also reports whether sequence flags in DEVICE memory work (csrc/ipc.hip: IRS_IPC_FLAGS=device); bench.py switches to
    them only where every rank's child says so.  One rank has no peer to wait for, so the probe passes; the hook simulates a node
    where it does not."""
    ok, info = ipc_preflight.run('irs_pre_test_flags', 0, 1, 0, timeout=120)
    assert ok and info['device_flags'] is True and info['exchange_us_device_flags'] >= 0
    monkeypatch.setenv('IRS_IPC_PREFLIGHT_SIMULATE_NO_DEVICE_FLAGS', '1')
    ok, info = ipc_preflight.run('irs_pre_test_flags2', 0, 1, 0, timeout=120)
    assert ok and info['device_flags'] is False and info['device_flags_error'] == 'simulated'
