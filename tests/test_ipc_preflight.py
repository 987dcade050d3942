"""ir_sgmcmc_amd.ipc_preflight.run: a child that cannot bring the peer-mapped transport up (here: no GPU at all) is an answer, not an
exception -- `bench.py --transport auto` then goes on with RCCL.  (The working path and a child that DIES mid-way run on the GPU:
tests/test_gpu_bench_cli.py.)"""
import torch

from ir_sgmcmc_amd import ipc_preflight


def test_a_failing_child_is_reported_not_raised():
    if torch.cuda.is_available():
        ok, info = ipc_preflight.run('irs_pre_test_bad_device', 0, 1, 10 ** 6, timeout=120)   # a device that does not exist
    else:
        ok, info = ipc_preflight.run('irs_pre_test_no_gpu', 0, 1, 0, timeout=120)
    assert ok is False
    assert 'pre-flight child of rank 0 ended with code' in info
