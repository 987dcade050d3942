"""Pre-flight of the peer-mapped slab transport (csrc/ipc.hip) in a CHILD process of every rank.

The transport's first contact with a peer GPU is a device-side store into memory mapped by hipIpcOpenMemHandle.  Where that
mapping is not reachable from the storing device the process does not get an error code, it gets a memory fault -- and a rank that
dies takes a whole `bench.py --gpus N` run (and every RCCL collective of the others) with it.  So before a multi-GPU run trusts the
transport, every rank starts this module as a child on its own device: the children bring the transport up among themselves under
a name of their own, run the library's self-test (neighbour exchanges of several sizes checked word by word, all-reduces of every
kind) and time a hand-over.  A child that faults, fails or hangs costs its parent an exit code, nothing else.

    python -m ir_sgmcmc_amd.ipc_preflight NAME RANK WORLD DEVICE     -> one JSON line, exit code 0 when the transport works

`run(...)` is what a rank calls; `bench.py` uses it in `--transport auto` runs with one rank per device.
"""
import json
import os
import subprocess
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(name, rank, world, device_index, timeout=120.0):
    """start the child and wait for it: (True, {'exchange_us': .., 'allreduce_us': .., 'info': ..}) or (False, 'what happened')"""
    cmd = [sys.executable, '-m', 'ir_sgmcmc_amd.ipc_preflight', name, str(rank), str(world), str(device_index)]
    env = dict(os.environ)
    env['PYTHONPATH'] = _ROOT + os.pathsep + env.get('PYTHONPATH', '')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'TORCHELASTIC_RUN_ID'):   # the child is nobody's rank but its argument's
        env.pop(k, None)
    env.setdefault('IRS_IPC_TIMEOUT_S', '10')
    # the children of all ranks start together: one that waits longer than this for a peer at a bootstrap step is waiting for a
    # child that died (the library's default of 120 s is for ranks whose start-up may be minutes apart)
    env.setdefault('IRS_IPC_BOOT_TIMEOUT_S', '45')
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=_ROOT)
    except subprocess.TimeoutExpired:
        return False, f'pre-flight child of rank {rank} did not finish within {timeout:.0f} s'
    if p.returncode != 0:
        tail = ' | '.join((p.stderr or p.stdout or '').strip().splitlines()[-3:])
        return False, f'pre-flight child of rank {rank} ended with code {p.returncode}: {tail}'
    try:
        return True, json.loads(p.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return False, f'pre-flight child of rank {rank} printed no result'


def _child(name, rank, world, device_index):
    import ctypes as C

    import torch

    from . import _lib as L
    torch.cuda.set_device(device_index)
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.irs_comm_create_ipc(name.encode(), rank, world, C.byref(h)))
    if os.environ.get('IRS_IPC_PREFLIGHT_SIMULATE_FAULT') == str(rank):   # (tests: a rank whose first peer access kills it)
        os.abort()
    try:
        L.check(lib.irs_comm_selftest(h, None))
        us = (C.c_double * 2)()
        L.check(lib.irs_comm_probe(h, 3 * 256 * 256 * 12, 21, 50, None, C.byref(us)))   # three ghost planes of a 256^3 field
        buf = C.create_string_buffer(256)
        L.check(lib.irs_comm_describe(h, buf, 256))
        res = {'exchange_us': round(us[0], 2), 'allreduce_us': round(us[1], 2), 'info': buf.value.decode()}
    finally:
        lib.irs_comm_destroy(h)
    res.update(_probe_device_flags(lib, name, rank, world))
    print(json.dumps(res), flush=True)


def _probe_device_flags(lib, name, rank, world):
    """Do sequence flags in DEVICE memory work here?  (csrc/ipc.hip: IRS_IPC_FLAGS=device puts them into the landing areas -- polled
    locally, raised by the peer through its mapping; with two processes on ONE device a poll never saw the other's store, across
    devices nobody could try.)  A second communicator with that placement and a 3 s device-side timeout runs the same self-test and
    timing: a flag that never arrives is a clean error (the transport's timeouts are fail-safe), reported as `device_flags: false`."""
    import ctypes as C

    from . import _lib as L
    if os.environ.get('IRS_IPC_PREFLIGHT_SIMULATE_NO_DEVICE_FLAGS') == '1':   # (tests: a node where the placement does not work)
        return {'device_flags': False, 'device_flags_error': 'simulated'}
    old = {k: os.environ.get(k) for k in ('IRS_IPC_FLAGS', 'IRS_IPC_TIMEOUT_S')}
    os.environ.update(IRS_IPC_FLAGS='device', IRS_IPC_TIMEOUT_S='3')
    h = C.c_void_p()
    try:
        L.check(lib.irs_comm_create_ipc((name + '_df').encode(), rank, world, C.byref(h)))
        try:
            L.check(lib.irs_comm_selftest(h, None))
            us = (C.c_double * 2)()
            L.check(lib.irs_comm_probe(h, 3 * 256 * 256 * 12, 21, 50, None, C.byref(us)))
            return {'device_flags': True, 'exchange_us_device_flags': round(us[0], 2), 'allreduce_us_device_flags': round(us[1], 2)}
        finally:
            lib.irs_comm_destroy(h)
    except L.IrsError as e:
        return {'device_flags': False, 'device_flags_error': str(e)[:200]}
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == '__main__':
    _child(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
