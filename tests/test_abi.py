"""The C-ABI library loads and exports every symbol include/irsgmcmc.h declares (no compute calls: CPU-safe)."""
import os
import re

import pytest

from ir_sgmcmc_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'irsgmcmc.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(irs_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    assert _header_functions() == sorted(L.SIGNATURES)


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(L.LIB_PATH):
        pytest.fail(f'{L.LIB_PATH} is missing: run __graft_entry__.build() first')
    lib = L.load()
    for name in _header_functions():
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.irs_version()


def test_library_exports_nothing_but_the_c_abi():
    """-fvisibility=hidden + the version script csrc/exports.map: no internal C++ symbol leaves the library"""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [l.split()[-1] for l in out.splitlines() if l.strip()]
    assert names and sorted(names) == _header_functions(), sorted(set(names) ^ set(_header_functions()))


def test_struct_layouts_match_the_header():
    import ctypes as C
    # sizes computed by hand from the header (natural alignment)
    assert C.sizeof(L.IrsConfig) == 240
    assert C.sizeof(L.IrsIO) == 3 * 8 + 16 + 10 * 8
    assert C.sizeof(L.IrsScalars) == 5 * 8 * L.IRS_MAX_CHAINS
    assert C.sizeof(L.IrsState) == 2 * 4 * 8 + 2 * 2 * 8 * 8 + 16 + 3 * 16 + 16 + 8


def test_cpu_tensors_are_refused():
    import torch
    from ir_sgmcmc_amd import ops
    with pytest.raises(L.IrsError):
        ops.svf_exp_fwd(torch.zeros(1, 3, 8, 8, 8))
