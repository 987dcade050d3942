"""Seeded random configurations of the z-slab decomposition against the fused single-GPU engine: ragged volumes cut into 2 .. 4
slabs of unequal thickness, every ghost-width limit, both data terms, one to three chains, SVF and SVFFD, small to large
displacements -- concurrent ranks on one device over the peer-mapped transport (asynchronous exchanges), three transitions each
(one measuring, two from predicted ghost widths).  The result must be the fused engine's chain."""
import os
import random

import pytest

from tests.conftest import fuzz_seeds
from tests.test_gpu_slab import _launch

pytestmark = pytest.mark.gpu


def _draw(seed):
    r = random.Random(7000 + seed)
    world = r.choice([2, 2, 3, 4])
    D = r.randint(12 * world, 14 * world + 10)         # slabs of >= 12 planes, not all the same thickness
    dims = (D, r.randint(14, 40), r.randint(14, 44))
    data_loss = r.choice(['GMM', 'GMM', 'SSD'])
    C = r.choice([1, 1, 2, 3])
    cps = r.choice([None, None, None, (4, 4, 4), (2, 2, 2)])
    ghost_max = r.choice([1, 2, 4, 6, 8, 12])
    # (what the library REFUSES, loudly, is not drawn: a squaring step whose reach exceeds the ghost planes a rank holds -- "larger
    # irs_slab_config.margin" -- or the thinnest slab -- "fewer ranks for this displacement"; 60 unconstrained draws of round 4 ended
    # in one of the two ten times and in the fused engine's chain fifty times)
    amp = r.choice([3.0] if ghost_max <= 2 else [3.0, 9.0] if ghost_max <= 6 or D // world < 16 else [3.0, 9.0, 14.0])
    return dict(world=world, dims=dims, data_loss=data_loss, C=C, cps=cps, vd=r.random() < 0.7, amp=amp,
                reg=r.choice(['RegLoss_LogNormal', 'RegLoss_L2']), ghost_max=ghost_max, split=r.choice([1, 1, 0]))


# default draws: 0 (three ranks, SVFFD_3D cps 2, three chains, round limit 1), 1 (two ranks, SSD, two chains, round limit 12, 9-voxel start),
# 5 (four ranks -- middle ranks with two neighbours --, unsplit launches, 9-voxel start).  IRS_LONG=1: the six of round 4;
# IRS_SLAB_FUZZ_SEEDS=60: a longer hunt
@pytest.mark.parametrize('seed', fuzz_seeds('IRS_SLAB_FUZZ_SEEDS', (0, 1, 5), range(6)))
def test_random_slab_configuration_equals_the_fused_engine(seed, monkeypatch):
    k = _draw(seed)
    monkeypatch.setenv('IRS_SLAB_SPLIT', str(k['split']))   # (inherited by the spawned ranks)
    dv, dd, ds, st = _launch(k['world'], k['data_loss'], k['C'], k['dims'], k['vd'], k['amp'], k['reg'], k['ghost_max'], k['cps'],
                             transport='ipc')
    from tests._report import check
    name = 'slab_fuzz/%d_%s_ranks%d_%s_C%d_cps%s_g%d_amp%g_split%d' % (seed, 'x'.join(map(str, k['dims'])), k['world'], k['data_loss'], k['C'],
                                                                      k['cps'][0] if k['cps'] else 0, k['ghost_max'], k['amp'], k['split'])
    # Strict: 1e-5 on every element.  (The slab chain is not the fused chain bit for bit -- the statistics are summed per rank, then
    # over the ranks -- and in a long hunt ONE draw, seed 62, had one gradient element of 117 936 at 1.8e-4 where a sampling position
    # sat within rounding of a cell face and took the other one-sided derivative.  Round 4 relaxed the criterion for <= 3 such elements
    # without asserting the cell-face condition; a one- or two-voxel defect of a boundary strip would have passed the same way, so the
    # relaxation is gone: a draw that fails here is diagnosed (tools/debug/slab_diff.py: per plane and per transition, with the
    # count of deviating elements), not waved through.)
    check(name, 'v_new (rel to max)', dv, 0.0, 1e-5)
    check(name, 'displacement [voxels]', dd, 0.0, 1e-5)
    check(name, 'loss terms (rel)', ds, 0.0, 1e-6)
