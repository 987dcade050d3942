#!/usr/bin/env python3
"""Fixture for the utility operators next to the hot path, generated from the real reference (build container only):

  separable_conv_3D, 2-argument branch (utils/util.py:362-392): pad the LAST axis by replicate, flatten, conv1d with zero
      padding, crop, permute -- three times.  The crop removes exactly the positions the zero padding / the neighbouring rows
      reach, so the branch equals one (2 p + 1)-tap filter per axis with replicate padding (checked here to 1.2e-7 against the
      4-argument branch and against the oracle's per-axis filter), applied in the order W, D, H.
  calc_norm (utils/util.py:215-225), calc_DSC_GPU (utils/util.py:123-148).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_utils.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402

from oracle import ops as O  # noqa: E402


def main():
    ref = import_reference()
    g = torch.Generator().manual_seed(4242)
    N, D = 2, 12
    v = torch.randn(N, 3, D, D, D, generator=g)
    store = {'field': v.numpy()}
    for tag, k in (('k3', torch.tensor([0.2, 0.5, 0.3])), ('k5', torch.tensor([0.1, 0.3, 0.2, 0.25, 0.15])),
                   ('sobolev', torch.as_tensor(O.sobolev_kernel_1d(3, 0.5)).float())):
        p = (k.numel() - 1) // 2
        S = torch.stack((k, k, k), 0).unsqueeze(1)
        out2 = ref.utils.separable_conv_3D(v, S, p)
        out4 = ref.utils.separable_conv_3D(v, S.unsqueeze(2).unsqueeze(2), S.unsqueeze(2).unsqueeze(4), S.unsqueeze(3).unsqueeze(4), (p,) * 6)
        mine = O.separable_conv3d_replicate(v, k)
        d24, d2o = float((out2 - out4).abs().max()), float((out2 - mine).abs().max())
        print(f'separable_conv_3D {tag}: 2-arg vs 4-arg {d24:.1e}, 2-arg vs oracle per-axis replicate filter {d2o:.1e}')
        assert d24 < 1e-6 and d2o < 1e-6
        store[f'{tag}_kernel'] = k.numpy()
        store[f'{tag}_out_2arg'] = out2.numpy()
    # per-channel kernels (groups = 3): channel c is filtered with ITS row of the kernel tensor
    kc = torch.tensor([[0.2, 0.5, 0.3], [0.6, 0.3, 0.1], [0.0, 1.0, 0.0]])
    store['kc_kernel'] = kc.numpy()
    store['kc_out_2arg'] = ref.utils.separable_conv_3D(v, kc.unsqueeze(1), 1).numpy()
    store['norm'] = ref.utils.calc_norm(v).numpy()
    seg_f = torch.randint(0, 4, (3, 1, D, D, D), generator=g).short()
    seg_m = torch.randint(0, 4, (3, 1, D, D, D), generator=g).short()
    seg_m[0] = seg_f[0]  # a perfect overlap
    structures = {'a': 1, 'b': 2, 'c': 3, 'absent': 9}
    store.update(seg_fixed=seg_f.numpy(), seg_moving=seg_m.numpy(), dsc=ref.utils.calc_DSC_GPU(3, seg_f, seg_m, structures),
                 dsc_labels=np.array(list(structures.values())))
    print('calc_DSC_GPU', store['dsc'])
    path = os.path.join(HERE, 'utils_ops.npz')
    np.savez_compressed(path, **store)
    print(f'wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    main()
