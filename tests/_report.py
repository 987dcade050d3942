"""check(): assert |a - b|_max <= tol with the numbers in the message, and log the deviation for the parity report."""
import torch

PARITY = {}

# Gradient tolerance, relative to max|grad|.  The north star states tolerances for the loss (1e-5 relative) and the
# displacement field (1e-4); for the gradient the yardstick is the reference itself: its fp32 autograd through the 12
# grid_sample calls deviates 1e-4 .. 1.7e-2 (relative) from an fp64 evaluation of the same chain, growing as the
# displacement shrinks and the volume grows (measured in this container, DESIGN.md "Numerics").  The HIP path is held
# to 1e-3 of the reference's fp32 values -- inside that band -- and typically lands at 1e-6 .. 7e-4.
GRAD_RTOL = 1e-3


def check(test, key, a, b, tol):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    dev = float((a - b).abs().max()) if a.numel() else 0.0
    rec = PARITY.setdefault(test, {}).setdefault(key, {'max_dev': 0.0, 'tol': tol})
    rec['max_dev'] = max(rec['max_dev'], dev)
    rec['tol'] = max(rec['tol'], tol)
    assert dev <= tol, f'{test}: {key}: max deviation {dev:.3e} > tolerance {tol:.3e}'
    return dev


def fp64_band(key):
    """tests/golden/fp64_bands.json: how far the fp32 evaluation of one transition is from the fp64 evaluation of the same
    transition -- measured on the unmodified reference (32^3, 64^3) and on the oracle at the full-size test inputs (128^3, 256^3)
    by tests/golden/make_golden_fp64.py.  Where a comparison needs more room than the north-star tolerances, THIS is the room
    it may take: the HIP path must be no further from the fp32 oracle than the fp32 oracle is from fp64."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fp64_bands.json')) as f:
        return json.load(f)[key]
