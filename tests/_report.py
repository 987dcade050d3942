"""check(): assert |a - b|_max <= tol with the numbers in the message, and log the deviation for the parity report."""
import torch

PARITY = {}


def check(test, key, a, b, tol):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    dev = float((a - b).abs().max()) if a.numel() else 0.0
    rec = PARITY.setdefault(test, {}).setdefault(key, {'max_dev': 0.0, 'tol': tol})
    rec['max_dev'] = max(rec['max_dev'], dev)
    rec['tol'] = max(rec['tol'], tol)
    assert dev <= tol, f'{test}: {key}: max deviation {dev:.3e} > tolerance {tol:.3e}'
    return dev
