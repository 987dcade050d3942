"""Import harness for the *read-only* Python reference at /root/reference.

Only used in the build container by ``make_golden.py`` to generate fixtures and to
validate ``oracle/``.  The reference cannot travel to the GPU box; nothing under
``tests/`` that runs there imports this module.

The reference's hot path needs torch/numpy/scipy only, but ``utils/util.py:6,13-14`` and
``logger/*`` import I/O packages (SimpleITK, vtk, nibabel, tvtk, seaborn, tensorboard) at
module scope.  None of them is *called* on the hot path, so empty placeholder modules are
registered in ``sys.modules`` before the import.  The reference tree is
not modified and no bytecode is written into it.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('IRSGMCMC_REFERENCE', '/root/reference')


def _placeholder(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def import_reference():
    """Returns a namespace with the reference modules needed by the golden generator."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f'reference tree not found at {REFERENCE_ROOT}')

    sys.dont_write_bytecode = True

    class _Unavailable:  # any accidental *use* of an I/O dependency must fail loudly
        def __init__(self, *a, **k):
            raise RuntimeError('I/O dependency placeholder was called')

    _placeholder('SimpleITK')
    vtk = _placeholder('vtk', vtkStructuredPointsReader=_Unavailable)
    vtk.util = _placeholder('vtk.util')
    vtk.util.numpy_support = _placeholder('vtk.util.numpy_support', vtk_to_numpy=_Unavailable)
    _placeholder('nibabel')
    tvtk = _placeholder('tvtk')
    tvtk.api = _placeholder('tvtk.api', tvtk=_Unavailable, write_data=_Unavailable)
    _placeholder('seaborn')
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        import torch.utils
        tb = _placeholder('torch.utils.tensorboard', SummaryWriter=type('SummaryWriter', (), {}))
        tb.summary = _placeholder('torch.utils.tensorboard.summary', hparams=_Unavailable)
        torch.utils.tensorboard = tb

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import utils as ref_utils
    import model.loss as ref_loss
    import model.distributions as ref_distr
    import optimizers as ref_optim
    import trainer.trainer as ref_trainer

    return types.SimpleNamespace(utils=ref_utils, loss=ref_loss, distr=ref_distr, optim=ref_optim,
                                 Trainer=ref_trainer.Trainer)
