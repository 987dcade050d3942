"""Data loaders with the reference's contract (data_loader/data_loaders.py, data_loader/datasets.py:107-145):
iterating yields ONE (fixed, moving, var_params_q_v) triple; fixed/moving are dicts with 'im' float32, 'mask' bool,
'seg' int16 of shape (1, 1, D, H, W); var_params_q_v has 'mu', 'log_var', 'u' of shape (1, 3, *dims_v)."""
from ..ops import control_grid_size
from .synthetic import init_var_params, synthetic_pair


class SyntheticDataLoader:
    """deterministic Gaussian-blob pair; the reference ships no image data"""

    def __init__(self, dims, sigma_v_init=0.5, u_v_init=0.1, cps=None, save_dirs=None, seed=0, **_unused):
        self.dims = tuple(dims)
        self.cps = tuple(cps) if cps else None
        self.save_dirs = save_dirs
        self.sigma_v_init, self.u_v_init, self.seed = sigma_v_init, u_v_init, seed
        self.im_spacing = None

    @property
    def dims_v(self):
        return control_grid_size(self.dims, self.cps) if self.cps else self.dims

    def __len__(self):
        return 1

    def __iter__(self):
        fixed, moving = synthetic_pair(self.dims, seed=self.seed)
        fixed = {k: v.unsqueeze(0) for k, v in fixed.items()}
        moving = {k: v.unsqueeze(0) for k, v in moving.items()}
        vp = {k: v.unsqueeze(0) for k, v in init_var_params(self.dims_v, self.sigma_v_init, self.u_v_init).items()}
        yield fixed, moving, vp


class BiobankDataLoader(SyntheticDataLoader):
    """Reference contract (data_loader/data_loaders.py:5-19): `.nii.gz` triples under `data_dir` through BiobankDataset
    (numpy NIfTI reader instead of SimpleITK).  An unusable `data_dir` is an ERROR, as in the reference (its `listdir`
    raises): a typo must not turn into a registration of fake data written out under the requested names.  The synthetic
    pair of the same `dims` is only used when the config asks for it: `"allow_synthetic_fallback": true` (the reference
    ships no image data, its configs point at the author's cluster), or the `SyntheticDataLoader` type."""

    def __init__(self, data_dir=None, dims=None, sigma_v_init=0.5, u_v_init=0.1, cps=None, save_dirs=None,
                 allow_synthetic_fallback=False, **kw):
        super().__init__(dims, sigma_v_init, u_v_init, cps, save_dirs, **kw)
        self.data_dir, self.dataset = data_dir, None
        from .datasets import BiobankDataset
        try:
            if not data_dir:  # a config without data_dir (None / ''): the same explanatory error or fallback as a missing directory
                raise FileNotFoundError('no data_dir given')
            self.dataset = BiobankDataset(dims, data_dir, save_dirs, sigma_v_init, u_v_init, cps=cps)
        except (FileNotFoundError, NotADirectoryError) as e:
            if not allow_synthetic_fallback:
                raise FileNotFoundError(f'BiobankDataLoader: no usable image pair under data_dir={data_dir!r} ({e}); set '
                                        f'"allow_synthetic_fallback": true in data_loader.args to run on the synthetic pair') from e
            import logging
            logging.getLogger('default').warning(f'BiobankDataLoader: {e}; allow_synthetic_fallback: using the synthetic pair at {dims}')

    @property
    def im_spacing(self):
        return self.dataset.im_spacing if self.dataset is not None else None

    @im_spacing.setter
    def im_spacing(self, value):   # the synthetic base class assigns None in its constructor
        pass

    def __iter__(self):
        if self.dataset is None:
            yield from super().__iter__()
            return
        fixed, moving, vp = self.dataset[0]
        add = lambda d: {k: v.unsqueeze(0) for k, v in d.items()}   # the DataLoader's batch dimension
        yield add(fixed), add(moving), add(vp)
