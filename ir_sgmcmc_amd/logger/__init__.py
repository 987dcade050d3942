"""Observability for the MCMC path: console/file logging, a scalar writer, and the reference's file writers.

The TensorBoard / seaborn figures of the reference's logger are out of scope; its output FILES
are not: `save_im_to_disk` (.nii.gz), `save_field_to_disk` / `save_grid_to_disk` (.vtk) and the helpers built on them
(logger/logger.py:35-240) are reproduced on top of utils/imageio.py (numpy; no nibabel / tvtk).
"""
import logging
from os import path

import numpy as np

from ..utils.imageio import write_nifti, write_vtk_field, write_vtk_grid


def setup_logging(save_dir=None, level=logging.INFO):
    handlers = [logging.StreamHandler()]
    if save_dir is not None:
        handlers.append(logging.FileHandler(str(save_dir) + '/info.log'))
    logging.basicConfig(level=level, format='%(message)s', handlers=handlers, force=True)


class ScalarWriter:
    """stand-in for TensorboardWriter (logger/visualization.py:12-55): remembers the last value of every scalar"""

    def __init__(self):
        self.step = 0
        self.scalars = {}

    def set_step(self, step):
        self.step = step

    def add_scalar(self, key, value):
        self.scalars[key] = (self.step, value)

    def write_hparams(self, *_):
        pass


# ------------------------------------------------------------------ file writers (logger/logger.py:35-240)
def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def save_field_to_disk(field, file_path, spacing=(1, 1, 1)):
    """vector field (3, nx, ny, nz) -> legacy .vtk, x fastest (logger/logger.py:35-60)"""
    write_vtk_field(_np(field), file_path, _np(spacing))


def save_grid_to_disk(grid, file_path):
    """sampling grid (3, nx, ny, nz) -> .vtk structured grid (logger/logger.py:63-80)"""
    write_vtk_grid(_np(grid), file_path)


def save_im_to_disk(im, file_path, spacing=(1, 1, 1)):
    """3-D image -> .nii.gz with identity affine, mm units, zooms = spacing (logger/logger.py:83-100)"""
    write_nifti(_np(im), file_path, _np(spacing))


def _folder(save_dirs, key, model=None):
    import os
    folder = path.join(str(save_dirs['samples']), model) if model is not None else str(save_dirs[key])
    os.makedirs(folder, exist_ok=True)
    return folder


def save_field(save_dirs, spacing, field, field_name, model=None):
    save_field_to_disk(field, path.join(_folder(save_dirs, 'fields', model), f'{field_name}.vtk'), spacing)


def save_im(save_dirs, spacing, im, name, model=None):
    save_im_to_disk(im, path.join(_folder(save_dirs, 'images', model), f'{name}.nii.gz'), spacing)


def save_fixed_im(save_dirs, spacing, im_fixed):
    save_im(save_dirs, spacing, im_fixed[0, 0], 'im_fixed')


def save_fixed_mask(save_dirs, spacing, mask_fixed):
    save_im(save_dirs, spacing, mask_fixed[0, 0].float(), 'mask_fixed')


def save_moving_im(save_dirs, spacing, im_moving_batch):
    save_im(save_dirs, spacing, im_moving_batch[0, 0], 'im_moving')


def save_moving_mask(save_dirs, spacing, mask_moving):
    save_im(save_dirs, spacing, mask_moving[0, 0].float(), 'mask_moving')


def save_displacement_mean_and_std_dev(logger, save_dirs, spacing, displacement_mean, displacement_std_dev, mask, model):
    """posterior mean / std of the displacement in mm, plain and masked (logger/logger.py:110-131)"""
    folder = _folder(save_dirs, 'samples')
    for name, field in (('mean', displacement_mean), ('std_dev', displacement_std_dev)):
        field = field * spacing[0]
        logger.info(f'{model} displacement {name.replace("_", ". ")} min.: {float(field.min()):.2f}, max.: {float(field.max()):.2f}')
        save_field_to_disk(field, path.join(folder, f'{model}_sample_{name}.vtk'), spacing)
        save_field_to_disk(field * mask[0], path.join(folder, f'{model}_sample_{name}_masked.vtk'), spacing)


def save_sample(save_dirs, spacing, sample_no, im_moving_warped_batch, displacement_batch, log_det_J_batch, model, chain_no=None):
    """warped image, displacement (mm) and log det J of one sample (logger/logger.py:215-240)"""
    prefix = f'chain_{chain_no}_sample_{sample_no:07}' if model == 'MCMC' else f'sample_{sample_no:07}'
    save_im(save_dirs, spacing, im_moving_warped_batch[0, 0], f'{prefix}_im_moving_warped', model)
    save_field(save_dirs, spacing, displacement_batch[0] * spacing[0], f'{prefix}_displacement', model)
    save_im(save_dirs, spacing, log_det_J_batch[0], f'{prefix}_log_det_J', model)
