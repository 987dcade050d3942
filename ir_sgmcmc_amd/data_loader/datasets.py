"""BiobankDataset (reference data_loader/datasets.py:13-145) on top of the numpy NIfTI reader.

Directory layout as in the reference: `<data_dir>/*.nii.gz` images, `<data_dir>/masks/*`, `<data_dir>/segs/*`, sorted by
name; image 0 is the fixed image, image idx + 1 the moving one (all-to-one, `__len__` = 1).  Every volume is padded with
its minimum to a cube of the largest extent, then resized to `dims`: trilinear / align_corners for images, nearest for
masks and segmentations -- the same torch calls the reference makes (:70-105)."""
import json
import os
from os import listdir, path

import numpy as np
import torch
import torch.nn.functional as F

from ..ops import control_grid_size
from ..utils.imageio import read_nifti


class BiobankDataset:
    def __init__(self, dims, im_paths, save_paths=None, sigma_v_init=0.5, u_v_init=0.1, cps=None):
        self.im_paths, self.save_paths = im_paths, save_paths
        self.sigma_v_init, self.u_v_init = sigma_v_init, u_v_init
        self.dims = tuple(dims)
        self.dims_im = (1, *self.dims)
        self.dims_v = (3, *self.dims) if cps is None else (3, *control_grid_size(self.dims, cps))
        self.padding, self.im_spacing = None, None

        im_filenames = self._get_filenames(im_paths)
        mask_filenames = self._get_filenames(path.join(im_paths, 'masks'))
        seg_filenames = self._get_filenames(path.join(im_paths, 'segs'))
        self.im_mask_seg_triples = [{'im': t[0], 'mask': t[1], 'seg': t[2]}
                                    for t in zip(im_filenames, mask_filenames, seg_filenames)]
        if len(self.im_mask_seg_triples) < 2 or any(not path.isfile(f) for t in self.im_mask_seg_triples[:2] for f in t.values()):
            raise FileNotFoundError(f'{im_paths}: need at least two image / mask / seg triples (fixed + moving)')
        if save_paths and save_paths.get('dir'):
            with open(os.path.join(str(save_paths['dir']), 'idx_to_biobank_ID.json'), 'w') as out:
                json.dump(dict(enumerate(self.im_mask_seg_triples)), out, indent=4, sort_keys=True)

    def __len__(self):
        return 1

    @staticmethod
    def _get_filenames(p):
        if path.isdir(p) and listdir(p):
            return sorted(path.join(p, f) for f in listdir(p) if path.isfile(path.join(p, f)))
        return ['' for _ in range(2)]

    def _padded(self, file_path):
        arr, _ = read_nifti(file_path, np.float32)
        if arr.ndim != 3:
            raise ValueError(f'{file_path}: expected a 3-D volume, got shape {arr.shape}')
        if self.im_spacing is None:
            self.im_spacing = torch.tensor(max(arr.shape) / np.asarray(self.dims), dtype=torch.float32)
        if self.padding is None:
            padding = (max(arr.shape) - np.asarray(arr.shape)) // 2
            self.padding = tuple((int(p), int(p)) for p in padding)
        return torch.from_numpy(np.pad(arr, self.padding, mode='minimum')).unsqueeze(0).unsqueeze(0)

    def _get_image(self, im_path):
        return F.interpolate(self._padded(im_path), size=self.dims, mode='trilinear', align_corners=True).squeeze(0)

    def _get_mask(self, mask_path):
        return F.interpolate(self._padded(mask_path), size=self.dims, mode='nearest').bool().squeeze(0)

    def _get_seg(self, seg_path):
        return F.interpolate(self._padded(seg_path), size=self.dims, mode='nearest').short().squeeze(0)

    def _load(self, triple):
        return {'im': self._get_image(triple['im']), 'mask': self._get_mask(triple['mask']), 'seg': self._get_seg(triple['seg'])}

    def __getitem__(self, idx):
        fixed = self._load(self.im_mask_seg_triples[0])
        moving = self._load(self.im_mask_seg_triples[idx + 1])
        var_v = (self.sigma_v_init ** 2) + torch.zeros(self.dims_v)
        var_params_q_v = {'mu': torch.zeros(self.dims_v), 'log_var': var_v.log(), 'u': self.u_v_init + torch.zeros(self.dims_v)}
        return fixed, moving, var_params_q_v
