"""File formats of the reference's data loader and logger, re-implemented without SimpleITK / nibabel / tvtk (CPU tests)."""
import gzip
import os
import struct

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ir_sgmcmc_amd.data_loader import BiobankDataLoader
from ir_sgmcmc_amd.data_loader.datasets import BiobankDataset
from ir_sgmcmc_amd.logger import save_field_to_disk, save_grid_to_disk, save_im_to_disk, save_sample
from ir_sgmcmc_amd.utils.imageio import read_nifti, read_vtk_vectors, write_nifti


def test_nifti_header_is_what_nibabel_writes(tmp_path):
    im = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    p = tmp_path / 'a.nii.gz'
    save_im_to_disk(im, str(p), spacing=(1.5, 2.0, 2.5))
    raw = gzip.open(p, 'rb').read()
    assert struct.unpack('<i', raw[:4])[0] == 348 and raw[344:348] == b'n+1\x00'
    assert struct.unpack('<8h', raw[40:56]) == (3, 2, 3, 4, 1, 1, 1, 1)
    assert struct.unpack('<hh', raw[70:74]) == (16, 32)                       # float32
    assert struct.unpack('<3f', raw[80:92]) == (1.5, 2.0, 2.5)                # zooms
    assert raw[123] == 2                                                      # xyzt_units = mm (set_xyzt_units(2))
    assert struct.unpack('<f', raw[108:112])[0] == 352.0
    # x runs fastest on disk
    data = np.frombuffer(raw, dtype='<f4', offset=352)
    assert data[0] == im[0, 0, 0] and data[1] == im[1, 0, 0] and data[2] == im[0, 1, 0]
    back, zooms = read_nifti(str(p))
    assert np.array_equal(back, im) and zooms == (1.5, 2.0, 2.5)


IO_FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'io')


def test_reader_and_writer_against_spec_derived_nibabel_bytes(tmp_path):
    """tests/golden/io/nifti1_nibabel_2x3x4_f32.nii.gz was packed field by field after nifti1.h the way nibabel -- the reference's
    writer, logger/logger.py:84-102 -- fills the header (tests/golden/make_io_fixtures.py; NOT with utils/imageio.py): the reader
    must decode it (including nibabel's NaN "no scaling" slope), and the writer must produce the same 352 header bytes."""
    raw = gzip.open(os.path.join(IO_FIXTURES, 'nifti1_nibabel_2x3x4_f32.nii.gz'), 'rb').read()
    im = np.arange(24, dtype=np.float32).reshape(2, 3, 4) * 0.25 - 1.0
    back, zooms = read_nifti(os.path.join(IO_FIXTURES, 'nifti1_nibabel_2x3x4_f32.nii.gz'))
    assert back.dtype == np.float32 and np.array_equal(back, im) and zooms == (1.5, 2.0, 2.5)
    p = tmp_path / 'mine.nii.gz'
    save_im_to_disk(im, str(p), spacing=(1.5, 2.0, 2.5))
    mine = gzip.open(p, 'rb').read()
    assert mine[:352] == raw[:352], [i for i in range(352) if mine[i] != raw[i]]
    assert mine[352:] == raw[352:]


def test_reader_against_spec_derived_scanner_style_bytes():
    back, zooms = read_nifti(os.path.join(IO_FIXTURES, 'nifti1_int16_be_scaled.nii'))
    lab = (np.arange(24, dtype=np.int16).reshape(2, 3, 4) - 5).astype(np.float32)
    assert np.allclose(back, lab * 0.5 + 10.0) and zooms == pytest.approx((0.9, 0.9, 3.0))


def test_vtk_reader_against_spec_derived_legacy_files(tmp_path):
    """the reference writes ASCII legacy files with double vectors (tvtk write_data, dtype=float) and reads them back with
    vtkStructuredPointsReader (utils/util.py:94-111; round trip in tests/test_utils.py:153-159)"""
    kind, dims, f = read_vtk_vectors(os.path.join(IO_FIXTURES, 'vtk_legacy_ascii_field.vtk'))
    x, y, z = np.meshgrid(np.arange(2.0), np.arange(3.0), np.arange(2.0), indexing='ij')
    want = np.stack([100.0 * c + x + 10.0 * y + 0.5 * z for c in range(3)]).astype(np.float32)
    assert kind == 'STRUCTURED_POINTS' and dims == (2, 3, 2) and np.array_equal(f, want)
    kind, dims, g = read_vtk_vectors(os.path.join(IO_FIXTURES, 'vtk_legacy_binary_grid.vtk'))
    wantg = np.stack([-1.0 + 2.0 * x, -1.0 + y, 0.25 * z]).astype(np.float32)
    assert kind == 'STRUCTURED_GRID' and np.array_equal(g, wantg)
    # and the writer's own files read back the same way
    save_field_to_disk(torch.from_numpy(want), str(tmp_path / 'f.vtk'), spacing=(1.5, 2.0, 2.5))
    assert np.array_equal(read_vtk_vectors(str(tmp_path / 'f.vtk'))[2], want)


@pytest.mark.parametrize('dtype', [np.uint8, np.int16, np.int32, np.float32, np.float64])
@pytest.mark.parametrize('ext', ['nii', 'nii.gz'])
def test_nifti_round_trip(tmp_path, dtype, ext):
    rng = np.random.default_rng(0)
    im = (rng.random((5, 7, 3)) * 100).astype(dtype)
    p = str(tmp_path / f'x.{ext}')
    write_nifti(im, p)
    back, _ = read_nifti(p, np.float64)
    assert np.array_equal(back, im.astype(np.float64))


def test_nifti_big_endian_and_scaling(tmp_path):
    im = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    hdr = bytearray(352)
    struct.pack_into('>i', hdr, 0, 348)
    struct.pack_into('>8h', hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into('>hh', hdr, 70, 4, 16)
    struct.pack_into('>8f', hdr, 76, 1, 1, 1, 1, 1, 1, 1, 1)
    struct.pack_into('>f', hdr, 108, 352.0)
    struct.pack_into('>ff', hdr, 112, 0.5, 10.0)     # scl_slope, scl_inter
    hdr[344:348] = b'n+1\x00'
    p = tmp_path / 'be.nii'
    p.write_bytes(bytes(hdr) + im.astype('>i2').tobytes(order='F'))
    back, _ = read_nifti(str(p))
    assert np.allclose(back, im * 0.5 + 10.0)
    with pytest.raises(ValueError):
        (tmp_path / 'bad.nii').write_bytes(b'\x00' * 400)
        read_nifti(str(tmp_path / 'bad.nii'))


def test_vtk_writers(tmp_path):
    rng = np.random.default_rng(1)
    field = rng.standard_normal((3, 4, 5, 6)).astype(np.float32)
    p = str(tmp_path / 'f.vtk')
    save_field_to_disk(torch.from_numpy(field), p, spacing=torch.tensor([2.0, 2.0, 2.0]))
    head = open(p, 'rb').read(200).decode(errors='replace')
    assert head.startswith('# vtk DataFile Version 3.0') and 'DATASET STRUCTURED_POINTS' in head
    assert 'DIMENSIONS 4 5 6' in head and 'SPACING 2.0 2.0 2.0' in head and 'VECTORS field float' in head
    kind, dims, back = read_vtk_vectors(p)
    assert kind == 'STRUCTURED_POINTS' and dims == (4, 5, 6) and np.array_equal(back, field)
    g = str(tmp_path / 'g.vtk')
    save_grid_to_disk(torch.from_numpy(field), g)
    kind, dims, back = read_vtk_vectors(g)
    assert kind == 'STRUCTURED_GRID' and np.array_equal(back, field)


def _write_triples(root, shapes, seed=0):
    rng = np.random.default_rng(seed)
    (root / 'masks').mkdir(parents=True)
    (root / 'segs').mkdir()
    vols = []
    for i, shp in enumerate(shapes):
        im = rng.random(shp).astype(np.float32)
        mask = (rng.random(shp) > 0.3).astype(np.uint8)
        seg = rng.integers(0, 5, shp).astype(np.int16)
        write_nifti(im, str(root / f'im_{i}.nii.gz'))
        write_nifti(mask, str(root / 'masks' / f'im_{i}.nii.gz'))
        write_nifti(seg, str(root / 'segs' / f'im_{i}.nii.gz'))
        vols.append((im, mask, seg))
    return vols


def test_biobank_dataset_pipeline(tmp_path):
    """pad with the minimum to a cube, resize: trilinear/align_corners for images, nearest for masks and segmentations
    (data_loader/datasets.py:70-105)"""
    vols = _write_triples(tmp_path / 'data', [(10, 14, 12), (10, 14, 12)])
    out = tmp_path / 'out'
    out.mkdir()
    dims = (8, 8, 8)
    ds = BiobankDataset(dims, str(tmp_path / 'data'), {'dir': out}, sigma_v_init=0.5, u_v_init=0.1, cps=None)
    assert len(ds) == 1 and (out / 'idx_to_biobank_ID.json').is_file()
    fixed, moving, vp = ds[0]
    im, mask, seg = vols[0]
    pad = ((2, 2), (0, 0), (1, 1))
    ref = F.interpolate(torch.from_numpy(np.pad(im, pad, mode='minimum'))[None, None], size=dims, mode='trilinear', align_corners=True)[0]
    assert torch.equal(fixed['im'], ref) and fixed['im'].shape == (1, 8, 8, 8)
    refm = F.interpolate(torch.from_numpy(np.pad(mask.astype(np.float32), pad, mode='minimum'))[None, None], size=dims, mode='nearest').bool()[0]
    assert torch.equal(fixed['mask'], refm) and fixed['mask'].dtype == torch.bool
    assert fixed['seg'].dtype == torch.int16 and moving['im'].shape == (1, 8, 8, 8)
    assert torch.allclose(ds.im_spacing, torch.tensor([14 / 8] * 3))
    assert vp['mu'].shape == (3, 8, 8, 8) and torch.allclose(vp['log_var'], torch.full((3, 8, 8, 8), 0.25).log())
    assert float(vp['u'][0, 0, 0, 0]) == pytest.approx(0.1)
    # loader contract: one (fixed, moving, var_params) triple with a batch dimension
    dl = BiobankDataLoader(data_dir=str(tmp_path / 'data'), dims=dims, save_dirs={'dir': out})
    (f, m, v), = list(dl)
    assert f['im'].shape == (1, 1, 8, 8, 8) and v['mu'].shape == (1, 3, 8, 8, 8) and dl.im_spacing is not None
    # no data: an error, as in the reference (its listdir raises) -- never a silent run on fake data
    with pytest.raises(FileNotFoundError):
        BiobankDataLoader(data_dir=str(tmp_path / 'nowhere'), dims=dims)
    # ... unless the config asks for the synthetic pair in so many words
    dl = BiobankDataLoader(data_dir=str(tmp_path / 'nowhere'), dims=dims, allow_synthetic_fallback=True)
    (f, m, v), = list(dl)
    assert f['im'].shape == (1, 1, 8, 8, 8) and dl.im_spacing is None
    # a config WITHOUT data_dir behaves the same: the explanatory error, or the fallback when asked for (not a raw TypeError)
    with pytest.raises(FileNotFoundError, match='allow_synthetic_fallback'):
        BiobankDataLoader(data_dir=None, dims=dims)
    (f, m, v), = list(BiobankDataLoader(data_dir=None, dims=dims, allow_synthetic_fallback=True))
    assert f['im'].shape == (1, 1, 8, 8, 8)


def test_save_sample_layout(tmp_path):
    dirs = {'samples': tmp_path / 'samples', 'images': tmp_path / 'images', 'fields': tmp_path / 'fields'}
    save_sample(dirs, torch.tensor([2.0, 2.0, 2.0]), 12, torch.rand(1, 1, 4, 4, 4), torch.rand(1, 3, 4, 4, 4), torch.rand(1, 4, 4, 4),
                'MCMC', chain_no=1)
    names = sorted(p.name for p in (tmp_path / 'samples' / 'MCMC').iterdir())
    assert names == ['chain_1_sample_0000012_displacement.vtk', 'chain_1_sample_0000012_im_moving_warped.nii.gz',
                     'chain_1_sample_0000012_log_det_J.nii.gz']
