"""Image and field file I/O without SimpleITK / nibabel / tvtk (none of which exist in the ROCm image).

Formats are the ones the reference reads and writes:
  * NIfTI-1 single file, optionally gzipped (`.nii`, `.nii.gz`): read by data_loader/datasets.py:70-105 through SimpleITK,
    written by logger/logger.py:83-100 through nibabel (identity affine, units mm, zooms = spacing);
  * legacy VTK `.vtk`: vector fields as STRUCTURED_POINTS + VECTORS "field" (logger/logger.py:35-60) and sampling grids as
    STRUCTURED_GRID (logger/logger.py:63-80), both with x running fastest as VTK requires.
Arrays use the reference's index order [x][y][z] (nibabel's `get_fdata()`; SimpleITK's array transposed (2, 1, 0)).
Host-side, numpy only; nothing here is on the SG-MCMC hot path.
"""
import gzip
import struct

import numpy as np

# NIfTI-1 datatype codes <-> numpy
_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
                 768: np.uint32, 1024: np.int64, 1280: np.uint64}
_NIFTI_CODES = {np.dtype(v): k for k, v in _NIFTI_DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith('.gz') else open(path, mode)


def read_nifti(path, dtype=np.float32):
    """-> (array of shape (nx, ny, nz[, nt...]) as `dtype` with scl_slope / scl_inter applied, zooms (dx, dy, dz))"""
    with _open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError(f'{path}: too short for a NIfTI-1 header')
    for endian in ('<', '>'):
        if struct.unpack(endian + 'i', raw[0:4])[0] == 348:
            break
    else:
        raise ValueError(f'{path}: sizeof_hdr is not 348 (not a NIfTI-1 file)')
    magic = raw[344:348]
    if magic[:3] not in (b'n+1', b'ni1'):
        raise ValueError(f'{path}: bad NIfTI magic {magic!r}')
    if magic[:3] == b'ni1':
        raise ValueError(f'{path}: header/image pairs (.hdr/.img) are not supported, only single-file NIfTI')
    dim = struct.unpack(endian + '8h', raw[40:56])
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError(f'{path}: dim[0] = {ndim}')
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    datatype, bitpix = struct.unpack(endian + 'hh', raw[70:74])
    if datatype not in _NIFTI_DTYPES:
        raise ValueError(f'{path}: unsupported NIfTI datatype code {datatype}')
    pixdim = struct.unpack(endian + '8f', raw[76:108])
    vox_offset = int(struct.unpack(endian + 'f', raw[108:112])[0])
    slope, inter = struct.unpack(endian + 'ff', raw[112:120])
    dt = np.dtype(_NIFTI_DTYPES[datatype]).newbyteorder(endian)
    n = int(np.prod(shape))
    if len(raw) < vox_offset + n * dt.itemsize:
        raise ValueError(f'{path}: truncated image data')
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=max(vox_offset, 352)).reshape(shape, order='F')
    arr = arr.astype(dtype)
    # nifti1.h: scl_slope == 0 means "no scaling"; nibabel -- the reference's writer -- stores NaN in both fields for the same
    if not np.isfinite(slope) or slope == 0.0:
        slope = 1.0
    if not np.isfinite(inter):
        inter = 0.0
    if slope != 1.0 or inter != 0.0:
        arr = arr * dtype(slope) + dtype(inter)
    while arr.ndim > 3 and arr.shape[-1] == 1:
        arr = arr[..., 0]
    return np.ascontiguousarray(arr), tuple(float(p) for p in pixdim[1:4])


def write_nifti(arr, path, spacing=(1.0, 1.0, 1.0)):
    """save_im_to_disk (logger/logger.py:83-100): identity affine, xyzt_units = mm, zooms = spacing."""
    arr = np.asarray(arr)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype == np.float16:
        arr = arr.astype(np.float32)
    if np.dtype(arr.dtype) not in _NIFTI_CODES:
        raise ValueError(f'unsupported dtype {arr.dtype}')
    if not 1 <= arr.ndim <= 7:
        raise ValueError('NIfTI-1 holds 1 to 7 dimensions')
    spacing = [float(s) for s in np.asarray(spacing).reshape(-1)][:3]
    hdr = bytearray(352)
    struct.pack_into('<i', hdr, 0, 348)
    hdr[38:39] = b'r'                              # regular
    dim = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into('<8h', hdr, 40, *dim)
    struct.pack_into('<hh', hdr, 70, _NIFTI_CODES[np.dtype(arr.dtype)], arr.dtype.itemsize * 8)
    pixdim = [1.0] + spacing + [1.0] * (7 - len(spacing))
    struct.pack_into('<8f', hdr, 76, *pixdim)
    struct.pack_into('<f', hdr, 108, 352.0)        # vox_offset
    struct.pack_into('<ff', hdr, 112, float('nan'), float('nan'))   # scl_slope, scl_inter: NaN = no scaling, as nibabel writes
    hdr[123] = 2                                   # xyzt_units: NIFTI_UNITS_MM
    struct.pack_into('<hh', hdr, 252, 0, 2)        # qform_code 0, sform_code 2 (aligned), as nibabel writes for an affine
    struct.pack_into('<4f', hdr, 280, 1.0, 0.0, 0.0, 0.0)   # srow_x .. srow_z = identity
    struct.pack_into('<4f', hdr, 296, 0.0, 1.0, 0.0, 0.0)
    struct.pack_into('<4f', hdr, 312, 0.0, 0.0, 1.0, 0.0)
    hdr[344:348] = b'n+1\x00'
    data = np.asarray(arr, dtype=arr.dtype.newbyteorder('<')).tobytes(order='F')
    with _open(path, 'wb') as f:
        f.write(bytes(hdr))
        f.write(data)


def _vtk_header(title, kind):
    return f'# vtk DataFile Version 3.0\n{title}\nBINARY\nDATASET {kind}\n'.encode()


def write_vtk_field(field, path, spacing=(1.0, 1.0, 1.0)):
    """save_field_to_disk (logger/logger.py:35-60): field (3, nx, ny, nz) -> STRUCTURED_POINTS with point VECTORS 'field'."""
    field = np.asarray(field, dtype=np.float32)
    if field.ndim != 4 or field.shape[0] != 3:
        raise ValueError('field must have shape (3, nx, ny, nz)')
    nx, ny, nz = field.shape[1:]
    sp = [float(s) for s in np.asarray(spacing).reshape(-1)][:3]
    vec = np.stack([field[0], field[1], field[2]], axis=-1).transpose(2, 1, 0, 3).reshape(-1, 3)   # x fastest
    with open(path, 'wb') as f:
        f.write(_vtk_header('field', 'STRUCTURED_POINTS'))
        f.write(f'DIMENSIONS {nx} {ny} {nz}\nORIGIN 0 0 0\nSPACING {sp[0]} {sp[1]} {sp[2]}\n'.encode())
        f.write(f'POINT_DATA {nx * ny * nz}\nVECTORS field float\n'.encode())
        f.write(vec.astype('>f4').tobytes())
        f.write(b'\n')


def write_vtk_grid(grid, path):
    """save_grid_to_disk (logger/logger.py:63-80): grid (3, nx, ny, nz) of point coordinates -> STRUCTURED_GRID."""
    grid = np.asarray(grid, dtype=np.float32)
    if grid.ndim != 4 or grid.shape[0] != 3:
        raise ValueError('grid must have shape (3, nx, ny, nz)')
    nx, ny, nz = grid.shape[1:]
    pts = np.stack([grid[0], grid[1], grid[2]], axis=-1).transpose(2, 1, 0, 3).reshape(-1, 3)
    with open(path, 'wb') as f:
        f.write(_vtk_header('grid', 'STRUCTURED_GRID'))
        f.write(f'DIMENSIONS {nx} {ny} {nz}\nPOINTS {nx * ny * nz} float\n'.encode())
        f.write(pts.astype('>f4').tobytes())
        f.write(b'\n')


def read_vtk_vectors(path):
    """Legacy `.vtk` reader for what the reference writes and reads back (logger/logger.py:35-80, utils/util.py:94-111):
    STRUCTURED_POINTS with point VECTORS, or STRUCTURED_GRID POINTS; ASCII (tvtk's default for a `.vtk` name) or BINARY
    (big-endian, as the legacy format prescribes); float or double.  -> (kind, dims, (3, nx, ny, nz) float32)."""
    raw = open(path, 'rb').read()
    lines, pos = [], 0
    kind = dims = dtype = None
    binary = False
    while pos < len(raw):
        end = raw.index(b'\n', pos) if b'\n' in raw[pos:] else len(raw)
        line = raw[pos:end].decode(errors='replace').strip()
        pos = end + 1
        lines.append(line)
        up = line.upper()
        if up == 'BINARY':
            binary = True
        elif up.startswith('DATASET'):
            kind = line.split()[1]
        elif up.startswith('DIMENSIONS'):
            dims = tuple(int(t) for t in line.split()[1:4])
        elif up.startswith('VECTORS') or up.startswith('POINTS'):
            dtype = line.split()[-1].lower()
            break
    if kind is None or dims is None or dtype not in ('float', 'double'):
        raise ValueError(f'{path}: not a legacy VTK structured points / grid file with float or double vectors')
    n = dims[0] * dims[1] * dims[2]
    if binary:
        vec = np.frombuffer(raw, dtype='>f4' if dtype == 'float' else '>f8', count=3 * n, offset=pos)
    else:
        vec = np.array(raw[pos:].split()[:3 * n], dtype=np.float64)
        if vec.size != 3 * n:
            raise ValueError(f'{path}: {vec.size} values for {n} points')
    vec = vec.reshape(dims[2], dims[1], dims[0], 3)     # x runs fastest
    return kind, dims, np.ascontiguousarray(vec.transpose(3, 2, 1, 0).astype(np.float32))
