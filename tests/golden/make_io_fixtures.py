#!/usr/bin/env python3
"""Byte fixtures for the file formats of the reference's logger / data loader, derived from the FORMAT SPECIFICATIONS -- not
from ir_sgmcmc_amd/utils/imageio.py, which they are there to check.

  nifti1_nibabel_2x3x4_f32.nii.gz   what `nib.Nifti1Image(im, np.eye(4))` + `header.set_xyzt_units(2)` + `header.set_zooms(spacing)`
                                    + `to_filename` produce (logger/logger.py:84-102), field by field after nifti1.h (NIfTI-1.1):
                                    sizeof_hdr 348, regular 'r', dim, datatype 16 / bitpix 32, pixdim = (qfac 1, zooms, 1 ...),
                                    vox_offset 352, scl_slope = scl_inter = NaN (nibabel's "no scaling"), xyzt_units 2,
                                    qform_code 0, sform_code 2 with the identity srow_* (set_zooms does not touch the affine),
                                    magic "n+1\0", a zero extension flag, then the voxels in Fortran order (x fastest), little endian.
  nifti1_int16_be_scaled.nii        a big-endian int16 volume with scl_slope 0.5 / scl_inter 10 (what a scanner export looks like)
  vtk_legacy_ascii_field.vtk        what tvtk's `write_data(ImageData)` writes for a `.vtk` name (logger/logger.py:35-60): the
                                    LEGACY format, ASCII (vtkDataWriter's default file type), STRUCTURED_POINTS, point VECTORS
                                    "field" of type double (the reference builds them with dtype=float), x running fastest.
  vtk_legacy_binary_grid.vtk        a BINARY legacy STRUCTURED_GRID (big-endian floats, as the legacy format prescribes)

    python tests/golden/make_io_fixtures.py      (numpy only; rewrites tests/golden/io/)
"""
import gzip
import os
import struct

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'io')


def nifti1_header(endian, dim, datatype, bitpix, pixdim, slope, inter, xyzt_units, sform_code, srow):
    h = bytearray(348)
    put = lambda fmt, off, *v: struct.pack_into(endian + fmt, h, off, *v)
    put('i', 0, 348)                  # sizeof_hdr
    # data_type[10] @4, db_name[18] @14: unused, blank;  extents @32 = 0;  session_error @36 = 0
    h[38:39] = b'r'                   # regular
    h[39] = 0                         # dim_info
    put('8h', 40, *dim)               # dim[8]
    put('3f', 56, 0.0, 0.0, 0.0)      # intent_p1..p3
    put('h', 68, 0)                   # intent_code
    put('h', 70, datatype)            # datatype
    put('h', 72, bitpix)              # bitpix
    put('h', 74, 0)                   # slice_start
    put('8f', 76, *pixdim)            # pixdim[8]
    put('f', 108, 352.0)              # vox_offset
    put('f', 112, slope)              # scl_slope
    put('f', 116, inter)              # scl_inter
    put('h', 120, 0)                  # slice_end
    h[122] = 0                        # slice_code
    h[123] = xyzt_units               # xyzt_units
    put('4f', 124, 0.0, 0.0, 0.0, 0.0)  # cal_max, cal_min, slice_duration, toffset
    put('2i', 140, 0, 0)              # glmax, glmin
    # descrip[80] @148, aux_file[24] @228: blank
    put('h', 252, 0)                  # qform_code
    put('h', 254, sform_code)         # sform_code
    put('6f', 256, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)  # quatern_b, c, d, qoffset_x, y, z
    for r in range(3):
        put('4f', 280 + 16 * r, *srow[r])
    # intent_name[16] @328: blank
    h[344:348] = b'n+1\x00'           # magic
    return bytes(h) + b'\x00\x00\x00\x00'   # + extension flag: no extensions; voxels start at 352


def main():
    os.makedirs(HERE, exist_ok=True)
    eye = [(1.0, 0.0, 0.0, 0.0), (0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 1.0, 0.0)]
    # ---- 1. nibabel-style float32
    im = (np.arange(24, dtype=np.float32).reshape(2, 3, 4) * 0.25 - 1.0)       # im[x, y, z]
    hdr = nifti1_header('<', (3, 2, 3, 4, 1, 1, 1, 1), 16, 32, (1.0, 1.5, 2.0, 2.5, 1.0, 1.0, 1.0, 1.0), float('nan'), float('nan'), 2, 2, eye)
    vox = b''.join(struct.pack('<f', float(im[x, y, z])) for z in range(4) for y in range(3) for x in range(2))   # x fastest
    with open(os.path.join(HERE, 'nifti1_nibabel_2x3x4_f32.nii.gz'), 'wb') as f:
        with gzip.GzipFile(filename='', mode='wb', fileobj=f, mtime=0) as g:   # mtime 0: reproducible bytes
            g.write(hdr + vox)
    # ---- 2. big-endian int16 with scaling
    lab = np.arange(24, dtype=np.int16).reshape(2, 3, 4) - 5
    hdr = nifti1_header('>', (3, 2, 3, 4, 1, 1, 1, 1), 4, 16, (1.0, 0.9, 0.9, 3.0, 0.0, 0.0, 0.0, 0.0), 0.5, 10.0, 10, 1, [(0.9, 0, 0, -12.0), (0, 0.9, 0, -20.0), (0, 0, 3.0, 5.0)])
    vox = b''.join(struct.pack('>h', int(lab[x, y, z])) for z in range(4) for y in range(3) for x in range(2))
    open(os.path.join(HERE, 'nifti1_int16_be_scaled.nii'), 'wb').write(hdr + vox)
    # ---- 3. legacy VTK, ASCII, double vectors (field[c, x, y, z] = 100 c + x + 10 y + 0.5 z)
    nx, ny, nz = 2, 3, 2
    lines = ['# vtk DataFile Version 3.0', 'vtk output', 'ASCII', 'DATASET STRUCTURED_POINTS', f'DIMENSIONS {nx} {ny} {nz}',
             'SPACING 1.5 2 2.5', 'ORIGIN 0 0 0', f'POINT_DATA {nx * ny * nz}', 'VECTORS field double']
    vals = []
    for z in range(nz):
        for y in range(ny):
            for x in range(nx):
                vals += [repr(100.0 * c + x + 10.0 * y + 0.5 * z) for c in range(3)]
    for i in range(0, len(vals), 9):                     # VTK breaks ASCII data lines after nine values
        lines.append(' '.join(vals[i:i + 9]) + ' ')
    open(os.path.join(HERE, 'vtk_legacy_ascii_field.vtk'), 'w').write('\n'.join(lines) + '\n')
    # ---- 4. legacy VTK, BINARY structured grid, float (big endian)
    head = f'# vtk DataFile Version 3.0\nvtk output\nBINARY\nDATASET STRUCTURED_GRID\nDIMENSIONS {nx} {ny} {nz}\nPOINTS {nx * ny * nz} float\n'.encode()
    body = b''.join(struct.pack('>3f', -1.0 + 2.0 * x, -1.0 + y, 0.25 * z) for z in range(nz) for y in range(ny) for x in range(nx))
    open(os.path.join(HERE, 'vtk_legacy_binary_grid.vtk'), 'wb').write(head + body + b'\n')
    for n in sorted(os.listdir(HERE)):
        print(n, os.path.getsize(os.path.join(HERE, n)), 'bytes')


if __name__ == '__main__':
    main()
