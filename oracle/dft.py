"""Exact (direct-DFT) measurement equation -- the definition parity is pinned to.

Conventions follow the reference's gridder call sites:
  * ``flip_u/flip_v/flip_w`` negate the corresponding baseline coordinate;
  * pixel ``(ix, iy)`` sits at ``l = lshift + (ix - nx/2) pixsize_x``,
    ``m = mshift + (iy - ny/2) pixsize_y`` with ``lshift = -center_x if flip_u
    else center_x``, ``mshift = -center_y if flip_v else center_y``
    (/root/reference/tests/test_hessian_approx.py:145-149 for the flip_v case;
    the flip_u rule is the symmetric assumption -- the reference never sets it);
  * ``vis = sum_pix dirty * exp(-2 pi i f/c (u l + v m - w (n-1))) [/ n]``
    (/root/reference/tests/test_hessian_approx.py:44-67) and ``vis2dirty`` is
    its adjoint with weights and mask.
"""

import numpy as np

from ._lib import P, cint, f64, i64, lib, ptr


def _signs(flip_u, flip_v, flip_w):
    return (-1.0 if flip_u else 1.0, -1.0 if flip_v else 1.0, -1.0 if flip_w else 1.0)


def _shifts(center_x, center_y, flip_u, flip_v):
    return (-center_x if flip_u else center_x, -center_y if flip_v else center_y)


def dft_vis2dirty(uvw, freq, vis, wgt, mask, npix_x, npix_y, pixsize_x, pixsize_y, center_x=0.0, center_y=0.0,
                  flip_u=False, flip_v=False, flip_w=False, do_wgridding=True, divide_by_n=True, pixels=None):
    """Direct DFT dirty image; ``pixels=(ix, iy)`` restricts to a subset (returns 1-D)."""
    uvw = np.ascontiguousarray(uvw, dtype=np.float64)
    freq = np.ascontiguousarray(freq, dtype=np.float64)
    vis = np.ascontiguousarray(vis, dtype=np.complex128)
    wgt = None if wgt is None else np.ascontiguousarray(wgt, dtype=np.float64)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    nrow, nchan = vis.shape
    su, sv, sw = _signs(flip_u, flip_v, flip_w)
    ls, ms = _shifts(center_x, center_y, flip_u, flip_v)
    if pixels is None:
        ix, iy = np.meshgrid(np.arange(npix_x), np.arange(npix_y), indexing="ij")
        ix = ix.ravel()
        iy = iy.ravel()
    else:
        ix, iy = pixels
    ix = np.ascontiguousarray(ix, dtype=np.int64)
    iy = np.ascontiguousarray(iy, dtype=np.int64)
    out = np.zeros(ix.size, dtype=np.float64)
    lib().pfbo_dft_vis2dirty(i64(nrow), i64(nchan), ptr(uvw), ptr(freq), ptr(vis.view(np.float64)), ptr(wgt), ptr(mask),
                             f64(su), f64(sv), f64(sw), i64(npix_x), i64(npix_y), f64(pixsize_x), f64(pixsize_y),
                             f64(ls), f64(ms), cint(int(do_wgridding)), cint(int(divide_by_n)), i64(ix.size),
                             ptr(ix), ptr(iy), ptr(out))
    return out.reshape(npix_x, npix_y) if pixels is None else out


def dft_dirty2vis(uvw, freq, dirty, pixsize_x, pixsize_y, center_x=0.0, center_y=0.0, flip_u=False, flip_v=False,
                  flip_w=False, do_wgridding=True, divide_by_n=True, rows=None, chans=None):
    """Direct DFT visibilities; ``rows``/``chans`` (equal-length index arrays) restrict to a subset."""
    uvw = np.ascontiguousarray(uvw, dtype=np.float64)
    freq = np.ascontiguousarray(freq, dtype=np.float64)
    dirty = np.ascontiguousarray(dirty, dtype=np.float64)
    nx, ny = dirty.shape
    su, sv, sw = _signs(flip_u, flip_v, flip_w)
    ls, ms = _shifts(center_x, center_y, flip_u, flip_v)
    full = rows is None
    if full:
        rows, chans = np.meshgrid(np.arange(uvw.shape[0]), np.arange(freq.size), indexing="ij")
        rows = rows.ravel()
        chans = chans.ravel()
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    chans = np.ascontiguousarray(chans, dtype=np.int64)
    out = np.zeros(rows.size, dtype=np.complex128)
    lib().pfbo_dft_dirty2vis(i64(rows.size), ptr(rows), ptr(chans), ptr(uvw), ptr(freq), f64(su), f64(sv), f64(sw),
                             i64(nx), i64(ny), f64(pixsize_x), f64(pixsize_y), f64(ls), f64(ms),
                             cint(int(do_wgridding)), cint(int(divide_by_n)), ptr(dirty), ptr(out.view(np.float64)))
    return out.reshape(uvw.shape[0], freq.size) if full else out
