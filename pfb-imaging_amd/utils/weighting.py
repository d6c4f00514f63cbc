"""uv-cell counts and Briggs weights on the GPU.

Mirrors /root/reference/src/pfb_imaging/utils/weighting.py: ``_compute_counts`` (:81-140) and
``counts_to_weights`` (:143-208).  The integer uv-cell index is bit-exact against the CPU
oracle; the scatter-add uses hardware f64 atomics, so counts agree to summation order.
"""

import numpy as np

from .. import _lib
from .._lib import as_c, check, f64, i64, lib, ptr


def uvcell_index(uvw, freq, mask, nx, ny, cell_size_x, cell_size_y, usign=1.0, vsign=-1.0):
    """Flat index ``u_idx*ny + v_idx`` per visibility, -1 when masked or out of bounds."""
    _lib.require_gpu()
    uvw, freq, mask = as_c(uvw, np.float64), as_c(freq, np.float64), as_c(mask, np.uint8)
    nrow, nchan = mask.shape
    cell = np.empty((nrow, nchan), dtype=np.int64)
    check(lib().pfbhip_uvcell_index(ptr(uvw), ptr(freq), ptr(mask), i64(nrow), i64(nchan), i64(nx), i64(ny),
                                    f64(cell_size_x), f64(cell_size_y), f64(usign), f64(vsign), ptr(cell)))
    return cell


def _compute_counts(uvw, freq, mask, wgt, nx, ny, cell_size_x, cell_size_y, dtype, ngrid=1, usign=1.0, vsign=-1.0):
    """weighting.py:81-140.  ``ngrid`` (private per-thread grids on the CPU) is irrelevant here."""
    _lib.require_gpu()
    uvw, freq, mask = as_c(uvw, np.float64), as_c(freq, np.float64), as_c(mask, np.uint8)
    wgt = as_c(wgt, np.float64)
    ncorr, nrow, nchan = wgt.shape
    counts = np.zeros((ncorr, nx, ny), dtype=np.float64)
    check(lib().pfbhip_compute_counts(ptr(uvw), ptr(freq), ptr(mask), ptr(wgt), i64(ncorr), i64(nrow), i64(nchan),
                                      i64(nx), i64(ny), f64(cell_size_x), f64(cell_size_y), f64(usign), f64(vsign),
                                      ptr(counts)))
    return counts.astype(dtype, copy=False)


def counts_to_weights(counts, uvw, freq, weight, mask, nx, ny, cell_size_x, cell_size_y, robust, usign=1.0,
                      vsign=-1.0):
    """weighting.py:143-208; mutates ``counts`` and ``weight`` in place like the reference."""
    if not counts.any():
        return weight
    _lib.require_gpu()
    ncorr, nrow, nchan = weight.shape
    if robust > -2:
        numsqrt = 5 * 10 ** (-robust)
        avgwnum = (counts.reshape(ncorr, -1) ** 2).sum(axis=1)
        avgwden = counts.reshape(ncorr, -1).sum(axis=1)
        ssq = numsqrt * numsqrt * avgwden / avgwnum
        counts *= ssq[:, None, None]
        counts += 1
    uvw, freq, mask = as_c(uvw, np.float64), as_c(freq, np.float64), as_c(mask, np.uint8)
    w = weight if (weight.flags.c_contiguous and weight.dtype == np.float64) else np.array(weight, dtype=np.float64)
    c = as_c(counts, np.float64)
    check(lib().pfbhip_counts_divide(ptr(uvw), ptr(freq), ptr(mask), ptr(c), i64(ncorr), i64(nrow), i64(nchan),
                                     i64(nx), i64(ny), f64(cell_size_x), f64(cell_size_y), f64(usign), f64(vsign),
                                     ptr(w)))
    if w is not weight:
        weight[...] = w
    return weight


def filter_extreme_counts(counts, level=10.0):
    """weighting.py:212-226: positive counts below ``median(positive)/level`` are raised to it (in place)."""
    if not level:
        return counts
    _lib.require_gpu()
    c = counts if (counts.flags.c_contiguous and counts.dtype == np.float64) else np.array(counts, dtype=np.float64)
    check(lib().pfbhip_filter_extreme_counts(ptr(c), i64(c.size), f64(level), None))
    if c is not counts:
        counts[...] = c
    return counts


def box_sum_counts(counts, npix_super):
    """weighting.py:229-254: (2 npix_super + 1)^2 box sum with zero padding; identity when disabled."""
    if npix_super is None or npix_super <= 0:
        return counts
    assert np.issubdtype(counts.dtype, np.floating), (
        f"box_sum_counts requires a floating-point counts array; got dtype={counts.dtype}"
    )
    _lib.require_gpu()
    c = as_c(counts, np.float64)
    ncorr, nx, ny = c.shape
    out = _lib.result_empty(c.shape, np.float64)
    check(lib().pfbhip_box_sum_counts(ptr(c), i64(ncorr), i64(nx), i64(ny), i64(int(npix_super)), ptr(out)))
    return out.astype(counts.dtype, copy=False)


def imaging_weights(uvw, freq, mask, weight, nx_pad, ny_pad, cell_size_x, cell_size_y, robust, filter_level=5.0,
                    npix_super=0, usign=1.0, vsign=-1.0, return_counts=False):
    """The whole imaging-weight chain of ``image_data_products`` (operators/gridder.py:534-576) in one device pipeline:
    ``_compute_counts`` -> ``filter_extreme_counts`` -> ``box_sum_counts`` -> ``counts_to_weights``.  ``weight``
    ``(ncorr, nrow, nchan)`` is updated in place and returned (with the final counts when ``return_counts``); the counts
    grid never visits the host otherwise and uvw / mask / weights are uploaded once instead of three times."""
    _lib.require_gpu()
    uvw, freq, mask = as_c(uvw, np.float64), as_c(freq, np.float64), as_c(mask, np.uint8)
    ncorr, nrow, nchan = weight.shape
    w = weight if (weight.flags.c_contiguous and weight.dtype == np.float64) else np.array(weight, dtype=np.float64)
    counts = _lib.result_empty((ncorr, nx_pad, ny_pad), np.float64) if return_counts else None
    check(lib().pfbhip_imaging_weights(ptr(uvw), ptr(freq), ptr(mask), ptr(w), i64(ncorr), i64(nrow), i64(nchan), i64(nx_pad),
                                       i64(ny_pad), f64(cell_size_x), f64(cell_size_y), f64(usign), f64(vsign), f64(robust),
                                       f64(filter_level or 0.0), i64(int(npix_super or 0)), ptr(counts)))
    if w is not weight:
        weight[...] = w
    return (weight, counts) if return_counts else weight
