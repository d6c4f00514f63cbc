"""uv-cell index map, counts and Briggs weights (TEST INFRASTRUCTURE).

Restated from /root/reference/src/pfb_imaging/utils/weighting.py:81-140
(``_compute_counts``) and :143-208 (``counts_to_weights``).  The integer cell
index is the bit-exact target of the HIP kernels; C loops live in
oracle/pfb_oracle.c.
"""

import numpy as np

from ._lib import f64, i64, lib, ptr


def uvcell_index(uvw, freq, mask, nx, ny, cell_x, cell_y, usign=1.0, vsign=-1.0):
    """Flat cell index ``u_idx*ny + v_idx`` per visibility, -1 if masked / out of bounds."""
    uvw = np.ascontiguousarray(uvw, dtype=np.float64)
    freq = np.ascontiguousarray(freq, dtype=np.float64)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    nrow, nchan = mask.shape
    cell = np.empty((nrow, nchan), dtype=np.int64)
    lib().pfbo_uvcell_index(i64(nrow), i64(nchan), ptr(uvw), ptr(freq), ptr(mask), i64(nx), i64(ny), f64(cell_x),
                            f64(cell_y), f64(usign), f64(vsign), ptr(cell))
    return cell


def compute_counts(uvw, freq, mask, wgt, nx, ny, cell_x, cell_y, dtype=np.float64, ngrid=1, usign=1.0, vsign=-1.0):
    """``_compute_counts`` (weighting.py:81-140); ngrid only changes summation order there."""
    wgt = np.ascontiguousarray(wgt, dtype=np.float64)
    ncorr, nrow, nchan = wgt.shape
    cell = uvcell_index(uvw, freq, mask, nx, ny, cell_x, cell_y, usign, vsign)
    counts = np.zeros((ncorr, nx, ny), dtype=np.float64)
    lib().pfbo_counts_accumulate(i64(ncorr), i64(nrow), i64(nchan), ptr(cell), ptr(wgt), i64(nx * ny), ptr(counts))
    return counts.astype(dtype, copy=False)


def counts_to_weights(counts, uvw, freq, weight, mask, nx, ny, cell_x, cell_y, robust, usign=1.0, vsign=-1.0):
    """``counts_to_weights`` (weighting.py:143-208). Mutates ``counts`` and ``weight`` in place like the reference."""
    if not counts.any():
        return weight
    assert counts.dtype == np.float64 and weight.dtype == np.float64
    assert counts.flags.c_contiguous and weight.flags.c_contiguous
    ncorr, nrow, nchan = weight.shape
    if robust > -2:
        numsqrt = 5 * 10 ** (-robust)
        num = np.zeros(ncorr)
        den = np.zeros(ncorr)
        lib().pfbo_briggs_sums(i64(ncorr), i64(nx * ny), ptr(counts), ptr(num), ptr(den))
        ssq = numsqrt * numsqrt * den / num
        counts *= ssq[:, None, None]
        counts += 1
    cell = uvcell_index(uvw, freq, mask, nx, ny, cell_x, cell_y, usign, vsign)
    lib().pfbo_counts_divide(i64(ncorr), i64(nrow), i64(nchan), ptr(cell), ptr(counts), i64(nx * ny), ptr(weight))
    return weight


def filter_extreme_counts(counts, level=10.0):
    """weighting.py:212-226 restated."""
    if not level:
        return counts
    pos = counts > 0
    if not pos.any():
        return counts
    lowval = np.median(counts[pos]) / level
    counts[pos] = np.maximum(counts[pos], lowval)
    return counts


def box_sum_counts(counts, npix_super):
    """weighting.py:229-254 restated (explicit zero-padded window sums instead of uniform_filter * size^2)."""
    if npix_super is None or npix_super <= 0:
        return counts
    s = int(npix_super)
    ncorr, nx, ny = counts.shape
    pad = np.zeros((ncorr, nx + 2 * s, ny + 2 * s), dtype=counts.dtype)
    pad[:, s:s + nx, s:s + ny] = counts
    out = np.zeros_like(counts)
    for dx in range(2 * s + 1):
        for dy in range(2 * s + 1):
            out += pad[:, dx:dx + nx, dy:dy + ny]
    return out
