"""l21 dual update, prox_21m and positivity on the GPU.

Mirrors /root/reference/src/pfb_imaging/prox/prox_21m.py (``prox_21m`` :5-26, ``dual_update_numba`` :73-102,
``dual_update_numba_fast`` :105-135) and prox/positivity.py:12-43.  Arrays are ``(nband, nbasis, n1, n2)``
coefficient cubes / ``(nband, nx, ny)`` image cubes on the host; ``v`` / ``x`` are updated in place.
"""

import numpy as np

from . import _lib
from ._lib import as_c, check, cint, f64, i64, lib, ptr


def _inplace(arr):
    if not (isinstance(arr, np.ndarray) and arr.flags.c_contiguous and arr.dtype == np.float64 and arr.flags.writeable):
        raise ValueError("in-place operand must be a writable C-contiguous float64 array")
    return arr


def dual_update_numba_fast(vp, v, lam, sigma=1.0, weight=None):
    """v <- vtilde * min(1, lam*w/|sum_band vtilde|), vtilde = vp + sigma*v (prox_21m.py:105-135)."""
    _lib.require_gpu()
    v = _inplace(v)
    nband = v.shape[0]
    n = v.size // max(nband, 1)
    vp = as_c(vp, np.float64)
    if vp.shape != v.shape:
        raise ValueError(f"vp shape {vp.shape} != v shape {v.shape}")
    w = np.ones(v.shape[1:]) if weight is None else as_c(np.broadcast_to(weight, v.shape[1:]), np.float64)
    check(lib().pfbhip_dual_update(ptr(vp), ptr(v), i64(nband), i64(n), f64(lam), f64(sigma), ptr(w)))


# the numerically fragile original (prox_21m.py:73-102) computes the same update
dual_update_numba = dual_update_numba_fast


def _vtilde_sum_gpu(vp_loc, v_loc, sigma):
    """GPU phase 1 on this rank's bands: v_loc <- vp_loc + sigma v_loc (in place), returns the local band sum."""
    from ._lib import DeviceArray

    nb = v_loc.shape[0]
    n = v_loc.size // max(nb, 1)
    dvp, dv = DeviceArray.from_host(vp_loc), DeviceArray.from_host(v_loc)
    ds = DeviceArray((n,), np.float64)
    check(lib().pfbhip_l21_vtilde_sum_dev(dvp.ptr, dv.ptr, i64(nb), i64(n), f64(sigma), ds.ptr))
    dv.download(v_loc)
    s = ds.download()
    for d in (dvp, dv, ds):
        d.free()
    return s.reshape(v_loc.shape[1:])


def _scale_gpu(v_loc, lam, weight, total):
    """GPU phase 2 on this rank's bands, given the band sum over ALL bands."""
    from ._lib import DeviceArray

    nb = v_loc.shape[0]
    n = v_loc.size // max(nb, 1)
    dv, dw, ds = DeviceArray.from_host(v_loc), DeviceArray.from_host(weight), DeviceArray.from_host(total)
    check(lib().pfbhip_l21_scale_dev(dv.ptr, i64(nb), i64(n), f64(lam), dw.ptr, ds.ptr))
    dv.download(v_loc)
    for d in (dv, dw, ds):
        d.free()


def dual_update_bands(vp, v, lam, sigma=1.0, weight=None, comm=None, bands=None, phases=None):
    """Band-sharded ``dual_update_numba_fast`` over full cubes ``(nband, nbasis, n1, n2)``, in place on ``v``.

    Every rank holds the full cubes (as the reference's driver does); rank r computes the bands in ``bands``.
    The band sum of vtilde -- the one quantity that couples the bands -- is completed with ONE all-reduce
    (`allreduce(sum)` of the band-sum inside the l21 dual update each PD iteration); a second
    all-reduce re-assembles the updated cube (each band is produced by exactly one rank).
    ``phases`` lets the CPU tests substitute the two device phases."""
    if comm is None or comm.world_size == 1:
        if phases is None:
            dual_update_numba_fast(vp, v, lam, sigma, weight)
            return
    v = _inplace(v)
    vtilde_sum, scale = phases if phases is not None else (_vtilde_sum_gpu, _scale_gpu)
    if phases is None:
        _lib.require_gpu()
    idx = list(range(v.shape[0])) if bands is None else list(bands)
    w = np.ones(v.shape[1:]) if weight is None else as_c(np.broadcast_to(weight, v.shape[1:]), np.float64)
    v_loc = np.ascontiguousarray(v[idx])
    vp_loc = np.ascontiguousarray(np.asarray(vp, dtype=np.float64)[idx])
    local = vtilde_sum(vp_loc, v_loc, sigma) if idx else np.zeros(v.shape[1:])
    total = local if comm is None else comm.allreduce_sum(local).reshape(local.shape)
    if idx:
        scale(v_loc, lam, w, np.ascontiguousarray(total))
    out = np.zeros_like(v)
    out[idx] = v_loc
    v[...] = out if comm is None else comm.allreduce_sum(out).reshape(v.shape)


def prox_21m(v, sigma, weight=1.0, axis=0):
    """prox_{sigma ||.||_21}(v) with the band axis first (prox_21m.py:5-26); returns a new array."""
    _lib.require_gpu()
    if axis != 0:
        raise NotImplementedError("prox_21m: only axis=0 (band axis first) is implemented on the GPU")
    v = as_c(v, np.float64)
    nband = v.shape[0]
    n = v.size // max(nband, 1)
    w = as_c(np.broadcast_to(weight, v.shape[1:]), np.float64)
    out = _lib.result_empty(v.shape, np.float64)
    check(lib().pfbhip_prox_21m(ptr(v), i64(nband), i64(n), f64(sigma), ptr(w), ptr(out)))
    return out


def prox_21m_numba(v, result, lam, sigma=1.0, weight=None):
    """result <- prox_{(lam/sigma) ||.||_21}(v / sigma) (prox_21m.py:29-58)."""
    w = np.ones(v.shape[1:]) if weight is None else weight
    result[...] = prox_21m(np.asarray(v) / sigma, lam / sigma, weight=w)


def positivity(x):
    """Clamp negative values to zero, in place (positivity.py:12-19)."""
    _lib.require_gpu()
    x = _inplace(x)
    check(lib().pfbhip_positivity(ptr(x), i64(1), i64(x.size), cint(1)))


def positivity_band(x):
    """Zero a pixel in all bands where any band is non-positive, in place (positivity.py:22-33)."""
    _lib.require_gpu()
    x = _inplace(x)
    nband = x.shape[0]
    check(lib().pfbhip_positivity(ptr(x), i64(nband), i64(x.size // max(nband, 1)), cint(2)))


def positivity_prox(mode):
    """positivity.py:36-43"""
    if mode == 0:
        return None
    if mode == 1:
        return positivity
    if mode == 2:
        return positivity_band
    raise ValueError(f"Unknown positivity mode {mode}")
