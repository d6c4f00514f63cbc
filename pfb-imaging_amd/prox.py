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


def prox_21m(v, sigma, weight=1.0, axis=0):
    """prox_{sigma ||.||_21}(v) with the band axis first (prox_21m.py:5-26); returns a new array."""
    _lib.require_gpu()
    if axis != 0:
        raise NotImplementedError("prox_21m: only axis=0 (band axis first) is implemented on the GPU")
    v = as_c(v, np.float64)
    nband = v.shape[0]
    n = v.size // max(nband, 1)
    w = as_c(np.broadcast_to(weight, v.shape[1:]), np.float64)
    out = np.empty_like(v)
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
