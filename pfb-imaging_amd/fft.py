"""``ducc0.fft`` replacements on the GPU: ``r2c``, ``c2r``, ``good_size``.

Only the call forms the reference uses are supported (2-D transforms over the last two axes,
/root/reference/src/pfb_imaging/operators/psf.py:20-32, operators/fft.py:10,39,
operators/gridder.py:659,912): ``r2c(a, axes, forward=True, inorm=0)`` and
``c2r(a, axes, forward=False, lastsize, inorm=2)``.  Other forms raise NotImplementedError
rather than silently computing on the CPU.
"""

import numpy as np

from . import _lib
from ._lib import as_c, check, i64, lib, ptr


def good_size(n, real=False):
    """Smallest 2-3-5-7-11-smooth (``real=False``) / 2-3-5-smooth (``real=True``) integer >= n
    (/root/reference/src/pfb_imaging/utils/misc.py:921-951)."""
    return int(lib().pfbhip_good_size(int(n), int(bool(real))))


def _last_two(a, axes):
    nd = a.ndim
    axes = tuple(ax % nd for ax in axes)
    if nd < 2 or axes != (nd - 2, nd - 1):
        raise NotImplementedError(f"only transforms over the last two axes are supported (axes={axes}, ndim={nd})")


def r2c(a, axes=(-2, -1), forward=True, inorm=0, out=None, nthreads=1, centred=False):
    """``centred=True`` (not a ducc0 keyword; used by ``operators.fft``): the transform of ``ifftshift(a)`` over both axes for
    even lengths, the shift applied to the spectrum on the device."""
    if not forward or inorm != 0:
        raise NotImplementedError("r2c supports forward=True, inorm=0 (the reference's only use)")
    _lib.require_gpu()
    a = as_c(a, np.float64)
    _last_two(a, axes)
    n0, n1 = a.shape[-2:]
    nbatch = int(np.prod(a.shape[:-2], dtype=np.int64))
    res = np.empty(a.shape[:-2] + (n0, n1 // 2 + 1), dtype=np.complex128)
    fn = lib().pfbhip_r2c_2d_centred if centred else lib().pfbhip_r2c_2d
    check(fn(ptr(a), i64(nbatch), i64(n0), i64(n1), ptr(res)))
    if out is not None:
        out[...] = res
        return out
    return res


def c2r(a, axes=(-2, -1), forward=False, lastsize=None, inorm=2, out=None, nthreads=1,
        allow_overwriting_input=False):
    if forward or inorm != 2:
        raise NotImplementedError("c2r supports forward=False, inorm=2 (the reference's only use)")
    _lib.require_gpu()
    a = as_c(a, np.complex128)
    _last_two(a, axes)
    n0, nh = a.shape[-2:]
    if lastsize is None:
        lastsize = 2 * (nh - 1)
    if lastsize // 2 + 1 != nh:
        raise ValueError(f"lastsize={lastsize} is inconsistent with a half-complex axis of {nh}")
    nbatch = int(np.prod(a.shape[:-2], dtype=np.int64))
    res = np.empty(a.shape[:-2] + (n0, int(lastsize)), dtype=np.float64)
    check(lib().pfbhip_c2r_2d(ptr(a), i64(nbatch), i64(n0), i64(lastsize), ptr(res)))
    if out is not None:
        out[...] = res
        return out
    return res
