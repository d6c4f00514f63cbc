"""``ducc0.fft`` replacements on the GPU: ``r2c``, ``c2r``, ``good_size``.

Two-axis transforms in every form ducc0 offers for them (any pair of axes, both exponent signs, ``inorm`` 0 / 1 / 2); the
reference's own calls are ``r2c(a, axes=last two, forward=True, inorm=0)`` and ``c2r(a, axes=last two, forward=False,
lastsize, inorm=2)`` (/root/reference/src/pfb_imaging/operators/psf.py:20-32, operators/fft.py:10,39,
operators/gridder.py:659,912), which run without any host pass.  One-axis and n > 2-axis transforms raise
NotImplementedError rather than silently computing on the CPU (the reference has none on this path).
"""

import numpy as np

from . import _lib
from ._lib import as_c, check, i64, lib, ptr


def good_size(n, real=False):
    """Smallest 2-3-5-7-11-smooth (``real=False``) / 2-3-5-smooth (``real=True``) integer >= n
    (/root/reference/src/pfb_imaging/utils/misc.py:921-951)."""
    return int(lib().pfbhip_good_size(int(n), int(bool(real))))


def _two_axes(a, axes):
    """The two transform axes of ``a`` as non-negative indices (ducc0 transforms over ``axes`` in the order given; the LAST one
    is the half-complex axis)."""
    nd = a.ndim
    try:
        axes = tuple(int(ax) % nd for ax in axes)
    except TypeError:
        axes = (int(axes) % nd,)
    if len(axes) != 2 or axes[0] == axes[1]:
        raise NotImplementedError(f"only two-axis transforms are supported on the GPU (axes={axes}); the reference has no other use")
    return axes


def _norm(inorm, n):
    if inorm == 0:
        return 1.0
    if inorm == 1:
        return 1.0 / np.sqrt(float(n))
    if inorm == 2:
        return 1.0 / float(n)
    raise ValueError(f"inorm must be 0, 1 or 2, got {inorm}")


def r2c(a, axes=(-2, -1), forward=True, inorm=0, out=None, nthreads=1, centred=False):
    """``ducc0.fft.r2c`` over two axes (the last of ``axes`` becomes the half-complex one): ``forward`` selects the sign of the
    exponent, ``inorm`` 0 / 1 / 2 the scaling 1, 1/sqrt(N), 1/N.  The reference only calls it as ``r2c(a, axes=last two,
    forward=True, inorm=0)`` (operators/psf.py:20-32, operators/fft.py:10,39); other axes go through a host-side axis move,
    the backward sign through a conjugation of the (real-input) result.  ``centred=True`` (not a ducc0 keyword; used by
    ``operators.fft``): the transform of ``ifftshift(a)`` over both axes for even lengths, the shift applied to the spectrum
    on the device."""
    _lib.require_gpu()
    a = np.asarray(a)
    ax = _two_axes(a, axes)
    last_two = ax == (a.ndim - 2, a.ndim - 1)
    src = as_c(a if last_two else np.moveaxis(a, ax, (-2, -1)), np.float64)
    n0, n1 = src.shape[-2:]
    nbatch = int(np.prod(src.shape[:-2], dtype=np.int64))
    res = _lib.result_empty(src.shape[:-2] + (n0, n1 // 2 + 1), np.complex128)  # (page-locked: the download runs at the PCIe rate)
    fn = lib().pfbhip_r2c_2d_centred if centred else lib().pfbhip_r2c_2d
    check(fn(ptr(src), i64(nbatch), i64(n0), i64(n1), ptr(res)))
    if not forward:
        np.conjugate(res, out=res)
    s = _norm(inorm, n0 * n1)
    if s != 1.0:
        res *= s
    if not last_two:
        res = np.moveaxis(res, (-2, -1), ax)
    if out is not None:
        out[...] = res
        return out
    return res


def c2r(a, axes=(-2, -1), forward=False, lastsize=None, inorm=2, out=None, nthreads=1,
        allow_overwriting_input=False):
    """``ducc0.fft.c2r`` over two axes (the last of ``axes`` is the half-complex one, ``lastsize`` its real length).  The
    reference's only form is ``c2r(a, axes=last two, forward=False, lastsize=n, inorm=2)`` (operators/psf.py:28-32); the other
    sign conjugates the input, other axes move on the host, ``inorm`` 0 / 1 / 2 scale by 1, 1/sqrt(N), 1/N."""
    _lib.require_gpu()
    a = np.asarray(a)
    ax = _two_axes(a, axes)
    last_two = ax == (a.ndim - 2, a.ndim - 1)
    src = a if last_two else np.moveaxis(a, ax, (-2, -1))
    if forward:
        src = np.conjugate(src)
    src = as_c(src, np.complex128)
    n0, nh = src.shape[-2:]
    if lastsize is None:
        lastsize = 2 * (nh - 1)
    if lastsize // 2 + 1 != nh:
        raise ValueError(f"lastsize={lastsize} is inconsistent with a half-complex axis of {nh}")
    nbatch = int(np.prod(src.shape[:-2], dtype=np.int64))
    res = _lib.result_empty(src.shape[:-2] + (n0, int(lastsize)), np.float64)
    check(lib().pfbhip_c2r_2d(ptr(src), i64(nbatch), i64(n0), i64(lastsize), ptr(res)))   # (scaled by 1 / N on the device)
    s = _norm(inorm, n0 * int(lastsize)) * float(n0 * int(lastsize))
    if abs(s - 1.0) > 1e-15:
        res *= s
    if not last_two:
        res = np.moveaxis(res, (-2, -1), ax)
    if out is not None:
        out[...] = res
        return out
    return res
