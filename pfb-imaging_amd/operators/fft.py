"""``pfb_imaging.operators.fft`` on the GPU: the PSF -> PSFHAT transforms.

``fft2d`` / ``fft_cube`` are ``r2c(ifftshift(x))`` over the last two axes, unnormalised
(/root/reference/src/pfb_imaging/operators/fft.py:9-61; used for PSFHAT at operators/gridder.py:659,912).  The reference
wraps the transform in a dask ``blockwise`` graph; here the arrays are computed eagerly on the device and returned as
numpy arrays of shape ``(..., nx, ny // 2 + 1)`` (call ``.compute()``-free).  ``nthreads`` is accepted and ignored.
"""

import numpy as np

from .. import fft as _fft


def _r2c_centred(x, axes):
    """``r2c(ifftshift(x))``: for even lengths (every PSF size the reference produces, utils/misc.py:917-951) the shift is a
    checkerboard sign on the spectrum, applied on the device; odd lengths shift on the host."""
    x = np.asarray(x)
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    if x.shape[-2] % 2 == 0 and x.shape[-1] % 2 == 0:
        return _fft.r2c(x, axes=axes, forward=True, inorm=0, centred=True)
    return _fft.r2c(np.fft.ifftshift(x, axes=(-2, -1)), axes=axes, forward=True, inorm=0)


def fft2d(x, nthreads=1):
    """(nx, ny) real -> (nx, ny // 2 + 1) complex: the half-complex spectrum of the image with its centre pixel moved to
    the origin."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError(f"fft2d expects a 2-D image, got shape {x.shape}")
    out = _r2c_centred(x, (0, 1))
    return out.astype(np.complex64) if x.dtype == np.float32 else out


def fft_cube(x, nthreads=1):
    """(nband, nx, ny) real -> (nband, nx, ny // 2 + 1) complex, band by band in one batched device transform."""
    x = np.asarray(x)
    if x.ndim != 3:
        raise ValueError(f"fft_cube expects a (nband, nx, ny) cube, got shape {x.shape}")
    out = _r2c_centred(x, (1, 2))
    return out.astype(np.complex64) if x.dtype == np.float32 else out
