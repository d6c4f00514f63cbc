"""Wavelet dictionary ``Psi`` on the GPU.

Mirrors /root/reference/src/pfb_imaging/operators/psi.py: ``Psi`` (psi.py:551-607, coefficient cubes
``(nband, nbasis, nymax, nxmax)``), ``PsiNocopyt`` (psi.py:610-665, ``(nband, nbasis, nxmax, nymax)``) and
``IdentityPsi`` (psi.py:714-734).  ``dot(x, alphao)`` is image -> coefficients, ``hdot(alpha, xo)`` is
coefficients -> image; both fill the output in place.  Bases are ``"self"`` and ``"db1".."db8"``; the
transform is the zero-padding multi-level 2-D DWT of wavelets/wavelets.py:216-343 in the packed layout of
psi.py:23-142 (``pfbhip_psi_*``, csrc/psi.hip).  ``nthreads`` is accepted and ignored.
"""

import ctypes as ct

import numpy as np

from .. import _lib
from .._lib import as_c, check, cint, i64, lib, ptr


def _basis_code(name):
    if name == "self":
        return 0
    if isinstance(name, str) and name.startswith("db") and name[2:].isdigit() and 1 <= int(name[2:]) <= 8:
        return int(name[2:])
    raise ValueError(f"unsupported basis {name!r}: only 'self' and 'db1'..'db8'")


class PsiBand:
    """One band's dictionary handle (pfbhip_psi)."""

    def __init__(self, nx, ny, bases, nlevel):
        _lib.require_gpu()
        self.nx, self.ny, self.nbasis, self.nlevel = int(nx), int(ny), len(bases), int(nlevel)
        codes = np.array([_basis_code(b) for b in bases], dtype=np.int32)
        self._h = ct.c_void_p()
        check(lib().pfbhip_psi_create(i64(self.nx), i64(self.ny), ct.c_int32(self.nbasis), ptr(codes), ct.c_int32(self.nlevel),
                                      ct.byref(self._h)))
        a, b = ct.c_int64(), ct.c_int64()
        check(lib().pfbhip_psi_shape(self._h, ct.byref(a), ct.byref(b)))
        self.nxmax, self.nymax = int(a.value), int(b.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().pfbhip_psi_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, transposed):
        return (self.nbasis, self.nymax, self.nxmax) if transposed else (self.nbasis, self.nxmax, self.nymax)

    def dot(self, x, alphao, transposed=False):
        x = as_c(x, np.float64)
        if x.shape != (self.nx, self.ny):
            raise ValueError(f"x shape {x.shape} != {(self.nx, self.ny)}")
        if alphao.shape != self._shape(transposed):
            raise ValueError(f"alpha shape {alphao.shape} != {self._shape(transposed)}")
        target = alphao if (alphao.flags.c_contiguous and alphao.dtype == np.float64) else np.empty(alphao.shape)
        check(lib().pfbhip_psi_dot(self._h, ptr(x), ptr(target), cint(int(transposed))))
        if target is not alphao:
            alphao[...] = target
        return alphao

    def hdot(self, alpha, xo, transposed=False):
        alpha = as_c(alpha, np.float64)
        if alpha.shape != self._shape(transposed):
            raise ValueError(f"alpha shape {alpha.shape} != {self._shape(transposed)}")
        if xo.shape != (self.nx, self.ny):
            raise ValueError(f"x shape {xo.shape} != {(self.nx, self.ny)}")
        target = xo if (xo.flags.c_contiguous and xo.dtype == np.float64) else np.empty(xo.shape)
        check(lib().pfbhip_psi_hdot(self._h, ptr(alpha), ptr(target), cint(int(transposed))))
        if target is not xo:
            xo[...] = target
        return xo


class _PsiCube:
    _transposed = False

    def __init__(self, nband, nx, ny, bases, nlevel, nthreads=1):
        self.nband, self.nx, self.ny = int(nband), int(nx), int(ny)
        self.bases, self.nlevel, self.nthreads = tuple(bases), int(nlevel), nthreads
        self.nbasis = len(self.bases)
        # the bands share one geometry, hence one handle (the reference builds one jitclass per band only
        # because each owns scratch buffers used concurrently by its thread pool)
        self._band = PsiBand(nx, ny, self.bases, nlevel)
        self.nxmax, self.nymax = self._band.nxmax, self._band.nymax

    def dot(self, x, alphao):
        """image to coeffs (psi.py:576-591 / 632-646)"""
        for b in range(self.nband):
            self._band.dot(x[b], alphao[b], self._transposed)

    def hdot(self, alpha, xo):
        """coeffs to image (psi.py:593-607 / 648-665)"""
        for b in range(self.nband):
            self._band.hdot(alpha[b], xo[b], self._transposed)


class PsiNocopyt(_PsiCube):
    """Coefficient cubes (nband, nbasis, nxmax, nymax) (psi.py:610-665)."""


class Psi(_PsiCube):
    """Coefficient cubes (nband, nbasis, nymax, nxmax): the transposed layout of psi.py:551-607."""

    _transposed = True


class PsiNocopytRay:
    """Facade over a ``BandWorkerPool`` (psi.py:670-711): one band per GPU process instead of one Ray actor
    per band; cubes are (nband, nbasis, nxmax, nymax)."""

    def __init__(self, nband, nx, ny, bases, nlevel, nthreads=1, workers=None):
        from .band_worker import BandWorkerPool

        self.nband, self.nx, self.ny = nband, nx, ny
        self.nbasis, self.nthreads = len(bases), nthreads
        if workers is None:
            workers = BandWorkerPool(nband, nthreads)
        elif workers.nband != nband:
            raise ValueError(f"workers pool has {workers.nband} bands, expected {nband}")
        self._pool = workers
        self.nxmax, self.nymax = self._pool.init_psi(nx, ny, bases, nlevel)

    def dot(self, x, alphao):
        self._pool.psi_dot(x, alphao)

    def hdot(self, alpha, xo):
        self._pool.psi_hdot(alpha, xo)


class IdentityPsi:
    """psi.py:714-734 (pure bookkeeping, no arithmetic)."""

    def __init__(self, nband, nx, ny):
        self.nband, self.nx, self.ny = nband, nx, ny
        self.nbasis = 1
        self.nymax, self.nxmax = nx, ny

    def dot(self, x, alphao):
        alphao[:, 0, :, :] = x

    def hdot(self, alpha, xo):
        xo[...] = alpha[:, 0, :, :]
