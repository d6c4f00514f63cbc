"""Preconditioners of ``pfb_imaging.operators.precond`` on the GPU.

/root/reference/src/pfb_imaging/operators/precond.py:12-154 holds a second ``HessPSF``: the cube form of the PSF-approximate
Hessian with a REQUIRED beam, ``eta`` per band or per pixel, and an ``idot`` that is a plain CG from ``x0`` (zeros by default,
``minit=2``) -- no direct-estimate start and no ``mode`` switch, which are what ``operators.hessian.HessPSF`` (hessian.py:251-436) adds.
Same device plan underneath (``psfconv.PsfConv``: PSFs and beams resident, five device passes per band, on-device CG).
"""

import numpy as np

from .. import _lib
from ..psfconv import PsfConv


class HessPSF(object):
    """``dot(x) = beam * (abspsf (*) (beam * x)) + eta * x`` per band; ``hdot`` is ``dot``; ``idot`` solves with CG
    (precond.py:12-154).  ``dot`` returns the internal ``self.xout`` (aliasing is part of the reference contract), ``idot`` a copy."""

    def __init__(self, nx, ny, abspsf, beam=None, eta=1.0, nthreads=1, cgtol=1e-3, cgmaxit=300, cgverbose=2, cgrf=25,
                 taper_width=32, min_beam=5e-3, memory_greedy=True):
        if not memory_greedy:
            raise NotImplementedError("Non-memory-greedy mode is not implemented yet")
        self.nx = nx
        self.ny = ny
        self.abspsf = abspsf
        self.nband, self.nx_psf, self.nyo2 = abspsf.shape
        if beam is None:
            raise ValueError("Beam is required for HessPSF preconditioner")
        assert self.nband == beam.shape[0]
        assert self.nx == beam.shape[1]
        assert self.ny == beam.shape[2]
        self.ny_psf = 2 * (self.nyo2 - 1)
        self.nx_pad = self.nx_psf - self.nx
        self.ny_pad = self.ny_psf - self.ny
        self.nthreads = nthreads
        # eta: one value, one per band, or one per pixel (precond.py:57-66)
        self._eta_pix = None
        if isinstance(eta, float):
            self.eta = np.tile(eta, self.nband)[:, None, None]
        elif isinstance(eta, np.ndarray):
            if eta.size == self.nband:
                self.eta = eta[:, None, None]
            else:
                assert eta.shape == (self.nband, self.nx, self.ny)
                self.eta = eta
                self._eta_pix = np.ascontiguousarray(eta, dtype=np.float64)
        else:
            raise ValueError("Unsupported type for eta")
        self._plan = PsfConv(nx, ny, self.nx_psf, self.ny_psf)
        for b in range(self.nband):
            self._plan.set_psfhat(b, abspsf[b])
        self.set_beam(beam)
        self.xout = _lib.result_empty((self.nband, self.nx, self.ny), np.float64)
        self.cgtol = cgtol
        self.cgmaxit = cgmaxit
        self.cgverbose = cgverbose
        self.cgrf = cgrf
        self.memory_greedy = memory_greedy

    def set_beam(self, beam):
        assert beam.shape == (self.nband, self.nx, self.ny)
        self.beam = beam
        for b in range(self.nband):
            self._plan.set_beam(b, beam[b])

    def _cube(self, x, what):
        if len(x.shape) == 3:
            xtmp = x
        elif len(x.shape) == 2:
            xtmp = x[None, :, :]
        else:
            raise ValueError(f"Unsupported number of {what}")
        nband, nx, ny = xtmp.shape
        assert nband == self.nband
        assert nx == self.nx
        assert ny == self.ny
        return xtmp

    def _eta_band(self, b):
        return 0.0 if self._eta_pix is not None else float(self.eta[b, 0, 0])

    def dot(self, x):
        xtmp = self._cube(x, "input dimensions")
        for b in range(self.nband):
            self._plan.apply(xtmp[b], b, beam_slot=b, eta=self._eta_band(b), out=self.xout[b])
        if self._eta_pix is not None:  # per-pixel Tikhonov term: one host pass (the device call folds in a scalar only)
            self.xout += xtmp * self._eta_pix
        return self.xout

    def hdot(self, x):
        return self.dot(x)

    def idot(self, x, x0=None):
        xtmp = self._cube(x, "dimensions")
        if x0 is None:
            x0 = np.zeros_like(xtmp)
        if self._eta_pix is not None:
            # (a per-pixel eta is not a parameter of the device CG: the reference's loop over this operator's dot)
            from ..opt import pcg_numba

            self.xout[...] = pcg_numba(self.dot, xtmp, x0=np.array(x0, dtype=np.float64), tol=self.cgtol, maxit=self.cgmaxit, minit=2,
                                       verbosity=0, report_freq=self.cgrf, backtrack=False, return_resid=False)
            return self.xout.copy()
        # The operator is block diagonal over the bands: the reference's single CG over the cube and one CG per band converge to
        # the same solution (the stopping test is per band here: never later than the cube's).
        for b in range(self.nband):
            self.xout[b] = self._plan.cg(xtmp[b], [b], [b], scale=1.0, eta=self._eta_band(b), x0=x0[b], tol=self.cgtol,
                                         maxit=self.cgmaxit, minit=2)
        return self.xout.copy()
