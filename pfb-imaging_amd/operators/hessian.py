"""Hessian family of ``pfb_imaging.operators.hessian`` on the GPU.

Names, arguments and return conventions follow
/root/reference/src/pfb_imaging/operators/hessian.py:
    hessian_slice      :15-100   exact ``beam R^H W R (beam x) / wsum + eta x`` (degrid + grid)
    hessian_psf_slice  :103-143  PSF-convolution approximation, one band
    hessian_psf_cube   :146-175
    hess_direct(_slice):178-248  tapered division / multiplication by (abspsf + eta)
    HessPSF            :251-436  per-band PSF Hessian with ``dot`` / ``hdot`` / ``idot``
    HessianTree        :439-522  sum-over-partitions PSF Hessian, per correlation
    HessTreeRay        :525-615  cube-level facade over a band-worker pool
Scratch arguments (``xpad``, ``xhat``) are accepted for signature compatibility; transform
scratch lives on the device.
"""

import numpy as np

from .. import _lib
from ..psfconv import PsfConv, cached_plan, cached_psf_slot
from ..wgridder import _get_gridder


def taperf(shape, taper_width):
    """Separable edge taper of ``hess_direct`` (the reference's ``utils.misc.taperf``, misc.py:968-975): along each axis
    the weight is 1 in the interior and rolls off over ``taper_width`` pixels at both ends as a raised cosine
    ``(1 + cos(theta)) / 2`` -- theta runs over [1.1 pi, 2 pi] on the leading edge (so the first pixel is ~0.024, not 0) and
    over [0, 0.9 pi] on the trailing edge; the image taper is the outer product of the axis tapers."""
    def axis_taper(n):
        w = np.ones(n)
        lead = np.linspace(1.1 * np.pi, 2.0 * np.pi, taper_width)
        trail = np.linspace(0.0, 0.9 * np.pi, taper_width)
        w[:taper_width] = (1.0 + np.cos(lead)) / 2.0
        w[n - taper_width:] = (1.0 + np.cos(trail)) / 2.0
        return w

    wx, wy = (axis_taper(int(n)) for n in shape)
    return wx[:, None] * wy[None, :]


def hessian_slice(x, xout=None, uvw=None, weight=None, vis_mask=None, freq=None, beam=None, cell=None, x0=0.0, y0=0.0,
                  flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, epsilon=1e-7, double_accum=True,
                  nthreads=1, eta=None, wsum=None):
    """Exact Hessian of one image slice, fused on the device: the model visibilities of
    ``dirty2vis`` feed ``vis2dirty`` without leaving HBM (reference: hessian.py:15-100).

    ``x0, y0, flip_*`` must follow ``wgridder_conventions``.  Returns zeros without touching the
    GPU when ``x`` is identically zero (hessian.py:47-48).  ``xout`` is filled in place if given.
    """
    if not _lib.any_nonzero(x):
        return np.zeros_like(x)
    nx, ny = x.shape
    # hessian_slice calls ducc0 with its default oversampling bounds (no sigma_min/sigma_max)
    g, cached = _get_gridder(uvw, freq, vis_mask, npix_x=nx, npix_y=ny, pixsize_x=float(cell), pixsize_y=float(cell),
                             center_x=float(x0), center_y=float(y0), epsilon=float(epsilon), flip_u=bool(flip_u),
                             flip_v=bool(flip_v), flip_w=bool(flip_w), do_wgridding=bool(do_wgridding),
                             divide_by_n=False, sigma_min=1.1, sigma_max=2.6)
    try:
        from ..wgridder import _fingerprint

        token = None if weight is None else _fingerprint(np.asarray(weight))
        if getattr(g, "_hess_weight_token", "unset") != token:
            g.set_weights(weight)
            g._hess_weight_token = token
        # wsum / eta are folded into the device call (the last fused kernel writes beam * acc * corr / wsum + eta * x): the
        # reference's host passes `convim /= wsum`, `convim += eta * x` and the copy into xout each cost as much as the
        # whole device apply at 8192^2.  None keeps the reference's meaning: no normalisation / no Tikhonov term.
        # (a float32 x takes the single-precision boundary: float32 across PCIe, sums in double, float32 result)
        direct = xout is not None and xout.dtype == x.dtype and x.dtype in (np.float32, np.float64) and xout.flags.c_contiguous \
            and xout.shape == x.shape and not np.shares_memory(xout, x)
        # The device call folds in a POSITIVE wsum only (it reads wsum <= 0 as "no normalisation").  The reference divides by
        # whatever it is given (`if wsum is not None: convim /= wsum`, hessian.py:91-92): wsum = 0 there yields inf / nan, a
        # negative one flips the sign -- those cases keep the reference's host arithmetic.
        odd_wsum = wsum is not None and not (float(wsum) > 0.0)
        if odd_wsum:
            convim = g.hessian(x, beam=beam, eta=0.0, wsum=0.0, out=xout if direct else None)
            with np.errstate(divide="ignore", invalid="ignore"):
                convim /= float(wsum)
            if eta is not None:
                convim += x * float(eta)
        else:
            convim = g.hessian(x, beam=beam, eta=0.0 if eta is None else float(eta), wsum=0.0 if wsum is None else float(wsum),
                               out=xout if direct else None)
    finally:
        if not cached:
            g.close()
    if xout is not None:
        if not direct:
            xout[...] = convim
        return xout
    return convim


def hessian_psf_slice(x, xpad=None, xhat=None, xout=None, abspsf=None, beam=None, lastsize=None, nthreads=1, eta=None):
    """Tikhonov-regularised PSF-approximate Hessian (hessian.py:103-143)."""
    nx, ny = x.shape
    plan = cached_plan(nx, ny, abspsf.shape[0], lastsize)
    slot = cached_psf_slot(plan, abspsf)
    bslot = -1
    if beam is not None:
        bslot = 0
        plan.set_beam(0, beam)
    if xout is None:
        xout = _lib.result_empty(x.shape, np.float64)
    plan.apply(x, slot, beam_slot=bslot, eta=eta if eta else 0.0, out=xout)
    return xout


def hessian_psf_cube(xpad, xhat, xout, beam, abspsf, lastsize, x, nthreads=1, eta=1, mode="forward"):
    """hessian.py:146-175 (only ``mode="forward"`` exists in the reference)."""
    if mode != "forward":
        raise NotImplementedError
    nband = x.shape[0]
    for b in range(nband):
        hessian_psf_slice(x[b], xout=xout[b], abspsf=abspsf[b], beam=None if beam is None else beam[b],
                          lastsize=lastsize, eta=eta)
    return xout


def hess_direct_slice(x, xpad=None, xhat=None, xout=None, abspsf=None, taperxy=None, lastsize=None, nthreads=1, eta=1,
                      mode="forward"):
    """hessian.py:215-248: ``taper * crop(irfft2(rfft2(pad(taper x)) (*|/) (abspsf + eta)))``.
    eta is relative to wsum (the PSF peak)."""
    nx, ny = x.shape
    plan = cached_plan(nx, ny, abspsf.shape[0], lastsize)
    slot = cached_psf_slot(plan, abspsf)
    plan.set_beam(1, taperxy)
    if xout is None:
        xout = _lib.result_empty(x.shape, np.float64)
    plan.apply(x, slot, beam_slot=1, mode=1 if mode == "forward" else 2, shift=float(eta), out=xout)
    return xout


def hess_direct(x, xpad=None, xhat=None, xout=None, abspsf=None, taperxy=None, lastsize=None, nthreads=1, eta=1,
                mode="forward"):
    """hessian.py:178-212 (band cube version of :func:`hess_direct_slice`)."""
    for b in range(x.shape[0]):
        hess_direct_slice(x[b], xout=xout[b], abspsf=abspsf[b], taperxy=taperxy, lastsize=lastsize, eta=eta, mode=mode)
    return xout


class HessPSF(object):
    """Per-band PSF-approximate Hessian / preconditioner (hessian.py:251-436).

    PSFs, beams and the taper are uploaded once; ``dot`` runs five device passes per band and
    ``idot(mode="psf")`` runs the whole CG on the device.  ``dot`` returns the internal
    ``self.xout`` buffer (aliasing is part of the reference contract); ``idot`` returns a copy.
    """

    def __init__(self, nx, ny, abspsf, beam=None, eta=1.0, nthreads=1, cgtol=1e-3, cgmaxit=300, cgverbose=2, cgrf=25,
                 taper_width=32, min_beam=5e-3):
        self.nx = nx
        self.ny = ny
        self.abspsf = abspsf
        self.nband, self.nx_psf, self.nyo2 = abspsf.shape
        self.ny_psf = 2 * (self.nyo2 - 1)
        self.nx_pad = self.nx_psf - self.nx
        self.ny_pad = self.ny_psf - self.ny
        self.nthreads = nthreads
        if isinstance(eta, float):
            self.eta = np.tile(eta, self.nband)
        else:
            self.eta = np.array(eta)
            assert self.eta.size == self.nband
        self._plan = PsfConv(nx, ny, self.nx_psf, self.ny_psf)
        for b in range(self.nband):
            self._plan.set_psfhat(b, abspsf[b])
        self._taper_slot = self.nband
        if beam is not None and not (beam == 1).all():
            self.set_beam(beam)
        else:
            self.beam = (None,) * self.nband
        self.xout = _lib.result_empty((self.nband, self.nx, self.ny), np.float64)  # (page-locked: downloads at the PCIe rate)
        self.cgtol = cgtol
        self.cgmaxit = cgmaxit
        self.cgverbose = cgverbose
        self.cgrf = cgrf
        self.taperxy = taperf((nx, ny), taper_width)
        self._plan.set_beam(self._taper_slot, self.taperxy)
        self.min_beam = min_beam

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

    def dot(self, x):
        xtmp = self._cube(x, "input dimensions")
        for b in range(self.nband):
            bslot = -1 if self.beam[b] is None else b
            self._plan.apply(xtmp[b], b, beam_slot=bslot, eta=float(self.eta[b]), out=self.xout[b])
        return self.xout

    def hdot(self, x):
        return self.dot(x)

    def _direct_beam(self, xb, b, raw=None):
        """The direct estimate with the reference's beam division (hessian.py:381-386, 395-399: ``xout /= beam**2`` where
        ``xout > 0`` and ``beam > min_beam``) done on the device; ``raw`` receives the estimate before the division."""
        shift = float(self.eta[b] * np.sqrt(self.nx * self.ny))
        bslot = -1 if self.beam[b] is None else b
        self._plan.direct(xb, b, self._taper_slot, shift, beam_slot=bslot, min_beam=float(self.min_beam), out=self.xout[b], raw=raw)
        return self.xout[b]

    def idot(self, x, mode="psf", x0=None, init_x0=True):
        xtmp = self._cube(x, "dimensions")
        if x0 is None and init_x0:
            # initialise with the direct estimate (hessian.py:369-387).  As in the reference the
            # beam division lands in self.xout after x0[b] has taken its copy.
            x0 = np.zeros_like(xtmp)
            for b in range(self.nband):
                self._direct_beam(xtmp[b], b, raw=x0[b])
        else:
            x0 = np.zeros_like(xtmp)

        if mode == "direct":
            for b in range(self.nband):
                self._direct_beam(xtmp[b], b)
        elif mode == "psf":
            for b in range(self.nband):
                bslot = -1 if self.beam[b] is None else b
                self.xout[b] = self._plan.cg(xtmp[b], [b], [bslot], scale=1.0, eta=float(self.eta[b]), x0=x0[b],
                                             tol=self.cgtol, maxit=self.cgmaxit, minit=3)
        else:
            raise ValueError(f"Unknown mode {mode}")
        return self.xout.copy()


class HessianTree(object):
    """``H x = (1/sum_p wsum_p) sum_p B_p (PSF_p (*) (B_p x)) + eta x`` per correlation
    (hessian.py:439-522).  ``partitions``: dicts with ``psfhat (corr, nx_psf, nyo2)`` (real,
    non-negative), ``beam (corr, nx, ny)``, ``wsum (corr,)``; everything is uploaded once."""

    def __init__(self, partitions, nx, ny, nx_psf, ny_psf, eta=0.0, nthreads=1, wsum=None):
        if not partitions:
            raise ValueError("HessianTree requires at least one partition")
        self.parts = partitions
        self.nx = nx
        self.ny = ny
        self.nx_psf = nx_psf
        self.ny_psf = ny_psf
        self.eta = eta
        self.nthreads = nthreads
        self.ncorr = partitions[0]["wsum"].size
        if wsum is None:
            self.wsum = np.zeros(self.ncorr)
            for p in partitions:
                self.wsum += p["wsum"]
        else:
            self.wsum = np.broadcast_to(np.asarray(wsum, dtype=float), (self.ncorr,)).copy()
        self._plan = PsfConv(nx, ny, nx_psf, ny_psf)
        for ip, p in enumerate(partitions):
            for c in range(self.ncorr):
                self._plan.set_psfhat(ip * self.ncorr + c, p["psfhat"][c])
                self._plan.set_beam(ip * self.ncorr + c, p["beam"][c])

    def _slots(self, c):
        return [ip * self.ncorr + c for ip in range(len(self.parts))]

    def dot(self, x, out=None):
        xtmp = x if x.ndim == 3 else x[None, :, :]
        ncorr, nx, ny = xtmp.shape
        assert ncorr == self.ncorr, f"expected {self.ncorr} correlations on axis 0, got {ncorr}"
        assert nx == self.nx and ny == self.ny
        if out is None:
            out = _lib.result_empty(xtmp.shape, np.float64)  # (every correlation's first partition overwrites its image)
        elif out.shape != xtmp.shape or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape (ncorr, nx, ny)")
        for c in range(self.ncorr):
            scale = 1.0 / self.wsum[c]
            for k, s in enumerate(self._slots(c)):
                self._plan.apply(xtmp[c], s, beam_slot=s, scale=scale, eta=self.eta if k == 0 else 0.0, out=out[c],
                                 accumulate=k > 0)
        return out

    def hdot(self, x):
        return self.dot(x)

    def cg(self, rhs, x0=None, tol=1e-3, maxit=150, minit=1):
        """On-device CG on correlation 0 of this band (what the band worker solves:
        /root/reference/src/pfb_imaging/operators/band_worker.py:124-140)."""
        s = self._slots(0)
        return self._plan.cg(rhs, s, s, scale=1.0 / self.wsum[0], eta=self.eta, x0=x0, tol=tol, maxit=maxit,
                             minit=minit)


class HessTreeRay:
    """Cube-level Hessian over per-band HessianTrees held in band workers (hessian.py:525-615).
    The pool is the GPU band-worker pool (one band per GPU when launched one process per GPU)."""

    def __init__(self, partitions_per_band, nx, ny, nx_psf, ny_psf, etas=0.0, nthreads=1, wsums=None, cg_tol=1e-3,
                 cg_maxit=150, cg_minit=1, cg_verbose=0, workers=None):
        from .band_worker import BandWorkerPool

        if partitions_per_band is None:
            if workers is None:
                raise ValueError("partitions_per_band=None requires a workers pool with loaded bands")
            self.nband = workers.nband
        else:
            self.nband = len(partitions_per_band)
        self.nx = nx
        self.ny = ny
        self.cg_tol = cg_tol
        self.cg_maxit = cg_maxit
        self.cg_minit = cg_minit
        self.cg_verbose = cg_verbose
        etas = np.broadcast_to(np.asarray(etas, dtype=float), (self.nband,))
        if wsums is None:
            wsums = [None] * self.nband
        else:
            wsums = np.broadcast_to(np.asarray(wsums, dtype=float), (self.nband,))
        if workers is None:
            workers = BandWorkerPool(self.nband, nthreads)
        elif workers.nband != self.nband:
            raise ValueError(f"workers pool has {workers.nband} bands, expected {self.nband}")
        self._pool = workers
        self._pool.init_hess(partitions_per_band, nx, ny, nx_psf, ny_psf, etas, wsums)

    def dot(self, x):
        return self._pool.hess_dot(x)

    def hdot(self, x):
        return self.dot(x)

    def cg(self, rhs, x0=None, tol=None, maxit=None, minit=None):
        tol = self.cg_tol if tol is None else tol
        maxit = self.cg_maxit if maxit is None else maxit
        minit = self.cg_minit if minit is None else minit
        return self._pool.hess_cg(rhs, x0, tol, maxit, minit, self.cg_verbose)

    def get_mem(self):
        return self._pool.get_mem()
