"""The backward (primal-dual) step of the SARA minor cycle on the GPU.

Mirrors /root/reference/src/pfb_imaging/opt/primal_dual.py:303-448 (``PrimalDual``) and
prox/l21.py:15-50 (``L21``).  When the gradient is the closure of the forward-backward splitting,
``grad(x) = -hess.dot(xtilde - x) / gamma`` (core/sara.py:288-289, deconv/pfb.py:158-161), expressed as a
``PsfGrad`` over a device-resident ``HessPSF``, and the dictionary is this package's ``Psi`` / ``PsiNocopyt``,
``solve`` runs the whole loop on the device (``pfbhip_primal_dual``: one scalar round trip per iteration).
Any other gradient callable runs the reference's loop with the GPU dictionary / dual update and the
reference's own host-side vector steps.

``power_method`` (opt/power_method.py:40-148) is here too: the spectral norm of ``hess.dot`` that sets the
primal-dual step sizes (core/sara.py:200-209, deconv/pfb.py:118-126), iterated on the device when ``aop`` is the
``dot`` of a device-resident Hessian.
"""

import ctypes as ct

import numpy as np

from . import _lib
from ._lib import PDInfo, PMInfo, check, cint, f64, i64, lib, ptr
from .operators.psi import Psi, PsiNocopyt
from .prox import dual_update_numba_fast, prox_21m_numba


class L21:
    """R(x) = ||W Psi^T x||_{2,1}, 2-norm over the band axis (prox/l21.py:15-50)."""

    def __init__(self, psi, bases, nu=1.0, rmsfactor=1.0, alpha=2.0):
        for name in ("dot", "hdot", "nband", "nbasis", "nxmax", "nymax"):
            if not hasattr(psi, name):
                raise TypeError(f"psi does not satisfy the PsiOperator protocol (missing {name})")
        self.psi = psi
        self.nu = nu
        self.bases = tuple(bases)
        self.rmsfactor = rmsfactor
        self.alpha = alpha
        self.l1weight = np.ones(self.coeff_shape()[1:])
        self._outvar = None
        self._rms_comps = None

    def coeff_shape(self):
        """(nband, nbasis, n1, n2) in the layout of ``psi`` (the reference allocates (.., nymax, nxmax))."""
        p = self.psi
        if isinstance(p, PsiNocopyt):
            return (p.nband, p.nbasis, p.nxmax, p.nymax)
        return (p.nband, p.nbasis, p.nymax, p.nxmax)

    def prox(self, v, vout, lam, sigma=1.0):
        prox_21m_numba(v, vout, lam, sigma=sigma, weight=self.l1weight)

    # ---- l1 reweighting (prox/l21.py:52-88, utils/misc.py:742-755): host bookkeeping on the GPU analysis ----
    @property
    def reweight_active(self):
        return self._rms_comps is not None

    def _band_sum(self, x):
        if self._outvar is None:
            self._outvar = np.zeros(self.coeff_shape())
        self.psi.dot(x, self._outvar)
        return np.sum(self._outvar, axis=0)

    def init_reweighting(self, update):
        """Estimate per-basis component rms from the update and arm reweighting (l21.py:56-78)."""
        tmp = self._band_sum(update)
        rms = np.ones(self.psi.nbasis, dtype=float)
        for i in range(self.psi.nbasis):
            nonzero = tmp[i][tmp[i] != 0]
            if nonzero.size:
                rms[i] = np.std(nonzero)
        self._rms_comps = rms

    def update_weights(self, x):
        """l1weight = (1 + rmsfactor) / (1 + |sum_band Psi^T x|^alpha / rms^alpha) (misc.py:742-755)."""
        if self._rms_comps is None:
            raise RuntimeError("reweighting not initialised; call init_reweighting() first")
        mcomps = np.abs(self._band_sum(x))
        self.l1weight = (1 + self.rmsfactor) / (1 + mcomps**self.alpha / self._rms_comps[:, None, None] ** self.alpha)

    def dual_update(self, vp, v, lam, sigma=1.0):
        dual_update_numba_fast(vp, v, lam, sigma=sigma, weight=self.l1weight)


class PsfGrad:
    """grad(x) = -hess.dot(xtilde - x) / gamma with ``hess`` a PSF-approximate Hessian (HessPSF)."""

    def __init__(self, hess, xtilde, gamma=1.0):
        self.hess, self.gamma = hess, float(gamma)
        self.xtilde = np.ascontiguousarray(xtilde, dtype=np.float64)

    def __call__(self, x):
        return -self.hess.dot(self.xtilde - x) / self.gamma


class PrimalDual:
    """primal_dual.py:303-448: same constructor, ``setup`` / ``set_grad`` / ``reset`` / ``solve`` contract."""

    def __init__(self, tol=1e-5, maxit=1000, report_freq=10, verbosity=1, gamma=1.0, sigma=None, on_converge=None,
                 primal_prox=None):
        self.tol, self.maxit, self.report_freq, self.verbosity = tol, maxit, report_freq, verbosity
        self.gamma, self._sigma_opt = gamma, sigma
        self.on_converge, self.primal_prox = on_converge, primal_prox
        self._grad = self._reg = self._v = None
        self.last = None

    def setup(self, prox, hessnorm):
        if not all(hasattr(prox, a) for a in ("psi", "nu", "prox")):
            raise TypeError("prox does not satisfy the Regulariser protocol")
        self._reg = prox
        self.hessnorm = hessnorm
        nu = prox.nu
        sigma = self._sigma_opt
        if sigma is None:
            sigma = hessnorm / (2.0 * self.gamma) / nu
        self.sigma = sigma
        self.tau = 0.98 / (hessnorm / (2.0 * self.gamma) + sigma * nu**2)
        shape = prox.coeff_shape() if hasattr(prox, "coeff_shape") else (prox.psi.nband, prox.psi.nbasis, prox.psi.nymax,
                                                                          prox.psi.nxmax)
        self._v = np.zeros(shape)

    def set_grad(self, grad):
        self._grad = grad

    def reset(self):
        if self._v is not None:
            self._v[...] = 0.0

    # ---- device-resident loop --------------------------------------------------------------
    @staticmethod
    def _hess_bands(hess, nband):
        """(bands, comm, local): per LOCAL band (plan, psf slots, beam slots, scale, eta) of a device-resident PSF
        Hessian -- HessPSF (all bands here, comm None) or HessTreeRay (this rank's bands of its pool) -- or None."""
        from .operators.hessian import HessPSF, HessTreeRay

        if isinstance(hess, HessPSF) and hess.nband == nband:
            bands = [(hess._plan, [b], [-1 if hess.beam[b] is None else b], 1.0, float(hess.eta[b])) for b in range(nband)]
            return bands, None, list(range(nband))
        if isinstance(hess, HessTreeRay) and hess.nband == nband:
            pool = hess._pool
            comm = pool.comm if (pool.comm is not None and pool.comm.world_size > 1) else None
            if comm is not None and (comm.transport != "rccl" or not pool.local):
                return None  # CPU transport (tests) or a rank without bands: the generic loop handles it
            out = []
            for b in pool.local:
                tree = getattr(pool.workers[b], "_hess", None)
                if tree is None or tree.ncorr != 1:
                    return None
                s = tree._slots(0)
                out.append((tree._plan, s, s, 1.0 / float(tree.wsum[0]), float(tree.eta)))
            return out, comm, list(pool.local)
        return None

    def _device_path(self):
        from .prox import positivity, positivity_band

        g, reg = self._grad, self._reg
        if not (isinstance(g, PsfGrad) and isinstance(reg, L21)) or self.on_converge is not None:
            return None
        if not isinstance(reg.psi, (Psi, PsiNocopyt)):
            return None
        if self._hess_bands(g.hess, reg.psi.nband) is None:
            return None
        return {None: 0, positivity: 1, positivity_band: 2}.get(self.primal_prox, None)

    def _solve_device(self, x, lam, mode):
        reg, psi = self._reg, self._reg.psi
        bands, comm, local = self._hess_bands(self._grad.hess, psi.nband)
        nloc = len(local)
        transposed = isinstance(psi, Psi)
        vfull = self._v.transpose(0, 1, 3, 2) if transposed else self._v
        w = reg.l1weight.transpose(0, 2, 1) if transposed else reg.l1weight
        v = np.ascontiguousarray(vfull[local], dtype=np.float64)
        w = np.ascontiguousarray(np.broadcast_to(w, v.shape[1:]), dtype=np.float64)
        xs = np.ascontiguousarray(np.asarray(x, dtype=np.float64)[local])
        xt = np.ascontiguousarray(self._grad.xtilde[local])
        handles = (ct.c_void_p * nloc)(*[b[0]._h for b in bands])
        nparts = np.array([len(b[1]) for b in bands], dtype=np.int64)
        psf_slots = np.array([s for b in bands for s in b[1]], dtype=np.int64)
        beam_slots = np.array([s for b in bands for s in b[2]], dtype=np.int64)
        scale = np.array([b[3] for b in bands], dtype=np.float64)
        eta = np.array([b[4] for b in bands], dtype=np.float64)
        info = PDInfo()
        check(lib().pfbhip_primal_dual(psi._band._h, handles, i64(nloc), ptr(nparts), ptr(psf_slots), ptr(beam_slots),
                                       ptr(scale), ptr(eta), ptr(xt), f64(self._grad.gamma), ptr(xs), ptr(v),
                                       ptr(w), f64(lam), f64(self.sigma), f64(self.tau), cint(mode), f64(self.tol),
                                       cint(self.maxit), None if comm is None else comm._h, ct.byref(info)))
        if comm is None:
            xall, vall = xs, v
        else:  # every rank holds the full cubes again: each band was produced by exactly one rank
            xall = np.zeros(x.shape)
            xall[local] = xs
            xall = comm.allreduce_sum(xall).reshape(x.shape)
            vall = np.zeros(vfull.shape)
            vall[local] = v
            vall = comm.allreduce_sum(vall).reshape(vfull.shape)
        self._v[...] = vall.transpose(0, 1, 3, 2) if transposed else vall
        self.last = dict(iters=info.iters, status=info.status, eps=info.eps)
        x[...] = xall
        return x

    # ---- the reference's loop (any gradient callable) ----------------------------------------
    def _dual_step(self, xp, v, vp, lam):
        reg = self._reg
        reg.psi.dot(xp, v)
        if hasattr(reg, "dual_update"):
            reg.dual_update(vp, v, lam, sigma=self.sigma)
        else:
            vtilde = vp + self.sigma * v
            reg.prox(vtilde, v, lam, sigma=self.sigma)
            np.subtract(vtilde, self.sigma * v, out=v)

    def solve(self, x, lam):
        if self._reg is None:
            raise RuntimeError("regulariser not bound; call setup() before solve()")
        if self._grad is None:
            raise RuntimeError("grad not set; call set_grad() before solve()")
        _lib.require_gpu()
        mode = self._device_path()
        if mode is not None:
            return self._solve_device(x, lam, mode)
        xp = x.copy()
        v = self._v
        vp = v.copy()
        xout = np.zeros_like(x)
        eps, k = 1.0, 0
        for k in range(self.maxit):
            self._dual_step(xp, v, vp, lam)
            vp[...] = 2.0 * v - vp
            self._reg.psi.hdot(vp, xout)
            xout += self._grad(xp)
            x[...] = xp - self.tau * xout
            if self.primal_prox is not None:
                self.primal_prox(x)
            if x.any():
                eps = float(np.sqrt(((x - xp) ** 2).sum() / max((x**2).sum(), 1e-12)))
            else:
                eps = 1.0
            if eps < self.tol:
                if self.on_converge is None or self.on_converge(x, k, eps):
                    break
            np.copyto(xp, x)
            np.copyto(vp, v)
        self.last = dict(iters=k, status=0 if eps < self.tol else 1, eps=eps)
        return x


def _pm_device(aop, imsize, b):
    """(call, comm) running the whole power iteration on the device for ``aop``, or None: ``aop`` must be the bound
    ``dot`` of a HessPSF / HessTreeRay (cube) or the bound ``hessian`` of a Gridder with no extra arguments."""
    from .operators.hessian import HessPSF, HessTreeRay
    from .wgridder import Gridder

    owner = getattr(aop, "__self__", None)
    fn = getattr(aop, "__func__", None)
    if isinstance(owner, Gridder) and fn is Gridder.hessian and tuple(imsize) == (owner.nx, owner.ny):
        def call(b, tol, maxit, info):
            check(lib().pfbhip_gridder_power_method(owner._h, None, f64(0.0), f64(0.0), ptr(b), f64(tol), cint(maxit),
                                                    ct.byref(info)))
            return b
        return call
    if isinstance(owner, (HessPSF, HessTreeRay)) and fn is type(owner).dot and len(imsize) == 3:
        hb = PrimalDual._hess_bands(owner, imsize[0])
        if hb is None:
            return None
        bands, comm, local = hb

        def call(b, tol, maxit, info):
            nloc = len(local)
            bs = np.ascontiguousarray(b[local])
            handles = (ct.c_void_p * nloc)(*[x[0]._h for x in bands])
            nparts = np.array([len(x[1]) for x in bands], dtype=np.int64)
            psf_slots = np.array([s for x in bands for s in x[1]], dtype=np.int64)
            beam_slots = np.array([s for x in bands for s in x[2]], dtype=np.int64)
            scale = np.array([x[3] for x in bands], dtype=np.float64)
            eta = np.array([x[4] for x in bands], dtype=np.float64)
            check(lib().pfbhip_psfconv_power_method(handles, i64(nloc), ptr(nparts), ptr(psf_slots), ptr(beam_slots),
                                                    ptr(scale), ptr(eta), ptr(bs), f64(tol), cint(maxit),
                                                    None if comm is None else comm._h, ct.byref(info)))
            if comm is None:
                return bs
            out = np.zeros(b.shape)
            out[local] = bs
            return comm.allreduce_sum(out).reshape(b.shape)
        return call
    return None


def power_method(aop, imsize, b0=None, tol=1e-5, maxit=250, verbosity=1, report_freq=25):
    """Largest eigenvalue of the symmetric operator ``aop`` by power iteration; returns ``(beta, b)`` like
    opt/power_method.py:40-93 (``power_method_numba``) and :96-148 (``power_method``): ``b`` starts at ``b0 / ||b0||``
    (``randn`` when None), ``beta = (bp . A bp) / (bp . bp)``, stop when ``|beta - beta_prev| / beta_prev <= tol``.

    The iteration stays on the device (one scalar round trip per iteration) when ``aop`` is ``HessPSF.dot`` /
    ``HessTreeRay.dot`` / ``Gridder.hessian`` of this package; any other callable runs the reference's host loop
    around it.
    """
    if b0 is None:
        b = np.random.randn(*imsize)
    else:
        b = np.array(b0, dtype=np.float64)
    if b.shape != tuple(imsize):
        raise ValueError(f"b0 shape {b.shape} != {tuple(imsize)}")
    dev = _pm_device(aop, imsize, b)
    if dev is not None:
        info = PMInfo()
        b = dev(np.ascontiguousarray(b), tol, maxit, info)
        power_method.last = dict(iters=int(info.iters), status=int(info.status), eps=float(info.eps))
        return float(info.beta), b
    b /= np.linalg.norm(b)
    beta, eps, k = 1.0, 1.0, 0
    bp = b.copy()
    while eps > tol and k < maxit:
        b = aop(bp)
        bnorm = np.linalg.norm(b)
        betap = beta
        beta = float(np.vdot(bp, b) / np.vdot(bp, bp))
        b = b / bnorm  # aop may return an internal buffer (HessPSF.dot does): never scale it in place
        eps = abs(beta - betap) / betap
        k += 1
        bp[...] = b
    power_method.last = dict(iters=k, status=int(k == maxit and eps > tol), eps=eps)
    return beta, b


power_method_numba = power_method
