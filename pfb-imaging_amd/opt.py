"""The backward (primal-dual) step of the SARA minor cycle on the GPU.

Mirrors /root/reference/src/pfb_imaging/opt/primal_dual.py:303-448 (``PrimalDual``; the legacy ``primal_dual`` /
``primal_dual_numba`` loops of :65-322 are here too), prox/l21.py:15-50 (``L21``) and prox/l1.py (``L1``).  When the gradient is the closure of the forward-backward splitting,
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


class L1:
    """g(alpha) = ||W alpha||_1 (prox/l1.py:8-28): the image-domain regulariser (ISTA / lasso when ``psi`` is
    ``IdentityPsi``).  Every band is thresholded on its own, which is the l21 prox of a one-band cube: the soft threshold runs
    in the same device kernel as ``L21`` (``pfbhip_prox_21m`` with ``nband = 1`` per band of the cube)."""

    def __init__(self, psi, nu=1.0):
        for name in ("dot", "hdot", "nband", "nbasis", "nxmax", "nymax"):
            if not hasattr(psi, name):
                raise TypeError(f"psi does not satisfy the PsiOperator protocol (missing {name})")
        self.psi = psi
        self.nu = nu
        self.weight = np.ones((psi.nbasis, psi.nymax, psi.nxmax))

    def prox(self, v, vout, lam, sigma=1.0):
        """vout = prox_{(lam / sigma) ||W .||_1}(v / sigma), in place on ``vout``."""
        from .prox import prox_21m

        v = np.asarray(v, dtype=np.float64)
        w = np.broadcast_to(self.weight, v.shape[1:])
        for b in range(v.shape[0]):  # one band at a time: the 2-norm over a single band is the absolute value
            vout[b] = prox_21m(v[b:b + 1] / sigma, lam / sigma, weight=w)[0]


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
            # Whether THIS rank could run the device loop: RCCL transport, at least one band here (pfbhip_primal_dual /
            # pfbhip_psfconv_power_method take nband >= 1), one correlation per band.
            ok = comm is None or (comm.transport == "rccl" and bool(pool.local))
            out = []
            for b in (pool.local if ok else ()):
                tree = getattr(pool.workers[b], "_hess", None)
                if tree is None or tree.ncorr != 1:
                    ok = False
                    break
                s = tree._slots(0)
                out.append((tree._plan, s, s, 1.0 / float(tree.wsum[0]), float(tree.eta)))
            # The choice is COLLECTIVE: the device loop's RCCL sequence (one all-reduce of a single-band coefficient cube,
            # then one of 3 doubles, per iteration) differs from the generic loop's (cube-level psi / hess all-reduces), so a
            # rank deciding on its own -- e.g. the ranks >= nband of a 4-band run on 8 GPUs, which hold no band -- would
            # deadlock the others.  Every rank evaluates this, in the same place, whenever the pool is distributed.
            if comm is not None:
                ok = comm.min_over_ranks(1.0 if ok else 0.0) == 1.0
            return (out, comm, list(pool.local)) if ok else None
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
        self.last = dict(iters=info.iters, status=info.status, eps=info.eps, loop_ms=float(info.loop_ms),
                         stages={n: (float(info.stage_ms[i]), int(info.stage_calls[i])) for i, n in enumerate(_lib.PD_STAGE_NAMES)})
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
            if _lib.any_nonzero(x):
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


def primal_dual_numba(x, v, lam, psih, psi, hessnorm, prox, l1weight, reweighter, grad, nu=1.0, sigma=None, mask=None,
                      tol=1e-5, maxit=1000, positivity=1, report_freq=10, gamma=1.0, verbosity=1, maxreweight=20):
    """The legacy fused-kernel primal-dual loop with inner l1 reweighting (opt/primal_dual.py:153-322), returning ``(x, v)``.

    Argument meaning follows the reference's BODY, not its parameter names: ``psi(image, coeffs_out)`` is the analysis and
    ``psih(coeffs, image_out)`` the synthesis operator, both in place; ``prox`` and ``mask`` are accepted and unused;
    ``reweighter(x)`` -- if given -- supplies new l1 weights whenever the inner loop converges, up to ``maxreweight``
    consecutive times.  The dual update, positivity and (with this package's ``Psi``) the dictionary run on the GPU; the
    reference's own test holds ``PrimalDual`` + ``L21`` to this trajectory (tests/test_primal_dual.py:57-105)."""
    from .prox import positivity as clamp
    from .prox import positivity_band as clamp_band

    xp, vp, xout = x.copy(), v.copy(), np.zeros_like(x)
    half = hessnorm / (2.0 * gamma)
    if sigma is None:
        sigma = half / nu
    tau = 0.98 / (half + sigma * nu**2)
    eps, run, last = 1.0, 0, 0
    k = 0
    for k in range(maxit):
        psi(xp, v)
        dual_update_numba_fast(vp, v, lam, sigma=sigma, weight=l1weight)
        vp[...] = 2.0 * v - vp
        psih(vp, xout)
        xout += grad(xp)
        x[...] = xp - tau * xout
        if positivity == 1:
            clamp(x)
        elif positivity == 2:
            clamp_band(x)
        eps = float(np.sqrt(((x - xp) ** 2).sum() / max(float((x**2).sum()), 1e-12))) if _lib.any_nonzero(x) else 1.0
        if eps < tol:
            if reweighter is None or run >= maxreweight:
                break
            l1weight = reweighter(x)
            run = run + 1 if k - last == 1 else 0
            last = k
        np.copyto(xp, x)
        np.copyto(vp, v)
        if verbosity > 1 and not k % report_freq:
            print(f"At iteration {k} eps = {eps:.3e}")
    if verbosity:
        print(f"Max iters reached. eps = {eps:.3e}" if k == maxit - 1 else f"Success, converged after {k} iterations")
    return x, v


def primal_dual(x, v, lam, psi, psih, hessnorm, prox, grad, nu=1.0, sigma=None, mask=None, tol=1e-5, maxit=1000, minit=10,
                positivity=1, report_freq=10, gamma=1.0, verbosity=1):
    """The legacy allocating primal-dual loop (opt/primal_dual.py:65-150), returning ``(x, v)``.

    Here ``psi(coeffs)`` is the SYNTHESIS and ``psih(image)`` the ANALYSIS operator, both returning new arrays, and
    ``prox(v, lam)`` returns the prox of the regulariser; the step is ``tau = 0.9 / (hessnorm / 2 gamma + sigma nu^2)`` and the
    loop runs at least ``minit`` iterations.  Pure composition of the caller's callables (GPU when they are this package's)."""
    xp, vp = x.copy(), v.copy()
    half = hessnorm / (2.0 * gamma)
    if sigma is None:
        sigma = half / nu
    tau = 0.9 / (half + sigma * nu**2)
    eps, k = 1.0, 0
    while (eps > tol or k < minit) and k < maxit:
        vtilde = v + sigma * psih(xp)
        v = vtilde - sigma * prox(vtilde / sigma, lam / sigma)
        x = xp - tau * (psi(2 * v - vp) + grad(xp))
        if positivity == 1:
            x[x < 0.0] = 0.0
        elif positivity == 2:
            x[:, np.any(x <= 0, axis=0)] = 0.0
        eps = float(np.linalg.norm(x - xp) / np.linalg.norm(x))
        xp[...] = x
        vp[...] = v
        if not np.isfinite(eps):
            raise FloatingPointError(f"primal_dual: eps = {eps} at iteration {k}")  # (the reference drops into pdb here)
        if verbosity > 1 and not k % report_freq:
            print(f"At iteration {k} eps = {eps:.3e}")
        k += 1
    if verbosity:
        print(f"Max iters reached. eps = {eps:.3e}" if k == maxit else f"Success, converged after {k} iterations")
    return x, v


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


# ---------------------------------------------------------------------------------------------------------
# Conjugate gradients (opt/pcg.py of the reference): pcg / pcg_numba / PCG / pcg_dds over the on-device solver
# ---------------------------------------------------------------------------------------------------------

def _device_cg_target(aop):
    """What the on-device CG can solve without leaving HBM, or None: ``aop`` is

    * ``functools.partial(hessian_slice, uvw=..., weight=..., vis_mask=..., freq=..., beam=..., cell=..., eta=..., wsum=...)``
      -- the operator ``pcg_dds`` builds (opt/pcg.py:529-548) --, or
    * the bound ``hessian`` of a :class:`~pfb_imaging_amd.wgridder.Gridder`, bare or in a ``partial`` carrying only
      ``beam`` / ``eta`` / ``wsum`` keywords.

    Returns ``(gridder, cached, beam, eta, wsum)``.
    """
    import functools

    from .operators.hessian import hessian_slice
    from .wgridder import Gridder

    fn, kw = aop, {}
    if isinstance(aop, functools.partial):
        if aop.args:
            return None
        fn, kw = aop.func, dict(aop.keywords)
    owner = getattr(fn, "__self__", None)
    if isinstance(owner, Gridder) and getattr(fn, "__func__", None) is Gridder.hessian:
        if set(kw) - {"beam", "eta", "wsum"}:
            return None
        return owner, True, kw.get("beam"), kw.get("eta") or 0.0, kw.get("wsum") or 0.0
    if fn is hessian_slice:
        need = {"uvw", "freq", "cell"}
        allowed = need | {"weight", "vis_mask", "beam", "x0", "y0", "flip_u", "flip_v", "flip_w", "do_wgridding", "epsilon",
                          "double_accum", "nthreads", "eta", "wsum"}
        if not need <= set(kw) or set(kw) - allowed:
            return None
        return ("slice", kw)
    return None


def _cg_host(aop, b, x0, precond, tol, maxit, minit, verbosity, report_freq, return_resid, name):
    """The reference's CG loop around an arbitrary callable (opt/pcg.py:88-314): start residual ``r = A x0 - b``, direction
    ``p = -M r``, per iteration ``alpha = (r.y) / (p.Ap)``, ``x += alpha p``, ``r += alpha Ap``, ``y = M r``,
    ``beta = (r.y)_new / (r.y)_old``, ``p = beta p - y``; it stops on ``||x - xp|| / ||x|| <= tol`` (and ``k >= minit``),
    on ``maxit``, or after five iterations whose stopping measure moved by less than ``1e-3 tol``.  ``x0`` is the iterate."""
    if precond is None:
        def precond(v):
            return v
    x = x0
    r = aop(x) - b
    y = precond(r)
    if not _lib.any_nonzero(y):
        print("Initial residual is zero")
        return (x, r) if return_resid else x
    p = -y
    ry = float(np.vdot(r, y).real)
    phi0 = 1.0 if (np.isnan(ry) or ry == 0.0) else ry
    k, eps, stalls = 0, 1.0, 0
    while (eps > tol or k < minit) and k < maxit and stalls < 5:
        ap = aop(p)
        ry = float(np.vdot(r, y).real)
        alpha = ry / float(np.vdot(p, ap).real)
        xprev = x.copy()
        x += alpha * p          # in place: x IS the caller's x0
        r = r + alpha * ap      # (aop may hand back an internal buffer: never scale it in place)
        y = precond(r)
        ry_next = float(np.vdot(r, y).real)
        p = (ry_next / ry) * p - y
        k += 1
        eps_prev = eps
        eps = float(np.sqrt(((x - xprev) ** 2).sum() / max(float((x ** 2).sum()), 1e-12)))
        if abs(eps_prev - eps) < 1e-3 * tol:
            stalls += 1
        if verbosity > 1 and not k % report_freq:
            print(f"At iteration {k} eps = {eps:.3e}, phi = {ry_next / phi0:.3e}")
    if verbosity:
        if k >= maxit:
            print(f"Max iters reached. eps = {eps:.3e}")
        elif stalls >= 5:
            print(f"Stalled after {k} iterations with eps = {eps:.3e}")
        else:
            print(f"Success, converged after {k} iterations")
    _cg_host.last = dict(iters=k, eps=eps, status=1 if k >= maxit else (2 if stalls >= 5 else 0), where=name)
    return (x, r) if return_resid else x


def _cg_device(target, aop, b, x0, tol, maxit, minit, verbosity, return_resid):
    """Whole solve on the device (pfbhip_gridder_cg: the Hessian applies and every CG vector stay in HBM); the result is
    written into ``x0`` (the reference's in-place contract) when one is given."""
    from .operators.hessian import hessian_slice  # noqa: F401
    from .wgridder import _fingerprint, _get_gridder

    b = np.asarray(b, dtype=np.float64)
    if target[0] == "slice":
        kw = target[1]
        nx, ny = b.shape
        g, cached = _get_gridder(kw["uvw"], kw["freq"], kw.get("vis_mask"), npix_x=nx, npix_y=ny,
                                 pixsize_x=float(kw["cell"]), pixsize_y=float(kw["cell"]), center_x=float(kw.get("x0", 0.0)),
                                 center_y=float(kw.get("y0", 0.0)), epsilon=float(kw.get("epsilon", 1e-7)),
                                 flip_u=bool(kw.get("flip_u", False)), flip_v=bool(kw.get("flip_v", True)),
                                 flip_w=bool(kw.get("flip_w", False)), do_wgridding=bool(kw.get("do_wgridding", True)),
                                 divide_by_n=False, sigma_min=1.1, sigma_max=2.6)
        weight = kw.get("weight")
        token = None if weight is None else _fingerprint(np.asarray(weight))
        if getattr(g, "_hess_weight_token", "unset") != token:
            g.set_weights(weight)
            g._hess_weight_token = token
        beam, eta, wsum = kw.get("beam"), kw.get("eta") or 0.0, kw.get("wsum") or 0.0
    else:
        g, cached, beam, eta, wsum = target
    try:
        sol = g.cg(b, x0=x0, beam=beam, eta=eta, wsum=wsum, tol=tol, maxit=maxit, minit=minit)
        info = dict(g.last_cg, where="device")
    finally:
        if not cached:
            g.close()
    _cg_host.last = info
    if verbosity:
        if info["status"] == 1:
            print(f"Max iters reached. eps = {info['eps']:.3e}")
        elif info["status"] == 2:
            print(f"Stalled after {info['iters']} iterations with eps = {info['eps']:.3e}")
        else:
            print(f"Success, converged after {info['iters']} iterations")
    if x0 is not None:
        x0[...] = sol
        sol = x0
    if return_resid:
        return sol, aop(sol) - b
    return sol


def pcg_numba(aop, b, x0=None, precond=None, tol=1e-5, maxit=500, minit=100, verbosity=1, report_freq=10, backtrack=True,
              return_resid=False):
    """CG solve of ``aop(x) = b`` with the contract of the reference's ``pcg_numba`` (opt/pcg.py:88-199): ``x0`` -- when
    given -- is the iterate, updated IN PLACE and returned; stopping rule as in :func:`_cg_host`.

    When ``aop`` is the exact Hessian of this package (see :func:`_device_cg_target`) and there is no preconditioner, the
    whole solve runs on the device (``Gridder.cg``); any other callable gets the same loop on the host around it.
    ``backtrack`` is accepted and unused, as in the reference.
    """
    _lib.require_gpu()
    if x0 is None:
        x0 = np.zeros(np.shape(b), dtype=np.asarray(b).dtype)
    target = _device_cg_target(aop) if precond is None else None
    if target is not None and np.asarray(b).ndim == 2:
        if not _lib.any_nonzero(b) and not _lib.any_nonzero(x0):
            print("Initial residual is zero")
            return (x0, np.zeros_like(x0)) if return_resid else x0
        return _cg_device(target, aop, b, x0, tol, maxit, minit, verbosity, return_resid)
    return _cg_host(aop, b, x0, precond, tol, maxit, minit, verbosity, report_freq, return_resid, "host")


def pcg(aop, b, x0=None, precond=None, tol=1e-5, maxit=500, minit=100, verbosity=1, report_freq=10, backtrack=True,
        return_resid=False):
    """``pcg`` of the reference (opt/pcg.py:200-314): the same solver with an optional preconditioner callable
    (``HessPSF.idot`` uses it).  A preconditioned solve runs the host loop around the two callables."""
    return pcg_numba(aop, b, x0=x0, precond=precond, tol=tol, maxit=maxit, minit=minit, verbosity=verbosity,
                     report_freq=report_freq, backtrack=backtrack, return_resid=return_resid)


class PCG:
    """ForwardSolver ``update ~= hess^-1 residual`` (opt/pcg.py:586-630): delegates to ``hess.cg`` when the operator has
    one (HessTreeRay / HessianTree / Gridder: on-device solves), else runs :func:`pcg_numba` over ``hess.dot``."""

    def __init__(self, tol=1e-3, maxit=150, minit=1, verbosity=0, report_freq=10):
        self.tol, self.maxit, self.minit, self.verbosity, self.report_freq = tol, maxit, minit, verbosity, report_freq

    def solve(self, hess, residual, x0=None):
        if hasattr(hess, "cg"):
            return hess.cg(residual, x0=x0, tol=self.tol, maxit=self.maxit, minit=self.minit)
        from .operators import LinearOperator, require_protocol

        require_protocol(hess, LinearOperator, "hess")
        return pcg_numba(hess.dot, residual, x0=x0, tol=self.tol, maxit=self.maxit, minit=self.minit,
                         verbosity=self.verbosity, report_freq=self.report_freq)


def pcg_psf(psfhat, b, x0, beam, lastsize, nthreads, eta, cgopts, compute=True):
    """Per-band CG on the PSF-approximate Hessian ``beam (PSF (*) (beam x)) + eta x`` (opt/pcg.py:317-441).  The reference wraps one
    ``pcg`` over ``hessian_psf_slice`` per band in a dask ``blockwise`` graph (its ``x / eta`` preconditioner is a scalar and leaves
    the CG iterates unchanged); here every band's solve runs on the device (``pfbhip_psfconv_cg``) and the model cube comes back
    as a numpy array (``compute`` is accepted; there is no graph to defer).  ``cgopts``: ``tol, maxit, minit`` (+ the reporting keys,
    ignored)."""
    from .psfconv import cached_plan, cached_psf_slot

    psfhat, b = np.asarray(psfhat), np.asarray(b)
    if b.ndim != 3 or psfhat.ndim != 3 or psfhat.shape[0] != b.shape[0]:
        raise ValueError(f"psfhat {psfhat.shape} / b {b.shape}: expected (nband, nx_psf, nyo2) and (nband, nx, ny)")
    nband, nx, ny = b.shape
    eta = np.tile(eta, nband) if isinstance(eta, float) else np.array(eta)
    assert eta.size == nband
    if beam is not None:
        beam = np.asarray(beam)
        if beam.ndim == 2:
            beam = beam[None]
        if beam.shape[0] == 1:
            beam = np.tile(beam, (nband, 1, 1))
        elif beam.shape[0] != nband:
            raise ValueError("Beam has incorrect shape")
    x0 = np.zeros_like(b) if x0 is None else np.asarray(x0)
    opts = {k: v for k, v in dict(cgopts or {}).items() if k in ("tol", "maxit", "minit")}
    plan = cached_plan(nx, ny, psfhat.shape[1], int(lastsize))
    model = np.zeros((nband, nx, ny), dtype=b.dtype)
    for k in range(nband):
        slot = cached_psf_slot(plan, np.abs(psfhat[k]))
        bslot = -1
        if beam is not None:
            bslot = 8 + k % 8   # (beam slots of the shared plan: a small ring, rebound per band)
            plan.set_beam(bslot, beam[k])
        model[k] = plan.cg(b[k], [slot], [bslot], scale=1.0, eta=float(eta[k]), x0=x0[k], **opts)
    return model


def _ds_get(ds, name):
    """Field ``name`` of an xarray-like dataset or a plain mapping, as a numpy array (None if absent)."""
    if isinstance(ds, dict):
        v = ds.get(name)
    else:
        v = getattr(ds, name, None) if name in ds else None
    if v is None:
        return None
    return np.asarray(getattr(v, "values", v))


def _ds_attr(ds, name, default=None):
    if isinstance(ds, dict):
        return ds.get(name, ds.get("attrs", {}).get(name, default))
    return getattr(ds, name, getattr(ds, "attrs", {}).get(name, default))


def pcg_dds(ds, eta, mask=1.0, use_psf=True, residual_name="RESIDUAL", model_name="MODEL", do_wgridding=True, epsilon=5e-4,
            double_accum=True, nthreads=1, zero_model_outside_mask=False, tol=1e-5, maxit=500, verbosity=1, report_freq=10):
    """The flux-mop solve of ``pfb fluxtractor`` for one band (opt/pcg.py:444-583): ``x = (beam R^H W R beam / wsum + eta)^-1
    (beam * residual / wsum)`` by CG on the exact Hessian, ``model += x``, and the exact residual of the new model.

    ``ds`` is the band's dataset: an xarray ``Dataset`` or a mapping with the arrays ``DIRTY, BEAM, UVW, WEIGHT, MASK,
    FREQ`` (optionally ``MODEL`` / ``RESIDUAL`` / ``UPDATE``) and the attributes ``cell_rad, x0, y0, flip_u, flip_v,
    flip_w, wsum, bandid``.  The reference's zarr round trip stays with the caller: the new fields come back in a dict
    (``MODEL_MOPPED, RESIDUAL_MOPPED, UPDATE, X0``) and are also ``assign``-ed when ``ds`` is an xarray Dataset.

    Returns ``(resid, bandid, fields)``.  Both exact-Hessian applications and the whole CG run on the device.
    """
    from functools import partial

    from .operators.hessian import hessian_slice

    if isinstance(ds, (list, tuple)):
        ds = ds[0]
    if isinstance(ds, (str, bytes)):
        raise TypeError("pcg_dds takes the band's dataset (xarray Dataset or mapping of arrays), not a store path: "
                        "open it with the reference's xds_from_list and pass the Dataset")
    dirty, beam0 = _ds_get(ds, "DIRTY"), _ds_get(ds, "BEAM")
    common = dict(uvw=_ds_get(ds, "UVW"), weight=_ds_get(ds, "WEIGHT"), vis_mask=_ds_get(ds, "MASK"), freq=_ds_get(ds, "FREQ"),
                  cell=_ds_attr(ds, "cell_rad"), x0=_ds_attr(ds, "x0", 0.0), y0=_ds_attr(ds, "y0", 0.0),
                  do_wgridding=do_wgridding, epsilon=epsilon, double_accum=double_accum, nthreads=nthreads)
    flips = dict(flip_u=bool(_ds_attr(ds, "flip_u", False)), flip_v=bool(_ds_attr(ds, "flip_v", True)),
                 flip_w=bool(_ds_attr(ds, "flip_w", False)))
    beam = mask * beam0
    model = _ds_get(ds, model_name)
    if zero_model_outside_mask:
        if model is None:
            raise RuntimeError(f"Asked to zero model outside mask but {model_name} not in dds")
        model = np.where(np.asarray(mask) > 0, model, 0.0)
        print("Zeroing model outside mask")
        j = (dirty - hessian_slice(model, beam=beam0, **common, **flips)) * beam
    else:
        model = np.zeros(dirty.shape, dtype=float) if model is None else np.array(model, dtype=float)
        resid_in = _ds_get(ds, residual_name)
        j = (dirty if resid_in is None else resid_in) * beam
    wsum = float(_ds_attr(ds, "wsum"))
    j = j / wsum
    upd = _ds_get(ds, "UPDATE")
    x0 = np.zeros_like(j) if upd is None else upd * mask
    hess = partial(hessian_slice, beam=beam, eta=eta, wsum=wsum, **common, **flips)
    x = pcg(hess, j, x0=np.array(x0, dtype=float), precond=None, tol=tol, maxit=maxit, minit=1, verbosity=verbosity,
            report_freq=report_freq, backtrack=False, return_resid=False)
    model = model + x
    resid = dirty - hessian_slice(model, beam=beam0, **common, **flips)
    fields = {"MODEL_MOPPED": model, "RESIDUAL_MOPPED": resid, "UPDATE": x, "X0": x0}
    if hasattr(ds, "assign"):
        ds = ds.assign(**{k: (("x", "y"), v) for k, v in fields.items()})
    elif isinstance(ds, dict):
        ds.update(fields)
    return resid, int(_ds_attr(ds, "bandid", 0)), fields
