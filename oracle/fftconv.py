"""numpy restatement of the PSF-convolution operators (TEST INFRASTRUCTURE).

Follows /root/reference/src/pfb_imaging/operators/psf.py:8-96 (and its jax twin
:99-104, the numpy-expressible form) and the Hessian compositions of
/root/reference/src/pfb_imaging/operators/hessian.py:103-143,313-349,487-518.

FFT conventions (ducc0.fft as called by the reference): ``r2c(forward=True,
inorm=0)`` is an unnormalised real-to-half-complex transform == numpy
``rfft2``; ``c2r(forward=False, inorm=2, lastsize=ny_psf)`` divides by the
total size == numpy ``irfft2``.
"""

import numpy as np


def r2c(a, axes=(-2, -1)):
    return np.fft.rfftn(a, axes=axes)


def c2r(a, lastsize, axes=(-2, -1)):
    s = [a.shape[ax] for ax in axes]
    s[-1] = lastsize
    return np.fft.irfftn(a, s=s, axes=axes)


def psf_convolve(x, psfhat, nx_psf, ny_psf):
    """crop(irfft2(rfft2(pad(x)) * psfhat)) over the last two axes (psf.py:8-96)."""
    nx, ny = x.shape[-2:]
    xhat = np.fft.rfftn(x, s=(nx_psf, ny_psf), axes=(-2, -1))
    out = np.fft.irfftn(xhat * psfhat, s=(nx_psf, ny_psf), axes=(-2, -1))
    return out[..., :nx, :ny]


def hessian_psf_slice(x, abspsf, ny_psf, beam=None, eta=None):
    """hessian.py:103-143."""
    nx_psf = abspsf.shape[0]
    xin = x if beam is None else x * beam
    out = psf_convolve(xin, abspsf, nx_psf, ny_psf)
    if beam is not None:
        out = out * beam
    if eta:
        out = out + x * eta
    return out


def hess_psf_dot(x, abspsf, ny_psf, beam=None, eta=0.0):
    """HessPSF.dot, hessian.py:313-349: per band pad*beam -> r2c -> *abspsf -> c2r -> crop*beam, + x*eta_b."""
    x3 = x if x.ndim == 3 else x[None]
    nband = x3.shape[0]
    eta = np.broadcast_to(np.asarray(eta, dtype=float), (nband,))
    out = np.empty_like(x3)
    for b in range(nband):
        bm = None if beam is None else beam[b]
        out[b] = hessian_psf_slice(x3[b], abspsf[b], ny_psf, beam=bm, eta=None)
    return out + x3 * eta[:, None, None]


def hess_direct_slice(x, abspsf, ny_psf, taperxy, eta, mode="forward"):
    """hessian.py:215-248."""
    nx, ny = x.shape
    nx_psf = abspsf.shape[0]
    xhat = np.fft.rfft2(x * taperxy, s=(nx_psf, ny_psf))
    xhat = xhat * (abspsf + eta) if mode == "forward" else xhat / (abspsf + eta)
    return np.fft.irfft2(xhat, s=(nx_psf, ny_psf))[:nx, :ny] * taperxy


def hessian_tree_dot(x, partitions, nx_psf, ny_psf, eta=0.0, wsum=None):
    """HessianTree.dot, hessian.py:487-518."""
    x3 = x if x.ndim == 3 else x[None]
    ncorr = x3.shape[0]
    if wsum is None:
        wsum = np.zeros(ncorr)
        for p in partitions:
            wsum = wsum + p["wsum"]
    else:
        wsum = np.broadcast_to(np.asarray(wsum, dtype=float), (ncorr,))
    out = np.zeros_like(x3)
    for p in partitions:
        for c in range(ncorr):
            out[c] += p["beam"][c] * psf_convolve(x3[c] * p["beam"][c], p["psfhat"][c], nx_psf, ny_psf)
    out /= wsum[:, None, None]
    out += eta * x3
    return out


def taperf(shape, taper_width):
    """/root/reference/src/pfb_imaging/utils/misc.py:968-975."""
    tapers = []
    for npix in shape:
        t = np.ones(npix)
        t[:taper_width] = 0.5 * (1 + np.cos(np.linspace(1.1 * np.pi, 2 * np.pi, taper_width)))
        t[-taper_width:] = 0.5 * (1 + np.cos(np.linspace(0, 0.9 * np.pi, taper_width)))
        tapers.append(t)
    return np.outer(*tapers)


def pcg(aop, b, x0=None, tol=1e-5, maxit=500, minit=100):
    """Restatement of pcg_numba without preconditioner (/root/reference/src/pfb_imaging/opt/pcg.py:88-199):
    r = A x0 - b; stop on ||x - xp|| / ||x|| <= tol (and k >= minit), maxit, or 5 stalls."""
    x = np.zeros_like(b) if x0 is None else x0
    r = aop(x) - b
    if not np.any(r):
        return x
    p = -r
    rnorm = np.vdot(r, r).real
    k, eps, stall = 0, 1.0, 0
    while (eps > tol or k < minit) and k < maxit and stall < 5:
        xp = x.copy()
        ap = aop(p)
        alpha = rnorm / np.vdot(p, ap).real
        x = x + alpha * p
        r = r + alpha * ap
        rnorm_next = np.vdot(r, r).real
        beta = rnorm_next / rnorm
        p = beta * p - r
        rnorm = rnorm_next
        k += 1
        epsp = eps
        eps = np.linalg.norm(x - xp) / np.linalg.norm(x)
        if abs(epsp - eps) < 1e-3 * tol:
            stall += 1
    return x


def power_method(aop, imsize, b0, tol=1e-5, maxit=250):
    """power_method_numba / power_method (/root/reference/src/pfb_imaging/opt/power_method.py:40-148) with an
    explicit start vector; returns (beta, b, iterations)."""
    b = b0 / np.linalg.norm(b0)
    beta, eps, k = 1.0, 1.0, 0
    bp = b.copy()
    while eps > tol and k < maxit:
        b = aop(bp)
        bnorm = np.linalg.norm(b)
        betap = beta
        beta = np.vdot(bp, b) / np.vdot(bp, bp)
        b = b / bnorm
        eps = np.abs(beta - betap) / betap
        k += 1
        bp[...] = b
    return float(beta), b, k
