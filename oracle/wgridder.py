"""CPU restatement of the w-stacking ES-kernel gridder/degridder (TEST INFRASTRUCTURE).

This is the algorithm the reference reaches through
``ducc0.wgridder.experimental.vis2dirty / dirty2vis``
(/root/reference/src/pfb_imaging/operators/hessian.py:50-89,
/root/reference/src/pfb_imaging/operators/gridder.py:78,128,590-613,972-1016)
restated from the published method (Arras et al. 2021, A&A 646 A58; Barnett et
al. 2019 for the exponential-of-semicircle kernel, the same form as
/root/reference/src/pfb_imaging/utils/weighting.py:25-35):

  vis2dirty:  weight/phase-shift vis -> for each w-plane: scatter with
              phi(u) phi(v) phi(w) onto an oversampled grid -> inverse FFT ->
              crop -> multiply by the w-screen exp(-2 pi i w_p (n-1+nshift))
              -> accumulate real part;  finally divide by the kernel's
              Fourier transform in l, m and n-1.
  dirty2vis:  exact adjoint, run backwards.

ducc0 (locked 0.41.0) is not installable here, so the kernel (W, beta) table is
this build's own (tools/make_kernel_table.py) and parity against ducc0's exact
output is unpinned; parity is pinned against oracle.dft -- itself pinned to the
reference's explicit_wdegridder / explicit_degridder -- to the requested
epsilon (see oracle/__init__.py).

Three w-schemes (GridParams.wmode), the restatement of the product's choice: 0 ES-kernel
planes, 1 polynomial planes through Chebyshev nodes in w, 2 ONE plane with the rest of
the w-term carried by differentiated gridding kernels (Plan._init_wd, pfb_oracle.c:
pfbo_grid_plane_wd).

The scatter/gather inner loops are C (oracle/pfb_oracle.c); FFTs are
scipy.fft (pocketfft, the ancestor of ducc0.fft) with all host cores.
"""

import json
import os
from dataclasses import dataclass, asdict

import numpy as np
import scipy.fft as sfft

from ._lib import cint, f64, i64, lib, ptr

SPEED_OF_LIGHT = 299792458.0
_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLE = None
TILE = 32
FFT_WORKERS = -1  # scipy.fft workers (-1 = all cores); bench sets it to its CPU share


def good_size(n, real=False):
    """Smallest 2-3-5-7-11-smooth (complex) or 2-3-5-smooth (real) integer >= n.

    Mirrors the contract of ``ducc0.fft.good_size`` used by
    /root/reference/src/pfb_imaging/utils/misc.py:921-951.
    """
    primes = (2, 3, 5) if real else (2, 3, 5, 7, 11)
    n = max(int(n), 1)
    while True:
        m = n
        for p in primes:
            while m % p == 0:
                m //= p
        if m == 1:
            return n
        n += 1


def _good_size_2357(n):
    n = max(int(n), 1)
    while True:
        m = n
        for p in (2, 3, 5, 7):
            while m % p == 0:
                m //= p
        if m == 1:
            return n
        n += 1


def grid_size(npix, sigma):
    """Even 2-3-5-7-smooth oversampled grid size >= sigma*npix (and >= 32)."""
    return max(2 * _good_size_2357(int(np.ceil(0.5 * sigma * npix - 1e-9))), 32)


def kernel_table():
    global _TABLE
    if _TABLE is None:
        with open(os.path.join(_HERE, "es_kernel_table.json")) as f:
            _TABLE = json.load(f)["rows"]
    return _TABLE


_GLX, _GLW = np.polynomial.legendre.leggauss(96)
_GS = 0.5 * (_GLX + 1.0)
_GW = 0.5 * _GLW


def kernel_ft(v, W, beta):
    """psi(v) = W * int_0^1 phi(s) cos(pi W v s) ds  (Fourier transform of the width-W kernel)."""
    v = np.asarray(v, dtype=np.float64)
    phi = np.exp(beta * (np.sqrt(1.0 - _GS * _GS) - 1.0)) * _GW
    out = np.zeros(v.shape, dtype=np.float64)
    for s, p in zip(_GS, phi):
        out += p * np.cos((np.pi * W * s) * v)
    return W * out


def kernel_poly_degree(W):
    return 12


def kernel_poly_table(W, beta):
    """Piecewise-polynomial form of the kernel (the form the gridding kernels evaluate; the
    reference's gridder does the same with its own kernels): for tap a and sub-cell offset
    f in [0,1), phi((a + 1 - W/2 - f) 2/W) ~= sum_k c[a,k] z^k, z = 2f - 1, degree 12,
    by Chebyshev interpolation in extended precision."""
    D = kernel_poly_degree(W)
    n = D + 1
    ld = np.longdouble
    k = np.arange(n, dtype=ld)
    znodes = np.cos(ld(np.pi) * (k + ld(0.5)) / n)
    tab = np.zeros((W, n))
    for a in range(W):
        f = ld(0.5) * (znodes + 1)
        x = (ld(a) + 1 - ld(0.5) * W - f) * 2 / W
        t = np.maximum(1 - x * x, ld(0))
        fv = np.exp(ld(beta) * (np.sqrt(t) - 1))
        cc = np.array([2 * np.sum(fv * np.cos(ld(np.pi) * j * (k + ld(0.5)) / n)) / n for j in range(n)], dtype=ld)
        cc[0] *= ld(0.5)
        mono = np.zeros(n, dtype=ld)
        t0 = np.zeros(n, dtype=ld)
        t1 = np.zeros(n, dtype=ld)
        t0[0] = 1
        t1[1] = 1
        mono += cc[0] * t0 + cc[1] * t1
        for j in range(2, n):
            t2 = -t0.copy()
            t2[1:] += 2 * t1[:-1]
            mono += cc[j] * t2
            t0, t1 = t1, t2
        tab[a] = mono.astype(np.float64)
    return np.ascontiguousarray(tab)


@dataclass
class GridParams:
    nu: int
    nv: int
    W: int
    beta: float
    sigma: float
    nplanes: int
    wmin: float
    dw: float
    nshift: float
    lshift: float
    mshift: float
    tile: int = TILE
    wmode: int = 0        # 0: ES-kernel w-planes (equispaced), 1: polynomial (Chebyshev-node) w-planes,
                          # 2: ONE plane, the w-term folded into differentiated gridding kernels (see Plan._init_wd)
    wcenter: float = 0.0  # wmode 1, 2: centre of the w range
    whalf: float = 0.0    # wmode 1, 2: half-width of the w range
    nderiv: int = 0       # wmode 2: number K of kernel functions per axis (phi, phi'', ... phi^(2K-2))

    def asdict(self):
        return asdict(self)


def nm1_image(nx, ny, px, py, lshift, mshift):
    x = lshift + (np.arange(nx) - nx // 2) * px
    y = mshift + (np.arange(ny) - ny // 2) * py
    r2 = x[:, None] ** 2 + y[None, :] ** 2
    out = np.empty_like(r2)
    ok = r2 <= 1.0
    out[ok] = -r2[ok] / (1.0 + np.sqrt(1.0 - r2[ok]))
    out[~ok] = -np.sqrt(r2[~ok] - 1.0) - 1.0
    return out


def nm1_range(nx, ny, px, py, lshift, mshift):
    """min/max of n-1 over the image (corners, plus axis crossings)."""
    x0 = lshift - (nx // 2) * px  # pixel i sits at (i - n // 2) * pixsize (odd sizes: symmetric lattice)
    y0 = mshift - (ny // 2) * py
    xs = [x0, x0 + (nx - 1) * px]
    ys = [y0, y0 + (ny - 1) * py]
    if xs[0] * xs[1] < 0:
        xs.append(0.0)
    if ys[0] * ys[1] < 0:
        ys.append(0.0)
    vals = []
    for xc in xs:
        for yc in ys:
            t = xc * xc + yc * yc
            vals.append(-t / (1.0 + np.sqrt(1.0 - t)) if t <= 1.0 else -np.sqrt(t - 1.0) - 1.0)
    return min(vals), max(vals)


def w_range(uvw, freq, mask, flip_w):
    """min/max of |w| f/c over unmasked visibilities (after the Hermitian fold w >= 0)."""
    aw = np.abs(uvw[:, 2])
    fc = freq / SPEED_OF_LIGHT
    if mask is None:
        if aw.size == 0:
            return 0.0, 0.0
        return float(aw.min() * fc.min()), float(aw.max() * fc.max())
    m = mask != 0
    if not m.any():
        return 0.0, 0.0
    prod = aw[:, None] * fc[None, :]
    return float(prod[m].min()), float(prod[m].max())


def choose_params(uvw, freq, mask, nx, ny, px, py, center_x, center_y, epsilon, do_wgridding, flip_u=False,
                  flip_v=False, flip_w=False, sigma_min=1.1, sigma_max=2.6, force=None, force_wmode=None):
    """Pick (sigma, W, beta), grid size and w-plane layout.

    ``force=(sigma, W)`` pins the kernel row (tests use it to mirror the product's
    choice).  Otherwise a CPU cost model in the spirit of the reference's
    gridder picks the cheapest admissible row.
    """
    lshift = -center_x if flip_u else center_x
    mshift = -center_y if flip_v else center_y
    nm1min, nm1max = nm1_range(nx, ny, px, py, lshift, mshift)
    nshift = -0.5 * (nm1max + nm1min) if do_wgridding else 0.0
    tmax = max(abs(nm1max + nshift), abs(nm1min + nshift))
    wlo, whi = w_range(uvw, freq, mask, flip_w) if do_wgridding else (0.0, 0.0)
    nvis = uvw.shape[0] * freq.size
    # (admissibility on the worst-sub-cell-position error with the product's margins, see choose_kernel in csrc/gridder.hip)
    eps_w = epsilon / 3.0
    eps1 = 0.8 / 1.25 * epsilon / (3.0 if do_wgridding else 2.0)
    best = None
    for r in kernel_table():
        if force is not None:
            if not (abs(r["sigma"] - force[0]) < 1e-9 and r["W"] == force[1]):
                continue
        else:
            if r["sigma"] < sigma_min - 1e-9 or r["sigma"] > sigma_max + 1e-9 or r.get("eps_sup", r["eps_max"]) > eps1:
                continue
        nu, nv = grid_size(nx, r["sigma"]), grid_size(ny, r["sigma"])
        # (rounding amplified by the image-side correction: the product's third budget term, see choose_kernel)
        f0 = float(kernel_ft(np.array([0.0]), r["W"], r["beta"])[0])
        amp2 = (f0 / float(kernel_ft(np.array([0.5 * nx / nu]), r["W"], r["beta"])[0])) * \
               (f0 / float(kernel_ft(np.array([0.5 * ny / nv]), r["W"], r["beta"])[0]))
        ampw = f0 / float(kernel_ft(np.array([0.5 / r["sigma"]]), r["W"], r["beta"])[0])
        for wmode in ((0, 1, 2) if (do_wgridding and tmax > 0 and force_wmode is None) else
                      ((force_wmode,) if (do_wgridding and tmax > 0) else (0,))):
            if force is None and 2.5e-19 * amp2 * (max(ampw, 1.0) if (do_wgridding and tmax > 0 and wmode == 0) else 1.0) > 0.2 * epsilon:
                continue
            nder = 0
            if do_wgridding and tmax > 0:
                if wmode == 0:
                    dw = 0.5 / r["sigma"] / tmax
                    npl = int((whi - wlo) / dw + r["W"])
                    touched = r["W"]
                else:
                    dw = 1.0
                    omega = 2.0 * np.pi * 0.5 * (whi - wlo) * tmax
                    npl = cheb_planes_needed(omega, 2.0 * eps_w)
                    if npl is None:
                        continue
                    touched = npl
                    if wmode == 2:
                        # one plane with K = npl kernel functions per axis: phase centre on axis only (t a function of
                        # l^2 + m^2), 2 <= K <= 4, and the aliases of the differentiated kernels -- amplified by
                        # ((1 + 2 sigma) l_max)^(2k) against l_max^(2k) -- still inside the row's share
                        if lshift != 0.0 or mshift != 0.0 or not 2 <= npl <= WD_MAX_K:
                            continue
                        if force is None and r.get("eps_sup", r["eps_max"]) * wd_alias_amplification(omega, npl, nu / nx, nv / ny) > eps1:
                            continue
                        nder, npl = npl, 1
            else:
                dw, npl, touched = 1.0, 1, 1
            fftcost = 2.5e-9 * npl * nu * nv * np.log2(nu * nv) / 8.0
            gridcost = 1.2e-9 * nvis * r["W"] ** 2 * touched / 8.0
            cost = fftcost + gridcost
            if best is None or cost < best[0]:
                best = (cost, r, nu, nv, dw, npl, wmode, nder)
    if best is None:
        raise ValueError(f"no ES kernel reaches epsilon={epsilon} within sigma in [{sigma_min},{sigma_max}]")
    _, r, nu, nv, dw, npl, wmode, nder = best
    wmin = 0.5 * (wlo + whi) - 0.5 * (npl - 1) * dw if (do_wgridding and wmode == 0) else 0.0
    return GridParams(nu=nu, nv=nv, W=r["W"], beta=r["beta"], sigma=r["sigma"], nplanes=npl, wmin=wmin, dw=dw,
                      nshift=nshift, lshift=lshift, mshift=mshift, wmode=wmode, wcenter=0.5 * (wlo + whi),
                      whalf=0.5 * (whi - wlo), nderiv=nder)


WD_MAX_K = 4


def wd_alias_amplification(omega, K, sig_u, sig_v):
    """wmode 2: factor by which the k-th correction term's aliases exceed the plain kernel's, summed over the terms:
    the alias of l^(2k) psi(l) at l + L (L = the grid's period, 2 sigma l_max) carries (l + L)^(2k) where the wanted term has
    l^(2k) <= l_max^(2k), and the term itself is of size omega^k / k!."""
    from math import factorial

    g = max(1.0 + 2.0 * sig_u, 1.0 + 2.0 * sig_v) ** 2
    return float(sum((omega * g) ** k / factorial(k) for k in range(K)))


def wd_smax(nx, ny, px, py):
    """Largest l^2 + m^2 over the pixel lattice of an on-axis image (pixel i sits at (i - n // 2) pixsize)."""
    return ((nx // 2) * px) ** 2 + ((ny // 2) * py) ** 2


def wd_nodes(K, smax):
    """Chebyshev nodes of the first kind on [0, smax], ascending."""
    return 0.5 * smax * (1.0 - np.cos(np.pi * (2.0 * np.arange(K) + 1.0) / (2.0 * K)))


def wd_matrix(K):
    """M[k, q]: coefficient of (s / smax)^k in the Lagrange basis polynomial of node q (nodes of wd_nodes)."""
    return np.ascontiguousarray(np.linalg.inv(np.vander(wd_nodes(K, 1.0), K, increasing=True)))


def wd_tables(ktab, W, K, alpha):
    """(K, W, D + 1): the 2k-th derivative of the kernel's piecewise polynomial (x = (a + 1 - W/2 - (z + 1)/2) 2/W, so
    d^2/dx^2 = W^2 d^2/dz^2), times (-alpha)^k."""
    D1 = ktab.shape[1]
    out = np.zeros((K, W, D1))
    cur = ktab.copy()
    out[0] = cur
    for k in range(1, K):
        nxt = np.zeros_like(cur)
        for j in range(D1 - 2):
            nxt[:, j] = cur[:, j + 2] * ((j + 2) * (j + 1))
        cur = nxt * (W * W)
        out[k] = cur * (-alpha) ** k
    return np.ascontiguousarray(out)


MAX_CHEB_PLANES = 24


def cheb_planes_needed(omega, eps):
    """Smallest K with interpolation error bound omega^K / (2^(K-1) K!) <= eps for exp(i omega s),
    |s| <= 1, interpolated at K Chebyshev nodes (None if K would exceed MAX_CHEB_PLANES)."""
    if omega <= 0.0:
        return 1
    bound = omega  # K = 1: omega^1 / (2^0 1!)
    k = 1
    while bound > eps:
        k += 1
        bound *= omega / (2.0 * k)
        if k > MAX_CHEB_PLANES:
            return None
    return k


def cheb_nodes(k):
    """Chebyshev nodes of the first kind on [-1, 1], ascending."""
    return -np.cos(np.pi * (2.0 * np.arange(k) + 1.0) / (2.0 * k))


class Plan:
    """Everything that depends only on geometry + uvw/freq/mask (cf. the pinned per-band
    inputs of /root/reference/src/pfb_imaging/operators/band_worker.py:61-106)."""

    def __init__(self, uvw, freq, mask, npix_x, npix_y, pixsize_x, pixsize_y, center_x=0.0, center_y=0.0,
                 epsilon=1e-7, flip_u=False, flip_v=False, flip_w=False, do_wgridding=True, divide_by_n=True,
                 sigma_min=1.1, sigma_max=2.6, params=None, force=None, force_wmode=None, use_poly_kernel=True):
        self.uvw = np.ascontiguousarray(uvw, dtype=np.float64)
        self.freq = np.ascontiguousarray(freq, dtype=np.float64)
        self.mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.nrow, self.nchan = self.uvw.shape[0], self.freq.size
        self.nx, self.ny = int(npix_x), int(npix_y)
        self.px, self.py = float(pixsize_x), float(pixsize_y)
        self.do_w = bool(do_wgridding)
        self.divide_by_n = bool(divide_by_n)
        self.signs = (-1.0 if flip_u else 1.0, -1.0 if flip_v else 1.0, -1.0 if flip_w else 1.0)
        if params is None:
            params = choose_params(self.uvw, self.freq, self.mask, self.nx, self.ny, self.px, self.py, center_x,
                                   center_y, epsilon, self.do_w, flip_u, flip_v, flip_w, sigma_min, sigma_max, force,
                                   force_wmode)
        elif isinstance(params, dict):
            params = GridParams(**params)
        self.p = p = params
        n = self.nrow * self.nchan
        self.fc = self.freq / SPEED_OF_LIGHT
        self.pu = np.empty(n)
        self.pv = np.empty(n)
        self.pw = np.empty(n)
        self.uvw_l = np.empty((n, 3))
        self.flip = np.empty(n, dtype=np.uint8)
        self.iu0 = np.empty(n, dtype=np.int32)
        self.iv0 = np.empty(n, dtype=np.int32)
        self.p0 = np.empty(n, dtype=np.int32)
        if p.wmode == 0:
            wmin_map, xdw = p.wmin, 1.0 / p.dw
            self.wplanes = p.wmin + p.dw * np.arange(p.nplanes)
            self.nodes = None
        elif p.wmode == 1:
            wmin_map, xdw = p.wcenter, (1.0 / p.whalf if p.whalf > 0 else 0.0)
            self.nodes = cheb_nodes(p.nplanes)
            self.wplanes = p.wcenter + p.whalf * self.nodes
        else:
            wmin_map, xdw = p.wcenter, (1.0 / p.whalf if p.whalf > 0 else 0.0)
            self.nodes = None
            self.wplanes = np.array([p.wcenter])
        lib().pfbo_vismap(i64(self.nrow), i64(self.nchan), ptr(self.uvw), ptr(self.fc), ptr(self.mask),
                          f64(self.signs[0]), f64(self.signs[1]), f64(self.signs[2]), f64(self.px), f64(self.py),
                          i64(p.nu), i64(p.nv), cint(p.W), cint(int(self.do_w)), f64(wmin_map), f64(xdw), ptr(self.pu),
                          ptr(self.pv), ptr(self.pw), ptr(self.uvw_l), ptr(self.flip), ptr(self.iu0), ptr(self.iv0),
                          ptr(self.p0))
        # tile sort (tile of the first tap, wrapped)
        T = p.tile
        self.ntu, self.ntv = -(-p.nu // T), -(-p.nv // T)
        self.tile_id = (np.mod(self.iu0, p.nu) // T).astype(np.int64) * self.ntv + (np.mod(self.iv0, p.nv) // T)
        active = np.ones(n, dtype=bool) if self.mask is None else (self.mask.ravel() != 0)
        self.active = active
        idx = np.flatnonzero(active)
        order = idx[np.argsort(self.tile_id[idx], kind="stable")]
        self.order = np.ascontiguousarray(order, dtype=np.int64)
        counts = np.bincount(self.tile_id[idx], minlength=self.ntu * self.ntv)
        self.tstart = np.zeros(self.ntu * self.ntv + 1, dtype=np.int64)
        np.cumsum(counts, out=self.tstart[1:])
        # image-domain quantities
        self.xi = np.mod(np.arange(self.nx) - self.nx // 2, p.nu)
        self.yi = np.mod(np.arange(self.ny) - self.ny // 2, p.nv)
        self.ktab = kernel_poly_table(p.W, p.beta) if use_poly_kernel else None
        self.kdeg = kernel_poly_degree(p.W)
        self._corr = None
        self.t = nm1_image(self.nx, self.ny, self.px, self.py, p.lshift, p.mshift) + p.nshift if self.do_w else None
        if self.do_w and p.wmode == 2:
            self._init_wd()
        self.shifting = (p.lshift != 0.0) or (p.mshift != 0.0) or (p.nshift != 0.0)
        if self.shifting:
            ph = self.uvw_l[:, 0] * p.lshift + self.uvw_l[:, 1] * p.mshift + self.uvw_l[:, 2] * p.nshift
            ph -= np.rint(ph)
            self.phase = np.exp(2j * np.pi * ph)
        else:
            self.phase = None

    def _init_wd(self):
        """wmode 2 (one plane; see pfb_oracle.c: pfbo_grid_plane_wd): the K complex coefficients C_k of every visibility and
        the scaled derivative tables of the kernel polynomial.

        exp(-2 pi i w t(s)) = exp(-2 pi i wc t(s)) E(dw; s), s = l^2 + m^2, dw = w - wc = pw * whalf; E is interpolated in s
        at K Chebyshev nodes of [0, smax]: E(dw; s) ~ sum_k C_k(dw) (s / smax)^k, C_k = sum_q M[k, q] E(dw; s_q); and
        (s / smax)^k in the image is D^k on the gridding kernel, D = -(alpha_u d^2/dx^2 + alpha_v d^2/dy^2) / smax,
        alpha_u = (nu px / (pi W))^2."""
        p = self.p
        assert p.lshift == 0.0 and p.mshift == 0.0, "wmode 2 needs the phase centre on axis"
        K = p.nderiv
        self.smax = wd_smax(self.nx, self.ny, self.px, self.py)
        sq = wd_nodes(K, self.smax)
        tq = -sq / (1.0 + np.sqrt(1.0 - sq)) + p.nshift
        M = wd_matrix(K)
        dw = self.pw * p.whalf
        ph = dw[:, None] * tq[None, :]
        ph -= np.rint(ph)
        self.cw = np.ascontiguousarray(np.exp(-2j * np.pi * ph) @ M.T)          # (n, K), gridding direction
        au = (p.nu * self.px / (np.pi * p.W)) ** 2 / self.smax
        av = (p.nv * self.py / (np.pi * p.W)) ** 2 / self.smax
        self.dtab_u = wd_tables(self.ktab, p.W, K, au)
        self.dtab_v = wd_tables(self.ktab, p.W, K, av)

    # -- helpers -------------------------------------------------------
    @property
    def corr(self):
        """Correction image 1/(psi_l psi_m psi_n) [/n], built on first use."""
        if self._corr is None:
            p = self.p
            cfu = 1.0 / kernel_ft((np.arange(self.nx) - self.nx // 2) / p.nu, p.W, p.beta)
            cfv = 1.0 / kernel_ft((np.arange(self.ny) - self.ny // 2) / p.nv, p.W, p.beta)
            corr = cfu[:, None] * cfv[None, :]
            if self.do_w and p.wmode == 0:
                z = self.t * p.dw
                zmax = float(np.abs(z).max())
                if zmax > 0 and z.size > 1_000_000:
                    # large images: Chebyshev interpolant of 1/psi in (z/zmax)^2 (error < 1e-15)
                    cheb = np.polynomial.chebyshev.Chebyshev.interpolate(
                        lambda y: 1.0 / kernel_ft(zmax * np.sqrt(0.5 * (y + 1.0)), p.W, p.beta), 64)
                    corr = corr * cheb(2.0 * (z / zmax) ** 2 - 1.0)
                else:
                    corr = corr / kernel_ft(z, p.W, p.beta)
            if self.do_w and self.divide_by_n:
                corr = corr / (self.t - p.nshift + 1.0)
            self._corr = corr
        return self._corr

    def plane_round_trip(self, dc, swgt_flat, plane, acc_img, sacc):
        """One w-plane of an exact-Hessian apply (bench cpu_baseline sample): pad+screen -> FFT ->
        gather into sacc; then scatter sacc*wgt -> inverse FFT -> crop+screen -> acc_img."""
        p = self.p
        grid = np.zeros((p.nu, p.nv), dtype=np.complex128)
        grid[np.ix_(self.xi, self.yi)] = dc * self._screen(plane, +1.0) if self.do_w else dc
        grid = sfft.fft2(grid, workers=FFT_WORKERS, overwrite_x=True)
        self._degrid(grid, plane, sacc)
        sval = sacc * swgt_flat
        grid = self.grid_plane(sval, plane)
        img = sfft.ifft2(grid, norm="forward", workers=FFT_WORKERS, overwrite_x=True)
        sub = img[np.ix_(self.xi, self.yi)]
        if self.do_w:
            sub *= self._screen(plane, -1.0)
        acc_img += sub.real

    def plane_weights(self, plane):
        """Per-visibility weight of w-plane `plane` (0 where the visibility does not touch it)."""
        p = self.p
        n = self.pw.size
        if not self.do_w:
            return np.ones(n)
        if p.wmode == 0:
            dp = plane - self.p0
            ok = (dp >= 0) & (dp < p.W)
            if self.ktab is not None:
                z = 2.0 * ((self.pw + (1.0 - 0.5 * p.W)) - self.p0) - 1.0
                c = self.ktab[np.clip(dp, 0, p.W - 1)]
                kw = c[:, self.kdeg].copy()
                for k in range(self.kdeg - 1, -1, -1):
                    kw = kw * z + c[:, k]
            else:
                x = (plane - self.pw) * (2.0 / p.W)
                kw = np.exp(p.beta * (np.sqrt(np.maximum(1.0 - x * x, 0.0)) - 1.0))
            kw[~ok] = 0.0
            return kw
        # Lagrange basis polynomial of node `plane` evaluated at s = pw
        kw = np.ones(n)
        for m, sm in enumerate(self.nodes):
            if m != plane:
                kw *= (self.pw - sm) / (self.nodes[plane] - sm)
        kw[kw == 0.0] = 1e-300  # keep "touches this plane" semantics (exact zeros are measure-zero)
        return kw

    def _screen(self, plane, sign):
        w = self.wplanes[plane]
        ph = w * self.t
        ph -= np.rint(ph)
        return np.exp((sign * 2j * np.pi) * ph)

    def _degrid(self, grid, plane, acc):
        p = self.p
        if self.do_w and p.wmode == 2:
            lib().pfbo_degrid_plane_wd(i64(self.ntu * self.ntv), ptr(self.tstart), ptr(self.order), ptr(self.pu),
                                       ptr(self.pv), ptr(self.iu0), ptr(self.iv0), ptr(self.cw.view(np.float64)),
                                       cint(p.nderiv), cint(p.W), ptr(self.dtab_u), ptr(self.dtab_v), cint(self.kdeg),
                                       i64(p.nu), i64(p.nv), cint(p.tile), ptr(grid.view(np.float64)),
                                       ptr(acc.view(np.float64)))
            return
        kwv = np.ascontiguousarray(self.plane_weights(plane))
        lib().pfbo_degrid_plane(i64(self.ntu * self.ntv), ptr(self.tstart), ptr(self.order), ptr(self.pu),
                                ptr(self.pv), ptr(self.iu0), ptr(self.iv0), ptr(kwv), cint(p.W), f64(p.beta),
                                ptr(self.ktab), cint(self.kdeg), i64(p.nu), i64(p.nv), cint(p.tile),
                                ptr(grid.view(np.float64)), ptr(acc.view(np.float64)))

    def grid_plane(self, sval, plane):
        """Scatter one w-plane (pre-FFT grid), exposed for intermediate parity tests."""
        p = self.p
        grid = np.zeros((p.nu, p.nv), dtype=np.complex128)
        if self.do_w and p.wmode == 2:
            lib().pfbo_grid_plane_wd(i64(self.ntu * self.ntv), ptr(self.tstart), ptr(self.order), ptr(self.pu),
                                     ptr(self.pv), ptr(self.iu0), ptr(self.iv0), ptr(self.cw.view(np.float64)),
                                     ptr(sval.view(np.float64)), cint(p.nderiv), cint(p.W), ptr(self.dtab_u),
                                     ptr(self.dtab_v), cint(self.kdeg), i64(p.nu), i64(p.nv), cint(p.tile),
                                     ptr(grid.view(np.float64)))
            return grid
        kwv = np.ascontiguousarray(self.plane_weights(plane))
        lib().pfbo_grid_plane(i64(self.ntu * self.ntv), ptr(self.tstart), ptr(self.order), ptr(self.pu), ptr(self.pv),
                              ptr(self.iu0), ptr(self.iv0), ptr(kwv), ptr(sval.view(np.float64)), cint(p.W),
                              f64(p.beta), ptr(self.ktab), cint(self.kdeg), i64(p.nu), i64(p.nv), cint(p.tile),
                              ptr(grid.view(np.float64)))
        return grid

    def prep_vis(self, vis, wgt):
        vis = np.ascontiguousarray(vis, dtype=np.complex128).reshape(-1)
        sval = vis.copy()
        if wgt is not None:
            sval *= np.ascontiguousarray(wgt, dtype=np.float64).reshape(-1)
        fl = self.flip != 0
        sval[fl] = np.conj(sval[fl])
        if self.phase is not None:
            sval *= self.phase
        sval[~self.active] = 0.0
        return np.ascontiguousarray(sval)

    # -- operators -----------------------------------------------------
    def vis2dirty(self, vis, wgt=None):
        p = self.p
        sval = self.prep_vis(vis, wgt)
        acc = np.zeros((self.nx, self.ny), dtype=np.float64)
        for plane in range(p.nplanes):
            grid = self.grid_plane(sval, plane)
            img = sfft.ifft2(grid, norm="forward", workers=FFT_WORKERS, overwrite_x=True)
            sub = img[np.ix_(self.xi, self.yi)]
            if self.do_w:
                sub *= self._screen(plane, -1.0)
            acc += sub.real
        return acc * self.corr

    def dirty2vis(self, dirty, wgt=None):
        p = self.p
        dc = np.ascontiguousarray(dirty, dtype=np.float64) * self.corr
        n = self.nrow * self.nchan
        acc = np.zeros(n, dtype=np.complex128)
        for plane in range(p.nplanes):
            grid = np.zeros((p.nu, p.nv), dtype=np.complex128)
            grid[np.ix_(self.xi, self.yi)] = dc * self._screen(plane, +1.0) if self.do_w else dc
            grid = sfft.fft2(grid, workers=FFT_WORKERS, overwrite_x=True)
            self._degrid(grid, plane, acc)
        if self.phase is not None:
            acc *= np.conj(self.phase)
        fl = self.flip != 0
        acc[fl] = np.conj(acc[fl])
        acc[~self.active] = 0.0
        if wgt is not None:
            acc *= np.ascontiguousarray(wgt, dtype=np.float64).reshape(-1)
        return acc.reshape(self.nrow, self.nchan)


def vis2dirty(*, uvw, freq, vis, wgt=None, mask=None, npix_x, npix_y, pixsize_x, pixsize_y, center_x=0.0,
              center_y=0.0, epsilon, flip_u=False, flip_v=False, flip_w=False, do_wgridding, divide_by_n=True,
              nthreads=1, sigma_min=1.1, sigma_max=2.6, double_precision_accumulation=False, verbosity=0,
              dirty=None, params=None, force=None, force_wmode=None):
    """Keyword-compatible with ducc0.wgridder.experimental.vis2dirty as called at
    /root/reference/src/pfb_imaging/operators/gridder.py:590-613."""
    plan = Plan(uvw, freq, mask, npix_x, npix_y, pixsize_x, pixsize_y, center_x, center_y, epsilon, flip_u, flip_v,
                flip_w, do_wgridding, divide_by_n, sigma_min, sigma_max, params, force, force_wmode)
    out = plan.vis2dirty(vis, wgt)
    if dirty is not None:
        dirty[...] = out
        return dirty
    return out


def dirty2vis(*, uvw, freq, dirty, wgt=None, mask=None, pixsize_x, pixsize_y, center_x=0.0, center_y=0.0, epsilon,
              flip_u=False, flip_v=False, flip_w=False, do_wgridding, divide_by_n=True, nthreads=1, sigma_min=1.1,
              sigma_max=2.6, verbosity=0, vis=None, params=None, force=None, force_wmode=None):
    """Keyword-compatible with ducc0.wgridder.experimental.dirty2vis as called at
    /root/reference/src/pfb_imaging/operators/hessian.py:50-66."""
    nx, ny = dirty.shape
    plan = Plan(uvw, freq, mask, nx, ny, pixsize_x, pixsize_y, center_x, center_y, epsilon, flip_u, flip_v, flip_w,
                do_wgridding, divide_by_n, sigma_min, sigma_max, params, force, force_wmode)
    out = plan.dirty2vis(dirty, wgt)
    if vis is not None:
        vis[...] = out
        return vis
    return out
