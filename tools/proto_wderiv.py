#!/usr/bin/env python3
"""Prototype (numpy) of the single-plane w-correction by DIFFERENTIATED gridding kernels ("wmode 2").

    exp(-2 pi i w t(s)) = exp(-2 pi i wc t) * E(dw; s),   s = l^2 + m^2,  dw = w - wc
    E(dw; s) ~ sum_k C_k(dw) (s / smax)^k          (Chebyshev interpolation in s, J + 1 nodes)
    multiplication by l^2 in the image  <->  -(nu px / (pi W))^2 d^2/dx^2 of the u-kernel phi(x)

so ONE uv-plane with the per-visibility kernel  sum_k C_k D^k[phi(x) phi(y)],  D = -(au d2/dx2 + av d2/dy2) / smax,
replaces the K = J + 1 Chebyshev planes.  Checked here against the direct DFT.
"""
import sys

import numpy as np
import scipy.fft as sfft

sys.path.insert(0, ".")
from oracle import dft  # noqa: E402
from oracle import wgridder as owg  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402


def deriv_tables(ktab, W, nder):
    """ktab[a] = monomials in z of phi((a + 1 - W/2 - f) 2/W), z = 2 f - 1  ->  tables of d^(2k)phi/dx^(2k), k = 0..nder.
    x = (a + 1 - W/2 - (z + 1)/2) 2/W  =>  dx/dz = -1/W  =>  d/dx = -W d/dz (even orders: W^(2k))."""
    tabs = [ktab]
    cur = ktab
    for _ in range(nder):
        d1 = np.polynomial.polynomial.polyder(cur, 2, axis=1) * W * W
        d1 = np.pad(d1, ((0, 0), (0, ktab.shape[1] - d1.shape[1])))
        tabs.append(d1)
        cur = d1
    return tabs


def run(nrow=3000, npix=64, J=2, eps=1e-7, widen=1.0, zscale=1e-3, seed=1):
    c = synth.make_case(nrow, 2, npix, zscale=zscale, seed=seed)
    cell = c["cell"] * widen
    plan = owg.Plan(c["uvw"], c["freq"], c["mask"], npix, npix, cell, cell, 0.0, 0.0, eps, False, True, False, True, False,
                    force_wmode=1)
    p = plan.p
    omega = 2 * np.pi * p.whalf * np.abs(plan.t).max()
    print(f"W={p.W} sigma={p.sigma} nu={p.nu} planes(wmode1)={p.nplanes} omega={omega:.4g} whalf={p.whalf:.4g} tmax={np.abs(plan.t).max():.3g}")
    nx = ny = npix
    x = (np.arange(nx) - nx // 2) * cell
    s_img = x[:, None] ** 2 + x[None, :] ** 2
    smax = s_img.max()
    tfun = lambda s: -s / (1 + np.sqrt(1 - s)) + p.nshift  # noqa: E731
    # Chebyshev nodes in s on [0, smax] and the monomial coefficients of the Lagrange basis in s' = s / smax
    q = np.arange(J + 1)
    sq = 0.5 * smax * (1 - np.cos(np.pi * (2 * q + 1) / (2 * (J + 1))))
    V = np.vander(sq / smax, J + 1, increasing=True)  # V[q, k] = s'_q^k
    M = np.linalg.inv(V)  # coefficient k of l_q: M[k, q]
    dw = plan.pw * p.whalf  # w - wcenter per visibility
    Eq = np.exp(-2j * np.pi * dw[:, None] * tfun(sq)[None, :])  # (n, J+1)
    C = Eq @ M.T  # (n, J+1): C_k
    # check the interpolation error on the image for extreme dw
    for d in (p.whalf, -p.whalf):
        e = np.exp(-2j * np.pi * d * tfun(sq)) @ M.T
        approx = sum(e[k] * (s_img / smax) ** k for k in range(J + 1))
        print("  s-interp err at dw=%+.3g: %.3g" % (d, np.abs(approx - np.exp(-2j * np.pi * d * tfun(s_img))).max()))
    W = p.W
    tabs = deriv_tables(plan.ktab, W, J)
    au = (p.nu * cell / (np.pi * W)) ** 2 / smax
    av = (p.nv * cell / (np.pi * W)) ** 2 / smax
    act = np.flatnonzero(plan.active)
    fu = plan.pu[act] + (1 - 0.5 * W) - plan.iu0[act]
    fv = plan.pv[act] + (1 - 0.5 * W) - plan.iv0[act]
    zu, zv = 2 * fu - 1, 2 * fv - 1
    taps = np.arange(W)

    def kvals(tab, z):  # (n, W)
        out = np.zeros((z.size, W))
        for a in range(W):
            out[:, a] = np.polynomial.polynomial.polyval(z, tab[a], tensor=False)
        return out

    a_k = [(-au) ** k * kvals(tabs[k], zu) for k in range(J + 1)]
    b_k = [(-av) ** k * kvals(tabs[k], zv) for k in range(J + 1)]
    from math import comb

    def kernel2d(Cv):  # Cv (n, J+1) complex -> (n, W, W) complex
        K = np.zeros((act.size, W, W), dtype=complex)
        for k in range(J + 1):
            Dk = sum(comb(k, r) * a_k[r][:, :, None] * b_k[k - r][:, None, :] for r in range(k + 1))
            K += Cv[:, k, None, None] * Dk
        return K

    iu = (plan.iu0[act][:, None] + taps[None, :]) % p.nu
    iv = (plan.iv0[act][:, None] + taps[None, :]) % p.nv
    # ---- vis2dirty ----
    sval = plan.prep_vis(c["vis"], c["wgt"])[act]
    K = kernel2d(C[act])
    grid = np.zeros((p.nu, p.nv), dtype=complex)
    np.add.at(grid, (iu[:, :, None].repeat(W, 2), iv[:, None, :].repeat(W, 1)), sval[:, None, None] * K)
    img = sfft.ifft2(grid, norm="forward")
    sub = img[np.ix_(plan.xi, plan.yi)]
    ph = p.wcenter * plan.t
    sub = sub * np.exp(-2j * np.pi * ph)
    got = sub.real * plan.corr
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], nx, ny, cell, cell, 0, 0, False, True, False, True, False)
    print("  vis2dirty rel l2 err vs DFT: %.3g   (wmode1 restatement: %.3g)" % (
        np.linalg.norm(got - ref) / np.linalg.norm(ref), np.linalg.norm(plan.vis2dirty(c["vis"], c["wgt"]) - ref) / np.linalg.norm(ref)))
    # ---- dirty2vis (adjoint) ----
    dc = c["x"] * plan.corr
    g2 = np.zeros((p.nu, p.nv), dtype=complex)
    g2[np.ix_(plan.xi, plan.yi)] = dc * np.exp(2j * np.pi * ph)
    g2 = sfft.fft2(g2)
    Kc = kernel2d(np.conj(C[act]))
    vals = (g2[iu[:, :, None].repeat(W, 2), iv[:, None, :].repeat(W, 1)] * Kc).sum(axis=(1, 2))
    acc = np.zeros(plan.pu.size, dtype=complex)
    acc[act] = vals
    if plan.phase is not None:
        acc *= np.conj(plan.phase)
    fl = plan.flip != 0
    acc[fl] = np.conj(acc[fl])
    acc = acc.reshape(plan.nrow, plan.nchan)
    refv = dft.dft_dirty2vis(c["uvw"], c["freq"], c["x"], cell, cell, 0, 0, False, True, False, True, False)
    refv[c["mask"] == 0] = 0
    print("  dirty2vis rel l2 err vs DFT: %.3g" % (np.linalg.norm(acc - refv) / np.linalg.norm(refv)))


if __name__ == "__main__":
    run()
    run(widen=3.0, J=2)
    run(widen=3.0, J=3)
    run(widen=6.0, J=3, zscale=3e-3)
