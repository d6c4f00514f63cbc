"""CPU oracle of the wavelet dictionary Psi and the l21 dual update -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product
(pfb_imaging_amd) never does.

Restates, in plain numpy:
  * the multi-level 2-D DWT / inverse DWT of /root/reference/src/pfb_imaging/wavelets/wavelets.py:216-343
    (dwt2d_nocopyt / idwt2d_nocopyt: zero-padding mode, x-first packed coefficient layout) built on the 1-D
    kernels of wavelets/convolutions.py:5-122 (downsampling_convolution: out[o] = sum_j f[j] in[2o+1-j]) and
    :125-327 (upsampling_convolution_valid_sf: out[2m+p] += sum_q f[2q+p] c[m + L/2 - 1 - q]);
  * the packing bookkeeping of operators/psi.py:23-142 (_build_wavelet_bookkeeping);
  * PsiBandNocopyt.dot / hdot (operators/psi.py:466-535) and the transposed layout of PsiBand (psi.py:273-345);
  * dual_update_numba_fast (prox/prox_21m.py:105-135), prox_21m (prox_21m.py:5-26), positivity /
    positivity_band (prox/positivity.py:12-33).

PARITY UNPINNED for this module: the reference pins its DWT against PyWavelets (tests/test_wavelets.py:72-132),
which is not installed here and whose outputs are not shipped as fixtures.  The filters are the exact
Daubechies extremal-phase filters (tools/make_wavelet_table.py); PyWavelets' tabulated ones differ from them
by ~1e-13.  What pins this restatement: perfect reconstruction, adjointness, the db1 (Haar) values worked by
hand in tests/test_oracle.py, and the reference's index formulas followed line by line.
"""

import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_FILTERS = None


def filters(name):
    """(dec_lo, dec_hi, rec_lo, rec_hi) of 'db1'..'db8' (pywt.Wavelet(name).filter_bank order)."""
    global _FILTERS
    if _FILTERS is None:
        _FILTERS = json.load(open(os.path.join(_HERE, "wavelet_filters.json")))
    f = _FILTERS[name]
    return tuple(np.array(f[k]) for k in ("dec_lo", "dec_hi", "rec_lo", "rec_hi"))


def coeff_size(n, L):  # wavelets.py:29-31
    return (n + L - 1) // 2


def signal_size(nc, L):  # wavelets.py:34-36
    return 2 * nc - L + 2


def down_conv(x, f, axis):
    """out[o] = sum_j f[j] x[2o+1-j] along `axis`, x zero outside (convolutions.py:5-122)."""
    x = np.moveaxis(x, axis, -1)
    n, L = x.shape[-1], len(f)
    no = coeff_size(n, L)
    xp = np.zeros(x.shape[:-1] + (n + 2 * L,))
    xp[..., L:L + n] = x
    out = np.zeros(x.shape[:-1] + (no,))
    o = np.arange(no)
    for j in range(L):
        out += f[j] * xp[..., 2 * o + 1 - j + L]
    return np.moveaxis(out, -1, axis)


def up_conv(c, f, nout, axis):
    """out[2m+p] = sum_q f[2q+p] c[m + L/2 - 1 - q], 0 <= 2m+p < nout (convolutions.py:125-327)."""
    c = np.moveaxis(c, axis, -1)
    L = len(f)
    h = L // 2
    out = np.zeros(c.shape[:-1] + (nout,))
    m = np.arange((nout + 1) // 2)
    for p in range(2):
        mm = m[2 * m + p < nout]
        acc = np.zeros(c.shape[:-1] + (len(mm),))
        for q in range(h):
            acc += f[2 * q + p] * c[..., mm + h - 1 - q]
        out[..., 2 * mm + p] = acc
    return np.moveaxis(out, -1, axis)


class Bookkeeping:
    """operators/psi.py:23-142 for one image size, list of bases and level count."""

    def __init__(self, nx, ny, bases, nlevel):
        self.nx, self.ny, self.bases, self.nlevel = nx, ny, tuple(bases), nlevel
        self.nbasis = len(bases)
        self.wavelets = [b for b in bases if b != "self"]
        nw = len(self.wavelets)
        self.ix = np.zeros((nw, nlevel, 2), dtype=np.int64)
        self.iy = np.zeros((nw, nlevel, 2), dtype=np.int64)
        self.sx = np.zeros((nw, nlevel), dtype=np.int64)
        self.sy = np.zeros((nw, nlevel), dtype=np.int64)
        self.spx = np.zeros((nw, nlevel), dtype=np.int64)
        self.spy = np.zeros((nw, nlevel), dtype=np.int64)
        self.ntotx = np.zeros(nw, dtype=np.int64)
        self.ntoty = np.zeros(nw, dtype=np.int64)
        nxmax = nymax = 0
        for w, name in enumerate(self.wavelets):
            L = 2 * int(name[-1])
            n_x, n_y = nx, ny
            cxs, cys = [], []
            for k in range(nlevel):
                cx, cy = coeff_size(n_x, L), coeff_size(n_y, L)
                cxs.append(cx)
                cys.append(cy)
                self.sx[w, k], self.sy[w, k] = cx, cy
                self.spx[w, k], self.spy[w, k] = signal_size(cx, L), signal_size(cy, L)
                n_x, n_y = cx + cx % 2, cy + cy % 2
            self.ntotx[w] = sum(cxs) + cxs[-1]
            self.ntoty[w] = sum(cys) + cys[-1]
            nxmax, nymax = max(nxmax, self.ntotx[w]), max(nymax, self.ntoty[w])
            lowx, lowy = cxs[-1], cys[-1]
            self.ix[w, nlevel - 1] = (lowx, 2 * lowx)
            self.iy[w, nlevel - 1] = (lowy, 2 * lowy)
            lowx, lowy = 2 * lowx, 2 * lowy
            for k in reversed(range(nlevel - 1)):
                self.ix[w, k] = (lowx, lowx + cxs[k])
                self.iy[w, k] = (lowy, lowy + cys[k])
                lowx += cxs[k]
                lowy += cys[k]
        self.nxmax, self.nymax = int(max(nxmax, nx)), int(max(nymax, ny))


def dwt2d(x, bk, w):
    """wavelets.py:243-276; returns the (ntotx, ntoty) packed coefficients (x-first)."""
    dec_lo, dec_hi, _, _ = filters(bk.wavelets[w])
    coeffs = np.zeros((bk.ntotx[w], bk.ntoty[w]))
    approx = x
    for i in range(bk.nlevel):
        sx, sy = bk.sx[w, i], bk.sy[w, i]
        hx, hy = bk.ix[w, i, 1], bk.iy[w, i, 1]
        lx, ly = hx - 2 * sx, hy - 2 * sy
        # axis 1 first (wavelets.py:233-236), then axis 0 (:238-240)
        rows = np.concatenate([down_conv(approx, dec_lo, 1), down_conv(approx, dec_hi, 1)], axis=1)
        blk = np.concatenate([down_conv(rows, dec_lo, 0), down_conv(rows, dec_hi, 0)], axis=0)
        coeffs[lx:hx, ly:hy] = blk
        approx = coeffs[lx:lx + sx, ly:ly + sy].copy()
    return coeffs


def idwt2d(coeffs, bk, w):
    """wavelets.py:306-343; returns the (nx, ny) image."""
    _, _, rec_lo, rec_hi = filters(bk.wavelets[w])
    alpha = coeffs.copy()
    image = np.zeros((bk.nx, bk.ny))
    for i in range(bk.nlevel - 1, -1, -1):
        sx, sy = bk.sx[w, i], bk.sy[w, i]
        hx, hy = bk.ix[w, i, 1], bk.iy[w, i, 1]
        lx, ly = hx - 2 * sx, hy - 2 * sy
        nxo, nyo = bk.spx[w, i], bk.spy[w, i]
        if i < bk.nlevel - 1:
            alpha[lx:lx + sx, ly:ly + sy] = image[0:sx, 0:sy]
        blk = alpha[lx:lx + 2 * sx, ly:ly + 2 * sy]
        cb = up_conv(blk[0:sx], rec_lo, nxo, 0) + up_conv(blk[sx:], rec_hi, nxo, 0)     # (nxo, 2 sy)
        image[0:nxo, 0:nyo] = up_conv(cb[:, 0:sy], rec_lo, nyo, 1) + up_conv(cb[:, sy:], rec_hi, nyo, 1)
    return image


class Psi:
    """PsiNocopyt (operators/psi.py:610-665): cubes (nband, nx, ny) <-> (nband, nbasis, nxmax, nymax).
    transposed=True gives the layout of the older Psi (psi.py:551-607): (nband, nbasis, nymax, nxmax)."""

    def __init__(self, nband, nx, ny, bases, nlevel, transposed=False):
        self.bk = Bookkeeping(nx, ny, bases, nlevel)
        self.nband, self.nx, self.ny = nband, nx, ny
        self.nbasis, self.nxmax, self.nymax = self.bk.nbasis, self.bk.nxmax, self.bk.nymax
        self.transposed = transposed

    def dot(self, x, alphao):
        a = np.zeros((self.nband, self.nbasis, self.nxmax, self.nymax))
        for b in range(self.nband):
            w = 0
            for i, name in enumerate(self.bk.bases):
                if name == "self":
                    a[b, i, :self.nx, :self.ny] = x[b]
                else:
                    a[b, i, :self.bk.ntotx[w], :self.bk.ntoty[w]] = dwt2d(x[b], self.bk, w)
                    w += 1
        alphao[...] = a.transpose(0, 1, 3, 2) if self.transposed else a
        return alphao

    def hdot(self, alpha, xo):
        a = alpha.transpose(0, 1, 3, 2) if self.transposed else alpha
        for b in range(self.nband):
            acc = np.zeros((self.nx, self.ny))
            w = 0
            for i, name in enumerate(self.bk.bases):
                if name == "self":
                    acc += a[b, i, :self.nx, :self.ny]
                else:
                    acc += idwt2d(a[b, i, :self.bk.ntotx[w], :self.bk.ntoty[w]], self.bk, w)
                    w += 1
            xo[b] = acc
        return xo


def dual_update(vp, v, lam, sigma, weight):
    """dual_update_numba_fast (prox_21m.py:105-135): v <- vtilde * min(1, lam w / |sum_band vtilde|),
    vtilde = vp + sigma v; in place on v, returns v."""
    vt = vp + sigma * v
    s = np.abs(vt.sum(axis=0))
    thr = lam * weight
    scale = np.where(s > thr, thr / np.where(s > 0, s, 1.0), 1.0)
    v[...] = vt * scale[None]
    return v


def prox_21m(v, sigma, weight=1.0, axis=0):
    """prox_21m.py:5-26."""
    l2 = np.sum(v, axis=axis)
    soft = np.maximum(np.abs(l2) - sigma * weight, 0.0) * np.sign(l2)
    ratio = np.zeros_like(l2)
    m = l2 != 0
    ratio[m] = soft[m] / l2[m]
    return v * np.expand_dims(ratio, axis=axis)


def positivity(x):  # positivity.py:12-19
    x[x < 0.0] = 0.0
    return x


def positivity_band(x):  # positivity.py:22-33
    bad = (x <= 0.0).any(axis=0)
    x[:, bad] = 0.0
    return x
