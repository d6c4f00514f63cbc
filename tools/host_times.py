#!/usr/bin/env python3
"""dev helper (GPU box): wall-clock of the host-side steps around the kernels at benchmark sizes -- handle creation (twice: the
second one finds the first one's device blocks in the cache), set_weights, one host-buffer call each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd import _lib
from pfb_imaging_amd.operators.hessian import HessPSF
from pfb_imaging_amd.operators.psi import PsiNocopyt
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.wgridder import Gridder


def t(label, f, n=1):
    _lib.check(_lib.lib().pfbhip_synchronize())
    t0 = time.perf_counter()
    for _ in range(n):
        out = f()
    _lib.check(_lib.lib().pfbhip_synchronize())
    print(f"{label:44s} {1e3 * (time.perf_counter() - t0) / n:9.1f} ms", flush=True)
    return out


rng = np.random.default_rng(0)
c = synth.make_case(1250000, 8, 8192, seed=0)
kw = dict(npix_x=8192, npix_y=8192, pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0, epsilon=1e-7, flip_u=False,
          flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False)
g = t("gridder plan C2 (first)", lambda: Gridder(c["uvw"], c["freq"], c["mask"], **kw))
g.close()
g = t("gridder plan C2 (second)", lambda: Gridder(c["uvw"], c["freq"], c["mask"], **kw))
t("set_weights", lambda: g.set_weights(c["wgt"]))
t("set_weights (again)", lambda: g.set_weights(c["wgt"]))
x = c["x"]
t("hessian, numpy in / out (first call: pinned staging is allocated)", lambda: g.hessian(x))
t("hessian, numpy in / out", lambda: g.hessian(x), 2)
t("vis2dirty, numpy in / out", lambda: g.vis2dirty(c["vis"], c["wgt"]), 2)
t("dirty2vis, numpy in / out", lambda: g.dirty2vis(x), 2)
g.close()

nband, nx, nxp = 4, 4096, 8192
psf = np.zeros((nband, nxp, nxp))
psf[:, 0, 0] = 1.0
psf += 0.01 * rng.standard_normal(psf.shape)
abspsf = t("abs(rfft2(psf)) on the host (numpy, 4 x 8192^2)", lambda: np.abs(np.fft.rfft2(psf, axes=(1, 2))))
h = t("HessPSF create (4 bands 4096^2 / 8192^2) first", lambda: HessPSF(nx, nx, abspsf, beam=None, eta=0.1))
xb = rng.standard_normal((nband, nx, nx))
t("HessPSF.dot, numpy in / out", lambda: h.dot(xb), 2)
del h
h = t("HessPSF create second", lambda: HessPSF(nx, nx, abspsf, beam=None, eta=0.1))
psi = t("PsiNocopyt create (4 bands, 4 bases, 3 levels)", lambda: PsiNocopyt(nband, nx, nx, ("self", "db1", "db2", "db3"), 3, 1))
