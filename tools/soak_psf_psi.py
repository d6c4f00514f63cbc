#!/usr/bin/env python3
"""dev helper (GPU box): randomised sweep of the PSF convolution (row-FFT pipeline and rocFFT path, odd image sizes inside
padded sizes of every supported form) and of the wavelet dictionary (bases, levels, even sizes) against the numpy oracles.
    python tools/soak_psf_psi.py [seed] [ncases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fftconv, psi as opsi
from pfb_imaging_amd.operators.hessian import HessPSF
from pfb_imaging_amd.operators.psi import PsiNocopyt
rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(seed)
bad = 0
own = [1024, 1280, 1536, 1792, 2048, 2304, 2560, 3072, 3584, 3840]
for k in range(ncases):
    if rng.random() < 0.7:
        nxp, nyp = int(rng.choice(own)), int(rng.choice(own))
    else:
        nxp, nyp = 2 * int(rng.integers(300, 700)), 2 * int(rng.integers(300, 700))
    nx, ny = int(rng.integers(nxp // 3, nxp // 2 + 1)), int(rng.integers(nyp // 3, nyp // 2 + 1))
    nband = int(rng.integers(1, 3))
    psf = rng.standard_normal((nband, nxp, nyp)) * np.exp(-np.linspace(-5, 5, nxp)[None, :, None] ** 2)
    abspsf = np.abs(np.fft.rfft2(np.fft.ifftshift(psf, axes=(1, 2)), axes=(1, 2)))
    x = rng.standard_normal((nband, nx, ny))
    beam = (0.5 + rng.random((nband, nx, ny))) if rng.random() < 0.5 else None
    eta = float(rng.choice([0.0, 0.1]))
    h = HessPSF(nx, ny, abspsf, beam=beam, eta=eta, cgtol=1e-6, cgmaxit=50)
    got = np.array(h.dot(x))
    ref = fftconv.hess_psf_dot(x, abspsf, nyp, beam=beam, eta=eta)
    e = rel(got, ref)
    ok = e < 1e-11
    bad += not ok
    print("psf", k, "OK " if ok else "BAD", (nx, ny, nxp, nyp, nband, eta, beam is not None), "%.1e" % e, flush=True)
for k in range(ncases):
    nx, ny = 2 * int(rng.integers(40, 400)), 2 * int(rng.integers(40, 400))
    pool = ["self", "db1", "db2", "db3", "db4", "db5", "db6", "db7", "db8"]
    bases = tuple(rng.choice(pool, size=int(rng.integers(1, 5)), replace=False))
    nlevel = int(rng.integers(1, 4))
    nband = int(rng.integers(1, 3))
    try:
        o = opsi.Psi(nband, nx, ny, bases, nlevel)
        g = PsiNocopyt(nband, nx, ny, bases, nlevel, nthreads=1)
    except Exception as ex:
        print("psi", k, "skip", (nx, ny, bases, nlevel), type(ex).__name__)
        continue
    x = rng.standard_normal((nband, nx, ny))
    a_ref = np.zeros((nband, o.nbasis, o.nxmax, o.nymax)); o.dot(x, a_ref)
    a = np.full_like(a_ref, np.nan); g.dot(x, a)
    co = rng.standard_normal(a_ref.shape); x_ref = np.zeros_like(x); o.hdot(co, x_ref)
    xo = np.full_like(x, np.nan); g.hdot(co, xo)
    e = (rel(a, a_ref), rel(xo, x_ref))
    ok = e[0] < 1e-13 and e[1] < 1e-13
    bad += not ok
    print("psi", k, "OK " if ok else "BAD", (nx, ny, bases, nlevel, nband), ["%.1e" % q for q in e], flush=True)
print("bad cases:", bad)
