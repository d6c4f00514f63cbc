"""Worker of tests/test_gpu_gridder.py::test_poisoned_device_blocks: run with PFBHIP_DEVCACHE_POISON=1 (every device block handed
to a handle is filled with 0xFF bytes = NaNs first).  Plans, PSF convolution, the wavelet dictionary and the primal-dual scratch are
built twice -- the second time from recycled blocks -- and must give the same results as the first: no kernel may rely on fresh or
recycled memory being zero."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pfb_imaging_amd import _lib  # noqa: E402
from pfb_imaging_amd.operators.hessian import HessPSF  # noqa: E402
from pfb_imaging_amd.operators.psi import PsiNocopyt  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402
from pfb_imaging_amd.wgridder import Gridder  # noqa: E402


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def main():
    assert os.environ.get("PFBHIP_DEVCACHE_POISON") == "1"
    _lib.require_gpu()
    rng = np.random.default_rng(3)
    n = 1100  # blocks of >= 32 MiB are the ones the cache recycles
    results = []
    for widen, center in ((8.0, 0.0), (8.0, 2e-4), (600.0, 0.0)):  # one-plane scheme, polynomial planes (off axis), ES-kernel planes
        c = synth.make_case(2000, 2, 64, zscale=0.3, seed=1)
        cell = c["cell"] * widen * 64.0 / n
        x = rng.standard_normal((n, n))
        kw = dict(npix_x=n, npix_y=n, pixsize_x=cell, pixsize_y=cell, center_x=center, center_y=0.0, epsilon=1e-7, flip_u=False,
                  flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False)
        outs = []
        for rep in range(2):
            g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
            g.set_weights(c["wgt"])
            outs.append((g.vis2dirty(c["vis"], c["wgt"]), g.dirty2vis(x), g.hessian(x, eta=0.1, wsum=3.0), g.info["wmode"]))
            g.close()
        for a, b in zip(outs[0][:3], outs[1][:3]):
            assert np.isfinite(a).all() and rel(b, a) < 1e-12, (widen, center)
        results.append(outs[0][3])
    assert sorted(set(results)) == [0, 1, 2], results  # every w-scheme was exercised
    # PSF-approximate Hessian and the wavelet dictionary
    nx, nxp = 1024, 2048
    psf = np.zeros((1, nxp, nxp))
    psf[0, 0, 0] = 1.0
    psf += 0.01 * rng.standard_normal(psf.shape)
    abspsf = np.abs(np.fft.rfft2(psf, axes=(1, 2)))
    xi = rng.standard_normal((1, nx, nx))
    hs = []
    for rep in range(2):
        h = HessPSF(nx, nx, abspsf, beam=None, eta=0.05)
        hs.append(h.dot(xi).copy())
        del h
    assert np.isfinite(hs[0]).all() and rel(hs[1], hs[0]) < 1e-12
    ps = []
    for rep in range(2):
        psi = PsiNocopyt(1, nx, nx, ("self", "db1", "db3"), 3, 1)
        alpha = np.zeros((1, 3, psi.nxmax, psi.nymax))
        psi.dot(xi, alpha)
        back = np.zeros_like(xi)
        psi.hdot(alpha, back)
        ps.append((alpha.copy(), back.copy()))
        del psi
    assert np.isfinite(ps[0][0]).all() and rel(ps[1][0], ps[0][0]) < 1e-12 and rel(ps[1][1], ps[0][1]) < 1e-12
    assert rel(ps[0][1], 3 * xi) < 1e-12
    print("poison ok", flush=True)


if __name__ == "__main__":
    main()
