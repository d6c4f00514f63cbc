"""Randomised geometry sweep of the GPU gridder against the direct DFT: odd and rectangular image
sizes, grid sizes that are not multiples of the 32-cell tile, all flip combinations, large centre
offsets, every kernel support from loose to tight epsilon, masked rows, both w-plane schemes.
Tolerance: relative L2 <= epsilon (the accuracy contract) for vis2dirty, dirty2vis and the Hessian."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dft  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


import os  # noqa: E402

# (PFB_FUZZ_SEED / PFB_FUZZ_N: a longer sweep from another seed, for soak runs)
CASES = []
_rng = np.random.default_rng(int(os.environ.get("PFB_FUZZ_SEED", "2024")))
for _i in range(int(os.environ.get("PFB_FUZZ_N", "24"))):
    CASES.append(dict(
        nx=int(_rng.choice([16, 18, 30, 33, 48, 50, 64, 71])), ny=int(_rng.choice([16, 20, 27, 40, 64, 66])),
        nrow=int(_rng.integers(1, 900)), nchan=int(_rng.integers(1, 5)),
        eps=float(_rng.choice([1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9, 1e-10])),
        widen=float(_rng.choice([0.5, 2.0, 10.0, 40.0])), zscale=float(_rng.choice([1e-3, 0.05, 0.5])),
        flips=tuple(bool(v) for v in _rng.integers(0, 2, 3)),
        center=(float(_rng.choice([0.0, 0.01, -0.2])), float(_rng.choice([0.0, -0.03, 0.35]))),
        do_w=bool(_rng.random() > 0.2), divn=bool(_rng.integers(0, 2)), wmode=[None, 0, 1][int(_rng.integers(0, 3))],
        seed=int(_rng.integers(0, 10_000)),
    ))


# Regressions a longer sweep (PFB_FUZZ_SEED=7) found: uv-grids of 36 cells -- a last 32-row block of 4 rows that the footprints
# of the block before it run through -- and odd image sizes with a shifted phase centre (the n - 1 range was half a pixel low).
CASES += [
    dict(nx=30, ny=27, nrow=841, nchan=3, eps=1e-3, widen=0.5, zscale=0.001, flips=(True, False, False), center=(0.01, -0.03),
         do_w=False, divn=True, wmode=0, seed=2779),
    dict(nx=64, ny=27, nrow=295, nchan=1, eps=1e-8, widen=2.0, zscale=0.5, flips=(False, True, False), center=(0.0, -0.03),
         do_w=True, divn=False, wmode=0, seed=2146),
    dict(nx=50, ny=27, nrow=72, nchan=3, eps=1e-5, widen=40.0, zscale=0.5, flips=(False, False, True), center=(0.0, 0.35),
         do_w=True, divn=True, wmode=0, seed=5308),
    dict(nx=64, ny=27, nrow=744, nchan=1, eps=1e-6, widen=40.0, zscale=0.5, flips=(True, False, True), center=(0.01, -0.03),
         do_w=True, divn=True, wmode=0, seed=3165),
]


@pytest.mark.parametrize("k", range(len(CASES)))
def test_fuzz_vs_dft(k):
    from pfb_imaging_amd.wgridder import Gridder

    p = CASES[k]
    c = synth.make_case(p["nrow"], p["nchan"], max(p["nx"], p["ny"]), zscale=p["zscale"], seed=p["seed"])
    cell = c["cell"] * p["widen"]
    # keep the field on the sky: |l|,|m| < 1 with the centre offset
    cell = min(cell, 0.4 / max(p["nx"], p["ny"]))
    fu, fv, fw = p["flips"]
    cx, cy = p["center"]
    nx, ny = p["nx"], p["ny"]
    x = np.random.default_rng(p["seed"]).standard_normal((nx, ny))
    try:
        g = Gridder(c["uvw"], c["freq"], c["mask"], npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell * 1.1, center_x=cx,
                    center_y=cy, epsilon=p["eps"], flip_u=fu, flip_v=fv, flip_w=fw, do_wgridding=p["do_w"],
                    divide_by_n=p["divn"], force_wmode=p["wmode"] if p["do_w"] else None)
    except ValueError as e:
        # a FORCED polynomial w-scheme is not admissible on every wide field (more planes than the scheme has)
        assert p["do_w"] and p["wmode"] == 1 and "force_wmode=1" in str(e), (p, str(e))
        pytest.skip("forced polynomial w-planes not admissible for this geometry")
    args = (cell, cell * 1.1, cx, cy, fu, fv, fw, p["do_w"], p["divn"])
    d = g.vis2dirty(c["vis"], c["wgt"])
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], nx, ny, *args)
    if np.linalg.norm(ref) > 0:
        assert rel(d, ref) < p["eps"], (p, g.info)
    v = g.dirty2vis(x)
    refv = dft.dft_dirty2vis(c["uvw"], c["freq"], x, *args)
    refv[c["mask"] == 0] = 0
    if np.linalg.norm(refv) > 0:
        assert rel(v, refv) < p["eps"], (p, g.info)
    g.set_weights(c["wgt"])
    h = g.hessian(x)
    refh = dft.dft_vis2dirty(c["uvw"], c["freq"], refv, c["wgt"], c["mask"], nx, ny, *args)
    if np.linalg.norm(refh) > 0:
        assert rel(h, refh) < 2 * p["eps"], (p, g.info)
    g.close()


def test_edge_cases_row_fft_paths():
    """Degenerate inputs on the paths that take the hand-written row FFT (grid >= 1024): no unmasked
    visibility, a single visibility, PSF convolution without padding and with a 2-row image."""
    from pfb_imaging_amd.psfconv import PsfConv
    from pfb_imaging_amd.wgridder import Gridder

    rng = np.random.default_rng(0)
    c = synth.make_case(40, 2, 1024, zscale=0.01, seed=2)
    kw = dict(npix_x=1024, npix_y=1024, pixsize_x=c["cell"], pixsize_y=c["cell"], epsilon=1e-6, flip_v=True,
              do_wgridding=True, divide_by_n=False)
    g = Gridder(c["uvw"], c["freq"], np.zeros_like(c["mask"]), **kw)  # everything masked
    assert g.info["nactive"] == 0
    assert not g.vis2dirty(c["vis"], c["wgt"]).any()
    assert not g.dirty2vis(c["x"]).any()
    g.close()
    one = np.zeros_like(c["mask"])
    one[7, 1] = 1
    g = Gridder(c["uvw"], c["freq"], one, **kw)  # a single visibility
    assert g.info["nactive"] == 1 and g.info["fft_mode"] & 3 == 3
    d = g.vis2dirty(c["vis"], c["wgt"])
    from oracle import dft

    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], one, 1024, 1024, c["cell"], c["cell"], 0.0, 0.0, False,
                            True, False, True, False)
    assert np.linalg.norm(d - ref) / np.linalg.norm(ref) < 1e-6
    v = g.dirty2vis(c["x"])
    assert np.count_nonzero(v) == 1
    g.close()
    # PSF convolution: no padding at all (circular convolution), and a 2-row image in a padded plan
    for nx, ny, nxp, nyp in ((1024, 1024, 1024, 1024), (2, 700, 1024, 2048)):
        psfhat = np.fft.rfft2(rng.standard_normal((nxp, nyp)))
        x = rng.standard_normal((nx, ny))
        pc = PsfConv(nx, ny, nxp, nyp)
        pc.set_psfhat(0, psfhat)
        got = pc.apply(x, 0)
        xp = np.zeros((nxp, nyp))
        xp[:nx, :ny] = x
        want = np.fft.irfft2(np.fft.rfft2(xp) * psfhat, s=(nxp, nyp))[:nx, :ny]
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-12
        pc.close()
