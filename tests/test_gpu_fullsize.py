"""Full-size parity (BASELINE config C2: 1e7 visibilities, 8192^2 image) through size-independent
properties, as the oracle cannot produce a whole image at this size in seconds:

  * direct-DFT spot checks on random pixels / random visibilities (the DFT of K samples is O(K N));
  * adjointness  <R x, y> == <x, R^H y>;
  * the fused Hessian equals the composition of its two halves, is symmetric and linear;
  * weights / mask semantics (masked visibilities come back exactly zero).

Tolerance: the requested epsilon (1e-7) relative to the RMS of the exact values for the spot checks,
1e-10 relative for the algebraic identities (same arithmetic, different summation order).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dft  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402

EPS = 1e-7


@pytest.fixture(scope="module")
def c2():
    from pfb_imaging_amd.wgridder import Gridder

    c = synth.make_config("C2", band=0)
    g = Gridder(c["uvw"], c["freq"], c["mask"], npix_x=c["nx"], npix_y=c["ny"], pixsize_x=c["cell"], pixsize_y=c["cell"],
                center_x=0.0, center_y=0.0, epsilon=EPS, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True,
                divide_by_n=False, sigma_min=1.1, sigma_max=3.0)
    yield c, g
    g.close()


def test_c2_vis2dirty_spot_check_vs_dft(c2):
    c, g = c2
    dirty = g.vis2dirty(c["vis"], c["wgt"])
    rng = np.random.default_rng(0)
    ix = np.concatenate([rng.integers(0, c["nx"], 45), [0, c["nx"] - 1, c["nx"] // 2]])
    iy = np.concatenate([rng.integers(0, c["ny"], 45), [0, c["ny"] - 1, c["ny"] // 2]])
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], c["nx"], c["ny"], c["cell"], c["cell"],
                            0.0, 0.0, False, True, False, True, False, pixels=(ix, iy))
    got = dirty[ix, iy]
    rms = np.sqrt(np.mean(dirty**2))
    assert np.abs(got - ref).max() <= EPS * rms * 3  # pointwise; the L2 contract is checked below
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < EPS


def test_c2_dirty2vis_spot_check_vs_dft_and_mask(c2):
    c, g = c2
    rng = np.random.default_rng(1)
    # a sparse model keeps the DFT affordable: 300 point sources over the whole field
    x = np.zeros((c["nx"], c["ny"]))
    x[rng.integers(0, c["nx"], 300), rng.integers(0, c["ny"], 300)] = rng.standard_normal(300)
    vis = g.dirty2vis(x)
    assert np.all(vis[c["mask"] == 0] == 0)
    rows = rng.integers(0, c["uvw"].shape[0], 4000)
    chans = rng.integers(0, c["freq"].size, 4000)
    keep = c["mask"][rows, chans] != 0
    rows, chans = rows[keep], chans[keep]
    ref = dft.dft_dirty2vis(c["uvw"], c["freq"], x, c["cell"], c["cell"], 0.0, 0.0, False, True, False, True, False,
                            rows=rows, chans=chans)
    got = vis[rows, chans]
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < EPS


def test_c2_adjointness_and_hessian_identities(c2):
    c, g = c2
    rng = np.random.default_rng(2)
    x = c["x"]
    y = c["vis"] * c["mask"]
    rx = g.dirty2vis(x)
    rhy = g.vis2dirty(y)
    lhs = np.vdot(rx, y).real
    rhs = np.vdot(x, rhy)
    assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), abs(rhs))
    # fused Hessian == vis2dirty(wgt * dirty2vis(x)), symmetric, linear
    g.set_weights(c["wgt"])
    hx = g.hessian(x)
    comp = g.vis2dirty(rx, c["wgt"])
    assert np.linalg.norm(hx - comp) / np.linalg.norm(comp) < 1e-10
    z = rng.standard_normal(x.shape)
    hz = g.hessian(z)
    a, b = np.vdot(z, hx), np.vdot(hz, x)
    assert abs(a - b) <= 1e-10 * abs(a)
    assert np.vdot(x, hx) > 0
    hxz = g.hessian(2.0 * x - 3.0 * z)
    assert np.linalg.norm(hxz - (2.0 * hx - 3.0 * hz)) / np.linalg.norm(hxz) < 1e-10
    # Tikhonov / wsum / beam handling of the fused call
    beam = 0.5 + rng.random(x.shape)
    wsum = float(c["wgt"][c["mask"] != 0].sum())
    hb = g.hessian(x, beam=beam, eta=0.25, wsum=wsum)
    ref = beam * g.hessian(x * beam) / wsum + 0.25 * x
    assert np.linalg.norm(hb - ref) / np.linalg.norm(ref) < 1e-10


def test_c3_device_cg_equals_host_loop_on_the_c2_operator(c2):
    """BASELINE config C3, single-GPU leg: the per-band PCG solve on the C2 operator.  K iterations of the on-device CG
    (pfbhip_gridder_cg: Hessian applies and CG vectors stay in HBM) give the iterate of the reference's loop
    (opt/pcg.py:122-199: alpha / beta updates, stop on the relative change of the iterate) run on the host around
    Gridder.hessian, the start value of the CG functional is undercut."""
    from pfb_imaging_amd import opt

    c, g = c2
    g.set_weights(c["wgt"])
    wsum = float(c["wgt"][c["mask"] != 0].sum())
    eta = 1e-3  # eta = 1e-3 of the wsum-normalised operator
    rng = np.random.default_rng(3)
    # a band-limited right-hand side: the Hessian of a sparse sky (what a residual looks like)
    sky = np.zeros((c["nx"], c["ny"]))
    sky[rng.integers(0, c["nx"], 200), rng.integers(0, c["ny"], 200)] = 1.0 + rng.random(200)
    b = g.hessian(sky, eta=eta, wsum=wsum)
    K = 4
    dev = g.cg(b, eta=eta, wsum=wsum, tol=0.0, maxit=K, minit=K)
    assert g.last_cg["iters"] == K
    host = opt._cg_host(lambda v: g.hessian(v, eta=eta, wsum=wsum), b, np.zeros_like(b), None, 0.0, K, K, 0, 10, False, "host")
    assert np.linalg.norm(dev - host) / np.linalg.norm(host) < 1e-9
    # CG minimises phi(x) = x.Ax / 2 - b.x over the Krylov space: after K steps phi is below its start value phi(0) = 0
    adev = g.hessian(dev, eta=eta, wsum=wsum)
    assert 0.5 * np.vdot(dev, adev) - np.vdot(b, dev) < 0.0
    # (the stopping rule itself -- relative change of the iterate -- is compared loop against loop at a size where the
    # solve converges in a few iterations: tests/test_gpu_callables.py::test_pcg_family_over_the_device_cg)


def test_record_scatter_past_32_bit_record_offsets():
    """1e8 active visibilities on a plan with polynomial w-planes: the record scatter's byte offsets into the record / value
    arrays pass 2^32 (32 bytes x 1.34e8 records; 16 bytes x planes x 6.7e7 .. 8.9e7 values).  They used to be absolute 32-bit
    numbers that wrapped -- silently addressing other visibilities' records -- and are now relative to the wave's first record
    (csrc/gridder_kernels_mp.hpp, k_grid_rec).  Gridding is additive over rows (tests/test_imager_pass2.py:45-63 of the
    reference): the image of all rows equals the sum of the images of the two halves, whose plans stay below every wrap point."""
    from pfb_imaging_amd.wgridder import Gridder

    nrow, nchan, npix = 12_500_000, 8, 256
    rng = np.random.default_rng(11)
    c = synth.make_case(nrow, nchan, npix, zscale=0.03, seed=11, with_vis=False)
    c["cell"] *= 2.0                                                      # (field and w range wide enough for three planes)
    vis = np.empty((nrow, nchan), dtype=np.complex128)
    vis.real = rng.standard_normal((nrow, nchan), dtype=np.float32)     # (float32 draws: half the time of float64 ones)
    vis.imag = rng.standard_normal((nrow, nchan), dtype=np.float32)
    kw = dict(npix_x=npix, npix_y=npix, pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0, epsilon=1e-9,
              flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False, force_wmode=1)
    g = Gridder(c["uvw"], c["freq"], None, **kw)
    assert g.info["scatter_mode"] == 2 and g.info["wmode"] == 1 and g.nactive == nrow * nchan, g.info
    assert g.info["nplanes"] >= 3, g.info
    assert (g.nactive + 136) * 16 * min(g.info["nplanes"], 4) > 2**32   # the value array is past the old wrap point
    whole = g.vis2dirty(vis)
    g.close()
    parts = np.zeros_like(whole)
    h = nrow // 2
    for sl in (slice(0, h), slice(h, nrow)):
        gp = Gridder(c["uvw"][sl], c["freq"], None, **kw)
        parts += gp.vis2dirty(vis[sl])
        gp.close()
    assert np.linalg.norm(whole - parts) / np.linalg.norm(parts) < 1e-7
