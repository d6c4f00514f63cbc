"""GPU parity tests of the w-stacking gridder / degridder / exact Hessian (through the C-ABI).

Oracle: oracle.dft (the measurement equation, exact) and oracle.wgridder (the algorithm
restated on the CPU, run with the GPU plan's own kernel/plane choice so that intermediate
quantities are comparable).  Tolerances:
  * index / bin / tile map ........ bit-exact
  * GPU vs algorithm restatement ... 1e-10 relative L2 (same arithmetic, different summation order)
  * GPU vs direct DFT .............. the requested epsilon (relative L2), the accuracy contract of
                                    the reference's gridder (ducc0)
"""


import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dft  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402
from oracle import wgridder as owg  # noqa: E402


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def make(nrow=3000, nchan=2, npix=64, zscale=0.3, seed=1, widen=40.0):
    c = synth.make_case(nrow, nchan, npix, zscale=zscale, seed=seed)
    c["cell"] = c["cell"] * widen
    return c


def gpu_plan(c, **over):
    from pfb_imaging_amd.wgridder import Gridder

    kw = dict(npix_x=c["nx"], npix_y=c["ny"], pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0,
              epsilon=1e-7, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False)
    kw.update(over)
    mask = kw.pop("mask", c["mask"])
    g = Gridder(c["uvw"], c["freq"], mask, **kw)
    kw.pop("force_wmode", None)
    return g, kw, mask


def oracle_plan(c, g, kw, mask):
    return owg.Plan(c["uvw"], c["freq"], mask, kw["npix_x"], kw["npix_y"], kw["pixsize_x"], kw["pixsize_y"],
                    kw["center_x"], kw["center_y"], kw["epsilon"], kw["flip_u"], kw["flip_v"], kw["flip_w"],
                    kw["do_wgridding"], kw["divide_by_n"], params=g.oracle_params())


@pytest.mark.parametrize("center", [(0.0, 0.0), (0.01, -0.02)])
@pytest.mark.parametrize("do_w", [True, False])
def test_binmap_bit_exact(center, do_w):
    c = make()
    g, kw, mask = gpu_plan(c, center_x=center[0], center_y=center[1], do_wgridding=do_w)
    o = oracle_plan(c, g, kw, mask)
    bm = g.binmap()
    assert np.array_equal(bm["iu0"], o.iu0)
    assert np.array_equal(bm["iv0"], o.iv0)
    assert np.array_equal(bm["p0"], o.p0)
    assert np.array_equal(bm["flip"], o.flip)
    # same set of active visibilities; the plan sorts by the tile id of the TRANSPOSED problem it works on
    # (v-tile major): the oracle's (u-tile, v-tile) pair re-keyed that way is non-decreasing along the GPU's order
    assert g.nactive == int(o.active.sum())
    assert np.array_equal(np.sort(bm["order"]), np.flatnonzero(o.active))
    ntv = -(-g.info["nv"] // 32)
    ntu = -(-g.info["nu"] // 32)
    tu, tv = o.tile_id // ntv, o.tile_id % ntv
    tiles = (tv * ntu + tu)[bm["order"]]
    assert np.all(np.diff(tiles) >= 0)
    g.close()


@pytest.mark.parametrize("wmode", [0, 1])
def test_grid_plane_matches_oracle(wmode):
    c = make()
    g, kw, mask = gpu_plan(c, force_wmode=wmode)
    o = oracle_plan(c, g, kw, mask)
    sval = o.prep_vis(c["vis"], c["wgt"])
    for plane in (0, g.info["nplanes"] // 2, g.info["nplanes"] - 1):
        got = g.grid_plane(c["vis"], c["wgt"], plane)
        ref = o.grid_plane(sval, plane)
        assert rel(got, ref) < 1e-12
    g.close()


@pytest.mark.parametrize("center", [(0.0, 0.0), (0.01, -0.02)])
@pytest.mark.parametrize("do_w,divn,wmode", [(True, False, 0), (True, True, 0), (True, False, 1), (True, True, 1),
                                             (False, False, None)])
def test_vis2dirty_dirty2vis_vs_dft(center, do_w, divn, wmode):
    """Both w-plane schemes (ES-kernel planes / polynomial planes through Chebyshev nodes)."""
    c = make()
    g, kw, mask = gpu_plan(c, center_x=center[0], center_y=center[1], do_wgridding=do_w, divide_by_n=divn,
                           force_wmode=wmode)
    if do_w:
        assert g.info["wmode"] == wmode
    o = oracle_plan(c, g, kw, mask)
    d = g.vis2dirty(c["vis"], c["wgt"])
    assert rel(d, o.vis2dirty(c["vis"], c["wgt"])) < 1e-10
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], mask, c["nx"], c["ny"], c["cell"], c["cell"],
                            center[0], center[1], False, True, False, do_w, divn)
    assert rel(d, ref) < kw["epsilon"]
    v = g.dirty2vis(c["x"])
    assert rel(v, o.dirty2vis(c["x"])) < 1e-10
    refv = dft.dft_dirty2vis(c["uvw"], c["freq"], c["x"], c["cell"], c["cell"], center[0], center[1], False, True,
                             False, do_w, divn)
    refv[mask == 0] = 0
    assert rel(v, refv) < kw["epsilon"]
    assert np.all(v[mask == 0] == 0)
    g.close()


@pytest.mark.parametrize("eps", [1e-3, 1e-5, 1e-9])
@pytest.mark.parametrize("widen,zscale", [(40.0, 0.3), (2.0, 1e-3)])
def test_epsilon_contract(eps, widen, zscale):
    """Requested accuracy holds for wide fields (kernel w-planes win) and narrow ones (polynomial planes win)."""
    c = make(nrow=1500, widen=widen, zscale=zscale)
    g, kw, mask = gpu_plan(c, epsilon=eps)
    d = g.vis2dirty(c["vis"], c["wgt"])
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], mask, c["nx"], c["ny"], c["cell"], c["cell"],
                            0.0, 0.0, False, True, False, True, False)
    assert rel(d, ref) < eps
    g.close()


def test_adjointness_and_linearity():
    c = make(nrow=2000)
    g, kw, mask = gpu_plan(c)
    y = c["vis"] * mask
    lhs = np.vdot(g.dirty2vis(c["x"]), y).real
    rhs = np.vdot(c["x"], g.vis2dirty(y))
    # exact adjoints up to rounding; the kernel correction amplifies FFT rounding by up to ~5e3 per axis at the
    # image edge for the low-sigma / W = 16 rows the plan prefers (scatter cost does not depend on W)
    assert abs(lhs - rhs) <= 1e-10 * abs(rhs) + 1e-9
    # row additivity (test_imager_pass2.py:45-63): grid(cat) == grid(p0) + grid(p1)
    from pfb_imaging_amd.wgridder import vis2dirty

    k = 1200
    common = dict(freq=c["freq"], npix_x=c["nx"], npix_y=c["ny"], pixsize_x=c["cell"], pixsize_y=c["cell"],
                  epsilon=1e-7, do_wgridding=True, flip_v=True, divide_by_n=False)
    d_all = vis2dirty(uvw=c["uvw"], vis=c["vis"], wgt=c["wgt"], mask=mask, **common)
    d0 = vis2dirty(uvw=c["uvw"][:k], vis=c["vis"][:k], wgt=c["wgt"][:k], mask=mask[:k], **common)
    d1 = vis2dirty(uvw=c["uvw"][k:], vis=c["vis"][k:], wgt=c["wgt"][k:], mask=mask[k:], **common)
    np.testing.assert_allclose(d_all, d0 + d1, rtol=1e-5, atol=1e-6 * np.abs(d_all).max())
    g.close()


def test_reference_conventions_two_point_sources(golden_dir):
    """/root/reference/tests/test_hessian_approx.py:128-185 (test_wgridder_conventions): dirty2vis
    of two unit point sources under the pfb conventions == the explicit DFT, atol 1e-4 at
    epsilon 1e-6, five centre offsets.  Expected values are the committed golden vectors."""
    from pfb_imaging_amd.operators.gridder import wgridder_conventions
    from pfb_imaging_amd.wgridder import dirty2vis

    gold = np.load(f"{golden_dir}/conventions_two_sources.npz")
    uvw, freqs = gold["uvw"], gold["freq"]
    npix = int(gold["npix"])
    pixsize = float(gold["pixsize"])
    dirty = np.zeros((npix, npix))
    dirty[npix // 2, npix // 2] = 1.0
    dirty[npix // 4, npix // 4] = 1.0
    for k, (l0, m0) in enumerate(gold["offsets"]):
        flip_u, flip_v, flip_w, x0, y0 = wgridder_conventions(l0, m0)
        vis = dirty2vis(uvw=uvw, freq=freqs, dirty=dirty, wgt=None, pixsize_x=pixsize, pixsize_y=pixsize,
                        center_x=x0, center_y=y0, epsilon=1e-6, do_wgridding=True, flip_u=flip_u, flip_v=flip_v,
                        flip_w=flip_w, divide_by_n=True, nthreads=2, verbosity=0)
        np.testing.assert_allclose(vis.real, gold["vis"][k].real, atol=1e-4)
        np.testing.assert_allclose(vis.imag, gold["vis"][k].imag, atol=1e-4)
        assert rel(vis, gold["vis"][k]) < 1e-6


def test_psfvis_identity():
    """/root/reference/tests/test_hessian_approx.py:188-231: dirty2vis(delta at centre) equals the
    analytic PSF phase ramp to epsilon = 1e-10."""
    from pfb_imaging_amd.operators.gridder import psf_visibilities, wgridder_conventions
    from pfb_imaging_amd.wgridder import dirty2vis

    c = make(nrow=2000, npix=128, widen=4.0)
    for l0, m0 in [(0.0, 0.0), (0.1, -0.17), (-0.15, -0.2)]:
        flip_u, flip_v, flip_w, x0, y0 = wgridder_conventions(l0, m0)
        x = np.zeros((128, 128))
        x[64, 64] = 1.0
        eps = 1e-10
        v = dirty2vis(uvw=c["uvw"], freq=c["freq"], dirty=x, pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=x0,
                      center_y=y0, flip_u=flip_u, flip_v=flip_v, flip_w=flip_w, epsilon=eps, do_wgridding=True,
                      divide_by_n=False)
        # the analytic ramp in extended precision: at these baseline lengths (3e5 wavelengths) the
        # float64 expression of gridder.py:616-629 is itself only good to ~2e-10
        ld = np.longdouble
        r2 = ld(x0) ** 2 + ld(y0) ** 2
        nm1 = -r2 / (1 + np.sqrt(1 - r2))
        ph = (c["uvw"][:, 0:1].astype(ld) * ld(x0) + c["uvw"][:, 1:2].astype(ld) * ld(y0)
              - c["uvw"][:, 2:].astype(ld) * nm1) * (c["freq"][None, :].astype(ld) / ld(299792458.0))
        ph = ph - np.rint(ph)
        exact = np.exp(-2j * np.pi * ph.astype(np.float64))
        assert np.abs(exact - v).max() <= eps
        # and the float64 formula the reference uses for its PSF visibilities agrees to its own rounding
        psf_vis = np.conj(psf_visibilities(c["uvw"], c["freq"], x0, y0, flip_u, flip_v))
        assert np.abs(psf_vis - v).max() <= 1e-9


def test_edge_cases():
    from pfb_imaging_amd.wgridder import Gridder, dirty2vis, vis2dirty

    c = make(nrow=200, npix=16)
    base = dict(freq=c["freq"], pixsize_x=c["cell"], pixsize_y=c["cell"], epsilon=1e-6, do_wgridding=True,
                flip_v=True, divide_by_n=False)
    # all masked -> zero image, zero vis
    zmask = np.zeros_like(c["mask"])
    d = vis2dirty(uvw=c["uvw"], vis=c["vis"], wgt=c["wgt"], mask=zmask, npix_x=16, npix_y=16, **base)
    assert d.shape == (16, 16) and not d.any()
    v = dirty2vis(uvw=c["uvw"], dirty=c["x"], mask=zmask, **base)
    assert not v.any()
    # no mask / no weights, non-square image, out-parameter filled in place, read-only inputs
    uvw_ro = c["uvw"].copy()
    uvw_ro.setflags(write=False)
    vis_ro = c["vis"].copy()
    vis_ro.setflags(write=False)
    out = np.zeros((16, 24))
    ret = vis2dirty(uvw=uvw_ro, vis=vis_ro, npix_x=16, npix_y=24, dirty=out, **base)
    assert ret is out
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], None, None, 16, 24, c["cell"], c["cell"], 0, 0, False,
                            True, False, True, False)
    assert rel(out, ref) < 1e-6
    # zero rows
    d0 = vis2dirty(uvw=np.zeros((0, 3)), vis=np.zeros((0, c["freq"].size), complex), npix_x=16, npix_y=16, **base)
    assert not d0.any()
    # broadcast (zero-stride) visibilities as the reference passes for the PSF (gridder.py:886)
    ones = np.broadcast_to(np.ones((1,), dtype=np.complex128), c["vis"].shape)
    dp = vis2dirty(uvw=c["uvw"], vis=ones, wgt=c["wgt"], mask=c["mask"], npix_x=16, npix_y=16, **base)
    refp = dft.dft_vis2dirty(c["uvw"], c["freq"], np.ones_like(c["vis"]), c["wgt"], c["mask"], 16, 16, c["cell"],
                             c["cell"], 0, 0, False, True, False, True, False)
    assert rel(dp, refp) < 1e-6
    # single precision in -> single precision out
    d32 = vis2dirty(uvw=c["uvw"], vis=c["vis"].astype(np.complex64), npix_x=16, npix_y=16, **base)
    assert d32.dtype == np.float32
    # shape errors are ValueErrors
    with pytest.raises(ValueError):
        vis2dirty(uvw=c["uvw"], vis=c["vis"][:-1], npix_x=16, npix_y=16, **base)
    with pytest.raises(ValueError):
        Gridder(c["uvw"], c["freq"], None, npix_x=16, npix_y=16, pixsize_x=-1.0, pixsize_y=1.0, epsilon=1e-6)
    with pytest.raises(ValueError):
        Gridder(c["uvw"], c["freq"], None, npix_x=16, npix_y=16, pixsize_x=1e-5, pixsize_y=1e-5, epsilon=1e-16)


def test_hessian_slice_matches_reference_composition():
    """hessian_slice == beam * vis2dirty(wgt * dirty2vis(beam x)) / wsum + eta x (hessian.py:15-100)."""
    from pfb_imaging_amd.operators.hessian import hessian_slice

    c = make()
    rng = np.random.default_rng(3)
    beam = 0.5 + rng.random((c["nx"], c["ny"]))
    wsum = c["wgt"][c["mask"] != 0].sum()
    eta = 0.05
    kw = dict(uvw=c["uvw"], weight=c["wgt"], vis_mask=c["mask"], freq=c["freq"], beam=beam, cell=c["cell"],
              do_wgridding=True, epsilon=1e-7, eta=eta, wsum=wsum)
    got = hessian_slice(c["x"], **kw)
    g, gkw, mask = gpu_plan(c)
    o = oracle_plan(c, g, gkw, mask)
    ref = o.vis2dirty(o.dirty2vis(c["x"] * beam), c["wgt"]) / wsum * beam + eta * c["x"]
    assert rel(got, ref) < 1e-10
    # zero shortcut and xout
    assert not hessian_slice(np.zeros_like(c["x"]), **kw).any()
    xout = np.empty_like(c["x"])
    assert hessian_slice(c["x"], xout=xout, **kw) is xout
    assert rel(xout, ref) < 1e-10
    # positive semi-definite, symmetric
    y = rng.standard_normal(c["x"].shape)
    kw0 = dict(kw, eta=None)
    a = np.vdot(y, hessian_slice(c["x"], **kw0))
    b = np.vdot(hessian_slice(y, **kw0), c["x"])
    assert abs(a - b) < 1e-10 * abs(a)
    g.close()


def test_gridder_cg_matches_oracle_pcg():
    from oracle.fftconv import pcg

    c = make(nrow=2500, npix=32, widen=20.0)
    g, gkw, mask = gpu_plan(c)
    g.set_weights(c["wgt"])
    wsum = c["wgt"][mask != 0].sum()
    eta = 0.5  # (the CPU restatement's solve is what this test's time goes into: a better-conditioned system, fewer iterations)
    rhs = g.hessian(c["x"], eta=eta, wsum=wsum)
    sol = g.cg(rhs, eta=eta, wsum=wsum, tol=1e-9, maxit=200, minit=1)
    assert rel(sol, c["x"]) < 1e-6
    o = oracle_plan(c, g, gkw, mask)
    ref = pcg(lambda z: o.vis2dirty(o.dirty2vis(z), c["wgt"]) / wsum + eta * z, rhs, tol=1e-9, maxit=200, minit=1)
    assert rel(sol, ref) < 1e-6
    g.close()


@pytest.mark.parametrize("eps,widen,zscale", [(1e-7, 8.0, 0.02), (1e-4, 30.0, 0.5), (1e-10, 8.0, 0.02)])
def test_scatter_forms_agree(eps, widen, zscale, monkeypatch):
    """The register-footprint scatter (default: block-sorted visibilities, footprints accumulated in registers) and the
    diagonal-walk scatter (PFBHIP_SCATTER=walk) are the same sum in a different order (the tile order of the sort is
    checked in test_binmap_bit_exact, which runs the default form)."""
    c = make(nrow=3000, npix=256, widen=widen, zscale=zscale)
    res = {}
    monkeypatch.setenv("PFBHIP_WMODE2", "0")  # the forms of the MULTI-plane scatter (the one-plane scheme has a kernel of its own)
    for mode in ("block", "walk", "rec", "rec_es"):
        monkeypatch.setenv("PFBHIP_SCATTER", mode)
        g, kw, mask = gpu_plan(c, epsilon=eps)
        # the record-driven form (k_grid_rec: per-visibility records, scalar loads) serves single-pass plans with polynomial
        # w-planes; "rec_es" extends it to ES-kernel plane stacks (their values written per pass by k_plane_values_es: an
        # option, not the default); asked for elsewhere, the plan keeps k_grid_blk
        single_pass = g.info["wmode"] == 1 and g.info["nplanes"] <= 4
        es = g.info["wmode"] == 0 or (g.info["wmode"] == 1 and g.info["nplanes"] > 4)  # (every multi-pass plan)
        assert g.info["scatter_mode"] == {"block": 1, "walk": 0, "rec": 2 if single_pass else 1,
                                          "rec_es": 2 if (single_pass or es) else 1}[mode], g.info
        g.set_weights(c["wgt"])
        res[mode] = (g.vis2dirty(c["vis"], c["wgt"]), g.hessian(c["x"], eta=0.1, wsum=3.0), g.info["W"], g.info["nplanes"])
        g.close()
    assert res["block"][2:] == res["walk"][2:] == res["rec"][2:] == res["rec_es"][2:]
    assert rel(res["block"][0], res["walk"][0]) < 1e-9 and rel(res["block"][1], res["walk"][1]) < 1e-9
    assert rel(res["rec"][0], res["block"][0]) < 1e-9 and rel(res["rec"][1], res["block"][1]) < 1e-9
    assert rel(res["rec_es"][0], res["block"][0]) < 1e-9 and rel(res["rec_es"][1], res["block"][1]) < 1e-9
    # left to itself the plan takes the single-launch walk kernel for a problem this small (a few hundred work items)
    monkeypatch.delenv("PFBHIP_SCATTER")
    g, kw, mask = gpu_plan(c, epsilon=eps)
    assert g.info["scatter_mode"] == 0 and g.info["scatter_launches"] == 1 and g.info["nwork"] < 2048
    g.close()
    # ... and, allowed to, the one-plane w-scheme where the field admits it (on-axis, 2..4 kernel functions): same image to epsilon
    monkeypatch.delenv("PFBHIP_WMODE2")
    g, kw, mask = gpu_plan(c, epsilon=eps)
    if g.info["wmode"] == 2:
        assert g.info["nplanes"] == 1 and g.info["scatter_launches"] == 1 and 2 <= g.info["nderiv"] <= 4
        assert rel(g.vis2dirty(c["vis"], c["wgt"]), res["block"][0]) < max(eps, 1e-9)
    g.close()


@pytest.mark.parametrize("W, sigma", [(13, 1.5), (14, 1.4), (15, 1.3), (16, 1.25)])
@pytest.mark.parametrize("wmode, widen, zscale", [(0, 30.0, 0.5), (1, 8.0, 0.02), (1, 30.0, 0.1)])
def test_multi_plane_scatter_frames(W, sigma, wmode, widen, zscale, monkeypatch):
    """The register frames of k_grid_blk / k_grid_rec (csrc/gridder_kernels_mp.hpp; round 4b): 16 x 16 cells on 4 x 16 lanes for
    W <= 15 (2 x 2-cell anchoring and the finer sort key at W = 14, 15), 3 x 20 lanes for W = 16 and under PFBHIP_WD_BLOCK=4 -- ES-kernel
    plane stacks and polynomial planes, both kernels, against the CPU restatement run with the plan's parameters and against
    the other anchoring (summation order only)."""
    c = make(nrow=3000, npix=256, widen=widen, zscale=zscale)
    monkeypatch.setenv("PFBHIP_WMODE2", "0")
    res = {}
    for mode in ("block", "rec_es"):
        for blk in ("2", "4"):
            monkeypatch.setenv("PFBHIP_SCATTER", mode)
            monkeypatch.setenv("PFBHIP_WD_BLOCK", blk)
            g, kw, mask = gpu_plan(c, epsilon=1e-7, force_wmode=wmode, force=(sigma, W))
            assert g.info["wmode"] == wmode and g.info["W"] == W, g.info
            assert (g.info["nplanes"] > 4) == (wmode == 0 or widen > 10), g.info  # (the third case: polynomial planes in several passes)
            assert g.info["scatter_block"] == (2 if (blk == "2" and W in (14, 15)) else 4), g.info
            o = oracle_plan(c, g, kw, mask)
            d = g.vis2dirty(c["vis"], c["wgt"])
            # (W = 16 at sigma = 1.25 under an image that fills its grid: the rounding of grid and FFT, amplified by the image-side
            # correction, is what separates the GPU from the restatement -- DESIGN.md section 3, "what the budget does not cover")
            assert rel(d, o.vis2dirty(c["vis"], c["wgt"])) < (1e-10 if W < 16 else 3e-8)
            g.set_weights(c["wgt"])
            res[mode, blk] = (d, g.hessian(c["x"], eta=0.1, wsum=3.0), g.info["scatter_mode"])
            g.close()
    assert res["block", "2"][2] == 1 and res["rec_es", "2"][2] == 2
    ref = res["block", "4"]
    for k, v in res.items():
        assert rel(v[0], ref[0]) < (1e-10 if W < 16 else 3e-8) and rel(v[1], ref[1]) < (1e-10 if W < 16 else 3e-8), k


def test_gridder_power_method_matches_oracle():
    """Spectral norm of the exact Hessian: the device-resident iteration against the numpy restatement
    (opt/power_method.py:40-93) around the oracle gridder, and through opt.power_method(g.hessian, ...)."""
    from oracle.fftconv import power_method as pm_oracle
    from pfb_imaging_amd.opt import power_method

    c = make(nrow=900, npix=32, widen=20.0)
    g, gkw, mask = gpu_plan(c)
    g.set_weights(c["wgt"])
    wsum = c["wgt"][mask != 0].sum()
    b0 = np.random.default_rng(8).standard_normal((c["nx"], c["ny"]))
    o = oracle_plan(c, g, gkw, mask)
    rbeta, rb, rk = pm_oracle(lambda z: o.vis2dirty(o.dirty2vis(z), c["wgt"]) / wsum + 0.1 * z, b0.shape, b0.copy(), tol=0.0,
                              maxit=6)
    beta, b = g.power_method(b0, eta=0.1, wsum=wsum, tol=0.0, maxit=6)
    assert g.last_pm["iters"] == 6 == rk
    assert abs(beta - rbeta) < 1e-8 * rbeta and rel(b, rb) < 1e-6
    # the reference's call form; eta = 0, wsum = 1 are Gridder.hessian's defaults
    beta2, b2 = power_method(g.hessian, b0.shape, b0=b0, tol=0.0, maxit=6, verbosity=0)
    rbeta2, rb2, _ = pm_oracle(lambda z: o.vis2dirty(o.dirty2vis(z), c["wgt"]), b0.shape, b0.copy(), tol=0.0, maxit=6)
    assert abs(beta2 - rbeta2) < 1e-8 * rbeta2 and rel(b2, rb2) < 1e-6
    g.close()


def test_residual_from_partitions_properties():
    """/root/reference/tests/test_imager_pass2.py:117-153."""
    from pfb_imaging_amd.operators.gridder import residual_from_partitions

    nx = ny = 16

    def part(nrow, seed, beam_val=1.0):
        rng = np.random.default_rng(seed)
        return {"UVW": rng.standard_normal((nrow, 3)) * 100.0, "FREQ": np.array([1.0e9]),
                "WEIGHT": np.abs(rng.standard_normal((1, nrow, 1))) + 0.1, "MASK": np.ones((nrow, 1), dtype=np.uint8),
                "BEAM": np.full((1, nx, ny), float(beam_val)), "attrs": {"l0": 0.0, "m0": 0.0}}

    dirty = np.random.default_rng(5).standard_normal((1, nx, ny))
    res = residual_from_partitions(dirty, [part(200, 0)], np.zeros((1, nx, ny)), cell_rad=1.0e-6)
    np.testing.assert_allclose(res, dirty, atol=1e-12)
    rng = np.random.default_rng(7)
    model = rng.standard_normal((1, nx, ny))
    p0, p1 = part(120, 0), part(80, 1)
    c01 = dirty - residual_from_partitions(dirty, [p0, p1], model, 1.0e-6)
    c0 = dirty - residual_from_partitions(dirty, [p0], model, 1.0e-6)
    c1 = dirty - residual_from_partitions(dirty, [p1], model, 1.0e-6)
    np.testing.assert_allclose(c01, c0 + c1, rtol=1e-5, atol=1e-8)
    z = np.zeros((1, nx, ny))
    c1 = z - residual_from_partitions(z, [part(200, 0, 1.0)], model, 1.0e-6)
    c2 = z - residual_from_partitions(z, [part(200, 0, 2.0)], model, 1.0e-6)
    np.testing.assert_allclose(c2, 2.0 * c1, rtol=1e-5, atol=1e-8)
    # against the DFT
    p = part(200, 0)
    mv = dft.dft_dirty2vis(p["UVW"], p["FREQ"], model[0], 1e-6, 1e-6, 0, 0, False, True, False, True, False)
    ref = dft.dft_vis2dirty(p["UVW"], p["FREQ"], mv, p["WEIGHT"][0], p["MASK"], nx, ny, 1e-6, 1e-6, 0, 0, False, True,
                            False, True, False)
    assert rel(z - residual_from_partitions(z, [p], model, 1.0e-6), ref[None]) < 1e-6


def test_handwritten_row_fft_matches_numpy():
    """The hand-written batched row FFT (rowfft.hpp) against numpy, forward and inverse, for every
    supported family of lengths (2^a, 3 2^a, 5 2^a)."""
    import ctypes as ct

    from pfb_imaging_amd._lib import check, cint, i64, lib, ptr

    rng = np.random.default_rng(0)
    for n in (1024, 1152, 1280, 1536, 1792, 1920, 2048, 2304, 3584, 3840, 4608, 5120, 6144, 7168, 7680, 8192, 9216, 10240, 12288,
              14336, 15360, 16384, 20480, 24576, 32768):
        a = rng.standard_normal((5, n)) + 1j * rng.standard_normal((5, n))
        for inverse in (0, 1):
            b = a.copy()
            check(lib().pfbhip_debug_rowfft(ptr(b), i64(n), i64(5), cint(inverse), cint(1), None))
            ref = np.fft.ifft(a, axis=1) * n if inverse else np.fft.fft(a, axis=1)
            assert rel(b, ref) < 5e-15
    with pytest.raises(ValueError):
        bad = np.zeros((1, 1000), dtype=complex)
        check(lib().pfbhip_debug_rowfft(ptr(bad), i64(1000), i64(1), cint(0), cint(1), None))


def test_fused_row_fft_path(monkeypatch):
    """Default plane transform (hand-written row FFTs, fused pad / crop / w-screen with the polynomial
    n-1, or its closed form for wide fields) against the rocFFT fallback (separate pad / crop kernels)
    and against the oracle.  W = 15 at sigma = 1.25 divides by a kernel transform of ~1e-7 at the image
    edge, so two correct FFTs (rounding ~1e-16) differ by up to ~1e-8 THERE; with ~170 planes summed the
    L2 difference is 3e-10 (measured: both paths sit 3.0e-10 from the oracle's numpy FFT)."""
    from pfb_imaging_amd.wgridder import Gridder

    for widen, zscale, tol in ((8.0, 0.02, 1e-10), (60.0, 1.0, 2e-9), (1150.0, 0.002, 2e-9)):
        c = make(nrow=2500, npix=1024, widen=widen, zscale=zscale)  # nu = 1280 = 5 * 2^8
        monkeypatch.delenv("PFBHIP_FUSED_FFT", raising=False)
        monkeypatch.delenv("PFBHIP_ROWFFT", raising=False)
        g1, kw, mask = gpu_plan(c, force=(1.25, 15))
        assert g1.info["nu"] == 1280 and g1.info["nplanes"] < 200
        d1, v1 = g1.vis2dirty(c["vis"], c["wgt"]), g1.dirty2vis(c["x"])
        if widen == 8.0:  # 5 polynomial planes: cheap enough for the CPU oracle
            o = oracle_plan(c, g1, {k: v for k, v in kw.items() if k != "force"}, mask)
            assert rel(d1, o.vis2dirty(c["vis"], c["wgt"])) < tol and rel(v1, o.dirty2vis(c["x"])) < tol
        g1.close()
        monkeypatch.setenv("PFBHIP_FUSED_FFT", "0")
        monkeypatch.setenv("PFBHIP_ROWFFT", "0")
        g0, _, _ = gpu_plan(c, force=(1.25, 15))
        d0, v0 = g0.vis2dirty(c["vis"], c["wgt"]), g0.dirty2vis(c["x"])
        g0.close()
        assert rel(d1, d0) < tol and rel(v1, v0) < tol
        monkeypatch.setenv("PFBHIP_FUSED_FFT", "1")  # mixed: fused second axis, rocFFT first axis
        g2, _, _ = gpu_plan(c, force=(1.25, 15))
        d2 = g2.vis2dirty(c["vis"], c["wgt"])
        g2.close()
        assert rel(d2, d0) < tol


def test_first_axis_variants_agree(monkeypatch):
    """The first axis of the plane transform in its four forms -- transposing row FFT with the degridding side stored
    transposed by the fused pad kernel (default), the same with gathered loads (PFBHIP_TPAD=0), without the XCD-aware
    row order (PFBHIP_TFFT=2), and the plain row FFT with separate transpose kernels (PFBHIP_TFFT=0) -- is the same
    arithmetic on the same numbers: results agree to rounding.  1100 x 1000 pixels: neither image axis is a multiple of
    the 64-row groups of the transposed stores."""
    c = make(nrow=2000, npix=64, widen=8.0, zscale=0.05)
    rng = np.random.default_rng(11)
    c["nx"], c["ny"] = 1100, 1000
    c["cell"] = c["cell"] * 64.0 / 1100
    c["x"] = rng.standard_normal((1100, 1000))
    outs = []
    # (+ the whole-row forms of what is pruned to the used cells of a plane: first-axis loads / stores, the Hessian's clear)
    for env in ({}, {"PFBHIP_TPAD": "0"}, {"PFBHIP_TFFT": "2"}, {"PFBHIP_TFFT": "0"}, {"PFBHIP_COLRUNS": "0"},
                {"PFBHIP_ASYNC_CLEAR": "0"}):
        for k in ("PFBHIP_TPAD", "PFBHIP_TFFT", "PFBHIP_COLRUNS", "PFBHIP_ASYNC_CLEAR"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g, kw, mask = gpu_plan(c)
        assert g.info["fft_mode"] & 3 == 3
        assert bool(g.info["fft_mode"] & 8) == (env.get("PFBHIP_TFFT") != "0")
        g.set_weights(c["wgt"])
        h1 = g.hessian(c["x"])
        # a second apply on the same handle sees planes the first one left behind (only their used cells are cleared)
        h2 = g.hessian(2.0 * c["x"])
        assert rel(h2, 2.0 * h1) < 1e-11
        outs.append((g.vis2dirty(c["vis"], c["wgt"]), g.dirty2vis(c["x"]), h1))
        g.close()
    for k in ("PFBHIP_COLRUNS", "PFBHIP_ASYNC_CLEAR"):
        monkeypatch.delenv(k, raising=False)
    for d, v, h in outs[1:]:
        assert rel(d, outs[0][0]) < 1e-11 and rel(v, outs[0][1]) < 1e-11 and rel(h, outs[0][2]) < 1e-11  # (LDS atomics: run-to-run 1e-13)


@pytest.mark.parametrize("nx,ny,center,widen,zscale", [
    (1200, 1000, (0.0, 0.0), 8.0, 0.02),        # grid 1536 x 1280 (3 * 2^9, 5 * 2^8): leading radix-3 and -5 passes
    (1600, 840, (0.003, -0.002), 8.0, 0.05),     # grid 2048 x 1280 / 1024: rectangular, shifted phase centre
    (1024, 1636, (0.0, 0.001), 30.0, 0.5),       # grid 1280 x 2048: wide field, ES-kernel planes, psi_w correction
    (1000, 1560, (0.0, 0.0), 8.0, 0.02),          # grid 1152 x 1920 (9 * 2^7, 15 * 2^7): radix-9 and radix-15 leads
    (1540, 1000, (-0.002, 0.001), 8.0, 0.05),     # grid 1792 x 1152 (7 * 2^8, 9 * 2^7)
])
def test_own_fft_shapes_vs_oracle(nx, ny, center, widen, zscale):
    """Rectangular images whose padded sizes take the hand-written row FFT on both axes (different
    shapes per axis), against the oracle and the DFT."""
    c = make(nrow=1500, npix=64, widen=widen, zscale=zscale)
    rng = np.random.default_rng(5)
    c["nx"], c["ny"] = nx, ny
    c["cell"] = c["cell"] * 64.0 / max(nx, ny)
    c["x"] = rng.standard_normal((nx, ny))
    g, kw, mask = gpu_plan(c, center_x=center[0], center_y=center[1])
    assert g.info["fft_mode"] & 3 == 3, g.info
    o = oracle_plan(c, g, kw, mask)
    d = g.vis2dirty(c["vis"], c["wgt"])
    v = g.dirty2vis(c["x"])
    assert rel(d, o.vis2dirty(c["vis"], c["wgt"])) < 2e-9
    assert rel(v, o.dirty2vis(c["x"])) < 2e-9
    # the DFT on a sub-sample of rows (2.5e3 vis x 1e6 pixels would take minutes)
    rows = slice(0, 200)
    refv = dft.dft_dirty2vis(c["uvw"][rows], c["freq"], c["x"], c["cell"], c["cell"], center[0], center[1], False, True,
                             False, True, False)
    refv[mask[rows] == 0] = 0
    assert rel(v[rows], refv) < kw["epsilon"]
    g.close()


@pytest.mark.parametrize("nx,ny", [(900, 14000), (14000, 900)])
def test_doubled_fft_shapes_vs_oracle(nx, ny, monkeypatch):
    """A 20480-point axis runs as a DOUBLED row-FFT shape (two 10240-point half transforms + combine): as the
    first axis (plain kernel) and as the second axis, there both with the separate pad / crop kernels (default)
    and with the dedicated fused kernels (PFBHIP_FUSED_DOUBLED=1)."""
    c = make(nrow=1500, npix=64, widen=8.0, zscale=0.02)
    rng = np.random.default_rng(6)
    c["nx"], c["ny"] = nx, ny
    c["cell"] = c["cell"] * 64.0 / max(nx, ny)
    c["x"] = rng.standard_normal((nx, ny))
    monkeypatch.delenv("PFBHIP_FUSED_DOUBLED", raising=False)
    g, kw, mask = gpu_plan(c)
    assert sorted((g.info["nu"], g.info["nv"])) == [1152, 20480], g.info
    assert g.info["fft_mode"] & 1 and g.info["fft_mode"] & 6, g.info
    o = oracle_plan(c, g, kw, mask)
    d = g.vis2dirty(c["vis"], c["wgt"])
    v = g.dirty2vis(c["x"])
    assert rel(d, o.vis2dirty(c["vis"], c["wgt"])) < 2e-9
    assert rel(v, o.dirty2vis(c["x"])) < 2e-9
    beam = 1.0 + 0.1 * rng.random((nx, ny))
    g.set_weights(c["wgt"])
    h = g.hessian(c["x"], beam=beam, eta=0.3, wsum=7.0)
    g.close()
    monkeypatch.setenv("PFBHIP_FUSED_DOUBLED", "1")
    g1, _, _ = gpu_plan(c)
    d1 = g1.vis2dirty(c["vis"], c["wgt"])
    v1 = g1.dirty2vis(c["x"])
    g1.set_weights(c["wgt"])
    h1 = g1.hessian(c["x"], beam=beam, eta=0.3, wsum=7.0)
    g1.close()
    assert rel(d1, d) < 1e-11 and rel(v1, v) < 1e-11 and rel(h1, h) < 1e-11


def test_device_block_cache_reuse_and_flush():
    """Device blocks of a destroyed plan wait in the library's cache for the next plan of the same sizes (hipMalloc costs
    ~45 ms per GB; csrc/common.hpp dev_alloc): the second plan takes them back out, computes the same image from recycled
    (not zeroed) memory, and a flush returns everything to the driver."""
    from pfb_imaging_amd import _lib

    c = make(nrow=2000, npix=64, widen=8.0)
    rng = np.random.default_rng(21)
    c["nx"] = c["ny"] = 1100
    c["cell"] = c["cell"] * 64.0 / 1100
    c["x"] = rng.standard_normal((1100, 1100))
    _lib.device_cache(flush=True)
    assert _lib.device_cache() == 0
    g, kw, mask = gpu_plan(c)
    d1 = g.vis2dirty(c["vis"], c["wgt"])
    v1 = g.dirty2vis(c["x"])
    held = g.info["device_bytes"]
    g.close()
    cached = _lib.device_cache()
    assert 0 < cached <= held, (cached, held)      # (blocks below 32 MiB are not kept)
    g2, _, _ = gpu_plan(c)
    assert _lib.device_cache() < cached            # the new plan took blocks of the old one
    d2 = g2.vis2dirty(c["vis"], c["wgt"])
    v2 = g2.dirty2vis(c["x"])
    assert rel(d2, d1) < 1e-12 and rel(v2, v1) < 1e-12
    g2.close()
    assert _lib.device_cache(flush=True) >= cached
    assert _lib.device_cache() == 0


def test_poisoned_device_blocks():
    """recycled device blocks are not zero.  One subprocess with PFBHIP_DEVCACHE_POISON=1 (every block filled with NaN
    bytes before use) builds every kind of plan twice and compares (tests/_poison_worker.py)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PFBHIP_DEVCACHE_POISON="1")
    env.pop("PFBHIP_SCATTER", None)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "_poison_worker.py")], env=env, cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "poison ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_separable_screen_matches_general_form(monkeypatch):
    """The w-screen of a pass in separable form (per-plane column table x row factor x residual polynomials,
    csrc/rowfft_api.hpp FusedPlanes::sep) against the same plan with n - 1 and sincos per pixel (PFBHIP_SEPSCREEN=0),
    and against the oracle: a 20480-point fused axis, ES-kernel planes, shifted phase centre, beam and eta."""
    c = make(nrow=1500, npix=64, widen=8.0, zscale=0.4)
    rng = np.random.default_rng(16)
    nx, ny = 14000, 900
    c["nx"], c["ny"] = nx, ny
    c["cell"] = c["cell"] * 64.0 / max(nx, ny)
    c["x"] = rng.standard_normal((nx, ny))
    center = (3e-3, -2e-3)
    monkeypatch.delenv("PFBHIP_SEPSCREEN", raising=False)
    g, kw, mask = gpu_plan(c, center_x=center[0], center_y=center[1])
    assert g.info["nu"] == 20480 and g.info["fft_mode"] & 2, g.info
    assert g.info["screen_separable"] > 0, g.info
    o = oracle_plan(c, g, kw, mask)
    d = g.vis2dirty(c["vis"], c["wgt"])
    assert rel(d, o.vis2dirty(c["vis"], c["wgt"])) < 2e-9
    beam = 1.0 + 0.1 * rng.random((nx, ny))
    g.set_weights(c["wgt"])
    h = g.hessian(c["x"], beam=beam, eta=0.3, wsum=7.0)
    g.close()
    monkeypatch.setenv("PFBHIP_SEPSCREEN", "0")
    g0, _, _ = gpu_plan(c, center_x=center[0], center_y=center[1])
    assert g0.info["screen_separable"] == 0, g0.info
    d0 = g0.vis2dirty(c["vis"], c["wgt"])
    g0.set_weights(c["wgt"])
    h0 = g0.hessian(c["x"], beam=beam, eta=0.3, wsum=7.0)
    g0.close()
    assert rel(d, d0) < 1e-11 and rel(h, h0) < 1e-11


@pytest.mark.parametrize("eps", [1e-4, 1e-10])
def test_epsilon_contract_own_fft_path(eps):
    """Accuracy contract at both ends of the range on a grid the hand-written FFT path serves (>= 1024)."""
    c = make(nrow=1200, npix=64, widen=10.0, zscale=0.1)
    rng = np.random.default_rng(9)
    c["nx"] = c["ny"] = 900
    c["cell"] = c["cell"] * 64.0 / 900
    c["x"] = rng.standard_normal((900, 900))
    g, kw, mask = gpu_plan(c, epsilon=eps)
    assert g.info["fft_mode"] & 3 == 3, g.info
    rows = slice(0, 150)
    v = g.dirty2vis(c["x"])
    refv = dft.dft_dirty2vis(c["uvw"][rows], c["freq"], c["x"], c["cell"], c["cell"], 0.0, 0.0, False, True, False, True, False)
    refv[mask[rows] == 0] = 0
    assert rel(v[rows], refv) < eps
    # adjointness ties vis2dirty to the same accuracy
    y = c["vis"] * mask
    lhs = np.vdot(v, y).real
    rhs = np.vdot(c["x"], g.vis2dirty(y))
    assert abs(lhs - rhs) <= 1e-9 * abs(rhs)
    g.close()


def test_fused_rmw_form_matches_lds_row_form(monkeypatch):
    """Grids too large for the LDS image row use the read-modify-write form of the fused kernels (every plane
    updates the accumulator, the last one finalizes): forced here with PFBHIP_FUSED_LDSROW=0, many ES-kernel
    planes in several launches, with beam and eta through the Hessian."""
    from pfb_imaging_amd.wgridder import Gridder

    c = make(nrow=1500, npix=1024, widen=60.0, zscale=1.0)
    kw = dict(npix_x=c["nx"], npix_y=c["ny"], pixsize_x=c["cell"], pixsize_y=c["cell"], epsilon=1e-7, flip_v=True,
              do_wgridding=True, divide_by_n=False)
    rng = np.random.default_rng(3)
    beam = 0.5 + rng.random((c["nx"], c["ny"]))
    res = {}
    for env in ("1", "0"):
        monkeypatch.setenv("PFBHIP_FUSED_LDSROW", env)
        g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
        assert g.info["fft_mode"] & 3 == 3 and g.info["nplanes"] > 8
        g.set_weights(c["wgt"])
        res[env] = (g.vis2dirty(c["vis"], c["wgt"]), g.dirty2vis(c["x"]), g.hessian(c["x"], beam=beam, eta=0.3, wsum=7.0))
        g.close()
    for a, b in zip(res["1"], res["0"]):
        # different summation order over ~150 planes and the run-to-run order of the LDS atomics (5e-11), edge-amplified
        assert rel(a, b) < 1e-9


@pytest.mark.parametrize("flips", [(False, True, False), (True, False, False), (True, True, True), (False, False, False)])
def test_anisotropic_pixels_and_flips_vs_dft(flips):
    """Different pixel sizes, image sizes, centre offsets and flip conventions per axis -- everything the
    plan's internal axis exchange has to carry over -- against the direct DFT (both directions)."""
    flip_u, flip_v, flip_w = flips
    c = make(nrow=1200, npix=64, widen=25.0, zscale=0.2)
    rng = np.random.default_rng(12)
    nx, ny = 72, 50
    px, py = c["cell"] * 0.8, c["cell"] * 1.3
    cx, cy = 0.011, -0.007
    x = rng.standard_normal((nx, ny))
    from pfb_imaging_amd.wgridder import Gridder

    g = Gridder(c["uvw"], c["freq"], c["mask"], npix_x=nx, npix_y=ny, pixsize_x=px, pixsize_y=py, center_x=cx, center_y=cy,
                epsilon=1e-8, flip_u=flip_u, flip_v=flip_v, flip_w=flip_w, do_wgridding=True, divide_by_n=False)
    assert g.info["nu"] >= nx and g.info["nv"] >= ny and g.info["nu"] > g.info["nv"]
    d = g.vis2dirty(c["vis"], c["wgt"])
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], nx, ny, px, py, cx, cy, flip_u, flip_v, flip_w,
                            True, False)
    assert d.shape == (nx, ny) and rel(d, ref) < 1e-8
    v = g.dirty2vis(x)
    refv = dft.dft_dirty2vis(c["uvw"], c["freq"], x, px, py, cx, cy, flip_u, flip_v, flip_w, True, False)
    refv[c["mask"] == 0] = 0
    assert rel(v, refv) < 1e-8
    o = owg.Plan(c["uvw"], c["freq"], c["mask"], nx, ny, px, py, cx, cy, 1e-8, flip_u, flip_v, flip_w, True, False,
                 params=g.oracle_params())
    bm = g.binmap()
    assert np.array_equal(bm["iu0"], o.iu0) and np.array_equal(bm["iv0"], o.iv0) and np.array_equal(bm["flip"], o.flip)
    g.close()


def test_stateless_plan_cache_sees_inplace_edits():
    """The ducc0-style calls are stateless in the reference; the plan cache behind them must therefore notice an
    input edited IN PLACE (same address, same shape) -- one flagged visibility, one changed weight -- and an equal-content
    array at another address must hit the cache."""
    from pfb_imaging_amd import wgridder as wg
    from pfb_imaging_amd.operators.hessian import hessian_slice

    wg.clear_cache()
    c = make(nrow=6000, nchan=4, npix=48)           # 24000 visibilities: the old fingerprint sampled every 5th
    mask = c["mask"].copy()
    wgt = c["wgt"].copy()
    kw = dict(uvw=c["uvw"], freq=c["freq"], wgt=wgt, mask=mask, npix_x=48, npix_y=48, pixsize_x=c["cell"], pixsize_y=c["cell"],
              center_x=0.0, center_y=0.0, epsilon=1e-7, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True,
              divide_by_n=False)
    d1 = wg.vis2dirty(vis=c["vis"], **kw)
    nplans = len(wg._cache)
    # (two runs agree to rounding, not bit for bit: the scatter adds into its tile with LDS atomics)
    assert rel(wg.vis2dirty(vis=c["vis"], **dict(kw, mask=mask.copy())), d1) < 1e-13 and len(wg._cache) == nplans
    idx = np.flatnonzero(mask.reshape(-1))
    idx = idx[(idx % 5 != 0) & (idx % 4 != 0)][len(idx) // 3]   # an element no strided sample would have read
    mask.reshape(-1)[idx] = 0
    d2 = wg.vis2dirty(vis=c["vis"], **kw)
    g, gkw, _ = gpu_plan(c, mask=mask)
    ref = g.vis2dirty(c["vis"], wgt)
    g.close()
    assert rel(d2, ref) < 1e-12 and rel(d2, d1) > 1e-9
    # weights: hessian_slice binds them to the cached plan; an in-place change of one weight must rebind
    hkw = dict(uvw=c["uvw"], weight=wgt, vis_mask=mask, freq=c["freq"], beam=None, cell=c["cell"], x0=0.0, y0=0.0,
               do_wgridding=True, epsilon=1e-7)
    h1 = hessian_slice(c["x"][:48, :48], **hkw)
    wgt.reshape(-1)[idx + 1] *= 3.0
    h2 = hessian_slice(c["x"][:48, :48], **hkw)
    g, gkw, _ = gpu_plan(c, mask=mask)
    g.set_weights(wgt)
    ref = g.hessian(np.ascontiguousarray(c["x"][:48, :48]))
    g.close()
    assert rel(h2, ref) < 1e-12 and rel(h2, h1) > 1e-9
    wg.clear_cache()


@pytest.mark.parametrize("K, widen, eps, npix", [(2, 30.0, 1e-4, 64), (3, 130.0, 1e-7, 64), (4, 200.0, 1e-7, 64), (3, 60.0, 1e-9, 96),
                                                 (3, 16.0, 1e-7, 512)])
def test_one_plane_w_scheme(K, widen, eps, npix, monkeypatch):
    """wmode 2 (round 4): ONE uv-plane, the rest of the w-term carried by differentiated gridding kernels
    (csrc/gridder_kernels_wd.hpp) -- against the direct DFT (epsilon), against its CPU restatement run with the plan's
    parameters (1e-10), against the polynomial-plane scheme it replaces (epsilon), in both directions and inside the fused
    Hessian apply; the 512^2 case has enough work items for the four colour launches, the others run the single launch."""
    c = synth.make_case(2500 if npix < 512 else 60000, 2, npix, zscale=1e-3, seed=5)
    cell = c["cell"] * widen
    if npix >= 512:
        monkeypatch.setenv("PFBHIP_WD_COLOURS", "1")  # the benchmark's four colour launches (plain tile flush) at a test's size
    nx, ny = npix, npix - 4
    x = np.ascontiguousarray(c["x"][:, :ny])
    g, kw, mask = gpu_plan(c, npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell * 1.1, epsilon=eps, force_wmode=2)
    assert g.info["wmode"] == 2 and g.info["nplanes"] == 1 and K <= g.info["nderiv"] <= 4, g.info
    assert g.info["scatter_launches"] == (4 if npix >= 512 else 1)
    K = g.info["nderiv"]
    o = oracle_plan(c, g, kw, mask)
    d = g.vis2dirty(c["vis"], c["wgt"])
    assert rel(d, o.vis2dirty(c["vis"], c["wgt"])) < 1e-10
    sub = dict(pixels=None) if npix < 512 else dict(pixels=(np.arange(0, nx * ny, 37) // ny, np.arange(0, nx * ny, 37) % ny))
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], nx, ny, cell, cell * 1.1, 0, 0, False, True, False, True,
                            False, **sub)
    got = d if npix < 512 else d[sub["pixels"]]
    assert rel(got, ref) < eps
    v = g.dirty2vis(x)
    assert rel(v, o.dirty2vis(x)) < 1e-10
    rows = None if npix < 512 else np.arange(0, c["uvw"].shape[0], 29)
    if rows is None:
        refv = dft.dft_dirty2vis(c["uvw"], c["freq"], x, cell, cell * 1.1, 0, 0, False, True, False, True, False)
        refv[c["mask"] == 0] = 0
        assert rel(v, refv) < eps
    else:
        refv = dft.dft_dirty2vis(c["uvw"], c["freq"], x, cell, cell * 1.1, 0, 0, False, True, False, True, False, rows=rows,
                                 chans=np.zeros_like(rows))
        assert rel(v[rows, 0] * 1.0, refv * c["mask"][rows, 0]) < eps
    # fused Hessian apply (the gather's epilogue writes the scatter's K values per visibility) == the two halves
    g.set_weights(c["wgt"])
    beam = 0.5 + np.random.default_rng(2).random((nx, ny))
    h = g.hessian(x, beam=beam, eta=0.3, wsum=7.0)
    mv = g.dirty2vis(beam * x)
    two = beam * g.vis2dirty(mv, c["wgt"]) / 7.0 + 0.3 * x
    assert rel(h, two) < 1e-10  # (the model visibilities stay in sorted order on the device: other rounding, same arithmetic)
    # the scheme it replaces, and the switch that disables it
    g1, _, _ = gpu_plan(c, npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell * 1.1, epsilon=eps, force_wmode=1)
    assert g1.info["wmode"] == 1 and g1.info["nplanes"] == K
    assert rel(d, g1.vis2dirty(c["vis"], c["wgt"])) < eps
    g1.close()
    g.close()
    monkeypatch.setenv("PFBHIP_WMODE2", "0")
    g0, _, _ = gpu_plan(c, npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell * 1.1, epsilon=eps)
    assert g0.info["wmode"] == 1
    g0.close()
    monkeypatch.delenv("PFBHIP_WMODE2")
    # off-axis phase centres keep the polynomial planes
    g2, _, _ = gpu_plan(c, npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell * 1.1, epsilon=eps, center_x=1e-4)
    assert g2.info["wmode"] in (0, 1)
    g2.close()


@pytest.mark.parametrize("env", [dict(PFBHIP_WD_COLOURS="1", PFBHIP_WD_CHUNK="64"), dict(PFBHIP_WD_COLOURS="0", PFBHIP_WD_CHUNK="96"),
                                 dict(PFBHIP_CHUNK="256", PFBHIP_WD_COLOURS="1")])
def test_one_plane_work_item_sizes(env, monkeypatch):
    """The one-plane scheme sizes its work items to the launch (csrc/gridder.hip: the gather's items, and the scatter's colour
    lists cut finer from them, parts of one tile flagged shared -> atomic tile flush): however the lists are cut, the results are
    those of the default cut up to the order of the additions."""
    c = synth.make_case(60000, 2, 512, zscale=1e-3, seed=8)
    cell = c["cell"] * 16.0
    x = np.ascontiguousarray(c["x"])

    def run():
        g, kw, mask = gpu_plan(c, npix_x=512, npix_y=512, pixsize_x=cell, pixsize_y=cell, epsilon=1e-7, force_wmode=2)
        assert g.info["wmode"] == 2
        g.set_weights(c["wgt"])
        out = g.vis2dirty(c["vis"], c["wgt"]), g.dirty2vis(x), g.hessian(x, eta=0.1, wsum=3.0), g.info["scatter_launches"]
        g.close()
        return out

    d0, v0, h0, _ = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    d1, v1, h1, nl = run()
    assert nl == (4 if env.get("PFBHIP_WD_COLOURS") == "1" else 1)
    # (the order of the additions into a cell changes with the cut; the image-side correction amplifies that rounding: 6e-12 seen)
    assert rel(d1, d0) < 1e-10 and rel(v1, v0) < 1e-10 and rel(h1, h0) < 1e-10


@pytest.mark.parametrize("W, sigma", [(13, 1.5), (14, 1.4), (15, 1.3), (16, 1.25)])
@pytest.mark.parametrize("colours", ["0", "1"])
def test_one_plane_scatter_frames(W, sigma, colours, monkeypatch):
    """The one-plane scatter's register frame (csrc/gridder_kernels_wd.hpp): 16 x 16 cells on 4 x 16 lanes for W <= 15 -- anchored on the
    4 x 4-cell blocks of the tile sort up to W = 13, on 2 x 2-cell blocks (finer sort key) for W = 14, 15 --, 3 x 20 lanes for W = 16 and
    under PFBHIP_WD_BLOCK=4.  Every form against the CPU restatement run with the plan's parameters (1e-10), against the direct DFT
    (epsilon), and the two anchorings of one plan against each other (summation order only)."""
    c = synth.make_case(60000, 2, 512, zscale=1e-3, seed=11)
    cell = c["cell"] * 16.0
    monkeypatch.setenv("PFBHIP_WD_COLOURS", colours)
    pix = (np.arange(0, 512 * 512, 41) // 512, np.arange(0, 512 * 512, 41) % 512)
    ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], 512, 512, cell, cell, 0, 0, False, True, False, True, False,
                            pixels=pix)

    def run():
        g, kw, mask = gpu_plan(c, npix_x=512, npix_y=512, pixsize_x=cell, pixsize_y=cell, epsilon=1e-7, force_wmode=2, force=(sigma, W))
        assert g.info["wmode"] == 2 and g.info["W"] == W, g.info
        o = oracle_plan(c, g, kw, mask)
        d = g.vis2dirty(c["vis"], c["wgt"])
        assert rel(d, o.vis2dirty(c["vis"], c["wgt"])) < 1e-10
        assert rel(d[pix], ref) < 1e-7
        g.set_weights(c["wgt"])
        h = g.hessian(np.ascontiguousarray(c["x"]), eta=0.1, wsum=3.0)
        g.close()
        return d, h

    d0, h0 = run()
    monkeypatch.setenv("PFBHIP_WD_BLOCK", "4")
    d1, h1 = run()
    assert rel(d1, d0) < 1e-10 and rel(h1, h0) < 1e-10


def test_apply_graph_replay(monkeypatch):
    """PFBHIP_GRAPH=1: the Hessian apply replayed from a captured hipGraph (csrc/gridder.hip: hessian_dev_impl): the first call with
    a set of buffers runs eagerly, the second captures, later ones replay -- with new contents of x, new weights, and in the
    device CG loop; another output buffer or another eta gets a graph of its own; results equal the eager path (to the
    summation order of the small plan's atomic tile flush)."""
    from pfb_imaging_amd._lib import DeviceArray

    c = make(nrow=4000, npix=64, widen=20.0, zscale=1e-3)
    nx = c["nx"] = c["ny"] = 1100  # (the hand-written row FFT path: plans on rocFFT are not captured)
    c["cell"] = c["cell"] * 64.0 / nx
    c["x"] = np.random.default_rng(4).standard_normal((nx, nx))
    monkeypatch.setenv("PFBHIP_GRAPH", "1")
    g, kw, mask = gpu_plan(c)
    g.set_weights(c["wgt"])
    g.hessian(c["x"])  # (the mode is read at the first apply)
    monkeypatch.setenv("PFBHIP_GRAPH", "0")
    g0, _, _ = gpu_plan(c)
    g0.set_weights(c["wgt"])
    rng = np.random.default_rng(5)
    xd, od, od2 = DeviceArray((nx, nx), np.float64), DeviceArray((nx, nx), np.float64), DeviceArray((nx, nx), np.float64)
    rd = DeviceArray((nx, nx), np.float64)
    for it in range(5):
        x = rng.standard_normal((nx, nx))
        xd.upload(x)
        g.hessian_dev(xd, od, eta=0.2, wsum=3.0)
        g0.hessian_dev(xd, rd, eta=0.2, wsum=3.0)
        assert rel(od.download(), rd.download()) < 1e-10, it
    assert g.refresh_info()["graph_replays"] == 4 and g0.refresh_info()["graph_replays"] == 0
    # new weights: same graph, new result
    w2 = c["wgt"] * (1.0 + rng.random(c["wgt"].shape))
    g.set_weights(w2)
    g0.set_weights(w2)
    g.hessian_dev(xd, od, eta=0.2, wsum=3.0)
    g0.hessian_dev(xd, rd, eta=0.2, wsum=3.0)
    assert rel(od.download(), rd.download()) < 1e-10 and g.refresh_info()["graph_replays"] == 5
    # another output buffer / another eta: their own graphs
    for _ in range(3):
        g.hessian_dev(xd, od2, eta=0.2, wsum=3.0)
        g.hessian_dev(xd, od, eta=0.0, wsum=3.0)
    g0.hessian_dev(xd, rd, eta=0.0, wsum=3.0)
    assert rel(od.download(), rd.download()) < 1e-10
    assert g.refresh_info()["graph_replays"] == 5 + 2 * 2
    # the host-array entry and the device CG go through the same path
    h = g.hessian(c["x"], eta=0.1, wsum=2.0)
    h = g.hessian(c["x"], eta=0.1, wsum=2.0)
    assert rel(h, g0.hessian(c["x"], eta=0.1, wsum=2.0)) < 1e-10
    rhs = g0.hessian(c["x"], eta=0.5, wsum=2.0)
    sol = g.cg(rhs, eta=0.5, wsum=2.0, tol=1e-10, maxit=50, minit=1)
    sol0 = g0.cg(rhs, eta=0.5, wsum=2.0, tol=1e-10, maxit=50, minit=1)
    assert g.last_cg["iters"] == g0.last_cg["iters"] and rel(sol, sol0) < 1e-5  # (50 iterations of an ill-conditioned solve: rounding order)
    assert g.refresh_info()["graph_replays"] > 12
    for d in (xd, od, od2, rd):
        d.free()
    g.close()
    g0.close()
