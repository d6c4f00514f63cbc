"""GPU tests of the reference callables around the hot path: vis2im / im2vis /
compute_residual arithmetic (operators/gridder.py:37-144, 1019-1148), fft2d / fft_cube (operators/fft.py:9-61), the
imaging-weight chain and the array-level image_data_products (operators/gridder.py:375-757), the pcg family
(opt/pcg.py:88-630).  Everything goes through the C-ABI; the checker is the oracle (DFT / algorithm restatement)."""

from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dft  # noqa: E402
from oracle import weighting as ow  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def case(nrow=2500, nchan=2, npix=48, seed=11, widen=30.0):
    c = synth.make_case(nrow, nchan, npix, zscale=0.2, seed=seed)
    c["cell"] *= widen
    return c


def dft_dirty(c, vis, wgt, nx, ny, l0=0.0, m0=0.0):
    return dft.dft_vis2dirty(c["uvw"], c["freq"], vis, wgt, c["mask"], nx, ny, c["cell"], c["cell"], -l0, -m0, False, True, False,
                             True, False)


@pytest.mark.parametrize("precision", ["double", "single"])
def test_vis2im_precision_and_coercions(precision):
    """vis2im coerces with np.require (gridder.py:58-76): float32 uvw / complex64 vis / bool mask are accepted, and the
    output precision follows ``precision`` (float32 image for "single").  Checked against the DFT."""
    from pfb_imaging_amd.operators.gridder import vis2im

    c = case()
    l0, m0 = 0.002, -0.001
    uvw32 = c["uvw"].astype(np.float32)  # (coerced back to float64: what is gridded is the float32-rounded uvw)
    args = (uvw32, c["freq"].astype(np.float32).astype(np.float64), c["vis"].astype(np.complex64), c["wgt"].astype(np.float32),
            c["mask"].astype(bool), c["nx"], c["ny"], c["cell"], c["cell"], l0, m0, 1e-6, precision, True, False, 1, 1.1, 3.0, True)
    got = vis2im(*args)
    assert got.dtype == (np.float64 if precision == "double" else np.float32) and got.shape == (c["nx"], c["ny"])
    cc = dict(c, uvw=uvw32.astype(np.float64))
    vis = c["vis"].astype(np.complex64).astype(np.complex128)
    wgt = c["wgt"].astype(np.float32).astype(np.float64)
    ref = dft_dirty(cc, vis, wgt, c["nx"], c["ny"], l0, m0)
    assert rel(got, ref) < (1e-6 if precision == "double" else 2e-6)
    with pytest.raises(ValueError):
        vis2im(*(args[:12] + ("half",) + args[13:]))


def test_im2vis_band_to_channel_slices():
    """im2vis degrids band i into its channel slice freq_bin_idx[i] : + freq_bin_counts[i] (gridder.py:103-144)."""
    from pfb_imaging_amd.operators.gridder import im2vis

    c = case(nrow=1200, nchan=6)
    rng = np.random.default_rng(2)
    image = rng.standard_normal((3, c["nx"], c["ny"]))
    idx, cnt = np.array([10, 12, 15]), np.array([2, 3, 1])  # offsets are relative to idx.min()
    vis = im2vis(c["uvw"], c["freq"], image, c["cell"], c["cell"], idx, cnt, l0=0.0, m0=0.0, epsilon=1e-7)
    assert vis.shape == (1200, 6) and vis.dtype == np.complex128
    for b, (lo, n) in enumerate(zip(idx - idx.min(), cnt)):
        ref = dft.dft_dirty2vis(c["uvw"][:300], c["freq"][lo:lo + n], image[b], c["cell"], c["cell"], 0.0, 0.0, False, True, False,
                                True, False)
        assert rel(vis[:300, lo:lo + n], ref) < 1e-7


def test_compute_residual_arrays_beam_once():
    """residual = dirty - R^H W R (beam * model): the beam enters once, on the degrid side (gridder.py:1070-1117)."""
    from pfb_imaging_amd.operators.gridder import compute_residual_arrays

    c = case()
    rng = np.random.default_rng(5)
    ncorr = 2
    nx, ny = c["nx"], c["ny"]
    model = rng.standard_normal((ncorr, nx, ny))
    beam = 0.5 + rng.random((ncorr, nx, ny))
    wgt = np.stack([c["wgt"], c["wgt"][::-1]])
    dirty = rng.standard_normal((ncorr, nx, ny)) * 1e3
    got = compute_residual_arrays(dirty, model, c["uvw"], c["freq"], wgt, c["mask"], beam, c["cell"])
    for k in range(ncorr):
        mv = dft.dft_dirty2vis(c["uvw"], c["freq"], beam[k] * model[k], c["cell"], c["cell"], 0.0, 0.0, False, True, False, True, False)
        conv = dft_dirty(c, mv, wgt[k], nx, ny)
        assert rel(dirty[k] - got[k], conv) < 1e-7   # compare the convolved image, not the (dirty-dominated) residual
    # zero model: the residual is the dirty image
    assert np.array_equal(compute_residual_arrays(dirty, np.zeros_like(model), c["uvw"], c["freq"], wgt, c["mask"], beam, c["cell"]),
                          dirty)


def test_fft2d_fft_cube_match_numpy():
    from pfb_imaging_amd.operators.fft import fft2d, fft_cube

    rng = np.random.default_rng(1)
    psf = rng.standard_normal((3, 96, 80))
    ref = np.fft.rfft2(np.fft.ifftshift(psf, axes=(1, 2)), axes=(1, 2))
    got = fft_cube(psf)
    assert got.shape == (3, 96, 41) and got.dtype == np.complex128 and rel(got, ref) < 1e-13
    one = fft2d(psf[1])
    assert one.shape == (96, 41) and rel(one, ref[1]) < 1e-13
    assert fft2d(psf[1].astype(np.float32)).dtype == np.complex64
    with pytest.raises(ValueError):
        fft2d(psf)
    # odd lengths: the shift is not a sign pattern on the spectrum (host ifftshift in front of the transform)
    odd = rng.standard_normal((2, 45, 38))
    assert rel(fft_cube(odd), np.fft.rfft2(np.fft.ifftshift(odd, axes=(1, 2)), axes=(1, 2))) < 1e-13


def test_imaging_weight_chain_equals_the_steps():
    """pfbhip_imaging_weights (one device pipeline) == _compute_counts -> filter_extreme_counts -> box_sum_counts ->
    counts_to_weights called one after the other, and == the oracle restatement of the numba kernels."""
    from pfb_imaging_amd.utils.weighting import (_compute_counts, box_sum_counts, counts_to_weights, filter_extreme_counts,
                                                 imaging_weights)

    c = synth.make_case(6000, 3, 96, seed=4)
    rng = np.random.default_rng(0)
    wgt0 = np.exp(rng.standard_normal((2,) + c["mask"].shape))
    nxp = nyp = 164
    for robust, level, sup in ((0.0, 5.0, 0), (-1.0, 10.0, 2), (-3.0, 0.0, 1)):
        args = (nxp, nyp, c["cell"], c["cell"])
        counts = _compute_counts(c["uvw"], c["freq"], c["mask"], wgt0, *args, np.float64, usign=-1.0, vsign=1.0)
        counts = filter_extreme_counts(counts, level=level)
        counts = box_sum_counts(counts, sup)
        ref_counts = counts.copy()
        ref = counts_to_weights(counts, c["uvw"], c["freq"], wgt0.copy(), c["mask"], *args, robust, usign=-1.0, vsign=1.0)
        got, gcounts = imaging_weights(c["uvw"], c["freq"], c["mask"], wgt0.copy(), *args, robust, filter_level=level,
                                       npix_super=sup, usign=-1.0, vsign=1.0, return_counts=True)
        np.testing.assert_allclose(got, ref, rtol=1e-11)
        np.testing.assert_allclose(gcounts, counts, rtol=1e-11)   # (counts_to_weights scaled `counts` in place)
        # oracle: the reference's loops restated
        oc = ow.compute_counts(c["uvw"], c["freq"], c["mask"], wgt0, *args, usign=-1.0, vsign=1.0)
        np.testing.assert_allclose(ref_counts if (not level and not sup) else oc, oc, rtol=1e-11)
    # natural weighting / empty counts leave the weights alone
    w = wgt0.copy()
    assert np.array_equal(imaging_weights(c["uvw"], c["freq"], np.zeros_like(c["mask"]), w, nxp, nyp, c["cell"], c["cell"], 0.0,
                                          usign=-1.0, vsign=1.0), wgt0)


def test_image_data_products_arrays():
    """The product chain of image_data_products on arrays: WEIGHT (Briggs on the 1.7x-padded grid), WSUM, DIRTY, PSF,
    PSFHAT = r2c(ifftshift(PSF)), RESIDUAL = R^H W (vis - R model); checked against the DFT and the reference's
    identities (uniform recount == 1: tests/test_weighting.py:47-118; PSF peak == wsum)."""
    from pfb_imaging_amd.operators.gridder import image_data_products_arrays
    from pfb_imaging_amd.utils.weighting import _compute_counts

    c = case(nrow=3000, nchan=2, npix=40)
    rng = np.random.default_rng(9)
    ncorr, nx, ny = 2, c["nx"], c["ny"]
    vis = np.stack([c["vis"], c["vis"].conj()])
    wgt = np.stack([c["wgt"], np.ones_like(c["wgt"])])
    model = rng.standard_normal((ncorr, nx, ny))
    kw = dict(model=model, epsilon=1e-7, do_noise=True, do_beam=True, rng=np.random.default_rng(1))
    # natural weights
    prod, out = image_data_products_arrays(c["uvw"], c["freq"], vis, wgt, c["mask"], nx, ny, 2 * nx, 2 * ny, c["cell"], c["cell"],
                                           robustness=None, **kw)
    assert np.array_equal(prod["WEIGHT"], wgt) and np.allclose(prod["WSUM"], wgt[:, c["mask"] != 0].sum(axis=1))
    for k in range(ncorr):
        assert rel(prod["DIRTY"][k], dft_dirty(c, vis[k], wgt[k], nx, ny)) < 1e-7
        mv = dft.dft_dirty2vis(c["uvw"], c["freq"], model[k], c["cell"], c["cell"], 0.0, 0.0, False, True, False, True, False)
        assert rel(prod["DIRTY"][k] - prod["RESIDUAL"][k], dft_dirty(c, mv, wgt[k], nx, ny)) < 1e-7
        assert abs(prod["PSF"][k, nx, ny] - prod["WSUM"][k]) < 1e-6 * prod["WSUM"][k]   # PSF peak at the centre pixel
    assert prod["PSF"].shape == (ncorr, 2 * nx, 2 * ny) and prod["PSFHAT"].shape == (ncorr, 2 * nx, ny + 1)
    assert rel(prod["PSFHAT"], np.fft.rfft2(np.fft.ifftshift(prod["PSF"], axes=(1, 2)), axes=(1, 2))) < 1e-13
    assert out["residual"] is prod["RESIDUAL"] and out["psf"] is prod["PSF"] and np.array_equal(out["wsum"], prod["WSUM"])
    assert prod["NOISE"].shape == (ncorr, nx, ny) and np.isfinite(prod["NOISE"]).all() and (prod["BEAM"] == 1).all()
    # uniform weighting (robustness <= -2): re-counting the imaging weights gives 1 in every occupied uv-cell
    prod_u, _ = image_data_products_arrays(c["uvw"], c["freq"], vis, np.ones_like(wgt), c["mask"], nx, ny, 2 * nx, 2 * ny, c["cell"],
                                           c["cell"], robustness=-3, filter_counts_level=0.0, do_psf=False, do_residual=False,
                                           do_dirty=False)
    nxp = int(np.ceil(1.7 * nx)) + int(np.ceil(1.7 * nx)) % 2
    recount = _compute_counts(c["uvw"], c["freq"], c["mask"], prod_u["WEIGHT"], nxp, nxp, c["cell"], c["cell"], np.float64,
                              usign=-1.0, vsign=1.0)
    assert np.allclose(recount[recount > 0], 1.0, rtol=1e-10)
    # no model: the returned residual is the dirty image; l2 reweighting without a model is an error
    prod_n, out_n = image_data_products_arrays(c["uvw"], c["freq"], vis, wgt, c["mask"], nx, ny, 2 * nx, 2 * ny, c["cell"], c["cell"],
                                               do_psf=False)
    assert out_n["residual"] is prod_n["DIRTY"] and "RESIDUAL" not in prod_n
    with pytest.raises(ValueError):
        image_data_products_arrays(c["uvw"], c["freq"], vis, wgt, c["mask"], nx, ny, 2 * nx, 2 * ny, c["cell"], c["cell"],
                                   l2_reweight_dof=5)


def test_pcg_family_over_the_device_cg():
    """pcg / pcg_numba / PCG / pcg_dds (opt/pcg.py): the exact-Hessian operator built as pcg_dds builds it
    (partial(hessian_slice, ...)) is solved on the device; x0 is the iterate, mutated in place; any other callable runs
    the reference's loop on the host; both agree with each other and with the oracle's restatement of pcg_numba."""
    from oracle.fftconv import pcg as pcg_oracle
    from pfb_imaging_amd import opt
    from pfb_imaging_amd.operators.hessian import hessian_slice
    from pfb_imaging_amd.wgridder import Gridder, clear_cache

    clear_cache()
    c = case(nrow=2500, npix=32, widen=20.0)
    rng = np.random.default_rng(7)
    nx, ny = c["nx"], c["ny"]
    beam = 0.7 + 0.3 * rng.random((nx, ny))
    wsum = float(c["wgt"][c["mask"] != 0].sum())
    eta = 0.1
    hess = partial(hessian_slice, uvw=c["uvw"], weight=c["wgt"], vis_mask=c["mask"], freq=c["freq"], beam=beam, cell=c["cell"],
                   x0=0.0, y0=0.0, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, epsilon=1e-7, double_accum=True,
                   nthreads=1, eta=eta, wsum=wsum)
    b = hess(c["x"])
    x0 = np.zeros_like(b)
    sol = opt.pcg_numba(hess, b, x0=x0, tol=1e-9, maxit=200, minit=1, verbosity=0)
    assert sol is x0 and opt._cg_host.last["where"] == "device"          # in place, on the device
    assert rel(sol, c["x"]) < 1e-6
    host = opt.pcg_numba(lambda v: hess(v), b, tol=1e-9, maxit=200, minit=1, verbosity=0)   # a bare callable: host loop
    assert opt._cg_host.last["where"] == "host" and rel(host, sol) < 1e-6
    ref = pcg_oracle(lambda v: hess(v), b, tol=1e-9, maxit=200, minit=1)
    assert rel(sol, ref) < 1e-6
    # same iteration count and stopping rule as the reference's loop (relative change of the iterate)
    opt.pcg_numba(hess, b, tol=1e-4, maxit=200, minit=1, verbosity=0)
    it_dev = opt._cg_host.last["iters"]
    opt.pcg_numba(lambda v: hess(v), b, tol=1e-4, maxit=200, minit=1, verbosity=0)
    assert abs(it_dev - opt._cg_host.last["iters"]) <= 1
    # preconditioned: host loop around both callables; return_resid
    xs, r = opt.pcg(hess, b, precond=lambda v: v / (1.0 + eta), tol=1e-9, maxit=200, minit=1, verbosity=0, return_resid=True)
    assert opt._cg_host.last["where"] == "host" and rel(xs, c["x"]) < 1e-6 and rel(r + b, hess(xs)) < 1e-9
    # bound Gridder.hessian, and the PCG solver class over an object with .cg
    g = Gridder(c["uvw"], c["freq"], c["mask"], npix_x=nx, npix_y=ny, pixsize_x=c["cell"], pixsize_y=c["cell"], epsilon=1e-7,
                flip_v=True, do_wgridding=True, divide_by_n=False)
    g.set_weights(c["wgt"])
    s2 = opt.pcg_numba(partial(g.hessian, beam=beam, eta=eta, wsum=wsum), b, tol=1e-9, maxit=200, minit=1, verbosity=0)
    assert opt._cg_host.last["where"] == "device" and rel(s2, sol) < 1e-8

    class Op:
        def cg(self, rhs, x0=None, tol=None, maxit=None, minit=None):
            return g.cg(rhs, x0=x0, beam=beam, eta=eta, wsum=wsum, tol=tol, maxit=maxit, minit=minit)

    assert rel(opt.PCG(tol=1e-9, maxit=200).solve(Op(), b), sol) < 1e-8
    g.close()
    # pcg_dds on an in-memory band dataset: x solves (H + eta) x = beam * residual / wsum, model += x, exact residual
    dirty = rng.standard_normal((nx, ny)) * wsum * 1e-2
    ds = dict(DIRTY=dirty, BEAM=beam, UVW=c["uvw"], WEIGHT=c["wgt"], MASK=c["mask"], FREQ=c["freq"], cell_rad=c["cell"], x0=0.0,
              y0=0.0, flip_u=False, flip_v=True, flip_w=False, wsum=wsum, bandid=3)
    resid, bandid, fields = opt.pcg_dds(ds, eta, mask=1.0, epsilon=1e-7, tol=1e-8, maxit=300, verbosity=0)
    assert bandid == 3 and set(fields) == {"MODEL_MOPPED", "RESIDUAL_MOPPED", "UPDATE", "X0"}
    x = fields["UPDATE"]
    assert rel(hess(x), dirty * beam / wsum) < 1e-5
    plain = partial(hessian_slice, uvw=c["uvw"], weight=c["wgt"], vis_mask=c["mask"], freq=c["freq"], beam=beam, cell=c["cell"],
                    do_wgridding=True, epsilon=1e-7)
    assert rel(resid, dirty - plain(fields["MODEL_MOPPED"])) < 1e-10 and ds["RESIDUAL_MOPPED"] is fields["RESIDUAL_MOPPED"]
    clear_cache()


def test_single_precision_boundary():
    """precision = "single" end to end (round 4): complex64 visibilities, float32 weights / images cross PCIe as they are
    (pfbhip_gridder_*_sp: widened on the device, sums in double = ducc0's double_precision_accumulation); results come back
    in single precision and equal the double path on the float32-rounded inputs to float32 rounding of the result."""
    from pfb_imaging_amd.operators.hessian import hessian_slice
    from pfb_imaging_amd.wgridder import Gridder, dirty2vis, vis2dirty

    c = case()
    nx, ny = c["nx"], c["ny"]
    kw = dict(npix_x=nx, npix_y=ny, pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0, epsilon=1e-6, flip_u=False,
              flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False)
    g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
    v32, w32, x32 = c["vis"].astype(np.complex64), c["wgt"].astype(np.float32), c["x"].astype(np.float32)
    d32 = g.vis2dirty(v32, w32)
    assert d32.dtype == np.float32 and d32.shape == (nx, ny)
    d64 = g.vis2dirty(v32.astype(np.complex128), w32.astype(np.float64))
    assert d64.dtype == np.float64 and rel(d32, d64) < 2e-7            # one rounding of the result to float32
    m32 = g.dirty2vis(x32, w32)
    assert m32.dtype == np.complex64
    assert rel(m32, g.dirty2vis(x32.astype(np.float64), w32.astype(np.float64))) < 2e-7
    # mixed precisions keep the double path (what ducc0 would refuse; here: the wider type wins)
    assert g.vis2dirty(v32, c["wgt"]).dtype == np.float64
    g.set_weights(w32)
    h32 = g.hessian(x32, eta=0.1, wsum=5.0)
    assert h32.dtype == np.float32
    g.set_weights(w32.astype(np.float64))
    assert rel(h32, g.hessian(x32.astype(np.float64), eta=0.1, wsum=5.0)) < 2e-7
    out = np.empty((nx, ny), np.float32)
    assert g.hessian(x32, eta=0.1, wsum=5.0, out=out) is out and np.array_equal(out, h32)
    with pytest.raises(ValueError):
        g.hessian(x32, out=np.empty((nx, ny), np.float64))
    g.close()
    # the ducc0-signature callables and hessian_slice follow the input precision
    sk = dict(uvw=c["uvw"], freq=c["freq"], mask=c["mask"], pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0,
              epsilon=1e-6, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False)
    assert vis2dirty(vis=v32, wgt=c["wgt"], npix_x=nx, npix_y=ny, **sk).dtype == np.float32
    assert rel(vis2dirty(vis=v32, wgt=w32, npix_x=nx, npix_y=ny, **sk), d64) < 2e-7
    assert dirty2vis(dirty=x32, **sk).dtype == np.complex64
    xo = np.empty((nx, ny), np.float32)
    hs = hessian_slice(x32, xout=xo, uvw=c["uvw"], weight=w32, vis_mask=c["mask"], freq=c["freq"], cell=c["cell"], epsilon=1e-6,
                       eta=0.1, wsum=5.0)
    assert hs is xo and rel(xo, h32) < 1e-6
