"""GPU parity tests of the FFT / PSF-convolution operator family and the band-worker pool.

Oracle: oracle.fftconv (numpy rfft2 / irfft2 restatement of the reference's operators).
Tolerance: 1e-11 relative (double-precision FFTs of different factorisations), the reference's
own tests use rtol 1e-12..1e-5 (test_hess_tree_ray.py:22-80, test_hessian_tree.py:20-66).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fftconv  # noqa: E402


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def _psf_case(nband=2, nx=48, ny=40, nxp=96, nyp=80, seed=0):
    rng = np.random.default_rng(seed)
    psf = rng.standard_normal((nband, nxp, nyp))
    psfhat = np.fft.rfft2(np.fft.ifftshift(psf, axes=(1, 2)), axes=(1, 2))
    x = rng.standard_normal((nband, nx, ny))
    beam = 0.5 + rng.random((nband, nx, ny))
    return psf, psfhat, np.abs(psfhat), x, beam


def test_r2c_c2r_good_size():
    from pfb_imaging_amd.fft import c2r, good_size, r2c

    rng = np.random.default_rng(1)
    a = rng.standard_normal((3, 36, 50))
    ah = r2c(a, axes=(1, 2), forward=True, inorm=0)
    assert ah.shape == (3, 36, 26)
    assert rel(ah, np.fft.rfft2(a, axes=(1, 2))) < 1e-13
    back = c2r(ah, axes=(1, 2), forward=False, lastsize=50, inorm=2, allow_overwriting_input=True)
    assert rel(back, a) < 1e-13
    out = np.empty_like(ah)
    assert r2c(a, axes=(-2, -1), out=out) is out
    assert good_size(11468) == 11520 and good_size(127) == 128 and good_size(1000, True) == 1000
    # the other forms ducc0 offers for two-axis transforms: any pair of axes, both signs, inorm 0 / 1 / 2
    ref01 = np.fft.rfftn(a, axes=(2, 0))                      # transform over axes (2, 0): axis 0 becomes half-complex
    got01 = r2c(a, axes=(2, 0), forward=True, inorm=0)
    assert got01.shape == ref01.shape == (2, 36, 50) and rel(got01, ref01) < 1e-13
    assert rel(r2c(a, axes=(1, 2), forward=False, inorm=1), np.conj(np.fft.rfft2(a, axes=(1, 2))) / np.sqrt(36 * 50)) < 1e-13
    assert rel(c2r(got01, axes=(2, 0), forward=False, lastsize=3, inorm=2), a) < 1e-13
    assert rel(c2r(np.conj(ah), axes=(1, 2), forward=True, lastsize=50, inorm=0), a * (36 * 50)) < 1e-13
    assert rel(c2r(ah, axes=(1, 2), forward=False, lastsize=50, inorm=1), a * np.sqrt(36 * 50)) < 1e-13
    with pytest.raises(NotImplementedError):
        r2c(a, axes=(0,))
    with pytest.raises(ValueError):
        r2c(a, axes=(1, 2), inorm=3)


def test_psf_convolve_slice_cube_fscube():
    """psf.py:8-96 == crop(irfft2(rfft2(pad(x)) psfhat)) (its jax twin, psf.py:99-104)."""
    from pfb_imaging_amd.operators.psf import psf_convolve_cube, psf_convolve_fscube, psf_convolve_slice

    psf, psfhat, abspsf, x, beam = _psf_case()
    nband, nx, ny = x.shape
    nxp, nyp = psf.shape[1:]
    ref = fftconv.psf_convolve(x, psfhat, nxp, nyp)
    xout = np.zeros((nx, ny))
    r = psf_convolve_slice(np.zeros((nxp, nyp)), np.zeros_like(psfhat[0]), xout, psfhat[0], nyp, x[0])
    assert r is xout and rel(xout, ref[0]) < 1e-11
    cube = np.zeros_like(x)
    psf_convolve_cube(np.zeros((nband, nxp, nyp)), np.zeros_like(psfhat), cube, psfhat, nyp, x)
    assert rel(cube, ref) < 1e-11
    fs = np.zeros((nband, 1, nx, ny))
    psf_convolve_fscube(None, None, fs, psfhat[:, None], nyp, x[:, None])
    assert rel(fs[:, 0], ref) < 1e-11
    # delta PSF -> identity (test_hessian_tree.py:20-33)
    d = np.zeros((nxp, nyp))
    d[0, 0] = 1.0
    dh = np.fft.rfft2(d)
    psf_convolve_slice(None, None, xout, dh, nyp, x[1])
    assert rel(xout, x[1]) < 1e-12


def test_hessian_psf_slice_and_direct():
    from pfb_imaging_amd.operators.hessian import hess_direct_slice, hessian_psf_slice, taperf

    psf, psfhat, abspsf, x, beam = _psf_case()
    nxp, nyp = psf.shape[1:]
    for bm, eta in ((None, None), (beam[0], 0.3), (beam[0], 0.0)):
        got = hessian_psf_slice(x[0], xout=np.zeros_like(x[0]), abspsf=abspsf[0], beam=bm, lastsize=nyp, eta=eta)
        assert rel(got, fftconv.hessian_psf_slice(x[0], abspsf[0], nyp, beam=bm, eta=eta)) < 1e-11
    taper = taperf(x[0].shape, 8)
    assert np.allclose(taper, fftconv.taperf(x[0].shape, 8))
    for mode in ("forward", "backward"):
        got = hess_direct_slice(x[0], xout=np.zeros_like(x[0]), abspsf=abspsf[0], taperxy=taper, lastsize=nyp,
                                eta=2.5, mode=mode)
        assert rel(got, fftconv.hess_direct_slice(x[0], abspsf[0], nyp, taper, 2.5, mode)) < 1e-10


def test_hesspsf_dot_idot():
    """HessPSF (hessian.py:251-436): dot aliases self.xout, idot returns a copy, psf-mode idot
    inverts dot to CG tolerance, protocol conformance."""
    from pfb_imaging_amd.operators import LinearOperator, Preconditioner, require_protocol
    from pfb_imaging_amd.operators.hessian import HessPSF

    psf, psfhat, abspsf, x, beam = _psf_case(nband=2, nx=32, ny=32, nxp=64, nyp=64, seed=3)
    # make the operator well conditioned: PSF = delta + small perturbation
    abspsf = 1.0 + 0.2 * abspsf / abspsf.max()
    eta = np.array([0.1, 0.2])
    h = HessPSF(32, 32, abspsf, beam=beam, eta=eta, cgtol=1e-10, cgmaxit=400)
    require_protocol(h, Preconditioner, "precond")
    assert isinstance(h, LinearOperator)
    out = h.dot(x)
    assert out is h.xout
    ref = fftconv.hess_psf_dot(x, abspsf, 64, beam=beam, eta=eta)
    assert rel(out, ref) < 1e-11
    assert rel(h.hdot(x), ref) < 1e-11
    # 2-D input is promoted for nband == 1
    h1 = HessPSF(32, 32, abspsf[:1], beam=None, eta=0.5)
    assert rel(h1.dot(x[0])[0], fftconv.hess_psf_dot(x[:1], abspsf[:1], 64, None, 0.5)[0]) < 1e-11
    rhs = ref.copy()
    sol = h.idot(rhs, mode="psf")
    assert sol is not h.xout
    assert rel(sol, x) < 1e-6
    d = h.idot(rhs, mode="direct")
    assert d.shape == x.shape and np.isfinite(d).all()
    # the beam division of the direct estimate (hessian.py:381-386, 395-399; on the device since round 4) == the reference's host
    # statement applied to the estimate of the same operator without a beam
    hn = HessPSF(32, 32, abspsf, beam=None, eta=eta)
    dn = hn.idot(rhs, mode="direct")
    want = dn.copy()
    msk = (want > 0) & (beam > h.min_beam)
    want[msk] /= beam[msk] ** 2
    assert msk.any() and (~msk).any() and rel(d, want) < 1e-13
    with pytest.raises(ValueError):
        h.idot(rhs, mode="nonsense")
    with pytest.raises(ValueError):
        h.dot(np.zeros((1, 2, 32, 32)))

def test_precond_hesspsf_cube():
    """The cube-level HessPSF of operators/precond.py (precond.py:12-154): beam required, eta per value / band / pixel, dot aliases
    self.xout, idot is a plain CG from x0 (no direct-estimate start) and inverts dot to CG tolerance."""
    from pfb_imaging_amd.operators import Preconditioner, require_protocol
    from pfb_imaging_amd.operators.precond import HessPSF

    psf, psfhat, abspsf, x, beam = _psf_case(nband=2, nx=32, ny=32, nxp=64, nyp=64, seed=5)
    abspsf = 1.0 + 0.2 * abspsf / abspsf.max()
    with pytest.raises(ValueError):
        HessPSF(32, 32, abspsf, beam=None)
    with pytest.raises(NotImplementedError):
        HessPSF(32, 32, abspsf, beam=beam, memory_greedy=False)
    with pytest.raises(ValueError):
        HessPSF(32, 32, abspsf, beam=beam, eta=1)
    for eta in (0.3, np.array([0.1, 0.2]), 0.05 + np.random.default_rng(1).random((2, 32, 32))):
        h = HessPSF(32, 32, abspsf, beam=beam, eta=eta, cgtol=1e-10, cgmaxit=400, cgverbose=0)
        require_protocol(h, Preconditioner, "precond")
        out = h.dot(x)
        assert out is h.xout
        e = eta if np.ndim(eta) == 3 else np.broadcast_to(np.reshape(np.atleast_1d(eta) * np.ones(2), (2, 1, 1)), x.shape)
        ref = fftconv.hess_psf_dot(x, abspsf, 64, beam=beam, eta=0.0) + e * x
        assert rel(out, ref) < 1e-11 and rel(h.hdot(x), ref) < 1e-11
        sol = h.idot(ref.copy())
        assert sol is not h.xout and rel(sol, x) < 1e-6
        # a start vector is honoured (and not required)
        assert rel(h.idot(ref.copy(), x0=0.9 * x), x) < 1e-6
    h1 = HessPSF(32, 32, abspsf[:1], beam=beam[:1], eta=0.5)
    assert rel(h1.dot(x[0])[0], fftconv.hess_psf_dot(x[:1], abspsf[:1], 64, beam[:1], 0.5)[0]) < 1e-11
    with pytest.raises(ValueError):
        h1.dot(np.zeros((1, 1, 32, 32)))



def test_pcg_psf_per_band_solves():
    """opt.pcg_psf (opt/pcg.py:317-441): per-band CG on beam (PSF (*) (beam x)) + eta x == the numpy restatement's solve."""
    from pfb_imaging_amd.opt import pcg_psf

    psf, psfhat, abspsf, x, beam = _psf_case(nband=2, nx=32, ny=32, nxp=64, nyp=64, seed=5)
    psfhat = 1.0 + 0.2 * psfhat / np.abs(psfhat).max()      # complex: pcg_psf takes the magnitude itself
    eta = np.array([0.1, 0.3])
    rhs = fftconv.hess_psf_dot(x, np.abs(psfhat), 64, beam=beam, eta=eta)
    sol = pcg_psf(psfhat, rhs, np.zeros_like(rhs), beam, 64, 1, eta, dict(tol=1e-11, maxit=400, minit=1, verbosity=0))
    assert sol.shape == x.shape and rel(sol, x) < 1e-7
    one = pcg_psf(psfhat, rhs, None, beam[0], 64, 1, 0.2, dict(tol=1e-9, maxit=300, minit=1), compute=False)   # 2-D beam, float eta
    ref = fftconv.hess_psf_dot(one, np.abs(psfhat), 64, beam=np.tile(beam[:1], (2, 1, 1)), eta=np.array([0.2, 0.2]))
    assert rel(ref, rhs) < 1e-6
    with pytest.raises(ValueError):
        pcg_psf(psfhat, rhs, None, beam[:, :16], 64, 1, eta, {})


def _tree_parts(nx, ny, nxp, nyp, nparts, ncorr, seed, delta=False):
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(nparts):
        if delta:
            psfhat = np.ones((ncorr, nxp, nyp // 2 + 1))
        else:
            psfhat = np.abs(np.fft.rfft2(rng.standard_normal((ncorr, nxp, nyp)), axes=(1, 2)))
        parts.append({"psfhat": psfhat, "beam": 0.5 + rng.random((ncorr, nx, ny)) if not delta else np.ones((ncorr, nx, ny)),
                      "wsum": 1.0 + rng.random(ncorr) if not delta else np.ones(ncorr)})
    return parts


def test_hessian_tree():
    """test_hessian_tree.py:20-66: delta-PSF identity, + eta, two identical partitions == one, 2-D input;
    plus a random multi-partition / multi-correlation case against the numpy restatement."""
    from pfb_imaging_amd.operators.hessian import HessianTree

    nx = ny = 16
    nxp = nyp = 32
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, nx, ny))
    one = _tree_parts(nx, ny, nxp, nyp, 1, 1, 0, delta=True)
    np.testing.assert_allclose(HessianTree(one, nx, ny, nxp, nyp).dot(x), x, atol=1e-6)
    np.testing.assert_allclose(HessianTree(one, nx, ny, nxp, nyp, eta=0.5).dot(x), 1.5 * x, atol=1e-6)
    np.testing.assert_allclose(HessianTree(one + one, nx, ny, nxp, nyp).dot(x), x, atol=1e-6)
    assert HessianTree(one, nx, ny, nxp, nyp).dot(x[0]).shape == (1, nx, ny)
    with pytest.raises(ValueError):
        HessianTree([], nx, ny, nxp, nyp)
    parts = _tree_parts(nx, ny, nxp, nyp, 3, 2, 1)
    x2 = rng.standard_normal((2, nx, ny))
    for wsum in (None, 7.0):
        got = HessianTree(parts, nx, ny, nxp, nyp, eta=0.2, wsum=wsum).dot(x2)
        assert rel(got, fftconv.hessian_tree_dot(x2, parts, nxp, nyp, eta=0.2, wsum=wsum)) < 1e-11
    # wsum override halves the output (test_hess_tree_ray.py:22-33)
    a = HessianTree(parts, nx, ny, nxp, nyp, wsum=1.0).dot(x2)
    b = HessianTree(parts, nx, ny, nxp, nyp, wsum=2.0).dot(x2)
    np.testing.assert_allclose(a, 2 * b, rtol=1e-12)


def test_hess_tree_ray_pool_and_cg():
    """test_hess_tree_ray.py:36-80: cube-level facade == per-band HessianTree == HessPSF (single
    partition); cg solves hess @ update = rhs; pool geometry errors."""
    from pfb_imaging_amd.operators.band_worker import BandWorkerPool
    from pfb_imaging_amd.operators.hessian import HessianTree, HessPSF, HessTreeRay

    nband, nx, ny, nxp, nyp = 3, 16, 16, 32, 32
    rng = np.random.default_rng(4)
    ppb = [_tree_parts(nx, ny, nxp, nyp, 1, 1, 10 + b) for b in range(nband)]
    for b in range(nband):  # well conditioned
        ppb[b][0]["psfhat"] = 1.0 + 0.3 * ppb[b][0]["psfhat"] / ppb[b][0]["psfhat"].max()
    etas = np.array([0.1, 0.2, 0.3])
    wsum_tot = sum(p[0]["wsum"][0] for p in ppb)
    hr = HessTreeRay(ppb, nx, ny, nxp, nyp, etas=etas, wsums=wsum_tot, cg_tol=1e-10, cg_maxit=300)
    x = rng.standard_normal((nband, nx, ny))
    got = hr.dot(x)
    for b in range(nband):
        loc = HessianTree(ppb[b], nx, ny, nxp, nyp, eta=etas[b], wsum=wsum_tot).dot(x[b])[0]
        np.testing.assert_allclose(got[b], loc, rtol=1e-12, atol=1e-13)
    abspsf = np.stack([p[0]["psfhat"][0] for p in ppb]) / wsum_tot
    beam = np.stack([p[0]["beam"][0] for p in ppb])
    hp = HessPSF(nx, ny, abspsf, beam=beam, eta=etas, taper_width=4)
    np.testing.assert_allclose(got, hp.dot(x).copy(), rtol=1e-10, atol=1e-12)
    sol = hr.cg(got, tol=1e-10, maxit=300, minit=1)
    assert rel(sol, x) < 1e-6
    x0 = rng.standard_normal((nband, nx, ny))
    x0c = x0.copy()
    sol2 = hr.cg(got, x0=x0, tol=1e-10, maxit=300, minit=1)
    assert rel(sol2, x) < 1e-6 and np.array_equal(x0, x0c)
    ref = fftconv.pcg(lambda z: fftconv.hessian_tree_dot(z, ppb[0], nxp, nyp, eta=etas[0], wsum=wsum_tot)[0], got[0],
                      tol=1e-10, maxit=300, minit=1)
    assert rel(sol[0], ref) < 1e-6
    with pytest.raises(ValueError):
        HessTreeRay(ppb, nx, ny, nxp, nyp, workers=BandWorkerPool(2))
    with pytest.raises(ValueError):
        HessTreeRay(None, nx, ny, nxp, nyp)
    assert isinstance(hr.get_mem(), list)


def test_band_pool_residual_and_mfs():
    """BandWorkerPool.residual == per-band residual_from_partitions; residual_mfs == sum_b / wsum
    (core/deconv.py:313-321)."""
    from pfb_imaging_amd.operators.band_worker import BandWorkerPool
    from pfb_imaging_amd.operators.gridder import residual_from_partitions

    nband, nx, ny = 2, 16, 16
    rng = np.random.default_rng(8)

    def part(seed):
        r = np.random.default_rng(seed)
        n = 150
        return {"UVW": r.standard_normal((n, 3)) * 100.0, "FREQ": np.array([1.0e9, 1.1e9]),
                "WEIGHT": np.abs(r.standard_normal((1, n, 2))) + 0.1, "MASK": (r.random((n, 2)) > 0.1).astype(np.uint8),
                "BEAM": 0.5 + r.random((1, nx, ny)), "attrs": {"l0": 0.0, "m0": 0.0}}

    parts = [[part(1), part(2)], [part(3)]]
    dirty = rng.standard_normal((nband, 1, nx, ny))
    model = rng.standard_normal((nband, 1, nx, ny))
    pool = BandWorkerPool(nband)
    pool.set_bands(dirty, parts)
    res = pool.residual(model, 1.0e-6)
    assert res.shape == (nband, 1, nx, ny)
    from pfb_imaging_amd.operators.gridder import PartitionResidual

    for b in range(nband):
        np.testing.assert_allclose(res[b], residual_from_partitions(dirty[b], parts[b], model[b], 1.0e-6), rtol=1e-9,
                                   atol=1e-9)
        # the device chain over the partitions (pfbhip_gridder_residual_dev) against the sum of the partitions' host-side applies
        st = PartitionResidual(parts[b], nx, ny, 1.0e-6)
        np.testing.assert_allclose(res[b], dirty[b] - st.convim(model[b]), rtol=1e-9, atol=1e-9)
        st.close()
    mfs = pool.residual_mfs(model, 1.0e-6, wsum=3.0)
    np.testing.assert_allclose(mfs, res.sum(axis=0) / 3.0, rtol=1e-12, atol=1e-12)
    assert pool.init_psi(nx, ny, ["self"], 2) == (nx, ny)  # the wavelet role (tests/test_gpu_psi.py)


@pytest.mark.parametrize("center_offset", [(0.0, 0.0), (0.1, -0.17), (0.2, 0.5), (-0.1, 0.2), (-0.15, -0.2)])
def test_hessian_equals_psf_convolution(center_offset):
    """/root/reference/tests/test_hessian_approx.py:234-307 (test_hessian), same steps, same tolerance:
    hessian_slice(delta) with do_wgridding=False == psf_convolve_slice(delta) with the PSF gridded from
    the phase-ramp visibilities; all through the GPU path."""
    from pfb_imaging_amd.fft import good_size, r2c
    from pfb_imaging_amd.operators.gridder import wgridder_conventions
    from pfb_imaging_amd.operators.hessian import hessian_slice
    from pfb_imaging_amd.operators.psf import psf_convolve_slice
    from pfb_imaging_amd.utils import synth
    from pfb_imaging_amd.wgridder import vis2dirty

    c = synth.make_case(3000, 2, 64, seed=9)
    uvw, freq = c["uvw"], c["freq"]
    nrow, nchan = c["vis"].shape
    nx = ny = 64
    nx_psf = ny_psf = good_size(int(1.5 * nx))
    cell_rad = c["cell"]
    x0, y0 = center_offset
    flip_u, flip_v, flip_w, x0, y0 = wgridder_conventions(x0, y0)
    epsilon = 1e-10
    signu = -1.0 if flip_u else 1.0
    signv = -1.0 if flip_v else 1.0
    signx = -1.0 if flip_u else 1.0
    signy = -1.0 if flip_v else 1.0
    freqfactor = -2j * np.pi * freq[None, :] / 299792458.0
    psf_vis = np.exp(freqfactor * (signu * uvw[:, 0:1] * x0 * signx + signv * uvw[:, 1:2] * y0 * signy))
    x = np.zeros((nx, ny), dtype="f8")
    x[nx // 2, ny // 2] = 1.0
    psf = vis2dirty(uvw=uvw, freq=freq, vis=psf_vis, wgt=None, npix_x=nx_psf, npix_y=ny_psf, pixsize_x=cell_rad,
                    pixsize_y=cell_rad, center_x=x0, center_y=y0, flip_u=flip_u, flip_v=flip_v, flip_w=flip_w,
                    epsilon=epsilon, do_wgridding=False, divide_by_n=False, nthreads=2, verbosity=0)
    psfhat = r2c(np.fft.ifftshift(psf, axes=(0, 1)), axes=(0, 1), nthreads=2, forward=True, inorm=0)
    res1 = hessian_slice(x, uvw=uvw, weight=np.ones((nrow, nchan), dtype="f8"),
                         vis_mask=np.ones((nrow, nchan), dtype=np.uint8), freq=freq, cell=cell_rad, x0=x0, y0=y0,
                         flip_u=flip_u, flip_v=flip_v, flip_w=flip_w, do_wgridding=False, epsilon=epsilon,
                         double_accum=True, nthreads=2)
    res2 = psf_convolve_slice(np.zeros((nx_psf, ny_psf)), np.zeros_like(psfhat), np.zeros_like(x), psfhat, ny_psf, x,
                              nthreads=2)
    scale = np.abs(res2).max()
    diff = (res2 - res1) / scale
    assert np.allclose(1 + diff, 1)


@pytest.mark.parametrize("is_complex,nx,nxp,nyp", [(False, 600, 2048, 1024), (True, 600, 2048, 1024), (False, 601, 2048, 1024),
                                                    (True, 600, 1280, 1536), (False, 602, 3072, 5120),
                                                    (False, 600, 1792, 1152), (True, 600, 1920, 2304)])
def test_psfconv_rowfft_pipeline(is_complex, nx, nxp, nyp, monkeypatch):
    """Power-of-two padded sizes take the three-pass row-FFT pipeline (csrc/psffft.hip); every mode, beam,
    eta and accumulate against numpy, and against the rocFFT fallback (PFBHIP_PSF_ROWFFT=0)."""
    from pfb_imaging_amd.psfconv import PsfConv

    rng = np.random.default_rng(11)
    ny = 520  # (odd nx: the last row has no partner in the paired transforms; 1280 / 1536 / 3072 / 5120: radix-3/5 leads)
    psf = rng.standard_normal((nxp, nyp))
    psfhat = np.fft.rfft2(np.fft.ifftshift(psf))
    ph = psfhat if is_complex else 1.0 + np.abs(psfhat) / np.abs(psfhat).max()
    x = rng.standard_normal((nx, ny))
    beam = 0.5 + rng.random((nx, ny))
    prev = rng.standard_normal((nx, ny))

    def ref(mode, shift, bm, scale, eta, acc):
        xp = np.zeros((nxp, nyp))
        xp[:nx, :ny] = x * (bm if bm is not None else 1.0)
        xh = np.fft.rfft2(xp)
        f = ph if mode == 0 else (ph + shift if mode == 1 else 1.0 / (ph + shift))
        r = np.fft.irfft2(xh * f, s=(nxp, nyp))[:nx, :ny] * (bm if bm is not None else 1.0) * scale + eta * x
        return r + prev if acc else r

    cases = [(0, 0.0, None, 1.0, 0.0, False), (0, 0.0, beam, 0.7, 0.3, False), (1, 2.5, beam, 1.0, 0.0, True),
             (2, 2.5, None, 1.3, 0.1, False)]
    results = {}
    for env in (None, "0"):
        if env is None:
            monkeypatch.delenv("PFBHIP_PSF_ROWFFT", raising=False)
        else:
            monkeypatch.setenv("PFBHIP_PSF_ROWFFT", env)
        pc = PsfConv(nx, ny, nxp, nyp)
        pc.set_psfhat(0, ph)
        pc.set_beam(0, beam)
        for i, (mode, shift, bm, scale, eta, acc) in enumerate(cases):
            if mode == 2 and is_complex:
                shift = 3.0 * np.abs(psfhat).max()  # keep the complex denominator away from zero
            out = prev.copy() if acc else None
            got = pc.apply(x, 0, beam_slot=0 if bm is not None else -1, mode=mode, shift=shift, scale=scale, eta=eta,
                           out=out, accumulate=acc)
            want = ref(mode, shift, bm, scale, eta, acc)
            assert rel(got, want) < 1e-12, (env, i)
            results[(env, i)] = got
        pc.close()
    for i in range(len(cases)):
        assert rel(results[(None, i)], results[("0", i)]) < 1e-12


def test_power_method_on_device():
    """power_method (opt/power_method.py:40-148) as called for hess_norm (core/sara.py:200-209): the device-resident
    iteration on HessPSF.dot / HessTreeRay.dot against the numpy restatement around the oracle's Hessian, the host
    loop around an arbitrary callable, and the dense eigenvalue on a tiny image."""
    from pfb_imaging_amd.opt import power_method, power_method_numba
    from pfb_imaging_amd.operators.hessian import HessPSF, HessTreeRay

    psf, psfhat, abspsf, x, beam = _psf_case(nband=2, nx=32, ny=32, nxp=64, nyp=64, seed=5)
    eta = np.array([0.1, 0.2])
    h = HessPSF(32, 32, abspsf, beam=beam, eta=eta)
    rng = np.random.default_rng(11)
    b0 = rng.standard_normal(x.shape)
    b0c = b0.copy()

    def oracle_op(z):
        return fftconv.hess_psf_dot(z, abspsf, 64, beam=beam, eta=eta)

    # a fixed number of iterations (tol = 0) compares the iterates themselves
    beta, b = power_method(h.dot, x.shape, b0=b0, tol=0.0, maxit=25, verbosity=0)
    rbeta, rb, rk = fftconv.power_method(oracle_op, x.shape, b0c.copy(), tol=0.0, maxit=25)
    assert power_method.last["iters"] == 25 == rk and power_method.last["status"] == 1
    assert abs(beta - rbeta) < 1e-11 * abs(rbeta) and rel(b, rb) < 1e-9
    assert np.array_equal(b0, b0c)  # the start vector is not mutated
    assert abs(np.linalg.norm(b) - 1.0) < 1e-12
    # converged run: same iteration count and eigenvalue; the alias the reference imports
    beta2, _ = power_method_numba(h.dot, x.shape, b0=b0, tol=1e-7, maxit=500, verbosity=0)
    rbeta2, _, rk2 = fftconv.power_method(oracle_op, x.shape, b0c.copy(), tol=1e-7, maxit=500)
    assert power_method.last["status"] == 0 and abs(power_method.last["iters"] - rk2) <= 1
    assert abs(beta2 - rbeta2) < 1e-6 * rbeta2
    # an arbitrary callable takes the host loop (and may return its internal buffer, like HessPSF.dot)
    beta3, b3 = power_method(lambda z: h.dot(z), x.shape, b0=b0, tol=0.0, maxit=25, verbosity=0)
    assert abs(beta3 - rbeta) < 1e-11 * abs(rbeta) and rel(b3, rb) < 1e-9
    # b0 = None draws the start vector
    beta4, _ = power_method(h.dot, x.shape, tol=1e-6, maxit=500, verbosity=0)
    assert abs(beta4 - rbeta2) < 1e-3 * rbeta2
    with pytest.raises(ValueError):
        power_method(h.dot, x.shape, b0=b0[0])
    with pytest.raises(ValueError):
        power_method(h.dot, x.shape, b0=np.zeros(x.shape))

    # dense check: 8 x 8 image, the matrix from unit vectors
    psf, psfhat, abspsf, x, beam = _psf_case(nband=1, nx=8, ny=8, nxp=16, nyp=16, seed=6)
    h8 = HessPSF(8, 8, abspsf, beam=beam, eta=0.05, taper_width=2)
    mat = np.stack([h8.dot(e.reshape(1, 8, 8)).copy().ravel() for e in np.eye(64)], axis=1)
    lam = np.linalg.eigvalsh(0.5 * (mat + mat.T))[-1]
    beta8, _ = power_method(h8.dot, (1, 8, 8), b0=rng.standard_normal((1, 8, 8)), tol=1e-13, maxit=5000, verbosity=0)
    assert abs(beta8 - lam) < 1e-8 * lam

    # HessTreeRay (one plan per band, wsum scaling)
    nband, nx, ny, nxp, nyp = 3, 16, 16, 32, 32
    ppb = [_tree_parts(nx, ny, nxp, nyp, 2, 1, 20 + k) for k in range(nband)]
    etas = np.array([0.1, 0.2, 0.3])
    wsum_tot = sum(sum(q["wsum"][0] for q in p) for p in ppb)
    hr = HessTreeRay(ppb, nx, ny, nxp, nyp, etas=etas, wsums=wsum_tot)
    b0 = rng.standard_normal((nband, nx, ny))

    def tree_op(z):
        return np.stack([fftconv.hessian_tree_dot(z[k], ppb[k], nxp, nyp, eta=etas[k], wsum=wsum_tot)[0] for k in range(nband)])

    betat, bt = power_method(hr.dot, b0.shape, b0=b0, tol=0.0, maxit=20, verbosity=0)
    rbt, rbv, _ = fftconv.power_method(tree_op, b0.shape, b0.copy(), tol=0.0, maxit=20)
    assert abs(betat - rbt) < 1e-11 * abs(rbt) and rel(bt, rbv) < 1e-9


def test_power_method_with_communicator():
    """The band-per-rank form of the power iteration (power_method_dist, opt/power_method.py:178-208): the three dots
    go through an RCCL all-reduce.  A 1-rank communicator must reproduce the single-process iteration bit for bit."""
    import ctypes as ct

    from pfb_imaging_amd._lib import PMInfo, cint, f64, i64, lib, check, ptr
    from pfb_imaging_amd.opt import PrimalDual
    from pfb_imaging_amd.operators.hessian import HessPSF
    from pfb_imaging_amd.parallel import BandComm

    psf, psfhat, abspsf, x, beam = _psf_case(nband=2, nx=32, ny=32, nxp=64, nyp=64, seed=7)
    h = HessPSF(32, 32, abspsf, beam=beam, eta=np.array([0.1, 0.3]))
    bands, _, local = PrimalDual._hess_bands(h, 2)
    handles = (ct.c_void_p * 2)(*[b[0]._h for b in bands])
    nparts = np.array([len(b[1]) for b in bands], dtype=np.int64)
    psf_slots = np.array([s for b in bands for s in b[1]], dtype=np.int64)
    beam_slots = np.array([s for b in bands for s in b[2]], dtype=np.int64)
    scale = np.array([b[3] for b in bands], dtype=np.float64)
    eta = np.array([b[4] for b in bands], dtype=np.float64)
    b0 = np.random.default_rng(12).standard_normal(x.shape)
    comm = BandComm.from_env(transport="rccl", set_device=False)
    out = []
    for c in (None, comm._h):
        b = b0.copy()
        info = PMInfo()
        check(lib().pfbhip_psfconv_power_method(handles, i64(2), ptr(nparts), ptr(psf_slots), ptr(beam_slots), ptr(scale),
                                                ptr(eta), ptr(b), f64(0.0), cint(12), c, ct.byref(info)))
        out.append((info.beta, info.iters, b))
    comm.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] == 12
    assert np.array_equal(out[0][2], out[1][2])


def test_band_pool_threads_match_one_band_after_the_other(monkeypatch):
    """The pool dispatches a method to all its bands at once like the reference's actors (band_worker.py:239-246); here the local
    bands run on host threads over their own HIP streams.  Same results as one band after the other (PFBHIP_BAND_THREADS=1),
    for the exact residual, the Hessian role and the wavelet role; an error in one band surfaces after the others finished."""
    from pfb_imaging_amd.operators.band_worker import BandWorkerPool

    nband, nx, ny, nxp, nyp = 4, 96, 80, 192, 160
    rng = np.random.default_rng(21)

    def part(seed):
        r = np.random.default_rng(seed)
        n = 4000
        return {"UVW": r.standard_normal((n, 3)) * 150.0, "FREQ": np.array([1.0e9, 1.05e9, 1.1e9]),
                "WEIGHT": np.abs(r.standard_normal((1, n, 3))) + 0.1, "MASK": (r.random((n, 3)) > 0.1).astype(np.uint8),
                "BEAM": 0.5 + r.random((1, nx, ny)), "attrs": {"l0": 0.0, "m0": 0.0}}

    parts = [[part(10 * b + 1), part(10 * b + 2)] for b in range(nband)]
    ppb = [_tree_parts(nx, ny, nxp, nyp, 2, 1, 40 + b) for b in range(nband)]
    dirty = rng.standard_normal((nband, 1, nx, ny))
    model = rng.standard_normal((nband, 1, nx, ny))
    x = rng.standard_normal((nband, nx, ny))

    def run():
        pool = BandWorkerPool(nband)
        pool.set_bands(dirty, parts)
        res = pool.residual(model, 2.0e-6)
        pool.init_hess(ppb, nx, ny, nxp, nyp, np.full(nband, 0.2), np.full(nband, 5.0))
        hd = pool.hess_dot(x)
        nxm, nym = pool.init_psi(nx, ny, ["self", "db2"], 2)
        al = np.zeros((nband, 2, nxm, nym))
        pool.psi_dot(x, al)
        xo = np.zeros((nband, nx, ny))
        pool.psi_hdot(al, xo)
        threaded = pool._exec is not None
        pool.close()
        return res, hd, al, xo, threaded

    a = run()
    assert a[4]
    monkeypatch.setenv("PFBHIP_BAND_THREADS", "1")
    b = run()
    assert not b[4]
    for u, v in zip(a[:4], b[:4]):  # (the atomic tile flush of small plans adds in whatever order the workgroups finish)
        assert rel(u, v) < 1e-10
    monkeypatch.delenv("PFBHIP_BAND_THREADS")
    pool = BandWorkerPool(nband)
    pool.set_bands(dirty, parts)
    with pytest.raises(Exception):
        pool.residual(model[:, :, :5], 2.0e-6)  # wrong image shape in every band: the first error is raised, no thread left behind
    assert pool.residual(model, 2.0e-6).shape == model.shape
    pool.close()

