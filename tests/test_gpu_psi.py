"""GPU parity of the wavelet dictionary Psi and the l21 / positivity proxes against the CPU oracle
(oracle/psi.py, a restatement of the reference's numba code; see its header: parity unpinned by fixtures)."""

import numpy as np
import pytest

from oracle import psi as opsi

pytestmark = pytest.mark.gpu

rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.mark.parametrize("bases,nlevel,shape", [
    (("self", "db1", "db2", "db3"), 3, (128, 96)),     # the reference's default dictionary (core/sara.py)
    (("db4", "db5"), 2, (128, 256)),                    # tests/test_wavelets.py:72-75
    (("db1",), 1, (512, 128)),
    (("db8", "self", "db6"), 3, (250, 300)),            # sizes that go odd at deeper levels
])
def test_psi_dot_hdot_vs_oracle(bases, nlevel, shape):
    from pfb_imaging_amd.operators.psi import Psi, PsiNocopyt

    nband = 2
    nx, ny = shape
    rng = np.random.default_rng(3)
    x = rng.standard_normal((nband, nx, ny))
    o = opsi.Psi(nband, nx, ny, bases, nlevel)
    g = PsiNocopyt(nband, nx, ny, bases, nlevel, nthreads=1)
    assert (g.nxmax, g.nymax, g.nbasis) == (o.nxmax, o.nymax, o.nbasis)
    a_ref = np.zeros((nband, o.nbasis, o.nxmax, o.nymax))
    o.dot(x, a_ref)
    a = np.full_like(a_ref, np.nan)  # every element must be written
    g.dot(x, a)
    assert rel(a, a_ref) < 1e-14
    coeffs = rng.standard_normal(a_ref.shape)
    x_ref = np.zeros_like(x)
    o.hdot(coeffs, x_ref)
    xo = np.full_like(x, np.nan)
    keep = coeffs.copy()
    g.hdot(coeffs, xo)
    assert rel(xo, x_ref) < 1e-14
    assert np.array_equal(coeffs, keep)  # hdot must not touch its input (psi.py "to avoid overwriting coeffs")
    # perfect reconstruction: Psi Psi^H = nbasis * I (orthonormal bases, zero-padding mode keeps all coefficients)
    back = np.zeros_like(x)
    g.hdot(a, back)
    assert rel(back / len(bases), x) < 1e-13
    # adjointness
    assert abs(np.vdot(a, coeffs) - np.vdot(x, xo)) < 1e-10 * abs(np.vdot(x, xo))
    # the older transposed layout
    gt = Psi(nband, nx, ny, bases, nlevel, nthreads=1)
    at = np.zeros((nband, o.nbasis, o.nymax, o.nxmax))
    gt.dot(x, at)
    assert np.array_equal(at, a.transpose(0, 1, 3, 2))
    xt = np.zeros_like(x)
    gt.hdot(np.ascontiguousarray(coeffs.transpose(0, 1, 3, 2)), xt)
    assert rel(xt, xo) < 1e-15


@pytest.mark.parametrize("bases,nlevel,shape", [
    (("self", "db1", "db2", "db3"), 3, (128, 96)),
    (("db8", "self", "db6"), 3, (250, 300)),
    (("db4",), 2, (64, 512)),
])
def test_psi_dot_writes_every_coefficient_of_a_dirty_device_cube(bases, nlevel, shape):
    """dot() zeroes only the cells no transform writes (margins of the packed layout, padding up to the largest basis): a
    device cube full of NaN must come back exactly as the oracle's, zeros included."""
    import ctypes as ct

    from pfb_imaging_amd._lib import DeviceArray, check, lib
    from pfb_imaging_amd.operators.psi import PsiBand

    nx, ny = shape
    rng = np.random.default_rng(8)
    x = rng.standard_normal((1, nx, ny))
    o = opsi.Psi(1, nx, ny, bases, nlevel)
    a_ref = np.zeros((1, o.nbasis, o.nxmax, o.nymax))
    o.dot(x, a_ref)
    g = PsiBand(nx, ny, bases, nlevel)
    xd = DeviceArray.from_host(x[0])
    ad = DeviceArray.from_host(np.full((o.nbasis, o.nxmax, o.nymax), np.nan))
    check(lib().pfbhip_psi_dot_dev(g._h, xd.ptr, ad.ptr))
    a = ad.download()
    assert not np.isnan(a).any()
    assert np.array_equal(a == 0.0, a_ref[0] == 0.0)
    assert rel(a, a_ref[0]) < 1e-14


def test_psi_errors():
    from pfb_imaging_amd.operators.psi import PsiNocopyt

    with pytest.raises(ValueError):
        PsiNocopyt(1, 64, 64, ("db9",), 2)
    with pytest.raises(ValueError):
        PsiNocopyt(1, 63, 64, ("db1",), 2)     # odd image size
    with pytest.raises(ValueError):
        PsiNocopyt(1, 16, 16, ("db8",), 3)     # level not possible
    p = PsiNocopyt(1, 64, 64, ("db2",), 2)
    with pytest.raises(ValueError):
        p.dot(np.zeros((1, 64, 64)), np.zeros((1, 1, 10, 10)))


def test_dual_update_prox_positivity():
    from pfb_imaging_amd import prox

    rng = np.random.default_rng(8)
    nband, nbasis, n1, n2 = 3, 4, 37, 29
    vp = rng.standard_normal((nband, nbasis, n1, n2))
    v = rng.standard_normal((nband, nbasis, n1, n2))
    w = np.abs(rng.standard_normal((nbasis, n1, n2))) + 0.1
    w[0, 0, :5] = 0.0  # zero threshold
    ref = opsi.dual_update(vp, v.copy(), 0.7, 1.3, w)
    got = v.copy()
    prox.dual_update_numba_fast(vp, got, 0.7, 1.3, w)
    assert rel(got, ref) < 1e-15
    # numpy form of the reference (prox_21m.py:61-70): v = vtilde - sigma prox(vtilde / sigma, lam / sigma)
    vt = vp + 1.3 * v
    ref2 = vt - 1.3 * opsi.prox_21m(vt / 1.3, 0.7 / 1.3, weight=w)
    assert rel(got, ref2) < 1e-13
    assert rel(prox.prox_21m(v, 0.4, weight=w), opsi.prox_21m(v, 0.4, weight=w)) < 1e-15
    res = np.empty_like(v)
    prox.prox_21m_numba(v, res, 0.7, 1.3, w)
    assert rel(res, opsi.prox_21m(v / 1.3, 0.7 / 1.3, weight=w)) < 1e-15
    x = rng.standard_normal((nband, 33, 21))
    a, b = x.copy(), x.copy()
    prox.positivity(a)
    assert np.array_equal(a, opsi.positivity(x.copy()))
    prox.positivity_band(b)
    assert np.array_equal(b, opsi.positivity_band(x.copy()))
    assert prox.positivity_prox(0) is None and prox.positivity_prox(2) is prox.positivity_band
    with pytest.raises(ValueError):
        prox.positivity_prox(3)


def test_band_pool_psi_role_and_dual_update():
    """BandWorkerPool's wavelet role (band_worker.py:144-163, 291-301) and the band-sharded dual update run the
    GPU kernels on a single rank exactly as the multi-rank path does (tests/_gloo_worker.py covers N = 2)."""
    from pfb_imaging_amd.operators.band_worker import BandWorkerPool
    from pfb_imaging_amd.operators.psi import PsiNocopytRay
    from pfb_imaging_amd.prox import dual_update_bands

    nband, nx, ny, bases = 3, 64, 48, ("self", "db1", "db2", "db3")
    rng = np.random.default_rng(2)
    pool = BandWorkerPool(nband)
    psi = PsiNocopytRay(nband, nx, ny, bases, 3, workers=pool)
    o = opsi.Psi(nband, nx, ny, bases, 3)
    x = rng.standard_normal((nband, nx, ny))
    a = np.zeros((nband, 4, psi.nxmax, psi.nymax))
    psi.dot(x, a)
    ref = np.zeros_like(a)
    o.dot(x, ref)
    assert rel(a, ref) < 1e-14
    xo = np.zeros_like(x)
    psi.hdot(a, xo)
    assert rel(xo, 4 * x) < 1e-13
    vp = rng.standard_normal(a.shape)
    w = np.abs(rng.standard_normal(a.shape[1:])) + 0.1
    v = a.copy()
    # force the two-phase device path (the one every rank runs around the all-reduce)
    from pfb_imaging_amd import prox

    dual_update_bands(vp, v, 0.6, 1.4, w, comm=None, bands=[0, 1, 2], phases=(prox._vtilde_sum_gpu, prox._scale_gpu))
    assert rel(v, opsi.dual_update(vp, a.copy(), 0.6, 1.4, w)) < 1e-15
    v2 = a.copy()
    pool.dual_update(vp, v2, 0.6, 1.4, w)
    assert np.array_equal(v2, v)


def _oracle_pd(x, v, lam, psi, weight, hess, xtilde, gamma, sigma, tau, tol, maxit, positivity):
    """PrimalDual.solve (primal_dual.py:406-448) in numpy on the oracle's dictionary (x-first layout)."""
    x, v = x.copy(), v.copy()
    xp, vp = x.copy(), v.copy()
    xout = np.zeros_like(x)
    eps, k = 1.0, 0
    for k in range(maxit):
        psi.dot(xp, v)
        opsi.dual_update(vp, v, lam, sigma, weight)
        vp = 2.0 * v - vp
        psi.hdot(vp, xout)
        xout = xout - hess(xtilde - xp) / gamma
        x = xp - tau * xout
        if positivity == 1:
            opsi.positivity(x)
        elif positivity == 2:
            opsi.positivity_band(x)
        eps = np.sqrt(((x - xp) ** 2).sum() / max((x**2).sum(), 1e-12)) if x.any() else 1.0
        if eps < tol:
            break
        xp, vp = x.copy(), v.copy()
    return x, v, k, eps


@pytest.mark.parametrize("layout,positivity,use_beam", [("nocopyt", 0, True), ("psi", 1, False), ("nocopyt", 2, True)])
def test_primal_dual_device_loop(layout, positivity, use_beam):
    """The device-resident backward step == the reference's loop on the oracle pieces, and == the generic
    (callable-gradient) path of this package."""
    from oracle import fftconv
    from pfb_imaging_amd import prox
    from pfb_imaging_amd.operators.hessian import HessPSF
    from pfb_imaging_amd.operators.psi import Psi, PsiNocopyt
    from pfb_imaging_amd.opt import L21, PrimalDual, PsfGrad

    nband, nx, ny, nxp, nyp = 2, 64, 48, 128, 96
    bases, nlevel = ("self", "db1", "db2"), 2
    rng = np.random.default_rng(21)
    psf = np.zeros((nband, nxp, nyp))
    psf[:, 0, 0] = 1.0
    psf += 0.02 * rng.standard_normal(psf.shape)
    abspsf = np.abs(np.fft.rfft2(psf, axes=(1, 2)))
    beam = 0.8 + 0.2 * rng.random((nband, nx, ny)) if use_beam else None
    eta = np.array([0.05, 0.1])
    hess = HessPSF(nx, ny, abspsf, beam=beam, eta=eta)
    model = np.abs(rng.standard_normal((nband, nx, ny))) * (rng.random((nband, nx, ny)) > 0.9)
    xtilde = model + 0.3 * rng.standard_normal(model.shape)
    # a valid bound on ||H|| (an underestimate makes the iteration expansive: one-step parity stays exact but
    # rounding differences between the FFT libraries then double every iteration under the positivity clamp)
    hessnorm = float(abspsf.max() * (beam.max() ** 2 if beam is not None else 1.0) + eta.max())
    gamma, lam = 1.0, 0.02
    cls = PsiNocopyt if layout == "nocopyt" else Psi
    psi = cls(nband, nx, ny, bases, nlevel, 1)
    reg = L21(psi, bases, nu=np.sqrt(len(bases)))  # ||Psi||: Psi Psi^H = nbasis I
    reg.l1weight = 0.5 + rng.random(reg.l1weight.shape)
    pp = prox.positivity_prox(positivity)
    pd = PrimalDual(tol=1e-6, maxit=40, verbosity=0, gamma=gamma, primal_prox=pp)
    pd.setup(reg, hessnorm)
    pd.set_grad(PsfGrad(hess, xtilde, gamma))
    assert pd._device_path() == positivity
    got = pd.solve(model.copy(), lam)
    # oracle loop (x-first layout)
    o = opsi.Psi(nband, nx, ny, bases, nlevel)
    w = reg.l1weight if layout == "nocopyt" else reg.l1weight.transpose(0, 2, 1)
    v0 = np.zeros((nband, o.nbasis, o.nxmax, o.nymax))
    href = lambda z: fftconv.hess_psf_dot(z, abspsf, nyp, beam=beam, eta=eta)
    xr, vr, kr, er = _oracle_pd(model, v0, lam, o, w, href, xtilde, gamma, pd.sigma, pd.tau, 1e-6, 40, positivity)
    assert pd.last["iters"] == kr
    assert rel(got, xr) < 1e-9
    vgot = pd._v if layout == "nocopyt" else pd._v.transpose(0, 1, 3, 2)
    # the dual integrates sigma Psi^H x_k over the iterations: rounding differences of the two FFT libraries in x
    # (1e-10 after 40 iterations) show up ~100x larger relative to the (small) unclipped dual coefficients
    assert rel(vgot, vr) < 1e-6
    # generic path (plain callable): same iterates
    pd2 = PrimalDual(tol=1e-6, maxit=40, verbosity=0, gamma=gamma, primal_prox=pp)
    pd2.setup(reg, hessnorm)
    g = PsfGrad(hess, xtilde, gamma)
    pd2.set_grad(lambda z: g(z))
    assert pd2._device_path() is None
    got2 = pd2.solve(model.copy(), lam)
    assert rel(got2, xr) < 1e-7 and pd2.last["iters"] == kr


def test_l21_reweighting():
    """L21.init_reweighting / update_weights (prox/l21.py:56-88, utils/misc.py:742-755) on the GPU analysis."""
    from pfb_imaging_amd.operators.psi import Psi
    from pfb_imaging_amd.opt import L21

    nband, nx, ny, bases = 2, 32, 48, ("self", "db2")
    rng = np.random.default_rng(4)
    psi = Psi(nband, nx, ny, bases, 2, 1)
    reg = L21(psi, bases, rmsfactor=0.7, alpha=2.0)
    assert not reg.reweight_active
    with pytest.raises(RuntimeError):
        reg.update_weights(np.zeros((nband, nx, ny)))
    upd = rng.standard_normal((nband, nx, ny))
    reg.init_reweighting(upd)
    assert reg.reweight_active
    o = opsi.Psi(nband, nx, ny, bases, 2, transposed=True)
    a = np.zeros((nband, 2, o.nymax, o.nxmax))
    o.dot(upd, a)
    s = a.sum(axis=0)
    rms = np.array([np.std(s[i][s[i] != 0]) for i in range(2)])
    x = rng.standard_normal((nband, nx, ny))
    reg.update_weights(x)
    o.dot(x, a)
    ref = 1.7 / (1 + np.abs(a.sum(axis=0)) ** 2 / rms[:, None, None] ** 2)
    assert reg.l1weight.shape == ref.shape and rel(reg.l1weight, ref) < 1e-12


def test_primal_dual_device_loop_hess_tree_ray():
    """The presets.py composition (deconv/presets.py:100-130): PsiNocopyt-layout dictionary + HessTreeRay (one PSF
    plan per band, two partitions each) runs the device loop too and matches the generic loop."""
    from pfb_imaging_amd import prox
    from pfb_imaging_amd.operators.hessian import HessTreeRay
    from pfb_imaging_amd.operators.psi import PsiNocopyt
    from pfb_imaging_amd.opt import L21, PrimalDual, PsfGrad

    nband, nx, ny, nxp, nyp = 2, 32, 48, 64, 96
    bases = ("self", "db1", "db3")
    rng = np.random.default_rng(5)
    parts = []
    for b in range(nband):
        pb = []
        for _ in range(2):
            psf = np.zeros((1, nxp, nyp))
            psf[:, 0, 0] = 1.0
            psf += 0.02 * rng.standard_normal(psf.shape)
            pb.append({"psfhat": np.abs(np.fft.rfft2(psf, axes=(1, 2))), "beam": 0.8 + 0.2 * rng.random((1, nx, ny)),
                       "wsum": np.array([1.0 + rng.random()])})
        parts.append(pb)
    hess = HessTreeRay(parts, nx, ny, nxp, nyp, etas=[0.05, 0.1])
    psi = PsiNocopyt(nband, nx, ny, bases, 2, 1)
    reg = L21(psi, bases, nu=np.sqrt(len(bases)))
    reg.l1weight = 0.5 + rng.random(reg.l1weight.shape)
    model = np.abs(rng.standard_normal((nband, nx, ny))) * (rng.random((nband, nx, ny)) > 0.8)
    xtilde = model + 0.3 * rng.standard_normal(model.shape)
    hessnorm = 1.4
    res = {}
    for name in ("device", "generic"):
        pd = PrimalDual(tol=1e-7, maxit=30, verbosity=0, gamma=1.0, primal_prox=prox.positivity_band)
        pd.setup(reg, hessnorm)
        g = PsfGrad(hess, xtilde, 1.0)
        pd.set_grad(g if name == "device" else (lambda z: g(z)))
        assert (pd._device_path() == 2) == (name == "device")
        res[name] = (pd.solve(model.copy(), 0.02), pd.last["iters"])
    assert res["device"][1] == res["generic"][1]
    assert rel(res["device"][0], res["generic"][0]) < 1e-9


def test_primal_dual_device_loop_with_communicator():
    """The band-per-rank form of the device loop (local band sum + all-reduce + update; all-reduced norms and
    positivity flags) on a 1-rank RCCL communicator: every collective is executed, results equal the
    single-process form."""
    from pfb_imaging_amd import prox
    from pfb_imaging_amd.operators.band_worker import BandWorkerPool
    from pfb_imaging_amd.operators.hessian import HessTreeRay
    from pfb_imaging_amd.operators.psi import PsiNocopyt
    from pfb_imaging_amd.opt import L21, PrimalDual, PsfGrad
    from pfb_imaging_amd.parallel import BandComm

    nband, nx, ny, nxp, nyp = 2, 32, 48, 64, 96
    bases = ("self", "db2")
    rng = np.random.default_rng(6)
    parts = [[{"psfhat": 1.0 + 0.1 * np.abs(rng.standard_normal((1, nxp, nyp // 2 + 1))), "beam": np.ones((1, nx, ny)),
               "wsum": np.array([1.0])}] for _ in range(nband)]
    model = np.abs(rng.standard_normal((nband, nx, ny))) * (rng.random((nband, nx, ny)) > 0.8)
    xtilde = model + 0.3 * rng.standard_normal(model.shape)
    res = {}
    for name in ("plain", "comm"):
        comm = BandComm.from_env(transport="rccl") if name == "comm" else None
        pool = BandWorkerPool(nband, comm=comm)
        hess = HessTreeRay(parts, nx, ny, nxp, nyp, etas=0.05, workers=pool)
        psi = PsiNocopyt(nband, nx, ny, bases, 2, 1)
        reg = L21(psi, bases, nu=np.sqrt(2.0))
        pd = PrimalDual(tol=1e-8, maxit=15, verbosity=0, gamma=1.0, primal_prox=prox.positivity_band)
        pd.setup(reg, 1.3)
        pd.set_grad(PsfGrad(hess, xtilde, 1.0))
        assert pd._device_path() == 2
        if name == "comm":  # a 1-rank communicator is "world_size 1": force the collective code path
            bands, _, local = pd._hess_bands(hess, nband)
            pd._hess_bands = lambda h, n: (bands, comm, local)
        res[name] = (pd.solve(model.copy(), 0.02), pd._v.copy(), pd.last["iters"])
        if comm is not None:
            comm.close()
    assert res["plain"][2] == res["comm"][2]
    assert rel(res["comm"][0], res["plain"][0]) < 1e-13 and rel(res["comm"][1], res["plain"][1]) < 1e-13
