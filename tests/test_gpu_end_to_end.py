"""End-to-end minor/major cycle on synthetic data, every operator on the GPU: the composition the reference's
`pfb sara` / `pfb kclean` run (core/sara.py:217-330, deconv/pfb.py:120-175) -- grid_partition products, PSF-approximate
Hessian + on-device CG for the forward step, the wavelet / l21 / positivity primal-dual for the backward step, and
the exact residual for the next major cycle.  Checks the physics (a few point sources are recovered, the residual
drops), not a reference number."""

import numpy as np
import pytest

from oracle import dft
from pfb_imaging_amd.utils import synth

pytestmark = pytest.mark.gpu


def test_major_cycles_recover_point_sources():
    from pfb_imaging_amd import prox
    from pfb_imaging_amd.fft import r2c
    from pfb_imaging_amd.operators.hessian import HessPSF
    from pfb_imaging_amd.operators.psi import Psi
    from pfb_imaging_amd.opt import L21, PrimalDual, PsfGrad
    from pfb_imaging_amd.wgridder import Gridder

    nx = ny = 128
    c = synth.make_case(6000, 2, nx, zscale=0.05, seed=11)
    cell = c["cell"]
    sky = np.zeros((nx, ny))
    for (ix, iy, flux) in ((40, 50, 1.0), (80, 70, 0.6), (64, 100, 0.3)):
        sky[ix, iy] = flux
    mask = c["mask"]
    vis = dft.dft_dirty2vis(c["uvw"], c["freq"], sky, cell, cell, 0.0, 0.0, False, True, False, True, False)
    vis = vis * mask
    wgt = np.ones_like(c["wgt"])
    wsum = float(wgt[mask != 0].sum())
    kw = dict(pixsize_x=cell, pixsize_y=cell, epsilon=1e-8, flip_v=True, do_wgridding=True, divide_by_n=False)
    g = Gridder(c["uvw"], c["freq"], mask, npix_x=nx, npix_y=ny, **kw)
    dirty = g.vis2dirty(vis, wgt) / wsum
    # PSF on the 2x padded grid and its transform (gridder.py:616-659)
    nxp = nyp = 2 * nx
    gp = Gridder(c["uvw"], c["freq"], mask, npix_x=nxp, npix_y=nyp, **kw)
    psf = gp.vis2dirty(np.ones_like(vis), wgt) / wsum
    gp.close()
    assert abs(psf[nxp // 2, nyp // 2] - 1.0) < 1e-7
    abspsf = np.abs(r2c(np.fft.ifftshift(psf)[None], axes=(1, 2), forward=True, inorm=0))
    hess = HessPSF(nx, ny, abspsf, beam=None, eta=1e-3, cgtol=1e-4, cgmaxit=200)
    bases = ("self", "db1", "db2")
    psi = Psi(1, nx, ny, bases, 2, 1)
    reg = L21(psi, bases, nu=len(bases))
    hessnorm = float(abspsf.max() + 1e-3)
    model = np.zeros((1, nx, ny))
    residual = dirty[None].copy()
    rms0 = residual.std()
    gamma = 1.0
    for major in range(3):
        update = hess.idot(residual, mode="psf")                      # forward step: H^-1 residual on the device
        xtilde = model + gamma * update
        pd = PrimalDual(tol=1e-5, maxit=200, verbosity=0, gamma=gamma, primal_prox=prox.positivity)
        pd.setup(reg, hessnorm)
        pd.set_grad(PsfGrad(hess, xtilde, gamma))
        assert pd._device_path() == 1
        model = pd.solve(model, lam=2e-3)                              # backward step on the device
        # exact residual for the next major cycle (gridder.py:926-1016): dirty - R^H W R model / wsum
        residual = (dirty - g.vis2dirty(g.dirty2vis(model[0]) * mask, wgt) / wsum)[None]
    g.close()
    assert residual.std() < 0.2 * rms0
    for (ix, iy, flux) in ((40, 50, 1.0), (80, 70, 0.6), (64, 100, 0.3)):
        got = model[0, ix - 1:ix + 2, iy - 1:iy + 2].sum()
        assert abs(got - flux) < 0.2 * flux, (ix, iy, got, flux)  # (soft-threshold bias on the faintest source)
    assert model.min() >= 0.0


def test_band_worker_loads_from_store_into_pinned_buffers():
    """the band worker reads its inputs from a store (here an in-memory mapping in the .dt layout) into
    page-locked buffers and runs the exact residual and the PSF Hessian from them."""
    import numpy as np

    from pfb_imaging_amd.operators.band_worker import _BandWorkerImpl
    from pfb_imaging_amd.operators.gridder import grid_partition, residual_from_partitions
    from pfb_imaging_amd.utils import synth

    c = synth.make_case(3000, 2, 48, zscale=0.2, seed=3)
    cell = c["cell"] * 30
    nx = ny = 48
    part = {"UVW": c["uvw"], "VIS": c["vis"][None], "WEIGHT": c["wgt"][None], "MASK": c["mask"], "FREQ": c["freq"],
            "BEAM": np.ones((1, nx, ny))}
    prod = grid_partition(part, None, nx, ny, 2 * nx, 2 * ny, cell)
    store = {"band0": {"arrays": {"DIRTY": prod["DIRTY"]}, "attrs": {}, "children": {"part0": {
        "arrays": {"UVW": c["uvw"], "WEIGHT": prod["WEIGHT"], "MASK": c["mask"], "FREQ": c["freq"], "BEAM": prod["BEAM"],
                   "PSFHAT": prod["PSFHAT"]}, "attrs": {"wsum": prod["WSUM"], "l0": 0.0, "m0": 0.0}}}}}
    w = _BandWorkerImpl(1)
    w.load_band(store, "band0")
    rng = np.random.default_rng(0)
    model = rng.standard_normal((1, nx, ny))
    res = w.residual(model, cell, 1e-7, True, True)
    ref = residual_from_partitions(prod["DIRTY"], [dict(part, WEIGHT=prod["WEIGHT"], BEAM=prod["BEAM"])], model, cell)
    assert np.linalg.norm(res - ref) / np.linalg.norm(ref) < 1e-10   # (same arithmetic; the scatter's LDS atomics reorder sums)
    w.init_hess(None, nx, ny, 2 * nx, 2 * ny, 0.1, None)
    h = w.hess_dot(model)
    assert h.shape == model.shape and np.isfinite(h).all()


def test_band_worker_loads_a_zarr_directory_store(tmp_path):
    """The same through REAL bytes: the band's inputs written as a zarr-v2 directory store in the reference's .dt layout
    (core/imager.py:138-194: group bandNNNN_timeNNNN with DIRTY, child group partNNNN with UVW / WEIGHT / MASK / FREQ / BEAM /
    PSFHAT and the wsum / l0 / m0 attributes), opened by PATH like the reference's load_band (band_worker.py:61-106); the chunk
    files are read straight into page-locked buffers by store.DirStore."""
    import os

    from pfb_imaging_amd.operators.band_worker import _BandWorkerImpl
    from pfb_imaging_amd.operators.gridder import grid_partition, residual_from_partitions
    from tests.test_store_cpu import _write_group, _write_zarr_array

    c = synth.make_case(3000, 2, 48, zscale=0.2, seed=4)
    cell = c["cell"] * 30
    nx = ny = 48
    part = {"UVW": c["uvw"], "VIS": c["vis"][None], "WEIGHT": c["wgt"][None], "MASK": c["mask"], "FREQ": c["freq"],
            "BEAM": np.ones((1, nx, ny))}
    prod = grid_partition(part, None, nx, ny, 2 * nx, 2 * ny, cell)
    root = str(tmp_path / "run.dt")
    _write_group(root)
    band = os.path.join(root, "band0000_time0000")
    _write_group(band, {"bandid": 0})
    _write_zarr_array(os.path.join(band, "DIRTY"), prod["DIRTY"], (1, nx, ny))
    p0 = os.path.join(band, "part0000")
    _write_group(p0, {"wsum": [float(w) for w in np.atleast_1d(prod["WSUM"])], "l0": 0.0, "m0": 0.0})
    nrow = c["uvw"].shape[0]
    _write_zarr_array(os.path.join(p0, "UVW"), c["uvw"], (1000, 3))
    _write_zarr_array(os.path.join(p0, "WEIGHT"), np.ascontiguousarray(prod["WEIGHT"]), (1, nrow, 2), {"id": "zlib", "level": 1})
    _write_zarr_array(os.path.join(p0, "MASK"), c["mask"], (700, 2), fill_value=0)
    _write_zarr_array(os.path.join(p0, "FREQ"), c["freq"], (2,))
    _write_zarr_array(os.path.join(p0, "BEAM"), np.ascontiguousarray(prod["BEAM"]), (1, nx, ny))
    _write_zarr_array(os.path.join(p0, "PSFHAT"), np.ascontiguousarray(prod["PSFHAT"]), (1, 32, prod["PSFHAT"].shape[-1]))
    w = _BandWorkerImpl(1)
    w.load_band(root, "band0000_time0000")
    model = np.random.default_rng(1).standard_normal((1, nx, ny))
    res = w.residual(model, cell, 1e-7, True, True)
    ref = residual_from_partitions(prod["DIRTY"], [dict(part, WEIGHT=prod["WEIGHT"], BEAM=prod["BEAM"])], model, cell)
    assert np.linalg.norm(res - ref) / np.linalg.norm(ref) < 1e-10
    w.init_hess(None, nx, ny, 2 * nx, 2 * ny, 0.1, None)
    assert np.isfinite(w.hess_dot(model)).all()
