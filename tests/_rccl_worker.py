"""Worker of tests/test_gpu_multi.py: run under torch.distributed.run, one rank per GPU (RCCL over xGMI).

Every multi-rank result is compared with the same quantity computed by THIS rank alone on its own GPU (or on the host), so
the script needs no cross-rank golden data.  With one rank it still walks every code path (RCCL communicator of size 1)."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pfb_imaging_amd import _lib, prox  # noqa: E402
from pfb_imaging_amd._lib import DeviceArray  # noqa: E402
from pfb_imaging_amd.operators.band_worker import BandWorkerPool  # noqa: E402
from pfb_imaging_amd.operators.hessian import HessPSF, HessTreeRay  # noqa: E402
from pfb_imaging_amd.operators.psi import PsiNocopyt  # noqa: E402
from pfb_imaging_amd.opt import L21, PrimalDual, PsfGrad, power_method  # noqa: E402
from pfb_imaging_amd.parallel import BandComm, RowShardedGridder, WShardedGridder, local_bands  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402
from pfb_imaging_amd.wgridder import Gridder  # noqa: E402


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def main():
    comm = BandComm.from_env()          # RCCL; binds this rank to GPU LOCAL_RANK
    rank, world = comm.rank, comm.world_size
    assert comm.transport == "rccl"
    rng = np.random.default_rng(0)      # the same data on every rank

    # --- collectives on device buffers and on host arrays (persistent staging) ---
    img = rng.standard_normal((world, 96, 80))
    send = DeviceArray.from_host(img[rank])
    recv = DeviceArray((96, 80), np.float64) if rank == 0 else None
    comm.reduce_sum_dev(send, recv, root=0)
    if rank == 0:
        assert rel(recv.download(), img.sum(axis=0)) < 1e-14
    comm.allreduce_sum_dev(send, send)
    assert rel(send.download(), img.sum(axis=0)) < 1e-14
    tot = comm.reduce_sum(img[rank], root=world - 1)
    assert (tot is None) == (rank != world - 1) and (tot is None or rel(tot, img.sum(axis=0)) < 1e-14)
    assert rel(comm.allreduce_sum(img[rank]), img.sum(axis=0)) < 1e-14
    assert np.array_equal(comm.allgather(img[rank]), img)
    assert comm.max_over_ranks(rank) == world - 1 and comm.sum_over_ranks(1.0) == world

    # --- per-collective timings (printed by rank 0: the first N > 1 run on hardware yields numbers without a code change) ---
    import time

    for n in (1 << 20, 8192 * 8192 // 8, 8192 * 8192):       # 8 MB, 67 MB, 537 MB (one C2 image) of f64
        a, b = DeviceArray((n,), np.float64), DeviceArray((n,), np.float64)
        a.upload(np.ones(n))
        for name, call in (("reduce", lambda: comm.reduce_sum_dev(a, b if rank == 0 else None, root=0)),
                           ("allreduce", lambda: comm.allreduce_sum_dev(a, b))):
            call()
            comm.barrier()
            _lib.check(_lib.lib().pfbhip_synchronize())
            t0 = time.perf_counter()
            for _ in range(3):
                call()
            _lib.check(_lib.lib().pfbhip_synchronize())
            comm.barrier()
            dt = (time.perf_counter() - t0) / 3
            if rank == 0:
                print(f"[rccl timing] world {world} {name:9s} {8 * n / 1e6:8.1f} MB  {dt * 1e3:8.3f} ms  "
                      f"{8 * n / dt / 1e9:7.1f} GB/s (algorithmic)", flush=True)
        a.free()
        b.free()

    # --- band pool: PSF Hessians per band, one band (or two) per GPU ---
    nband = 2 * world
    nx, ny, nxp, nyp = 64, 48, 128, 96
    psf = np.zeros((nband, nxp, nyp))
    psf[:, 0, 0] = 1.0
    psf += 0.02 * rng.standard_normal(psf.shape)
    abspsf = np.abs(np.fft.rfft2(psf, axes=(1, 2)))
    eta = 0.05 + 0.01 * np.arange(nband)
    parts = [[dict(psfhat=abspsf[b][None], beam=np.ones((1, nx, ny)), wsum=np.ones(1))] for b in range(nband)]
    pool = BandWorkerPool(nband, comm=comm)
    assert sorted(pool.workers) == local_bands(nband, rank, world)
    tree = HessTreeRay(parts, nx, ny, nxp, nyp, etas=eta, wsums=np.ones(nband), workers=pool)
    single = HessPSF(nx, ny, abspsf, beam=None, eta=eta)        # all bands on this GPU
    x = rng.standard_normal((nband, nx, ny))
    ref = single.dot(x).copy()
    assert rel(tree.dot(x), ref) < 1e-12                        # all-gather exchange (every rank owns 2 bands)
    sol = tree.cg(ref, tol=1e-10, maxit=200, minit=1)
    assert rel(sol, x) < 1e-6
    # spectral norm: per-rank bands + all-reduced dots == all bands on one GPU
    b0 = rng.standard_normal((nband, nx, ny))
    beta_d, _ = power_method(tree.dot, (nband, nx, ny), b0=b0, tol=1e-8, maxit=60)
    beta_s, _ = power_method(single.dot, (nband, nx, ny), b0=b0, tol=1e-8, maxit=60)
    assert abs(beta_d - beta_s) < 1e-8 * beta_s

    # --- band-sharded primal-dual loop (l21 band sum all-reduced every iteration) vs the single-process loop ---
    bases, nlevel = ("self", "db1", "db2"), 2
    model = np.abs(rng.standard_normal((nband, nx, ny))) * (rng.random((nband, nx, ny)) > 0.9)
    xtilde = model + 0.3 * rng.standard_normal(model.shape)
    hessnorm = float(abspsf.max() + eta.max())
    sols = {}
    for name, hess in (("sharded", tree), ("single", single)):
        psi = PsiNocopyt(nband, nx, ny, bases, nlevel, 1)
        reg = L21(psi, bases, nu=np.sqrt(len(bases)))
        reg.l1weight = 0.5 + np.random.default_rng(1).random(reg.l1weight.shape)
        pd = PrimalDual(tol=1e-7, maxit=30, verbosity=0, gamma=1.0, primal_prox=prox.positivity_prox(2))
        pd.setup(reg, hessnorm)
        pd.set_grad(PsfGrad(hess, xtilde, 1.0))
        assert pd._device_path() is not None, name               # both run the device-resident loop
        sols[name] = (pd.solve(model.copy(), 0.02), pd.last["iters"])
    assert sols["sharded"][1] == sols["single"][1]
    assert rel(sols["sharded"][0], sols["single"][0]) < 1e-9

    # --- one band split by row blocks (config-5 style) vs the unsharded plan ---
    c = synth.make_case(4000, 2, 64, zscale=0.3, seed=7)
    kw = dict(npix_x=64, npix_y=64, pixsize_x=c["cell"] * 40, pixsize_y=c["cell"] * 40, center_x=0.0, center_y=0.0, epsilon=1e-7,
              flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False)
    sh = RowShardedGridder(comm, c["uvw"], c["freq"], c["mask"], **kw)
    full = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
    dref = full.vis2dirty(c["vis"], c["wgt"])
    assert rel(sh.vis2dirty(c["vis"], c["wgt"]), dref) < 1e-6    # each block picks its own kernel / planes: epsilon-level
    d0 = sh.vis2dirty(c["vis"], c["wgt"], root=0)
    assert (d0 is None) == (rank != 0) and (d0 is None or rel(d0, dref) < 1e-6)
    mv = sh.dirty2vis(c["x"])
    assert rel(mv, full.dirty2vis(c["x"])[sh.r0:sh.r1]) < 1e-6
    sh.set_weights(c["wgt"])
    full.set_weights(c["wgt"])
    assert rel(sh.hessian(c["x"], eta=0.3, wsum=7.0), full.hessian(c["x"], eta=0.3, wsum=7.0)) < 1e-6
    sh.close()

    # --- the same band split by |w| (the split that shards the w-planes): wider field so that the plan stacks ES-kernel planes ---
    kw2 = dict(kw, pixsize_x=c["cell"] * 120, pixsize_y=c["cell"] * 120)
    full2 = Gridder(c["uvw"], c["freq"], c["mask"], **kw2)
    ws = WShardedGridder(comm, c["uvw"], c["freq"], c["mask"], **kw2)
    planes = ws.planes_per_rank()
    assert len(planes) == world and planes[rank] == ws.local.info["nplanes"]
    if world > 1 and full2.info["wmode"] == 0:
        assert max(planes) < full2.info["nplanes"], (planes, full2.info["nplanes"])   # every rank stacks fewer planes than the whole band
    dref2 = full2.vis2dirty(c["vis"], c["wgt"])
    assert rel(ws.vis2dirty(c["vis"], c["wgt"]), dref2) < 1e-6
    assert rel(ws.dirty2vis(c["x"]), full2.dirty2vis(c["x"])[ws.rows]) < 1e-6
    ws.set_weights(c["wgt"])
    full2.set_weights(c["wgt"])
    assert rel(ws.hessian(c["x"], eta=0.3, wsum=7.0), full2.hessian(c["x"], eta=0.3, wsum=7.0)) < 1e-6
    ws.close()
    full2.close()
    full.close()

    comm.barrier()
    print(f"rank {rank} of {world} ok", flush=True)
    comm.close()


if __name__ == "__main__":
    main()
