"""Worker for tests/test_parallel_cpu.py, 2 ranks on the CPU, two ways:
  * ``_gloo_worker.py socket``: plain processes, the product's own TCP rendezvous / host transport (parallel.HostGroup);
  * ``_gloo_worker.py gloo`` under torch.distributed.run: torch.distributed (gloo) stands in for the host transport through
    :class:`GlooHost` -- test-only, the product itself imports no torch."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pfb_imaging_amd.operators.band_worker import BandWorkerPool  # noqa: E402
from pfb_imaging_amd.parallel import BandComm, local_bands  # noqa: E402


class FakeWorker:
    """Stands in for the GPU band worker: a diagonal 'Hessian' and an affine 'residual' in numpy
    (the reference fakes its Hessian the same way: tests/test_pcg_solver.py:10-46)."""

    def __init__(self, nthreads):
        self.scale = None

    def set_band(self, dirty, parts, hess_parts=None):
        self.dirty = dirty
        self.scale = parts

    def init_hess(self, partitions, nx, ny, nx_psf, ny_psf, eta, wsum):
        self.d = partitions + eta

    def hess_dot(self, x):
        return (self.d * x)[None]

    def cg(self, rhs, x0, tol, maxit, minit, verbosity):
        return rhs / self.d

    def residual(self, model, cell_rad, epsilon, do_wgridding, double_accum):
        return self.dirty - self.scale * model

    # wavelet role: the oracle's dictionary stands in for the GPU one
    def init_psi(self, nx, ny, bases, nlevel):
        from oracle import psi as opsi

        self._psi = opsi.Psi(1, nx, ny, bases, nlevel)
        return self._psi.nxmax, self._psi.nymax

    def psi_dot(self, x):
        out = np.zeros((1, self._psi.nbasis, self._psi.nxmax, self._psi.nymax))
        self._psi.dot(x[None], out)
        return out[0]

    def psi_hdot(self, alpha):
        out = np.zeros((1, self._psi.nx, self._psi.ny))
        self._psi.hdot(alpha[None], out)
        return out[0]


class GlooHost:
    """HostGroup's methods over torch.distributed (gloo): ``allgather_bytes`` is the one primitive there too."""

    def __init__(self):
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo")
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_bytes(self, payload):
        torch, dist = self.torch, self.dist
        n = torch.tensor([len(payload)], dtype=torch.int64)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        m = max(int(s[0]) for s in sizes)
        buf = torch.zeros(max(m, 1), dtype=torch.uint8)
        if payload:
            buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        parts = [torch.zeros(max(m, 1), dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(parts, buf)
        return [bytes(p[:int(s[0])].numpy().tobytes()) for p, s in zip(parts, sizes)]

    def bcast_bytes(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src else b"")[src]

    def allgather(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        return np.stack([np.frombuffer(p, dtype=np.float64).reshape(arr.shape) for p in self.allgather_bytes(arr.tobytes())])

    def barrier(self):
        self.dist.barrier()

    def close(self):
        pass


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "socket"
    comm = BandComm.from_env(transport="host", host=GlooHost() if mode == "gloo" else None)
    if mode == "socket":
        from pfb_imaging_amd.parallel import HostGroup

        assert isinstance(comm._host, HostGroup) and "torch" not in sys.modules
        # the rendezvous payload of the product path: 128 bytes from rank 0 (RCCL's unique id travels like this)
        uid = bytes(range(128)) if comm.rank == 0 else b""
        assert comm._host.bcast_bytes(uid, src=0) == bytes(range(128))
        assert comm._host.allgather_bytes(bytes([comm.rank]) * (comm.rank + 1)) == [b"\x00", b"\x01\x01"]
    assert comm.world_size == 2
    nband, nx, ny = 5, 6, 4
    rng = np.random.default_rng(0)  # same data on every rank
    x = rng.standard_normal((nband, nx, ny))
    assert local_bands(nband, comm.rank, 2) == ([0, 2, 4] if comm.rank == 0 else [1, 3])

    # collectives
    total = comm.reduce_sum(np.full((3, 3), comm.rank + 1.0), root=0)
    if comm.rank == 0:
        assert np.array_equal(total, np.full((3, 3), 3.0))
    else:
        assert total is None
    assert np.array_equal(comm.allreduce_sum(np.full(4, comm.rank + 1.0)), np.full(4, 3.0))
    assert comm.max_over_ranks(comm.rank * 2.5) == 2.5
    assert comm.sum_over_ranks(1.0) == 2.0

    # band pool: each rank owns bands b % 2 == rank; cube-level results are identical on all ranks
    pool = BandWorkerPool(nband, comm=comm, worker_cls=FakeWorker)
    assert sorted(pool.workers) == local_bands(nband, comm.rank, 2)
    diag = [1.0 + b + rng.random((nx, ny)) for b in range(nband)]
    pool.init_hess(diag, nx, ny, 2 * nx, 2 * ny, np.full(nband, 0.5), [None] * nband)
    got = pool.hess_dot(x)
    ref = np.stack([(diag[b] + 0.5) * x[b] for b in range(nband)])
    np.testing.assert_allclose(got, ref, rtol=1e-14)
    sol = pool.hess_cg(ref, None, 1e-6, 10, 1, 0)
    np.testing.assert_allclose(sol, x, rtol=1e-13)

    dirty = rng.standard_normal((nband, 1, nx, ny))
    model = rng.standard_normal((nband, 1, nx, ny))
    scales = [2.0 + b for b in range(nband)]
    pool.set_bands(dirty, scales)
    res = pool.residual(model, 1e-6)
    ref = np.stack([dirty[b] - scales[b] * model[b] for b in range(nband)])
    np.testing.assert_allclose(res, ref, rtol=1e-14)
    mfs = pool.residual_mfs(model, 1e-6, wsum=4.0, root=0)
    if comm.rank == 0:
        np.testing.assert_allclose(mfs, ref.sum(axis=0) / 4.0, rtol=1e-13)
    else:
        assert mfs is None
    # Psi role + band-sharded l21 dual update (config-4 style): one all-reduce completes the band sum
    from oracle import psi as opsi
    from pfb_imaging_amd.operators.psi import PsiNocopytRay
    from pfb_imaging_amd.prox import dual_update_bands

    px, py = 16, 12
    psi = PsiNocopytRay(nband, px, py, ("self", "db1", "db2"), 2, workers=pool)
    full_psi = opsi.Psi(nband, px, py, ("self", "db1", "db2"), 2)
    assert (psi.nxmax, psi.nymax) == (full_psi.nxmax, full_psi.nymax)
    img = rng.standard_normal((nband, px, py))
    alpha = np.zeros((nband, 3, psi.nxmax, psi.nymax))
    psi.dot(img, alpha)
    ref_alpha = np.zeros_like(alpha)
    full_psi.dot(img, ref_alpha)
    np.testing.assert_allclose(alpha, ref_alpha, rtol=0, atol=1e-13)
    back = np.zeros_like(img)
    psi.hdot(alpha, back)
    np.testing.assert_allclose(back, 3 * img, rtol=0, atol=1e-12)
    vp = rng.standard_normal(alpha.shape)
    wgt = np.abs(rng.standard_normal(alpha.shape[1:])) + 0.05

    def vtilde_sum(vp_loc, v_loc, sigma):  # CPU stand-ins of the two device phases (oracle arithmetic)
        v_loc[...] = vp_loc + sigma * v_loc
        return v_loc.sum(axis=0)

    def scale(v_loc, lam, w, total):
        a = np.abs(total)
        v_loc *= np.where(a > lam * w, lam * w / np.where(a > 0, a, 1.0), 1.0)[None]

    v = alpha.copy()
    dual_update_bands(vp, v, 0.8, 1.7, wgt, comm=comm, bands=pool.local, phases=(vtilde_sum, scale))
    np.testing.assert_allclose(v, opsi.dual_update(vp, alpha.copy(), 0.8, 1.7, wgt), rtol=0, atol=1e-13)

    # as many bands per rank everywhere (nband % world == 0): the cube exchange is an all-gather of the local bands
    pool4 = BandWorkerPool(4, comm=comm, worker_cls=FakeWorker)
    pool4.init_hess(diag[:4], nx, ny, 2 * nx, 2 * ny, np.full(4, 0.25), [None] * 4)
    np.testing.assert_allclose(pool4.hess_dot(x[:4]), np.stack([(diag[b] + 0.25) * x[b] for b in range(4)]), rtol=1e-14)
    assert np.array_equal(comm.allgather(np.full((2, 3), float(comm.rank))), np.stack([np.zeros((2, 3)), np.ones((2, 3))]))

    # fewer bands than ranks (e.g. 4 bands on 8 GPUs): rank 1 holds no band, takes part in every collective, and the
    # choice between the device-resident and the generic primal-dual / power-method loop is made by ALL ranks together
    from pfb_imaging_amd.operators.hessian import HessTreeRay
    from pfb_imaging_amd.opt import PrimalDual

    pool1 = BandWorkerPool(1, comm=comm, worker_cls=FakeWorker)
    assert pool1.local == ([0] if comm.rank == 0 else [])
    tree1 = HessTreeRay([diag[0]], nx, ny, 2 * nx, 2 * ny, etas=0.5, workers=pool1)
    np.testing.assert_allclose(tree1.dot(x[:1]), ((diag[0] + 0.5) * x[0])[None], rtol=1e-14)
    np.testing.assert_allclose(tree1.cg(((diag[0] + 0.5) * x[0])[None]), x[:1], rtol=1e-13)
    assert PrimalDual._hess_bands(tree1, 1) is None  # host transport: every rank agrees on the generic loop
    psi1 = PsiNocopytRay(1, px, py, ("self", "db1"), 2, workers=pool1)
    full1 = opsi.Psi(1, px, py, ("self", "db1"), 2)
    assert (psi1.nxmax, psi1.nymax) == (full1.nxmax, full1.nymax)  # the bandless rank learned the shape from the others
    a1 = np.zeros((1, 2, psi1.nxmax, psi1.nymax))
    psi1.dot(img[:1], a1)
    r1 = np.zeros_like(a1)
    full1.dot(img[:1], r1)
    np.testing.assert_allclose(a1, r1, rtol=0, atol=1e-13)

    # single band sharded by row blocks (config-5 style): partial images are summed, degridding is local
    from oracle import wgridder as owg
    from pfb_imaging_amd.parallel import RowShardedGridder, row_block
    from pfb_imaging_amd.utils import synth

    class OracleGridder:
        """CPU stand-in with the Gridder interface (the oracle restatement)."""

        def __init__(self, uvw, freq, mask, **kw):
            self.p = owg.Plan(uvw, freq, mask, kw["npix_x"], kw["npix_y"], kw["pixsize_x"], kw["pixsize_y"], 0.0, 0.0,
                              kw["epsilon"], False, True, False, True, False)
            self.wgt = None

        def vis2dirty(self, vis, wgt=None):
            return self.p.vis2dirty(vis, wgt)

        def dirty2vis(self, dirty, wgt=None):
            return self.p.dirty2vis(dirty, wgt)

        def set_weights(self, wgt):
            self.wgt = wgt

        def hessian(self, x, beam=None, eta=0.0, wsum=0.0):
            xb = x if beam is None else x * beam
            out = self.p.vis2dirty(self.p.dirty2vis(xb), self.wgt)
            if beam is not None:
                out = out * beam
            return out / wsum if wsum else out

        def close(self):
            pass

    assert [row_block(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    c = synth.make_case(401, 2, 24, zscale=0.2, seed=3)
    kw = dict(npix_x=24, npix_y=24, pixsize_x=c["cell"] * 30, pixsize_y=c["cell"] * 30, epsilon=1e-6)
    sh = RowShardedGridder(comm, c["uvw"], c["freq"], c["mask"], gridder_cls=OracleGridder, **kw)
    full = OracleGridder(c["uvw"], c["freq"], c["mask"], **kw)
    ref = full.vis2dirty(c["vis"], c["wgt"])
    got = sh.vis2dirty(c["vis"], c["wgt"])
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-6
    got0 = sh.vis2dirty(c["vis"], c["wgt"], root=0)
    assert (got0 is None) == (comm.rank != 0)
    mv = sh.dirty2vis(c["x"])
    refv = full.dirty2vis(c["x"])[sh.r0:sh.r1]
    assert np.linalg.norm(mv - refv) / np.linalg.norm(refv) < 1e-6
    sh.set_weights(c["wgt"])
    full.set_weights(c["wgt"])
    hs = sh.hessian(c["x"], eta=0.3, wsum=7.0)
    hr = full.hessian(c["x"], wsum=7.0) + 0.3 * c["x"]
    assert np.linalg.norm(hs - hr) / np.linalg.norm(hr) < 1e-6

    # single band sharded by |w| (the split that shards the w-planes): the two ranks own disjoint, complementary, |w|-ordered
    # row sets; partial images sum to the unsharded image, each rank degrids its own rows, the Hessian needs one all-reduce
    from pfb_imaging_amd.parallel import WShardedGridder, partition_rows_by_w

    parts = partition_rows_by_w(c["uvw"], c["freq"], 2, plane_cost=0.73, vis_cost=0.27, support=0.125, mask=c["mask"])
    assert sorted(np.concatenate(parts).tolist()) == list(range(c["uvw"].shape[0]))
    wabs = np.abs(c["uvw"][:, 2])
    assert wabs[parts[0]].max() <= wabs[parts[1]].min() and np.all(np.diff(wabs[parts[1]]) >= 0)
    span = [wabs[p].max() - wabs[p].min() for p in parts]
    assert max(span) < 0.75 * (wabs.max() - wabs.min())          # neither rank spans (nearly) the whole w range
    ws = WShardedGridder(comm, c["uvw"], c["freq"], c["mask"], gridder_cls=OracleGridder, **kw)
    assert np.array_equal(ws.rows, parts[comm.rank]) and ws.planes_per_rank() == [0, 0]
    gw = ws.vis2dirty(c["vis"], c["wgt"])
    assert np.linalg.norm(gw - ref) / np.linalg.norm(ref) < 1e-6
    mw = ws.dirty2vis(c["x"])
    refw = full.dirty2vis(c["x"])[ws.rows]
    assert mw.shape == refw.shape and np.linalg.norm(mw - refw) / np.linalg.norm(refw) < 1e-6
    ws.set_weights(c["wgt"])
    hw = ws.hessian(c["x"], eta=0.3, wsum=7.0)
    assert np.linalg.norm(hw - hr) / np.linalg.norm(hr) < 1e-6
    # one rank: the partition is everything, in |w| order
    p1 = partition_rows_by_w(c["uvw"], c["freq"], 1)
    assert len(p1) == 1 and sorted(p1[0].tolist()) == list(range(c["uvw"].shape[0]))

    comm.barrier()
    print(f"rank {comm.rank} ok", flush=True)


if __name__ == "__main__":
    main()
