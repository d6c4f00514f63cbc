"""Per-band worker co-locating a band's deconvolution state on its GPU.

Mirrors /root/reference/src/pfb_imaging/operators/band_worker.py:23-319.  The reference runs
one Ray actor per band; here a band is owned by one GPU process (``parallel.BandComm``:
band ``b`` -> rank ``b % world_size``), holding its HessianTree (PSFs/beams resident in HBM,
CG on the device) and its gridding inputs (tile-sorted visibilities resident in HBM) for the
exact residual.  ``BandWorkerPool`` keeps the reference's cube-level methods; with a
communicator every rank calls them collectively (SPMD) and per-band results are exchanged
with one RCCL all-reduce of the cube (each band is produced by exactly one rank), while
``residual_mfs`` uses the RCCL sum-to-root that replaces the driver-side band sum
(core/deconv.py:320-321).  Without a communicator all bands run in-process on the current
GPU, like the reference's ``nband == 1`` local path (band_worker.py:220-223).

The wavelet role (``init_psi`` / ``psi_dot`` / ``psi_hdot``, band_worker.py:144-163, 291-301) runs the
SARA dictionary of ``operators.psi`` on the band's GPU; ``dual_update`` is the band-sharded form of
``dual_update_numba_fast`` (prox/prox_21m.py:105-135): the band sum is completed with ONE all-reduce.
"""

import os

import numpy as np

from .. import _lib
from ..misc import resize_thread_pool
from ..parallel import local_bands


class _BandWorkerImpl:
    def __init__(self, nthreads):
        resize_thread_pool(nthreads)
        self._nthreads = nthreads
        self._hess = None
        self._parts = None
        self._hess_parts = None
        self._dirty = None
        self._resid = None

    # --- band loading ---
    def set_band(self, dirty, parts, hess_parts=None):
        """In-memory equivalent of ``load_band``: ``dirty (corr,nx,ny)``, gridding partitions
        (UVW/WEIGHT/MASK/FREQ/BEAM + l0/m0) and optionally the Hessian partitions."""
        self._dirty = None if dirty is None else np.asarray(dirty)
        self._parts = list(parts) if parts is not None else None
        self._hess_parts = hess_parts
        if self._resid is not None:
            self._resid.close()
            self._resid = None

    def load_band(self, store_url, node_name):
        """Read this band's inputs from the ``.dt`` store (band_worker.py:61-106): ``store_url`` is a store path -- a local
        zarr-v2 directory store is read by this package's own reader (``store.DirStore``: chunk files straight into pinned
        memory, no xarray / zarr needed), anything else is opened with xarray, as the reference does -- or an already-open
        store / in-memory mapping (see ``pfb_imaging_amd.store``).
        Every array is decoded straight into page-locked memory and stays with this worker; the device plans built from
        them upload at the PCIe rate."""
        import gc

        from ..store import load_band as _load

        try:
            store = store_url
            if isinstance(store_url, bytes):
                store_url = store_url.decode()
            local = store_url[len("file://"):] if isinstance(store_url, str) and store_url.startswith("file://") else store_url
            if isinstance(local, (str, os.PathLike)) and os.path.exists(os.path.join(os.fspath(local), ".zgroup")):
                from ..store import open_store

                store = open_store(local)
            elif isinstance(store_url, str):
                try:
                    import xarray as xr
                except ImportError as e:  # pragma: no cover - the storage stack is not part of this package
                    raise ImportError("opening a store path needs xarray + zarr (the reference's storage layer); pass an "
                                      "open store / mapping, or use set_band() with in-memory arrays") from e
                store = xr.open_datatree(store_url, engine="zarr", chunks=None)
            dirty, parts, hess_parts = _load(store, node_name)
            self.set_band(dirty, parts, hess_parts)
        finally:
            gc.collect()

    # --- Hessian role ---
    def init_hess(self, partitions, nx, ny, nx_psf, ny_psf, eta, wsum):
        from .hessian import HessianTree

        if partitions is None:
            partitions = self._hess_parts
            if partitions is None:
                raise RuntimeError("no partitions passed and none loaded; call load_band first")
        self._hess = HessianTree(partitions, nx, ny, nx_psf, ny_psf, eta=eta, nthreads=self._nthreads, wsum=wsum)

    def hess_dot(self, x, out=None):
        return self._hess.dot(x, out=out)

    def cg(self, rhs, x0, tol, maxit, minit, verbosity):
        # whole solve on the device; x0 is copied (Ray-style read-only inputs are never written)
        return self._hess.cg(rhs, x0=x0, tol=tol, maxit=maxit, minit=minit)

    # --- exact residual role ---
    def residual(self, model, cell_rad, epsilon, do_wgridding, double_accum, out=None):
        from .gridder import PartitionResidual

        ncorr, nx, ny = self._dirty.shape
        key = (cell_rad, epsilon, bool(do_wgridding))
        if self._resid is None or self._resid_key != key:
            if self._resid is not None:
                self._resid.close()
            self._resid = PartitionResidual(self._parts, nx, ny, cell_rad, epsilon=epsilon, do_wgridding=do_wgridding)
            self._resid_key = key
        if not _lib.any_nonzero(model):
            if out is None:
                return self._dirty - np.zeros_like(self._dirty)
            out[...] = self._dirty
            return out
        return self._resid.residual(self._dirty, model, out=out)

    # --- wavelet role (band_worker.py:144-163) ---
    def init_psi(self, nx, ny, bases, nlevel):
        from .psi import PsiBand

        self._psib = PsiBand(nx, ny, tuple(bases), nlevel)
        self._alphao = _lib.result_empty((self._psib.nbasis, self._psib.nxmax, self._psib.nymax), np.float64)
        self._xo = _lib.result_empty((nx, ny), np.float64)
        return int(self._psib.nxmax), int(self._psib.nymax)

    def psi_dot(self, x, out=None):
        out = self._alphao if out is None else out
        self._psib.dot(x, out)
        return out

    def psi_hdot(self, alpha, out=None):
        out = self._xo if out is None else out
        self._psib.hdot(alpha, out)
        return out

    # --- telemetry ---
    def get_mem(self):
        import ctypes as ct
        import os

        free, total = ct.c_size_t(0), ct.c_size_t(0)
        _lib.check(_lib.lib().pfbhip_mem_info(ct.byref(free), ct.byref(total)))
        return {"pid": os.getpid(), "hbm_used_gb": (total.value - free.value) / 2**30, "hbm_total_gb": total.value / 2**30}


class BandWorkerPool:
    """nband band workers plus cube-level dispatch (band_worker.py:209-319).

    Args:
        nband: number of imaging bands.
        nthreads: kept for signature compatibility.
        comm: optional ``parallel.BandComm``; bands are then sharded ``b % world_size``.
    """

    def __init__(self, nband, nthreads=1, comm=None, worker_cls=_BandWorkerImpl):
        self.nband = nband
        self.nthreads_per_band = max(1, nthreads)
        self.comm = comm
        rank = 0 if comm is None else comm.rank
        world = 1 if comm is None else comm.world_size
        self.local = local_bands(nband, rank, world)
        self.workers = {b: worker_cls(self.nthreads_per_band) for b in self.local}
        self.actors = None  # no Ray
        import inspect

        self._residual_takes_out = "out" in inspect.signature(worker_cls.residual).parameters
        self._hess_takes_out = "out" in inspect.signature(worker_cls.hess_dot).parameters
        self._psi_takes_out = all("out" in inspect.signature(getattr(worker_cls, m)).parameters for m in ("psi_dot", "psi_hdot"))
        # The reference dispatches a method to every band's actor at once and gathers (band_worker.py:239-246).  Here the
        # local bands run on host threads: every band's handle owns its HIP stream and its buffers and the C calls release
        # the GIL, so one band's transfers overlap another's kernels -- and the two PCIe directions each other
        # (PFBHIP_BAND_THREADS, default 4; 1 = one band after the other).
        nthr = min(len(self.local), max(int(os.environ.get("PFBHIP_BAND_THREADS", "4")), 1))
        self._exec = None
        if nthr > 1:
            from concurrent.futures import ThreadPoolExecutor

            # HIP's current device is per host thread and new threads start on device 0: bind them to this thread's device
            # (worker classes of the CPU tests run without a GPU: nothing to bind)
            bind = dict(initializer=_lib.set_device, initargs=(_lib.get_device(),)) if _lib.device_count() > 0 else {}
            self._exec = ThreadPoolExecutor(max_workers=nthr, thread_name_prefix="pfbhip-band", **bind)

    def _map(self, method, per_band_args):
        """Run ``method(*args)`` on every LOCAL band worker; returns {band: result}."""
        if self._exec is None:
            return {b: getattr(self.workers[b], method)(*per_band_args[b]) for b in self.local}
        futs = {b: self._exec.submit(getattr(self.workers[b], method), *per_band_args[b]) for b in self.local}
        out, err = {}, None
        for b, f in futs.items():  # every band finishes before an error propagates: no call is left running behind it
            try:
                out[b] = f.result()
            except BaseException as e:  # noqa: BLE001 -- re-raised below
                err = err or e
        if err is not None:
            raise err
        return out

    def close(self):
        if self._exec is not None:
            self._exec.shutdown(wait=True)
            self._exec = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _exchange(self, cube):
        """Every band of ``cube`` was filled by exactly one rank (zeros elsewhere).  When every rank owns the same number of
        bands the exchange is an ALL-GATHER of the local bands (each byte crosses xGMI once, (N-1)/N of the cube per rank);
        otherwise the zero-padded cube is all-reduced (sum = gather).  Both run on the communicator's persistent device
        staging buffers."""
        if self.comm is None or self.comm.world_size == 1:
            return cube
        world = self.comm.world_size
        if self.nband % world == 0 and hasattr(self.comm, "allgather"):
            nloc = self.nband // world
            mine = np.ascontiguousarray(cube[self.local])            # bands rank, rank + N, ... (b % N == rank)
            got = self.comm.allgather(mine)                           # (world, nloc, ...)
            out = np.empty_like(cube)
            for r in range(world):
                out[r::world] = got[r][:nloc]
            return out
        return self.comm.allreduce_sum(cube).reshape(cube.shape)

    # --- band loading ---
    def load_bands(self, store_url, node_names):
        if len(node_names) != self.nband:
            raise ValueError(f"got {len(node_names)} band nodes for {self.nband} workers")
        self._map("load_band", [(store_url, node_names[b]) for b in range(self.nband)])

    def set_bands(self, dirty_per_band, parts_per_band, hess_parts_per_band=None):
        for b in self.local:
            self.workers[b].set_band(dirty_per_band[b], parts_per_band[b],
                                     None if hess_parts_per_band is None else hess_parts_per_band[b])

    # --- Hessian role ---
    def init_hess(self, partitions_per_band, nx, ny, nx_psf, ny_psf, etas, wsums):
        self._map("init_hess", [(None if partitions_per_band is None else partitions_per_band[b], nx, ny, nx_psf,
                                 ny_psf, etas[b], wsums[b]) for b in range(self.nband)])

    def hess_dot(self, x):
        out = _lib.result_empty(x.shape, np.float64)
        for b in range(self.nband):
            if b not in self.workers:
                out[b] = 0.0
        if self._hess_takes_out and x.ndim == 3:  # (nband, nx, ny): every band writes its (1, nx, ny) slice of the cube
            self._map("hess_dot", [(x[b], out[b:b + 1]) for b in range(self.nband)])
        else:
            for b, res in self._map("hess_dot", [(x[b],) for b in range(self.nband)]).items():
                out[b] = res[0]
        return self._exchange(out)

    def hess_cg(self, rhs, x0, tol, maxit, minit, verbosity):
        out = np.zeros_like(rhs, dtype=np.float64)
        args = [(rhs[b], None if x0 is None else x0[b], tol, maxit, minit, verbosity) for b in range(self.nband)]
        for b, res in self._map("cg", args).items():
            out[b] = res
        return self._exchange(out)

    # --- Psi role (band_worker.py:291-301) ---
    def init_psi(self, nx, ny, bases, nlevel):
        shapes = self._map("init_psi", [(nx, ny, tuple(bases), nlevel)] * self.nband)
        shape = next(iter(shapes.values()), (0, 0))  # (nxmax, nymax), identical across bands
        if self.comm is not None and self.comm.world_size > 1:  # a rank without bands (nband < world size) learns it too
            shape = (int(self.comm.max_over_ranks(shape[0])), int(self.comm.max_over_ranks(shape[1])))
        self._psi_shape = shape
        return self._psi_shape

    def _band_cube(self, method, inputs, target):
        """``method(inputs[b])`` of every local band, written straight into ``target[b]`` when no exchange follows (one
        process holds every band and the caller's array can take the results in place), else into a page-locked cube that
        the exchange completes."""
        solo = (self.comm is None or self.comm.world_size == 1) and target.dtype == np.float64 and target.flags.c_contiguous \
            and self._psi_takes_out
        out = target if solo else _lib.result_empty(target.shape, np.float64)
        if not solo:
            for b in range(self.nband):
                if b not in self.workers:
                    out[b] = 0.0
        if self._psi_takes_out:
            self._map(method, [(inputs[b], out[b]) for b in range(self.nband)])
        else:
            for b, res in self._map(method, [(inputs[b],) for b in range(self.nband)]).items():
                out[b] = res
        if not solo:
            target[...] = self._exchange(out)

    def psi_dot(self, x, alphao):
        self._band_cube("psi_dot", x, alphao)

    def psi_hdot(self, alpha, xo):
        self._band_cube("psi_hdot", alpha, xo)

    def dual_update(self, vp, v, lam, sigma=1.0, weight=None):
        """``dual_update_numba_fast`` (prox/prox_21m.py:105-135) over cubes ``(nband, nbasis, n1, n2)``, in place
        on ``v``.  Each rank updates the bands it owns; the band sum of vtilde is completed with one
        all-reduce (the only collective of the primal-dual iteration), after which every rank
        holds the full updated cube again."""
        from ..prox import dual_update_bands

        dual_update_bands(vp, v, lam, sigma, weight, comm=self.comm, bands=self.local)

    # --- exact residual role ---
    def residual(self, model, cell_rad, epsilon=1e-7, do_wgridding=True, double_accum=True):
        """Exact per-band residual for a ``(nband, corr, nx, ny)`` model cube."""
        # every band's result comes down straight into its slice of ONE page-locked cube (no zero-filled cube, no copy of the
        # band images on the host: at 8192^2 that copy cost as much as the residual itself)
        out = _lib.result_empty(model.shape, np.float64)
        for b in range(self.nband):
            if b not in self.workers:
                out[b] = 0.0
        args = [(model[b], cell_rad, epsilon, do_wgridding, double_accum) for b in range(self.nband)]
        if self._residual_takes_out:
            self._map("residual", [a + (out[b],) for b, a in enumerate(args)])
        else:  # (a worker class without the out argument: its arrays are copied in)
            for b, r in self._map("residual", args).items():
                out[b] = r
        return self._exchange(out)

    def residual_mfs(self, model, cell_rad, wsum, epsilon=1e-7, do_wgridding=True, double_accum=True, root=0):
        """``sum_b residual_b / wsum`` on the root rank (None elsewhere): the band reduce of
        core/deconv.py:320-321 as ONE sum-to-root instead of a gather + host sum."""
        args = [(model[b], cell_rad, epsilon, do_wgridding, double_accum) for b in range(self.nband)]
        res = self._map("residual", args)
        local = np.zeros(model.shape[1:], dtype=np.float64)
        for r in res.values():
            local += r
        total = local if self.comm is None else self.comm.reduce_sum(local, root=root)
        return None if total is None else total.reshape(local.shape) / wsum

    # --- telemetry ---
    def get_mem(self):
        return list(self._map("get_mem", [()] * self.nband).values())
