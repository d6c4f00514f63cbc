"""MI355X w-stacking gridder: drop-in for ``ducc0.wgridder.experimental.vis2dirty / dirty2vis``.

The two module-level functions accept exactly the keywords the reference passes
(/root/reference/src/pfb_imaging/operators/gridder.py:590-613, 485-503, 972-1016;
/root/reference/src/pfb_imaging/operators/hessian.py:50-89) and write ``dirty=`` / ``vis=``
out-parameters in place like ducc0 does.  Behind them a :class:`Gridder` handle binds what is
constant over a run (geometry + uvw + freq + mask: the inputs the reference pins per band,
/root/reference/src/pfb_imaging/operators/band_worker.py:61-106) and owns the tile-sorted
visibility index, kernel choice, correction image, rocFFT plans and device scratch.  Handles
are cached, so the stateless calls inside a CG loop pay the set-up once.
"""

import collections
import ctypes as ct
import os
import threading

import numpy as np

from . import _lib
from ._lib import CGInfo, PMInfo, GridderInfo, GridderParams, as_c, check, cint, f64, i64, lib, ptr


class Gridder:
    """Device-resident plan for one (geometry, uvw, freq, mask)."""

    def __init__(self, uvw, freq, mask=None, *, npix_x, npix_y, pixsize_x, pixsize_y, center_x=0.0, center_y=0.0,
                 epsilon, flip_u=False, flip_v=False, flip_w=False, do_wgridding=True, divide_by_n=True,
                 sigma_min=1.1, sigma_max=2.6, verbosity=0, force=None, force_wmode=None):
        _lib.require_gpu()
        uvw = as_c(uvw, np.float64)
        freq = as_c(freq, np.float64)
        if uvw.ndim != 2 or uvw.shape[1] != 3:
            raise ValueError(f"uvw must have shape (nrow, 3), got {uvw.shape}")
        if freq.ndim != 1:
            raise ValueError(f"freq must be one-dimensional, got {freq.shape}")
        self.nrow, self.nchan = uvw.shape[0], freq.size
        if mask is not None:
            mask = as_c(mask, np.uint8)
            if mask.shape != (self.nrow, self.nchan):
                raise ValueError(f"mask shape {mask.shape} != {(self.nrow, self.nchan)}")
        self.nx, self.ny = int(npix_x), int(npix_y)
        p = GridderParams(
            nrow=self.nrow, nchan=self.nchan, nx=self.nx, ny=self.ny, pixsize_x=pixsize_x, pixsize_y=pixsize_y,
            center_x=center_x, center_y=center_y, epsilon=epsilon, sigma_min=sigma_min, sigma_max=sigma_max,
            flip_u=int(bool(flip_u)), flip_v=int(bool(flip_v)), flip_w=int(bool(flip_w)),
            do_wgridding=int(bool(do_wgridding)), divide_by_n=int(bool(divide_by_n)), verbosity=int(verbosity),
            force_W=0 if force is None else int(force[1]), force_sigma=0.0 if force is None else float(force[0]),
            force_wmode=0 if force_wmode is None else int(force_wmode) + 1,
        )
        self._h = ct.c_void_p()
        check(lib().pfbhip_gridder_create(ct.byref(p), ptr(uvw), ptr(freq), ptr(mask), ct.byref(self._h)))
        info = GridderInfo()
        check(lib().pfbhip_gridder_get_info(self._h, ct.byref(info)))
        self.info = info.asdict()
        self.nactive = self.info["nactive"]
        self._weights_token = None

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().pfbhip_gridder_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- checks -----------------------------------------------------------
    def _vis(self, vis):
        vis = as_c(vis, np.complex128)
        if vis.shape != (self.nrow, self.nchan):
            raise ValueError(f"vis shape {vis.shape} != {(self.nrow, self.nchan)}")
        return vis

    def _wgt(self, wgt):
        if wgt is None:
            return None
        wgt = as_c(wgt, np.float64)
        if wgt.shape != (self.nrow, self.nchan):
            raise ValueError(f"wgt shape {wgt.shape} != {(self.nrow, self.nchan)}")
        return wgt

    def _img(self, x, name="dirty"):
        x = as_c(x, np.float64)
        if x.shape != (self.nx, self.ny):
            raise ValueError(f"{name} shape {x.shape} != {(self.nx, self.ny)}")
        return x

    # -- single precision at the boundary --------------------------------------
    # complex64 visibilities / float32 images and weights (the reference's precision="single", operators/gridder.py:58-100) cross
    # PCIe as they are -- half the bytes -- and are widened on the device; sums are formed in double (ducc0's
    # double_precision_accumulation=True, the reference's default).  Results come back in single precision like ducc0's.
    @staticmethod
    def _is_sp(a, kind):
        return isinstance(a, np.ndarray) and a.dtype == (np.complex64 if kind == "c" else np.float32)

    def _sp_wgt(self, wgt):
        if wgt is None:
            return None
        wgt = as_c(wgt, np.float32)
        if wgt.shape != (self.nrow, self.nchan):
            raise ValueError(f"wgt shape {wgt.shape} != {(self.nrow, self.nchan)}")
        return wgt

    def _sp_img(self, x, name):
        x = as_c(x, np.float32)
        if x.shape != (self.nx, self.ny):
            raise ValueError(f"{name} shape {x.shape} != {(self.nx, self.ny)}")
        return x

    # -- operators (host arrays) --------------------------------------------
    def vis2dirty(self, vis, wgt=None, out=None):
        """``out`` (double precision only: C-contiguous float64 of the image shape, e.g. one correlation of a caller's cube)
        receives the image in place."""
        if out is not None and (out.shape != (self.nx, self.ny) or out.dtype != np.float64 or not out.flags.c_contiguous):
            raise ValueError("out must be a C-contiguous float64 array of the image shape")
        if out is None and self._is_sp(vis, "c") and (wgt is None or self._is_sp(wgt, "r")):
            vis = as_c(vis, np.complex64)
            if vis.shape != (self.nrow, self.nchan):
                raise ValueError(f"vis shape {vis.shape} != {(self.nrow, self.nchan)}")
            out = _lib.result_empty((self.nx, self.ny), np.float32)
            check(lib().pfbhip_gridder_vis2dirty_sp(self._h, ptr(vis), ptr(self._sp_wgt(wgt)), ptr(out)))
            return out
        vis, wgt = self._vis(vis), self._wgt(wgt)
        if out is None:
            out = _lib.result_empty((self.nx, self.ny), np.float64)
        check(lib().pfbhip_gridder_vis2dirty(self._h, ptr(vis), ptr(wgt), ptr(out)))
        return out

    def vis2dirty_dev(self, vis, wgt, out_dev):
        """:meth:`vis2dirty` with the image left in HBM (``out_dev``: DeviceArray of the image shape)."""
        vis, wgt = self._vis(vis), self._wgt(wgt)
        check(lib().pfbhip_gridder_vis2dirty_dev(self._h, ptr(vis), ptr(wgt), out_dev.ptr))

    def dirty2vis(self, dirty, wgt=None):
        if self._is_sp(dirty, "r") and (wgt is None or self._is_sp(wgt, "r")):
            out = _lib.result_empty((self.nrow, self.nchan), np.complex64)
            check(lib().pfbhip_gridder_dirty2vis_sp(self._h, ptr(self._sp_img(dirty, "dirty")), ptr(self._sp_wgt(wgt)), ptr(out)))
            return out
        dirty, wgt = self._img(dirty), self._wgt(wgt)
        out = _lib.result_empty((self.nrow, self.nchan), np.complex128)
        check(lib().pfbhip_gridder_dirty2vis(self._h, ptr(dirty), ptr(wgt), ptr(out)))
        return out

    def set_weights(self, wgt):
        if self._is_sp(wgt, "r"):
            check(lib().pfbhip_gridder_set_weights_sp(self._h, ptr(self._sp_wgt(wgt))))
            self._weights_token = object()
            return
        wgt = self._wgt(wgt)
        check(lib().pfbhip_gridder_set_weights(self._h, ptr(wgt)))
        self._weights_token = object()

    def hessian(self, x, beam=None, eta=0.0, wsum=0.0, out=None):
        """beam * R^H W R (beam * x) / wsum + eta x with the weights bound by :meth:`set_weights`; ``out`` (C-contiguous
        float64, not ``x``) receives the result in place.  A float32 ``x`` (with a float32 or no beam) takes the
        single-precision boundary: float32 in, float32 out, sums in double."""
        if self._is_sp(x, "r") and (beam is None or self._is_sp(beam, "r")):
            x = self._sp_img(x, "x")
            beam = None if beam is None else self._sp_img(beam, "beam")
            if out is None:
                out = _lib.result_empty(x.shape, np.float32)
            elif out.shape != x.shape or out.dtype != np.float32 or not out.flags.c_contiguous or np.shares_memory(out, x):
                raise ValueError("out must be a C-contiguous float32 array of the image shape that does not alias x")
            check(lib().pfbhip_gridder_hessian_sp(self._h, ptr(x), ptr(beam), f64(eta or 0.0), f64(wsum or 0.0), ptr(out)))
            return out
        x = self._img(x, "x")
        beam = None if beam is None else self._img(beam, "beam")
        if out is None:
            out = _lib.result_empty(x.shape, np.float64)
        elif out.shape != x.shape or out.dtype != np.float64 or not out.flags.c_contiguous or np.shares_memory(out, x):
            raise ValueError("out must be a C-contiguous float64 array of the image shape that does not alias x")
        check(lib().pfbhip_gridder_hessian(self._h, ptr(x), ptr(beam), f64(eta or 0.0), f64(wsum or 0.0), ptr(out)))
        return out

    def hessian_dev(self, x_dev, out_dev, beam_dev=None, eta=0.0, wsum=0.0):
        check(lib().pfbhip_gridder_hessian_dev(self._h, x_dev.ptr, None if beam_dev is None else beam_dev.ptr,
                                               f64(eta or 0.0), f64(wsum or 0.0), out_dev.ptr))

    def residual_dev(self, model_dev, acc_dev, out_dev, beam_dev=None):
        """``out = acc - R^H W R (beam * model)`` with every image a ``DeviceArray`` (``out`` may be ``acc``): the exact
        residual of one partition, weights as bound by :meth:`set_weights`."""
        check(lib().pfbhip_gridder_residual_dev(self._h, model_dev.ptr, None if beam_dev is None else beam_dev.ptr, acc_dev.ptr,
                                                out_dev.ptr))

    def degrid_dev(self, dirty_dev, vis_sorted_dev):
        check(lib().pfbhip_gridder_degrid_dev(self._h, dirty_dev.ptr, vis_sorted_dev.ptr))

    def grid_dev(self, vis_sorted_dev, dirty_dev):
        check(lib().pfbhip_gridder_grid_dev(self._h, vis_sorted_dev.ptr, dirty_dev.ptr))

    def cg(self, rhs, x0=None, beam=None, eta=0.0, wsum=0.0, tol=1e-5, maxit=500, minit=100):
        """On-device CG solve of ``hessian(x) = rhs`` (pcg_numba semantics, x0 is not mutated)."""
        rhs = self._img(rhs, "rhs")
        x = np.zeros_like(rhs) if x0 is None else self._img(x0, "x0").copy()
        beam = None if beam is None else self._img(beam, "beam")
        info = CGInfo()
        check(lib().pfbhip_gridder_cg(self._h, ptr(beam), f64(eta or 0.0), f64(wsum or 0.0), ptr(rhs), ptr(x),
                                      cint(0 if x0 is None else 1), f64(tol), cint(maxit), cint(minit),
                                      ct.byref(info)))
        self.last_cg = dict(iters=info.iters, status=info.status, eps=info.eps, phi=info.phi)
        return x

    def cg_dev(self, rhs_dev, x_dev, beam_dev=None, eta=0.0, wsum=0.0, has_x0=False, tol=1e-5, maxit=500, minit=100):
        """The same solve with rhs / x (and beam) resident in HBM (``DeviceArray``); returns the CG info dict."""
        info = CGInfo()
        check(lib().pfbhip_gridder_cg_dev(self._h, None if beam_dev is None else beam_dev.ptr, f64(eta or 0.0), f64(wsum or 0.0),
                                          rhs_dev.ptr, x_dev.ptr, cint(int(bool(has_x0))), f64(tol), cint(maxit), cint(minit),
                                          ct.byref(info)))
        self.last_cg = dict(iters=info.iters, status=info.status, eps=info.eps, phi=info.phi)
        return self.last_cg

    def power_method(self, b0, beam=None, eta=0.0, wsum=0.0, tol=1e-5, maxit=250):
        """On-device power iteration on :meth:`hessian` (power_method_numba semantics): returns ``(beta, b)``."""
        b = self._img(b0, "b0").copy()
        beam = None if beam is None else self._img(beam, "beam")
        info = PMInfo()
        check(lib().pfbhip_gridder_power_method(self._h, ptr(beam), f64(eta or 0.0), f64(wsum or 0.0), ptr(b), f64(tol),
                                                cint(maxit), ct.byref(info)))
        self.last_pm = dict(iters=info.iters, status=info.status, eps=info.eps)
        return float(info.beta), b

    # -- introspection (parity tests, bench) --------------------------------
    def binmap(self):
        n = self.nrow * self.nchan
        iu0 = np.empty(n, dtype=np.int32)
        iv0 = np.empty(n, dtype=np.int32)
        p0 = np.empty(n, dtype=np.int32)
        flip = np.empty(n, dtype=np.uint8)
        order = np.empty(max(self.nactive, 1), dtype=np.int64)
        check(lib().pfbhip_gridder_get_binmap(self._h, ptr(iu0), ptr(iv0), ptr(p0), ptr(flip), ptr(order)))
        return dict(iu0=iu0, iv0=iv0, p0=p0, flip=flip, order=order[: self.nactive])

    def grid_plane(self, vis, wgt, plane):
        vis, wgt = self._vis(vis), self._wgt(wgt)
        out = np.empty((self.info["nu"], self.info["nv"]), dtype=np.complex128)
        check(lib().pfbhip_gridder_grid_plane(self._h, ptr(vis), ptr(wgt), i64(plane), ptr(out)))
        return out

    def profile(self, enable=True):
        check(lib().pfbhip_gridder_profile(self._h, cint(int(enable))))

    def profile_get(self, reset=True):
        ms = (f64 * _lib.NSTAGES)()
        calls = (i64 * _lib.NSTAGES)()
        check(lib().pfbhip_gridder_profile_get(self._h, ms, calls, cint(int(reset))))
        return {n: (ms[i], calls[i]) for i, n in enumerate(_lib.STAGE_NAMES)}

    def oracle_params(self):
        """The plan's choices in the keyword form oracle.wgridder.GridParams takes (tests only)."""
        i = self.info
        return dict(nu=i["nu"], nv=i["nv"], W=i["W"], beta=i["beta"], sigma=i["sigma"], nplanes=i["nplanes"],
                    wmin=i["wmin"], dw=i["dw"], nshift=i["nshift"], lshift=i["lshift"], mshift=i["mshift"],
                    tile=i["tile"], wmode=i["wmode"], wcenter=i["wcenter"], whalf=i["whalf"], nderiv=i["nderiv"])

    def refresh_info(self):
        """Re-read the plan's info (counters such as ``graph_replays`` change over its life)."""
        info = GridderInfo()
        check(lib().pfbhip_gridder_get_info(self._h, ct.byref(info)))
        self.info = info.asdict()
        return self.info

    def planes(self):
        w = np.empty(self.info["nplanes"], dtype=np.float64)
        check(lib().pfbhip_gridder_get_planes(self._h, ptr(w)))
        return w


# ---------------------------------------------------------------------------
# handle cache for the stateless ducc0-style calls
# ---------------------------------------------------------------------------

_CACHE_SIZE = int(os.environ.get("PFBHIP_PLAN_CACHE", "4"))
_cache = collections.OrderedDict()
_cache_lock = threading.Lock()  # (the map only: a plan evicted while another thread still runs on it is the caller's cache size to choose)


def _fingerprint(a):
    """Content key of an input array: shape, dtype and a hash of every byte (see _lib.content_key)."""
    return _lib.content_key(a)


def clear_cache():
    while _cache:
        _, g = _cache.popitem()
        g.close()


def _get_gridder(uvw, freq, mask, **kw):
    uvw = as_c(uvw, np.float64)
    freq = as_c(freq, np.float64)
    mask = None if mask is None else as_c(mask, np.uint8)
    if _CACHE_SIZE <= 0:
        return Gridder(uvw, freq, mask, **kw), False
    key = (_fingerprint(uvw), _fingerprint(freq), _fingerprint(mask), tuple(sorted(kw.items())))
    with _cache_lock:
        g = _cache.get(key)
        if g is not None:
            _cache.move_to_end(key)
            return g, True
    g = Gridder(uvw, freq, mask, **kw)
    evicted = []
    with _cache_lock:
        if key in _cache:  # (another thread planned the same inputs meanwhile: keep its plan)
            evicted.append(g)
            g = _cache[key]
        else:
            _cache[key] = g
            while len(_cache) > _CACHE_SIZE:
                evicted.append(_cache.popitem(last=False)[1])
    for old in evicted:
        old.close()
    return g, True


def vis2dirty(*, uvw, freq, vis, wgt=None, mask=None, npix_x, npix_y, pixsize_x, pixsize_y, center_x=0.0,
              center_y=0.0, epsilon, flip_u=False, flip_v=False, flip_w=False, do_wgridding, divide_by_n=True,
              nthreads=1, sigma_min=1.1, sigma_max=2.6, double_precision_accumulation=False, verbosity=0,
              dirty=None, allow_nshift=True, gpu=False):
    """``ducc0.wgridder.experimental.vis2dirty`` on the GPU.

    ``nthreads``, ``double_precision_accumulation`` (the grid and image are always accumulated in
    double here), ``allow_nshift`` and ``gpu`` are accepted and ignored.  Output precision follows
    ``vis`` (complex64 -> float32) like ducc0; complex64 visibilities (and float32 weights) are uploaded as they
    are and widened on the device (``pfbhip_gridder_vis2dirty_sp``).
    """
    vis_arr = np.asarray(vis)
    single = vis_arr.dtype == np.complex64
    if single and wgt is not None and np.asarray(wgt).dtype != np.float32:
        wgt = np.asarray(wgt, dtype=np.float32)  # (ducc0 wants the weights in the visibilities' precision)
    g, cached = _get_gridder(uvw, freq, mask, npix_x=int(npix_x), npix_y=int(npix_y), pixsize_x=float(pixsize_x),
                             pixsize_y=float(pixsize_y), center_x=float(center_x), center_y=float(center_y),
                             epsilon=float(epsilon), flip_u=bool(flip_u), flip_v=bool(flip_v), flip_w=bool(flip_w),
                             do_wgridding=bool(do_wgridding), divide_by_n=bool(divide_by_n),
                             sigma_min=float(sigma_min), sigma_max=float(sigma_max))
    try:
        out = g.vis2dirty(vis_arr, wgt)
    finally:
        if not cached:
            g.close()
    if dirty is not None:
        if dirty.shape != out.shape:
            raise ValueError(f"dirty shape {dirty.shape} != {out.shape}")
        dirty[...] = out
        return dirty
    return out.astype(np.float32, copy=False) if single else out


def dirty2vis(*, uvw, freq, dirty, wgt=None, mask=None, pixsize_x, pixsize_y, center_x=0.0, center_y=0.0, epsilon,
              flip_u=False, flip_v=False, flip_w=False, do_wgridding, divide_by_n=True, nthreads=1, sigma_min=1.1,
              sigma_max=2.6, verbosity=0, vis=None, allow_nshift=True, gpu=False):
    """``ducc0.wgridder.experimental.dirty2vis`` on the GPU (output precision follows ``dirty``)."""
    dirty_arr = np.asarray(dirty)
    if dirty_arr.ndim != 2:
        raise ValueError(f"dirty must be two-dimensional, got {dirty_arr.shape}")
    single = dirty_arr.dtype == np.float32
    if single and wgt is not None and np.asarray(wgt).dtype != np.float32:
        wgt = np.asarray(wgt, dtype=np.float32)
    nx, ny = dirty_arr.shape
    g, cached = _get_gridder(uvw, freq, mask, npix_x=int(nx), npix_y=int(ny), pixsize_x=float(pixsize_x),
                             pixsize_y=float(pixsize_y), center_x=float(center_x), center_y=float(center_y),
                             epsilon=float(epsilon), flip_u=bool(flip_u), flip_v=bool(flip_v), flip_w=bool(flip_w),
                             do_wgridding=bool(do_wgridding), divide_by_n=bool(divide_by_n),
                             sigma_min=float(sigma_min), sigma_max=float(sigma_max))
    try:
        out = g.dirty2vis(dirty_arr, wgt)
    finally:
        if not cached:
            g.close()
    if vis is not None:
        if vis.shape != out.shape:
            raise ValueError(f"vis shape {vis.shape} != {out.shape}")
        vis[...] = out
        return vis
    return out.astype(np.complex64, copy=False) if single else out


# ducc0 exposes the same callables under ducc0.wgridder.experimental
class _Experimental:
    vis2dirty = staticmethod(vis2dirty)
    dirty2vis = staticmethod(dirty2vis)


experimental = _Experimental()
