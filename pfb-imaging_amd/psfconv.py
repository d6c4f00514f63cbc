"""Device-resident PSF-convolution plan (Python face of ``pfbhip_psfconv_*``).

One plan per (nx, ny, nx_psf, ny_psf); Fourier-domain PSFs and image-plane beams/tapers are
bound to numbered slots and stay on the GPU.  Serves psf_convolve_*, hessian_psf_slice,
hess_direct_slice, HessPSF and HessianTree (/root/reference/src/pfb_imaging/operators/psf.py,
operators/hessian.py).
"""

import collections
import ctypes as ct

import numpy as np

from . import _lib
from ._lib import CGInfo, as_c, check, cint, f64, i64, lib, ptr


class PsfConv:
    def __init__(self, nx, ny, nx_psf, ny_psf):
        _lib.require_gpu()
        self.nx, self.ny, self.nx_psf, self.ny_psf = int(nx), int(ny), int(nx_psf), int(ny_psf)
        self.nyo2 = self.ny_psf // 2 + 1
        self._h = ct.c_void_p()
        check(lib().pfbhip_psfconv_create(i64(self.nx), i64(self.ny), i64(self.nx_psf), i64(self.ny_psf),
                                          ct.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().pfbhip_psfconv_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_psfhat(self, slot, psfhat):
        psfhat = np.asarray(psfhat)
        if psfhat.shape != (self.nx_psf, self.nyo2):
            raise ValueError(f"psfhat shape {psfhat.shape} != {(self.nx_psf, self.nyo2)}")
        is_complex = np.iscomplexobj(psfhat)
        a = as_c(psfhat, np.complex128 if is_complex else np.float64)
        check(lib().pfbhip_psfconv_set_psfhat(self._h, i64(slot), ptr(a), cint(int(is_complex))))

    def set_beam(self, slot, beam):
        if beam is not None:
            beam = as_c(beam, np.float64)
            if beam.shape != (self.nx, self.ny):
                raise ValueError(f"beam shape {beam.shape} != {(self.nx, self.ny)}")
        check(lib().pfbhip_psfconv_set_beam(self._h, i64(slot), ptr(beam)))

    def apply(self, x, psf_slot, beam_slot=-1, mode=0, shift=0.0, scale=1.0, eta=0.0, out=None, accumulate=False):
        x = as_c(x, np.float64)
        if x.shape != (self.nx, self.ny):
            raise ValueError(f"x shape {x.shape} != {(self.nx, self.ny)}")
        if out is None:
            if accumulate:
                raise ValueError("accumulate needs an output array")
            out = _lib.result_empty(x.shape, np.float64)
        target = out if (out.flags.c_contiguous and out.dtype == np.float64) else np.array(out, dtype=np.float64)
        check(lib().pfbhip_psfconv_apply(self._h, ptr(x), i64(psf_slot), i64(beam_slot), cint(mode), f64(shift),
                                         f64(scale), f64(eta or 0.0), cint(int(accumulate)), ptr(target)))
        if target is not out:
            out[...] = target
        return out

    def direct(self, x, psf_slot, taper_slot, shift, beam_slot=-1, min_beam=0.0, out=None, raw=None):
        """``hess_direct`` backward (division by ``psfhat + shift`` under the taper) followed, where ``beam_slot >= 0``, by the beam
        division of ``HessPSF.idot`` -- ``x /= beam**2`` where ``x > 0`` and ``beam > min_beam`` -- on the device.  ``raw`` (optional)
        receives the estimate before the division, ``out`` after it."""
        x = as_c(x, np.float64)
        out = _lib.result_empty(x.shape, np.float64) if out is None else out
        assert out.flags.c_contiguous and out.dtype == np.float64 and (raw is None or (raw.flags.c_contiguous and raw.dtype == np.float64))
        check(lib().pfbhip_psfconv_direct(self._h, ptr(x), i64(psf_slot), i64(taper_slot), f64(shift), i64(beam_slot), f64(min_beam),
                                          ptr(raw), ptr(out)))
        return out

    def cg(self, rhs, psf_slots, beam_slots, scale=1.0, eta=0.0, x0=None, tol=1e-5, maxit=500, minit=100):
        rhs = as_c(rhs, np.float64)
        x = np.zeros_like(rhs) if x0 is None else np.array(x0, dtype=np.float64, order="C")
        ps = np.ascontiguousarray(psf_slots, dtype=np.int64)
        bs = np.ascontiguousarray(beam_slots, dtype=np.int64)
        info = CGInfo()
        check(lib().pfbhip_psfconv_cg(self._h, i64(ps.size), ptr(ps), ptr(bs), f64(scale), f64(eta or 0.0), ptr(rhs),
                                      ptr(x), cint(0 if x0 is None else 1), f64(tol), cint(maxit), cint(minit),
                                      ct.byref(info)))
        self.last_cg = dict(iters=info.iters, status=info.status, eps=info.eps, phi=info.phi)
        return x


# ---- caches for the stateless reference-style calls ---------------------------------------

_plans = collections.OrderedDict()
_MAX_PLANS = 2
_SLOT_RING = 4


def _fingerprint(a):
    """Content key of psfhat: shape, dtype and a hash of every byte (an in-place edit or a reallocation at the same
    address must not reuse the resident copy)."""
    return _lib.content_key(a)


def cached_plan(nx, ny, nx_psf, ny_psf):
    key = (int(nx), int(ny), int(nx_psf), int(ny_psf))
    plan = _plans.get(key)
    if plan is None:
        plan = PsfConv(*key)
        plan._slot_keys = {}
        plan._slot_next = 0
        _plans[key] = plan
        while len(_plans) > _MAX_PLANS:
            _, old = _plans.popitem(last=False)
            old.close()
    else:
        _plans.move_to_end(key)
    return plan


def cached_psf_slot(plan, psfhat):
    """Upload ``psfhat`` into a small ring of slots unless an identical array is already resident."""
    psfhat = np.asarray(psfhat)
    key = _fingerprint(psfhat)
    slot = plan._slot_keys.get(key)
    if slot is None:
        slot = plan._slot_next % _SLOT_RING
        plan._slot_next += 1
        for k in [k for k, v in plan._slot_keys.items() if v == slot]:
            del plan._slot_keys[k]
        plan.set_psfhat(slot, psfhat)
        plan._slot_keys[key] = slot
    return slot


def clear_cache():
    while _plans:
        _, p = _plans.popitem()
        p.close()
