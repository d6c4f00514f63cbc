"""ctypes binding of libpfbhip.so (C-ABI: include/pfbhip.h).

The library is built in-tree by ``pfb-imaging_amd/csrc/Makefile`` (hipcc, gfx950).  There is
no fallback: if it is missing, importing any operator raises ImportError telling how to build.
"""

import ctypes as ct
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpfbhip.so")

i32 = ct.c_int32
i64 = ct.c_int64
f64 = ct.c_double
cint = ct.c_int
vp = ct.c_void_p
NSTAGES = 8
STAGE_NAMES = ("grid", "degrid", "fft_rows", "pad", "crop", "other", "fft_crop", "pad_fft")
UNIQUE_ID_BYTES = 128


class GridderParams(ct.Structure):
    _fields_ = [
        ("nrow", i64), ("nchan", i64), ("nx", i64), ("ny", i64),
        ("pixsize_x", f64), ("pixsize_y", f64), ("center_x", f64), ("center_y", f64),
        ("epsilon", f64), ("sigma_min", f64), ("sigma_max", f64),
        ("flip_u", i32), ("flip_v", i32), ("flip_w", i32), ("do_wgridding", i32), ("divide_by_n", i32),
        ("verbosity", i32), ("force_W", i32), ("force_wmode", i32), ("force_sigma", f64),
    ]


class GridderInfo(ct.Structure):
    _fields_ = [
        ("nu", i64), ("nv", i64), ("nplanes", i64), ("nactive", i64), ("ntiles", i64), ("nwork", i64),
        ("W", i32), ("tile", i32), ("beta", f64), ("sigma", f64), ("wmin", f64), ("dw", f64),
        ("nshift", f64), ("lshift", f64), ("mshift", f64), ("kernel_eps", f64),
        ("wmode", i32), ("occ_rows", i32), ("wcenter", f64), ("whalf", f64), ("device_bytes", ct.c_size_t),
        ("fft_mode", i32), ("screen_poly", i32), ("scatter_mode", i32), ("scatter_launches", i32),
        ("used_cells", i64), ("screen_composite", i32), ("screen_separable", i32), ("nderiv", i32), ("smax", f64), ("graph_replays", i64),
        ("scatter_block", i32), ("reserved0", i32),
    ]

    def asdict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


PD_NSTAGES = 5
PD_STAGE_NAMES = ("psi_analysis", "dual_update", "psi_synthesis", "psf_hessian", "primal_step")


class PDInfo(ct.Structure):
    _fields_ = [("iters", i32), ("status", i32), ("eps", f64), ("loop_ms", f64), ("stage_ms", f64 * PD_NSTAGES),
                ("stage_calls", i64 * PD_NSTAGES)]


class PMInfo(ct.Structure):
    _fields_ = [("iters", i32), ("status", i32), ("eps", f64), ("beta", f64)]


class CGInfo(ct.Structure):
    _fields_ = [("iters", i32), ("status", i32), ("eps", f64), ("phi", f64)]


# every symbol include/pfbhip.h declares (tests check the library exports all of them)
SYMBOLS = (
    "pfbhip_last_error", "pfbhip_device_count", "pfbhip_set_device", "pfbhip_get_device", "pfbhip_device_name",
    "pfbhip_mem_info", "pfbhip_device_cache", "pfbhip_resize_thread_pool", "pfbhip_thread_pool_size", "pfbhip_good_size",
    "pfbhip_hash64", "pfbhip_host_alloc", "pfbhip_host_free", "pfbhip_malloc", "pfbhip_free", "pfbhip_memcpy_h2d", "pfbhip_memcpy_d2h", "pfbhip_memcpy_d2d", "pfbhip_memset",
    "pfbhip_synchronize",
    "pfbhip_gridder_create", "pfbhip_gridder_destroy", "pfbhip_gridder_get_info", "pfbhip_gridder_get_binmap",
    "pfbhip_gridder_get_planes",
    "pfbhip_gridder_vis2dirty", "pfbhip_gridder_vis2dirty_dev", "pfbhip_gridder_dirty2vis", "pfbhip_gridder_grid_plane",
    "pfbhip_gridder_set_weights", "pfbhip_gridder_hessian", "pfbhip_gridder_hessian_dev", "pfbhip_gridder_residual_dev",
    "pfbhip_gridder_vis2dirty_sp", "pfbhip_gridder_dirty2vis_sp", "pfbhip_gridder_set_weights_sp", "pfbhip_gridder_hessian_sp",
    "pfbhip_gridder_degrid_dev", "pfbhip_gridder_grid_dev", "pfbhip_gridder_profile", "pfbhip_gridder_profile_get",
    "pfbhip_gridder_debug_stamps",
    "pfbhip_gridder_cg", "pfbhip_gridder_cg_dev",
    "pfbhip_r2c_2d", "pfbhip_r2c_2d_centred", "pfbhip_c2r_2d", "pfbhip_debug_rowfft",
    "pfbhip_psi_create", "pfbhip_psi_destroy", "pfbhip_psi_shape", "pfbhip_psi_dot", "pfbhip_psi_hdot",
    "pfbhip_psi_dot_dev", "pfbhip_psi_hdot_dev", "pfbhip_dual_update", "pfbhip_l21_vtilde_sum_dev",
    "pfbhip_l21_scale_dev", "pfbhip_prox_21m", "pfbhip_positivity", "pfbhip_positivity_dev", "pfbhip_primal_dual",
    "pfbhip_psfconv_power_method", "pfbhip_gridder_power_method",
    "pfbhip_psfconv_create", "pfbhip_psfconv_destroy", "pfbhip_psfconv_set_psfhat", "pfbhip_psfconv_set_beam",
    "pfbhip_psfconv_apply", "pfbhip_psfconv_apply_dev", "pfbhip_psfconv_direct", "pfbhip_psfconv_cg",
    "pfbhip_uvcell_index", "pfbhip_compute_counts", "pfbhip_counts_divide", "pfbhip_box_sum_counts",
    "pfbhip_filter_extreme_counts", "pfbhip_imaging_weights",
    "pfbhip_comm_unique_id", "pfbhip_comm_create", "pfbhip_comm_destroy", "pfbhip_comm_reduce_sum",
    "pfbhip_comm_allreduce_sum", "pfbhip_comm_allgather", "pfbhip_comm_allreduce_sum_host", "pfbhip_comm_reduce_sum_host",
    "pfbhip_comm_allgather_host", "pfbhip_comm_barrier",
)

_lib = None


def lib():
    """The loaded library; raises ImportError (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(hipcc, --offload-arch=gfx950) or `python -c 'import __graft_entry__ as g; g.build()'`. "
                "pfb-imaging_amd has no CPU fallback."
            )
        L = ct.CDLL(LIB_PATH)
        L.pfbhip_last_error.restype = ct.c_char_p
        L.pfbhip_good_size.restype = i64
        L.pfbhip_good_size.argtypes = [i64, cint]
        L.pfbhip_thread_pool_size.restype = cint
        L.pfbhip_hash64.restype = ct.c_uint64
        L.pfbhip_hash64.argtypes = [ct.c_void_p, ct.c_size_t]
        _lib = L
    return _lib


def last_error():
    return lib().pfbhip_last_error().decode("utf-8", "replace")


def check(status):
    if status == 0:
        return
    msg = last_error()
    if status == 1:
        raise ValueError(msg)
    raise RuntimeError(msg)


def device_count():
    n = cint(0)
    check(lib().pfbhip_device_count(ct.byref(n)))
    return n.value


def get_device():
    """The HIP device of the CALLING thread (HIP keeps one current device per host thread; new threads start on device 0)."""
    d = cint(0)
    check(lib().pfbhip_get_device(ct.byref(d)))
    return d.value


def set_device(device):
    check(lib().pfbhip_set_device(cint(int(device))))


def device_cache(flush=False):
    """Bytes of released device blocks the library keeps for the next plan of the same sizes (``pfbhip_device_cache``);
    ``flush=True`` returns them to the driver.  Returns the cached bytes before the flush."""
    n = ct.c_size_t(0)
    check(lib().pfbhip_device_cache(ct.byref(n), cint(1 if flush else 0)))
    return n.value


def require_gpu():
    if device_count() < 1:
        raise RuntimeError("pfb-imaging_amd: no AMD GPU visible (hipGetDeviceCount == 0); there is no CPU fallback")


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    if not a.flags.c_contiguous:
        raise ValueError("internal error: array is not C-contiguous")
    return a.ctypes.data_as(vp)


def as_c(a, dtype):
    """C-contiguous array of ``dtype`` without copying when possible (inputs may be read-only views)."""
    return np.ascontiguousarray(a, dtype=dtype)


# Page-locked result buffers whose arrays were garbage-collected wait here for the next result of the same size.  The pool is
# bounded (PFBHIP_PINNED_POOL_MB, default 2048) and evicts its OLDEST buffers first: a long-running process that returns
# arrays of many different sizes must not pile up locked memory -- the HIP runtime aborts the process when its own pinned
# allocations fail.
_pinned_pool = []  # [(nbytes, address), ...] oldest first
_PINNED_POOL_BYTES = int(os.environ.get("PFBHIP_PINNED_POOL_MB", "2048")) << 20
_pinned_pooled = [0]
_PINNED_MIN_BYTES = 1 << 20
# LIVE page-locked bytes (pooled + held by arrays the caller still references) are bounded too: a caller that keeps the
# DIRTY / PSF cubes of many bands would otherwise lock tens of GB.  Past the cap (PFBHIP_PINNED_MAX_MB; default a quarter of
# the physical memory, at least 4 GiB) results are ordinary pageable numpy arrays -- slower device-to-host copies, nothing else.
_pinned_live = [0]
# (BandWorkerPool runs its bands on threads, and finalizers run wherever the collector happens to: one re-entrant lock over the pool)
_pinned_lock = threading.RLock()


def _pinned_cap():
    env = os.environ.get("PFBHIP_PINNED_MAX_MB")
    if env is not None:
        return int(env) << 20
    try:
        phys = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    except (ValueError, OSError, AttributeError):
        phys = 64 << 30
    return max(phys // 4, 4 << 30)


def _pinned_free(addr, nbytes=0):
    _pinned_live[0] -= nbytes
    try:
        lib().pfbhip_host_free(vp(addr))
    except Exception:
        pass


def _pinned_trim(limit):
    while _pinned_pool and _pinned_pooled[0] > limit:
        nbytes, addr = _pinned_pool.pop(0)
        _pinned_pooled[0] -= nbytes
        _pinned_free(addr, nbytes)


def _pinned_release(addr, nbytes):
    with _pinned_lock:
        if nbytes > _PINNED_POOL_BYTES:
            _pinned_free(addr, nbytes)
            return
        _pinned_pool.append((nbytes, addr))
        _pinned_pooled[0] += nbytes
        _pinned_trim(_PINNED_POOL_BYTES)


def _pinned_take(nbytes):
    """Address of a page-locked buffer of ``nbytes`` (pooled or new), or None when the cap or the allocator says no.  Caller
    holds ``_pinned_lock``."""
    for i in range(len(_pinned_pool) - 1, -1, -1):  # newest first
        if _pinned_pool[i][0] == nbytes:
            addr = _pinned_pool.pop(i)[1]
            _pinned_pooled[0] -= nbytes
            return addr
    if _pinned_live[0] + nbytes > _pinned_cap():
        _pinned_trim(0)  # idle buffers first
        if _pinned_live[0] + nbytes > _pinned_cap():
            return None
    p = vp()
    if lib().pfbhip_host_alloc(ct.byref(p), ct.c_size_t(nbytes)) != 0:
        _pinned_trim(0)
        if lib().pfbhip_host_alloc(ct.byref(p), ct.c_size_t(nbytes)) != 0:
            return None
    _pinned_live[0] += nbytes
    return p.value


def result_empty(shape, dtype):
    """``np.empty(shape, dtype)`` for an array this package RETURNS (dirty image, visibilities), backed by page-locked host
    memory so that the device-to-host copy runs at the PCIe rate.  The buffer goes back to a small pool when the array (and
    every view of it) has been garbage-collected; small arrays are ordinary numpy allocations, and so is any array whose
    page-locked allocation fails (after the pool has been emptied)."""
    import weakref

    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in (shape if np.iterable(shape) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes < _PINNED_MIN_BYTES or os.environ.get("PFBHIP_PINNED_RESULTS", "1") == "0":
        return np.empty(shape, dtype=dtype)
    with _pinned_lock:
        addr = _pinned_take(nbytes)
    if addr is None:
        return np.empty(shape, dtype=dtype)
    buf = (ct.c_char * nbytes).from_address(addr)
    weakref.finalize(buf, _pinned_release, addr, nbytes)  # runs when the last array / view over `buf` is gone
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class DeviceArray:
    """A device allocation with numpy-like shape/dtype (host<->device copies are explicit)."""

    def __init__(self, shape, dtype=np.float64):
        self.shape = tuple(int(s) for s in (shape if np.iterable(shape) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = vp()
        check(lib().pfbhip_malloc(ct.byref(p), ct.c_size_t(max(self.nbytes, 1))))
        self.ptr = p

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        check(lib().pfbhip_memcpy_h2d(self.ptr, ptr(a), ct.c_size_t(self.nbytes)))

    def download(self, out=None):
        if out is None:
            out = result_empty(self.shape, self.dtype)
        assert out.nbytes == self.nbytes and out.flags.c_contiguous
        check(lib().pfbhip_memcpy_d2h(ptr(out), self.ptr, ct.c_size_t(self.nbytes)))
        return out

    def zero(self):
        check(lib().pfbhip_memset(self.ptr, 0, ct.c_size_t(self.nbytes)))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            lib().pfbhip_free(self.ptr)
            self.ptr = vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_ro_memo = {}  # (address, shape, dtype) -> (array kept alive, key, sample): read-only inputs only
_RO_MEMO_MAX = 32
_RO_MEMO_BYTES = int(os.environ.get("PFBHIP_RO_MEMO_MB", "4096")) << 20  # bound on the bytes the memo keeps alive
_ro_memo_bytes = [0]
_ro_lock = threading.Lock()


def _immutable(a):
    """True if nothing can write to ``a``'s memory through numpy: it and every ndarray it is a view of are read-only
    (the form in which Ray hands out pinned inputs, operators/band_worker.py:61-106 of the reference)."""
    obj = a
    while isinstance(obj, np.ndarray):
        if obj.flags.writeable:
            return False
        obj = obj.base
    return True


def _sample(c):
    """Cheap fingerprint of a C-contiguous array: <= 4096 evenly spaced 8-byte words plus the tail.  It guards the memo of
    content_key against memory that changes underneath a read-only view (an owner that flips ``writeable`` on, edits and flips
    it off again; a read-only view over shared memory or a writable mmap): not a proof of equality -- the full hash is --
    but an edit would have to miss every sampled word to go unnoticed."""
    b = c.reshape(-1).view(np.uint8)
    nw = b.size // 8
    if nw == 0:
        return b.tobytes()
    w = b[: nw * 8].view(np.uint64)
    step = max(nw // 4096, 1)
    return w[::step].tobytes() + b[nw * 8:].tobytes() + w[-1:].tobytes()


def content_key(a):
    """(shape, dtype, 64-bit hash of EVERY byte) of a host array, or None: the plan-cache key component for an input.

    The stateless ducc0-style calls may reuse a cached plan only for byte-identical inputs; a sampled fingerprint keyed on
    the address misses in-place edits and reallocations at the same address.  The hash (pfbhip_hash64, multi-threaded)
    costs a pass over the array per call; arrays that cannot change through numpy -- read-only views all the way down -- are
    hashed once and remembered by address for as long as the memo keeps them alive (bounded by entries and by bytes); a hit
    is confirmed against a sampled fingerprint of the current contents (see _sample)."""
    if a is None:
        return None
    a = np.asarray(a)
    ro = a.flags.c_contiguous and _immutable(a)
    if ro:
        mk = (a.ctypes.data, a.shape, a.dtype.str)
        with _ro_lock:
            hit = _ro_memo.get(mk)
            if hit is not None:
                if hit[2] == _sample(a):
                    return hit[1]
                _ro_memo_bytes[0] -= hit[0].nbytes
                del _ro_memo[mk]
    c = np.ascontiguousarray(a)
    key = (c.shape, c.dtype.str, int(lib().pfbhip_hash64(c.ctypes.data_as(ct.c_void_p), ct.c_size_t(c.nbytes))))
    if ro and a.nbytes <= _RO_MEMO_BYTES:
        with _ro_lock:
            if mk not in _ro_memo:
                while _ro_memo and (len(_ro_memo) >= _RO_MEMO_MAX or _ro_memo_bytes[0] + a.nbytes > _RO_MEMO_BYTES):
                    old = _ro_memo.pop(next(iter(_ro_memo)))
                    _ro_memo_bytes[0] -= old[0].nbytes
                _ro_memo[mk] = (a, key, _sample(a))  # holding `a` keeps its buffer from being freed and the address from being reused
                _ro_memo_bytes[0] += a.nbytes
    return key


def any_nonzero(a):
    """``bool(np.any(a))`` that answers from a 1024-point strided sample when the array is obviously non-zero: numpy's
    ``any`` on a float array scans every element (0.1 s for an 8192^2 image), and the reference's zero-input shortcuts
    (operators/hessian.py:47-48) sit in front of every Hessian apply."""
    a = np.asarray(a)
    if a.size > 4096:
        flat = a.reshape(-1) if a.flags.c_contiguous else a.ravel()
        if flat[:: a.size // 1024].any():
            return True
    return bool(a.any())
