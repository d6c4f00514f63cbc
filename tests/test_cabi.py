"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/pfbhip.h declares, its structs match the ctypes mirror, and it fails loudly without a GPU."""

import ctypes as ct
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pfbhip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pfbhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pfb_imaging_amd import _lib

    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 45
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == declared


def test_struct_layout_matches_header(tmp_path):
    from pfb_imaging_amd import _lib

    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "pfbhip.h"\n'
        "int main(void){\n"
        'printf("%zu %zu %zu %zu %zu\\n", sizeof(pfbhip_gridder_params), offsetof(pfbhip_gridder_params, epsilon),'
        " offsetof(pfbhip_gridder_params, flip_u), offsetof(pfbhip_gridder_params, force_wmode),"
        " offsetof(pfbhip_gridder_params, force_sigma));\n"
        'printf("%zu %zu %zu %zu %zu\\n", sizeof(pfbhip_gridder_info), offsetof(pfbhip_gridder_info, W),'
        " offsetof(pfbhip_gridder_info, beta), offsetof(pfbhip_gridder_info, wmode),"
        " offsetof(pfbhip_gridder_info, device_bytes));\n"
        'printf("%zu %zu\\n", sizeof(pfbhip_cg_info), offsetof(pfbhip_cg_info, eps));\n'
        'printf("%zu %zu %zu %d\\n", sizeof(pfbhip_pd_info), offsetof(pfbhip_pd_info, stage_ms),'
        " offsetof(pfbhip_pd_info, stage_calls), PFBHIP_PD_NSTAGES);\n"
        "return 0;}\n"
    )
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    P, I, C, D = _lib.GridderParams, _lib.GridderInfo, _lib.CGInfo, _lib.PDInfo
    expect = [ct.sizeof(P), P.epsilon.offset, P.flip_u.offset, P.force_wmode.offset, P.force_sigma.offset,
              ct.sizeof(I), I.W.offset, I.beta.offset, I.wmode.offset, I.device_bytes.offset, ct.sizeof(C), C.eps.offset,
              ct.sizeof(D), D.stage_ms.offset, D.stage_calls.offset, _lib.PD_NSTAGES]
    assert len(_lib.PD_STAGE_NAMES) == _lib.PD_NSTAGES
    assert [int(v) for v in out] == expect


def test_no_gpu_paths_fail_loudly_or_work_on_host():
    from pfb_imaging_amd import _lib, fft, misc

    L = _lib.lib()
    assert fft.good_size(11468) == 11520
    assert fft.good_size(127, True) == 128
    misc.resize_thread_pool(7)
    assert misc.thread_pool_size() == 7
    assert misc.empty_noncritical((3, 4), "c16").shape == (3, 4)
    # invalid arguments are reported through the status / last-error channel (no exceptions cross the ABI)
    st = L.pfbhip_gridder_get_info(None, None)
    assert st == 1 and "NULL" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(st)
    if _lib.device_count() == 0:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            _lib.require_gpu()
        from pfb_imaging_amd.wgridder import vis2dirty

        with pytest.raises(RuntimeError):
            vis2dirty(uvw=np.zeros((3, 3)), freq=np.ones(1), vis=np.zeros((3, 1), complex), npix_x=16, npix_y=16,
                      pixsize_x=1e-5, pixsize_y=1e-5, epsilon=1e-5, do_wgridding=True)


def test_product_never_imports_oracle():
    """The product path must not reach into the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "pfb-imaging_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "libpfb_oracle" in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    code = ("import sys; sys.path.insert(0, %r); import pfb_imaging_amd.operators.hessian, pfb_imaging_amd.operators.gridder, "
            "pfb_imaging_amd.operators.band_worker, pfb_imaging_amd.operators.psf, pfb_imaging_amd.utils.weighting; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])


def test_host_side_helpers():
    from pfb_imaging_amd.operators import LinearOperator, Preconditioner, require_protocol
    from pfb_imaging_amd.operators.gridder import psf_visibilities, wgridder_conventions
    from pfb_imaging_amd.operators.hessian import hessian_slice, taperf

    assert wgridder_conventions(0.1, -0.2) == (False, True, False, -0.1, 0.2)
    rng = np.random.default_rng(0)
    uvw = rng.standard_normal((10, 3)) * 100
    freq = np.array([1e9, 1.1e9])
    assert np.array_equal(psf_visibilities(uvw, freq, 0.0, 0.0), np.ones((10, 2)))
    x0, y0 = -0.1, 0.17
    n = np.sqrt(1 - x0**2 - y0**2)
    ref = np.exp(2j * np.pi * freq[None] / 299792458.0 * (uvw[:, 0:1] * x0 + uvw[:, 1:2] * y0 - uvw[:, 2:] * (n - 1)))
    np.testing.assert_allclose(psf_visibilities(uvw, freq, x0, y0), ref)
    t = taperf((64, 48), 8)
    assert t.shape == (64, 48) and t[32, 24] == 1.0 and t[0, 0] < 1e-2
    # zero-input shortcut never touches the device (hessian.py:47-48)
    assert not hessian_slice(np.zeros((8, 8))).any()

    class Bad:
        def dot(self, x):
            return x

    with pytest.raises(TypeError, match="hdot"):
        require_protocol(Bad(), LinearOperator, "hess")
    with pytest.raises(TypeError, match="Preconditioner"):
        require_protocol(Bad(), Preconditioner, "precond")


def test_content_hash_sees_every_byte():
    """The plan caches of the stateless calls are keyed on a hash of EVERY byte of uvw / freq / mask / weights / psfhat
    (the reference's ducc0 calls are stateless): a single flipped mask element, an in-place weight tweak, a swap of two
    elements and a reallocation with equal content must all be told apart / recognised."""
    from pfb_imaging_amd import _lib

    rng = np.random.default_rng(0)
    a = rng.standard_normal(1_000_003)
    k = _lib.content_key(a)
    assert k == _lib.content_key(a.copy())            # content, not address
    a[777_777] = np.nextafter(a[777_777], 10.0)       # one ulp, far from any "sample"
    k2 = _lib.content_key(a)
    assert k2 != k
    b = a.copy()
    b[[5, 6]] = b[[6, 5]]
    assert _lib.content_key(b) != k2                  # position-dependent
    m = np.ones((1001, 7), dtype=np.uint8)            # size not a multiple of 8: the tail bytes count too
    km = _lib.content_key(m)
    m[1000, 6] = 0
    assert _lib.content_key(m) != km
    assert _lib.content_key(m) != _lib.content_key(m.reshape(7, 1001))
    assert _lib.content_key(None) is None
    # read-only inputs are hashed once and remembered by address; anything writeable (also through a base) is not
    r = rng.standard_normal(4096)
    r.flags.writeable = False
    assert _lib._immutable(r) and not _lib._immutable(a)
    assert _lib.content_key(r) == _lib.content_key(r) and (r.ctypes.data, r.shape, r.dtype.str) in _lib._ro_memo
    base = rng.standard_normal(64)
    view = base[:32]
    view.flags.writeable = False
    assert not _lib._immutable(view)                  # still writeable through its base
    kv = _lib.content_key(view)
    base[3] += 1.0
    assert _lib.content_key(view) != kv
    # an owner that flips `writeable` on, edits in place and flips it off again: the memo's sampled fingerprint notices
    # (every word of a small array is sampled) and the entry is re-hashed
    own = rng.standard_normal(2048)
    own.flags.writeable = False
    ko = _lib.content_key(own)
    own.flags.writeable = True
    own[1234] += 1.0
    own.flags.writeable = False
    assert _lib.content_key(own) != ko
    # the memo is bounded by bytes as well as by entries
    assert _lib._ro_memo_bytes[0] == sum(v[0].nbytes for v in _lib._ro_memo.values()) <= _lib._RO_MEMO_BYTES


def test_pinned_results_are_capped(monkeypatch):
    """Live page-locked result bytes are bounded (PFBHIP_PINNED_MAX_MB): past the cap result arrays are ordinary numpy
    allocations.  (No GPU here: the allocator entry is replaced by one that records what it is asked for.)"""
    import ctypes as ct

    from pfb_imaging_amd import _lib

    bufs = []

    class FakeLib:
        def pfbhip_host_alloc(self, pp, nbytes):
            b = ct.create_string_buffer(nbytes.value)
            bufs.append(b)
            ct.cast(pp, ct.POINTER(ct.c_void_p))[0] = ct.addressof(b)
            return 0

        def pfbhip_host_free(self, p):
            return 0

    monkeypatch.setattr(_lib, "lib", lambda: FakeLib())
    monkeypatch.setenv("PFBHIP_PINNED_RESULTS", "1")
    monkeypatch.setenv("PFBHIP_PINNED_MAX_MB", "5")
    monkeypatch.setattr(_lib, "_pinned_live", [0])
    monkeypatch.setattr(_lib, "_pinned_pool", [])
    monkeypatch.setattr(_lib, "_pinned_pooled", [0])
    a = _lib.result_empty((2 << 20,), np.uint8)    # 2 MiB: pinned
    b = _lib.result_empty((2 << 20,), np.uint8)    # 4 MiB live
    assert len(bufs) == 2 and _lib._pinned_live[0] == 4 << 20
    c = _lib.result_empty((2 << 20,), np.uint8)    # would be 6 MiB > 5: pageable
    assert len(bufs) == 2 and c.shape == a.shape
    del b                                           # back to the pool (still live), reused by the next request of that size
    import gc

    gc.collect()
    d = _lib.result_empty((2 << 20,), np.uint8)
    assert len(bufs) == 2 and _lib._pinned_live[0] == 4 << 20 and d.ctypes.data in (ct.addressof(x) for x in bufs)
    del a, c, d
    gc.collect()
