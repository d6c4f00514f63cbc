"""ctypes loader for libpfb_oracle.so (built from oracle/pfb_oracle.c by oracle/Makefile)."""

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpfb_oracle.so")

i64 = ct.c_int64
f64 = ct.c_double
cint = ct.c_int
P = ct.c_void_p


def build(force=False):
    src = os.path.join(_HERE, "pfb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ct.CDLL(_SO)
        _lib.pfbo_num_threads.restype = cint
    return _lib


def ptr(a):
    """Pointer to a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags.c_contiguous, "oracle expects C-contiguous arrays"
    return a.ctypes.data_as(P)


def c128_as_f64(a):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    return a, a.view(np.float64)
