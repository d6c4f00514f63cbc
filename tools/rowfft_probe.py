#!/usr/bin/env python3
"""Run the hand-written row FFT once per size (for rocprofv3 counter passes)."""
import ctypes as ct, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd._lib import lib, check, ptr, i64, cint
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
a = np.ones((rows, n), dtype=np.complex128)
ms = ct.c_double(0)
check(lib().pfbhip_debug_rowfft(ptr(a), i64(n), i64(rows), cint(1), cint(3), ct.byref(ms)))
print(n, rows, ms.value)
