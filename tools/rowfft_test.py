#!/usr/bin/env python3
"""Check and time the hand-written row FFT against numpy / rocFFT-sized expectations (dev tool)."""
import ctypes as ct, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd._lib import lib, check, ptr, i64, cint

def run(n, nrows, inverse, reps=1):
    rng = np.random.default_rng(n + nrows)
    a = rng.standard_normal((nrows, n)) + 1j * rng.standard_normal((nrows, n))
    b = a.copy()
    ms = ct.c_double(0)
    check(lib().pfbhip_debug_rowfft(ptr(b), i64(n), i64(nrows), cint(int(inverse)), cint(reps), ct.byref(ms)))
    ref = np.fft.ifft(a, axis=1) * n if inverse else np.fft.fft(a, axis=1)
    err = np.linalg.norm(b - ref) / np.linalg.norm(ref)
    return err, ms.value

for n in (1024, 1152, 1280, 1536, 1792, 1920, 2048, 2304, 3072, 3584, 3840, 4096, 4608, 5120, 6144, 7168, 7680, 8192, 9216, 10240,
          12288, 14336, 15360, 16384, 20480, 24576, 32768):
    for inv in (0, 1):
        err, _ = run(n, 7, inv)
        print(f"n={n:6d} inverse={inv} rel err {err:.2e}")
for n, rows in ((10240, 8192), (8192, 8192), (9216, 8192), (12288, 8192), (14336, 8192), (15360, 8192), (16384, 8192), (20480, 4096),
                (24576, 4096), (32768, 2048)):
    err, ms = run(n, rows, 1, reps=5)
    print(f"n={n} rows={rows}: {ms:.3f} ms  ({2*n*rows*16/ms/1e6:.0f} GB/s)  err {err:.1e}")
print("compute-bound probe (one row per CU, data cache-resident):")
for n, rows in ((10240, 256), (10240, 512), (8192, 256), (12288, 256)):
    err, ms = run(n, rows, 1, reps=20)
    print(f"n={n} rows={rows}: {ms*1e3:.1f} us total, {ms*1e3/ (rows/256):.1f} us per row per CU")
