#!/usr/bin/env python3
"""Secondary measurement: PSF-approximate Hessian (HessPSF.dot / HessianTree.dot path) on the GPU.

    python tools/bench_psfconv.py [--nx 8192] [--oversize 2.0] [--steps 10]

One apply = pad*beam -> r2c -> *abspsf -> c2r -> crop*beam + eta*x on device-resident data
(pfbhip_psfconv_apply_dev).  Algorithmic bytes: B_psf = 6 I + 3 Xr + 6.5 Xc.
Prints one JSON line; the numpy (pocketfft) restatement is timed beside it as the CPU baseline.
"""

import argparse
import ctypes as ct
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=8192)
    ap.add_argument("--oversize", type=float, default=2.0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    from pfb_imaging_amd import _lib
    from pfb_imaging_amd._lib import DeviceArray, check, cint, f64, i64, lib
    from pfb_imaging_amd.fft import good_size
    from pfb_imaging_amd.psfconv import PsfConv

    _lib.require_gpu()
    nx = ny = args.nx
    nxp = good_size(int(args.oversize * nx))
    nxp += nxp % 2
    nyp = nxp
    rng = np.random.default_rng(0)
    x = rng.standard_normal((nx, ny))
    beam = 0.5 + rng.random((nx, ny))
    # |FFT| of a real array: Hermitian-consistent on the ky = 0 and Nyquist columns, as a real PSFHAT is (a random
    # half-spectrum is not, and c2r implementations differ in how they treat the inconsistent part)
    abspsf = np.abs(np.fft.rfft2(rng.standard_normal((nxp, nyp)) * np.exp(-np.linspace(-6, 6, nxp)[:, None] ** 2)))
    plan = PsfConv(nx, ny, nxp, nyp)
    plan.set_psfhat(0, abspsf)
    plan.set_beam(0, beam)
    xd, od = DeviceArray.from_host(x), DeviceArray((nx, ny))

    def apply():
        check(lib().pfbhip_psfconv_apply_dev(plan._h, xd.ptr, i64(0), i64(0), cint(0), f64(0.0), f64(1.0), f64(0.1),
                                             cint(0), od.ptr))

    apply()
    apply()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        apply()
    check(lib().pfbhip_synchronize())
    t = (time.perf_counter() - t0) / args.steps
    I, Xr, Xc = nx * ny * 8, nxp * nyp * 8, nxp * (nyp // 2 + 1) * 16
    b_psf = 6 * I + 3 * Xr + 6.5 * Xc
    out = {"metric": "PSF-approximate Hessian applies/s (HessPSF.dot, one band)", "value": 1.0 / t, "unit": "applies/s",
           "ms_per_apply": t * 1e3, "config": {"image": [nx, ny], "psf": [nxp, nyp]},
           "roofline": {"bound": "hbm", "achieved": b_psf / t / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": b_psf / t / 1e9 / 8000.0, "alg_bytes": b_psf,
                        "frac_note": "SURVEY.md 8(d) UNPRUNED accounting (full padded r2c / c2r); the pipeline that runs stores "
                                     "nothing outside the nx x (nyp/2+1) corner",
                        # x, beam read + T2 written | T2, psfhat read + T1 written | T1, beam, x read + out written
                        "actual_bytes": 6 * I + 4 * nx * (nyp // 2 + 1) * 16 + nxp * (nyp // 2 + 1) * 8,
                        "actual_frac": (6 * I + 4 * nx * (nyp // 2 + 1) * 16 + nxp * (nyp // 2 + 1) * 8) / t / 1e9 / 8000.0}}
    if not args.no_cpu:
        from oracle import fftconv

        t0 = time.perf_counter()
        ref = fftconv.hessian_psf_slice(x, abspsf, nyp, beam=beam, eta=0.1)
        tc = time.perf_counter() - t0
        got = od.download()
        out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "applies/s", "cores": 1, "kind": "port",
                               "sample": "one apply, numpy pocketfft single thread"}
        out["rel_err_vs_oracle"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
