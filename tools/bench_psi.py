#!/usr/bin/env python3
"""Secondary measurement: the SARA dictionary Psi (BASELINE config C4: 4096^2, bases self,db1,db2,db3, 3 levels).

    python tools/bench_psi.py [--nx 4096] [--steps 20] [--no-cpu]

One step = dot (image -> coefficients) + hdot (coefficients -> image) of one band on device-resident data
(pfbhip_psi_dot_dev / pfbhip_psi_hdot_dev).  Compulsory bytes: per wavelet basis and level l the two FIR passes
read and write ~4 * I / 4^l (I = nx ny 8 B), the identity basis 2 I, plus clearing the coefficient cube in dot.
The oracle (numpy restatement of the reference's numba kernels) is timed beside it as the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--nlevel", type=int, default=3)
    ap.add_argument("--bases", default="self,db1,db2,db3")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    from pfb_imaging_amd import _lib
    from pfb_imaging_amd._lib import DeviceArray, check, lib
    from pfb_imaging_amd.operators.psi import PsiBand

    _lib.require_gpu()
    nx = ny = args.nx
    bases = tuple(args.bases.split(","))
    band = PsiBand(nx, ny, bases, args.nlevel)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((nx, ny))
    xd = DeviceArray.from_host(x)
    ad = DeviceArray((band.nbasis, band.nxmax, band.nymax), np.float64)
    od = DeviceArray((nx, ny), np.float64)

    def step():
        check(lib().pfbhip_psi_dot_dev(band._h, xd.ptr, ad.ptr))
        check(lib().pfbhip_psi_hdot_dev(band._h, ad.ptr, od.ptr))

    for _ in range(3):
        step()
    check(lib().pfbhip_synchronize())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    check(lib().pfbhip_synchronize())
    t = (time.perf_counter() - t0) / args.steps
    back = od.download()
    err = np.abs(back / len(bases) - x).max()
    I = nx * ny * 8.0
    nw = sum(b != "self" for b in bases)
    per_dir = nw * sum(4.0 * I / 4 ** l for l in range(args.nlevel)) + (len(bases) - nw) * 2.0 * I
    cube = band.nbasis * band.nxmax * band.nymax * 8.0
    alg = 2 * per_dir + cube
    out = {"metric": "Psi dot+hdot per band", "ms_per_step": t * 1e3, "config": {"image": [nx, ny], "bases": bases,
           "nlevel": args.nlevel, "coeff_shape": [band.nbasis, band.nxmax, band.nymax]},
           "roofline": {"bound": "hbm", "achieved": alg / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / t / 8e12,
                        "alg_bytes": alg}, "reconstruction_error": err}
    if not args.no_cpu:
        from oracle import psi as opsi

        o = opsi.Psi(1, nx, ny, bases, args.nlevel)
        a = np.zeros((1, o.nbasis, o.nxmax, o.nymax))
        xo = np.zeros((1, nx, ny))
        t0 = time.perf_counter()
        o.dot(x[None], a)
        o.hdot(a, xo)
        out["cpu_baseline"] = {"ms_per_step": (time.perf_counter() - t0) * 1e3, "kind": "port", "cores": 1}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
