#!/usr/bin/env python3
"""Secondary measurement: device-resident primal-dual iterations (BASELINE config C4 shape on ONE GPU).

    python tools/bench_pd.py [--nx 4096] [--nband 4] [--iters 20]

nband bands of nx^2 pixels, PSF 2x oversized, bases self,db1,db2,db3, 3 levels, positivity mode 1.  One iteration =
per band: Psi^H, Psi, one PSF-approximate Hessian apply; plus the l21 dual update over all bands and the vector steps
(pfbhip_primal_dual).  Prints one JSON line with ms per iteration.
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
    ap.add_argument("--nband", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from pfb_imaging_amd import _lib, prox
    from pfb_imaging_amd.operators.hessian import HessPSF
    from pfb_imaging_amd.operators.psi import PsiNocopyt
    from pfb_imaging_amd.opt import L21, PrimalDual, PsfGrad

    _lib.require_gpu()
    nx = ny = args.nx
    nband, bases = args.nband, ("self", "db1", "db2", "db3")
    rng = np.random.default_rng(0)
    abspsf = 1.0 + 0.1 * np.abs(rng.standard_normal((nband, 2 * nx, ny + 1)))
    hess = HessPSF(nx, ny, abspsf, beam=None, eta=0.01)
    psi = PsiNocopyt(nband, nx, ny, bases, 3, 1)
    reg = L21(psi, bases, nu=np.sqrt(len(bases)))
    model = np.abs(rng.standard_normal((nband, nx, ny))) * (rng.random((nband, nx, ny)) > 0.99)
    xtilde = model + 0.1 * rng.standard_normal(model.shape)
    timings = {}
    for maxit in (2, 2 + args.iters):
        pd = PrimalDual(tol=0.0, maxit=maxit, verbosity=0, gamma=1.0, primal_prox=prox.positivity)
        pd.setup(reg, float(abspsf.max() + 0.01))
        pd.set_grad(PsfGrad(hess, xtilde, 1.0))
        t0 = time.perf_counter()
        pd.solve(model.copy(), 1e-3)
        timings[maxit] = time.perf_counter() - t0
    per_iter = (timings[2 + args.iters] - timings[2]) / args.iters
    print(json.dumps({"metric": "primal-dual iterations (device-resident)", "ms_per_iteration": per_iter * 1e3,
                      "iterations_per_s": 1.0 / per_iter,
                      "config": {"nband": nband, "image": [nx, ny], "psf": [2 * nx, 2 * ny], "bases": bases, "nlevel": 3,
                                 "positivity": 1},
                      "host_overhead_s_2_iterations": timings[2]}))


if __name__ == "__main__":
    main()
