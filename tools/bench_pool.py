#!/usr/bin/env python3
"""dev helper (GPU box): the band pool's exact residual over several bands of one GPU, threaded against one band after the other.
   python tools/bench_pool.py [nband] [npix] [nrow]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd.operators.band_worker import BandWorkerPool  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402

nband = int(sys.argv[1]) if len(sys.argv) > 1 else 4
npix = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nrow = int(sys.argv[3]) if len(sys.argv) > 3 else 250000
cases = [synth.make_case(nrow, 8, npix, zscale=1e-3, seed=b, with_vis=False) for b in range(nband)]
parts = [[{"UVW": c["uvw"], "FREQ": c["freq"], "WEIGHT": c["wgt"][None], "MASK": c["mask"], "BEAM": np.ones((1, npix, npix)),
           "attrs": {"l0": 0.0, "m0": 0.0}}] for c in cases]
dirty = np.zeros((nband, 1, npix, npix))
model = np.random.default_rng(0).standard_normal((nband, 1, npix, npix))
cell = cases[0]["cell"]
for thr in ("1", "4"):
    os.environ["PFBHIP_BAND_THREADS"] = thr
    pool = BandWorkerPool(nband)
    pool.set_bands(dirty, parts)
    pool.residual(model, cell)  # plans
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = pool.residual(model, cell)
        t.append(time.perf_counter() - t0)
    print(f"PFBHIP_BAND_THREADS={thr}: residual over {nband} bands of {npix}^2, {nrow * 8} vis each: {min(t) * 1e3:.1f} ms "
          f"({min(t) / nband * 1e3:.1f} per band)", flush=True)
    pool.close()
    del pool
