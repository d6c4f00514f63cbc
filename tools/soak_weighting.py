#!/usr/bin/env python3
"""dev helper (GPU box): randomised sweep of the imaging-weight pipeline (uv-cell index bit-exact; counts, filter, box sum,
Briggs divide against the C oracle; the one-call device pipeline against the four separate calls)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import weighting as ow
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.utils.weighting import (_compute_counts, box_sum_counts, counts_to_weights, filter_extreme_counts, imaging_weights,
                                             uvcell_index)
rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    nrow, nchan, ncorr = int(rng.integers(10, 4000)), int(rng.integers(1, 6)), int(rng.integers(1, 3))
    c = synth.make_case(nrow, nchan, 64, zscale=0.05, seed=int(rng.integers(0, 9999)))
    nx, ny = 2 * int(rng.integers(20, 300)), 2 * int(rng.integers(20, 300))
    cell = c["cell"] * float(rng.choice([0.3, 1.0, 3.0]))
    us, vs = float(rng.choice([-1.0, 1.0])), float(rng.choice([-1.0, 1.0]))
    robust = float(rng.choice([-3.0, -1.0, 0.0, 0.7, 2.0]))
    level = float(rng.choice([0.0, 5.0, 10.0])); nsup = int(rng.choice([0, 1, 2]))
    wgt = rng.random((ncorr, nrow, nchan)) + 0.1
    idx = uvcell_index(c["uvw"], c["freq"], c["mask"], nx, ny, cell, cell, us, vs)
    e0 = np.array_equal(idx, ow.uvcell_index(c["uvw"], c["freq"], c["mask"], nx, ny, cell, cell, us, vs))
    cg = _compute_counts(c["uvw"], c["freq"], c["mask"], wgt, nx, ny, cell, cell, np.float64, usign=us, vsign=vs)
    co = ow.compute_counts(c["uvw"], c["freq"], c["mask"], wgt, nx, ny, cell, cell, usign=us, vsign=vs)
    e1 = rel(cg, co)
    co2 = ow.box_sum_counts(ow.filter_extreme_counts(co.copy(), level) if level else co.copy(), nsup)
    cg2 = box_sum_counts(filter_extreme_counts(cg.copy(), level), nsup)
    e2 = rel(cg2, co2)
    wo = ow.counts_to_weights(co2.copy(), c["uvw"], c["freq"], wgt.copy(), c["mask"], nx, ny, cell, cell, robust, usign=us, vsign=vs)
    wg = counts_to_weights(cg2.copy(), c["uvw"], c["freq"], wgt.copy(), c["mask"], nx, ny, cell, cell, robust, usign=us, vsign=vs)
    e3 = rel(wg, wo)
    w1 = imaging_weights(c["uvw"], c["freq"], c["mask"], wgt.copy(), nx, ny, cell, cell, robust, filter_level=level, npix_super=nsup,
                         usign=us, vsign=vs)
    e4 = rel(w1, wo)
    ok = e0 and e1 < 1e-12 and e2 < 1e-12 and e3 < 1e-12 and e4 < 1e-12
    bad += not ok
    print(k, "OK " if ok else "BAD", (nrow, nchan, ncorr, nx, ny, us, vs, robust, level, nsup), e0, ["%.1e" % q for q in (e1, e2, e3, e4)], flush=True)
print("bad:", bad)
