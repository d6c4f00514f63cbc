#!/usr/bin/env python3
"""dev helper (GPU box): many different small problems through the STATELESS ducc0-style calls (plan cache churn: creation,
reuse, eviction, destruction while earlier results are still alive), every result checked against the DFT."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dft
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.wgridder import dirty2vis, vis2dirty
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
keep, bad = [], 0
cases = [synth.make_case(int(rng.integers(50, 600)), int(rng.integers(1, 4)), 48, zscale=0.05, seed=int(s)) for s in rng.integers(0, 9999, 12)]
for it in range(90):
    c = cases[int(rng.integers(0, len(cases)))]
    nx, ny = int(rng.choice([40, 48, 51, 64])), int(rng.choice([36, 48, 57]))
    cell = c["cell"] * 8 * 48.0 / max(nx, ny)
    kw = dict(uvw=c["uvw"], freq=c["freq"], pixsize_x=cell, pixsize_y=cell, center_x=0.0, center_y=0.0, epsilon=1e-6, flip_u=False,
              flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False, nthreads=1, sigma_min=1.1, sigma_max=3.0)
    if rng.random() < 0.5:
        d = vis2dirty(vis=c["vis"], wgt=c["wgt"], mask=c["mask"], npix_x=nx, npix_y=ny, double_precision_accumulation=True, **kw)
        ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], nx, ny, cell, cell, 0.0, 0.0, False, True, False, True, False)
        e = rel(d, ref); keep.append((d, ref))
    else:
        x = rng.standard_normal((nx, ny))
        v = dirty2vis(dirty=x, mask=c["mask"], **kw)
        ref = dft.dft_dirty2vis(c["uvw"], c["freq"], x, cell, cell, 0.0, 0.0, False, True, False, True, False); ref[c["mask"] == 0] = 0
        e = rel(v, ref); keep.append((v, ref))
    if not e < 1e-6:
        bad += 1; print("BAD", it, nx, ny, e)
    keep = keep[-25:]
for a, b in keep:  # earlier results still intact (pinned pool / handle destruction must not touch them)
    if not rel(a, b) < 1e-6:
        bad += 1; print("BAD kept result")
print("bad:", bad)
