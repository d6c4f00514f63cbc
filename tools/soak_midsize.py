#!/usr/bin/env python3
"""dev helper (GPU box): randomised mid-size sweep -- image sizes whose grids take the hand-written row FFT (fused second axis,
transposing first axis, column runs, rectangle clear) -- against the oracle restatement run with the plan's own parameters and,
for dirty2vis, the direct DFT on a subset of rows.   python tools/soak_midsize.py [seed] [ncases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PFBHIP_SCATTER", "rec")
from oracle import dft, wgridder as owg
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.wgridder import Gridder

rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
bad = 0
for k in range(ncases):
    lo, hi = (int(os.environ.get("SOAK_LO", "820")), int(os.environ.get("SOAK_HI", "1500")))
    nx, ny = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    eps = float(rng.choice([1e-4, 1e-6, 1e-7, 1e-9]))
    widen = float(rng.choice([4.0, 8.0, 30.0, 100.0])); zscale = float(rng.choice([0.002, 0.05, 0.5]))
    fu, fv, fw = (bool(v) for v in rng.integers(0, 2, 3))
    cx, cy = float(rng.choice([0.0, 0.004, -0.02])), float(rng.choice([0.0, -0.003, 0.03]))
    do_w, divn = bool(rng.random() > 0.15), bool(rng.integers(0, 2))
    c = synth.make_case(int(rng.integers(200, 3000)), int(rng.integers(1, 4)), 64, zscale=zscale, seed=int(rng.integers(0, 9999)))
    cell = c["cell"] * widen * 64.0 / max(nx, ny)
    cell = min(cell, 0.6 / max(nx, ny))
    x = rng.standard_normal((nx, ny))
    kw = dict(npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell * 1.07, center_x=cx, center_y=cy, epsilon=eps, flip_u=fu, flip_v=fv,
              flip_w=fw, do_wgridding=do_w, divide_by_n=divn)
    g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
    o = owg.Plan(c["uvw"], c["freq"], c["mask"], nx, ny, cell, cell * 1.07, cx, cy, eps, fu, fv, fw, do_w, divn, params=g.oracle_params())
    d, v = g.vis2dirty(c["vis"], c["wgt"]), g.dirty2vis(x)
    g.set_weights(c["wgt"]); h1 = g.hessian(x); h2 = g.hessian(x)
    e = [rel(d, o.vis2dirty(c["vis"], c["wgt"])), rel(v, o.dirty2vis(x)), rel(h1, g.vis2dirty(v, c["wgt"])), rel(h2, h1)]
    rows = slice(0, 60)
    refv = dft.dft_dirty2vis(c["uvw"][rows], c["freq"], x, cell, cell * 1.07, cx, cy, fu, fv, fw, do_w, divn)
    refv[c["mask"][rows] == 0] = 0
    e.append(rel(v[rows], refv) / eps)
    # vs the restatement: FFT rounding x the image-side correction -- the plan admits up to 0.2 epsilon of it (choose_kernel);
    # e[3]: run-to-run, LDS atomics reorder sums (the same rounding, an order of magnitude lower)
    lim = max(3e-8, 0.2 * eps)
    ok = e[0] < lim and e[1] < lim and e[2] < lim and e[3] < max(1e-9, 0.02 * eps) and e[4] < 1.0
    bad += not ok
    print(k, "OK " if ok else "BAD", (nx, ny), {q: g.info[q] for q in ("nu", "nv", "nplanes", "W", "wmode", "fft_mode", "scatter_mode", "used_cells")},
          dict(eps=eps, flips=(fu, fv, fw), center=(cx, cy), do_w=do_w, divn=divn), ["%.1e" % q for q in e], flush=True)
    g.close()
print("bad cases:", bad)
