#!/usr/bin/env python3
"""dev helper (GPU box): uv coverages with holes (a core and an outer ring: three or more column runs per tile row, the
whole-row fallback of the first-axis pruning) and coverages that wrap around the grid edge, on the hand-written FFT paths."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PFBHIP_SCATTER", "rec")
from oracle import dft
from pfb_imaging_amd.wgridder import Gridder
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
rng = np.random.default_rng(4)
bad = 0
for case in range(6):
    nx, ny = int(rng.integers(900, 1400)), int(rng.integers(900, 1400))
    cell = 1e-5
    umax = 0.5 / cell * (0.95 if case % 2 else 0.6)  # odd cases reach the edge of the grid (footprints wrap)
    nrow = 6000
    ang = rng.random(nrow) * 2 * np.pi
    rad = np.where(rng.random(nrow) < 0.5, rng.random(nrow) * 0.08, 0.8 + 0.2 * rng.random(nrow)) * umax
    freq = np.array([1.0e9])
    lam = 299792458.0 / freq[0]
    uvw = np.stack([rad * np.cos(ang), rad * np.sin(ang), rng.standard_normal(nrow) * 30.0], axis=1) * lam
    vis = rng.standard_normal((nrow, 1)) + 1j * rng.standard_normal((nrow, 1)); wgt = rng.random((nrow, 1)) + 0.5
    mask = np.ones((nrow, 1), np.uint8)
    g = Gridder(uvw, freq, mask, npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell, center_x=0.0, center_y=0.0, epsilon=1e-7, flip_u=False,
                flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False)
    x = rng.standard_normal((nx, ny))
    v = g.dirty2vis(x); d = g.vis2dirty(vis, wgt)
    g.set_weights(wgt); h1 = g.hessian(x); h2 = g.hessian(0.5 * x)
    rows = slice(0, 80)
    refv = dft.dft_dirty2vis(uvw[rows], freq, x, cell, cell, 0.0, 0.0, False, True, False, True, False)
    ix = rng.integers(0, nx, 40); iy = rng.integers(0, ny, 40)
    refd = dft.dft_vis2dirty(uvw, freq, vis, wgt, mask, nx, ny, cell, cell, 0.0, 0.0, False, True, False, True, False, pixels=(ix, iy))
    e = [rel(v[rows], refv) / 1e-7, np.linalg.norm(d[ix, iy] - refd) / np.linalg.norm(refd) / 1e-7, rel(h1, g.vis2dirty(v, wgt)), rel(h2, 0.5 * h1)]
    ok = e[0] < 1 and e[1] < 1 and e[2] < 1e-9 and e[3] < 1e-9
    bad += not ok
    print(case, "OK " if ok else "BAD", (nx, ny), {q: g.info[q] for q in ("nu", "nv", "nplanes", "wmode", "fft_mode", "occ_rows", "used_cells")}, ["%.1e" % q for q in e], flush=True)
print("bad:", bad)
