#!/usr/bin/env python3
"""dev helper (GPU box): randomised sweep of ON-AXIS geometries -- where the plan may take the one-plane w-scheme (wmode 2) --
against the oracle restatement run with the plan's own parameters (1e-9), the direct DFT (epsilon; dirty2vis on a subset of rows,
vis2dirty on a subset of pixels) and the fused Hessian's two halves.   python tools/soak_wd.py [seed] [ncases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PFBHIP_SCATTER", "rec")
from oracle import dft, wgridder as owg
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.wgridder import Gridder

rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
bad, modes = 0, {0: 0, 1: 0, 2: 0, -1: 0}
for k in range(ncases):
    big = rng.random() < 0.4
    lo, hi = (820, 1500) if big else (24, 400)
    nx, ny = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    eps = float(rng.choice([1e-3, 1e-4, 1e-6, 1e-7, 1e-9, 1e-10]))
    # field of view chosen so that omega = 2 pi (w range / 2) max|n - 1| lands between 1e-4 and 1 (K = 2..4 around the middle)
    zscale = float(rng.choice([0.001, 0.003, 0.02, 0.1]))
    fu, fv, fw = (bool(v) for v in rng.integers(0, 2, 3))
    do_w, divn = bool(rng.random() > 0.1), bool(rng.integers(0, 2))
    nrow, nchan = int(rng.integers(50, 4000)), int(rng.integers(1, 4))
    c = synth.make_case(nrow, nchan, 64, zscale=zscale, seed=int(rng.integers(0, 9999)))
    wmax = float(np.abs(c["uvw"][:, 2]).max() * c["freq"].max() / 299792458.0)
    omega = 10.0 ** rng.uniform(-4.0, 0.3)
    # omega ~ 2 pi (wmax / 2) (fov^2 / 8) => fov
    fov = min(np.sqrt(8.0 * omega / (np.pi * max(wmax, 1e-9))), 0.9)
    aniso = float(rng.choice([1.0, 1.07, 0.8]))
    cellx = fov / max(nx, ny * aniso)
    celly = cellx * aniso
    x = rng.standard_normal((nx, ny))
    kw = dict(npix_x=nx, npix_y=ny, pixsize_x=cellx, pixsize_y=celly, center_x=0.0, center_y=0.0, epsilon=eps, flip_u=fu, flip_v=fv,
              flip_w=fw, do_wgridding=do_w, divide_by_n=divn)
    try:
        g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
    except RuntimeError as e:
        if "no ES kernel reaches" in str(e):
            modes[-1] += 1
            continue
        raise
    modes[g.info["wmode"]] += 1
    o = owg.Plan(c["uvw"], c["freq"], c["mask"], nx, ny, cellx, celly, 0.0, 0.0, eps, fu, fv, fw, do_w, divn, params=g.oracle_params())
    d, v = g.vis2dirty(c["vis"], c["wgt"]), g.dirty2vis(x)
    g.set_weights(c["wgt"]); h1 = g.hessian(x); h2 = g.hessian(x)
    e = [rel(d, o.vis2dirty(c["vis"], c["wgt"])), rel(v, o.dirty2vis(x)), rel(h1, g.vis2dirty(v, c["wgt"])), rel(h2, h1)]
    rows = slice(0, 40)
    refv = dft.dft_dirty2vis(c["uvw"][rows], c["freq"], x, cellx, celly, 0.0, 0.0, fu, fv, fw, do_w, divn)
    refv[c["mask"][rows] == 0] = 0
    e.append(rel(v[rows], refv) / eps)
    ix, iy = rng.integers(0, nx, 60), rng.integers(0, ny, 60)
    refd = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], nx, ny, cellx, celly, 0.0, 0.0, fu, fv, fw, do_w, divn,
                             pixels=(ix, iy))
    e.append(np.linalg.norm(d[ix, iy] - refd) / max(np.linalg.norm(d) * np.sqrt(60.0 / d.size), 1e-300) / eps)
    info = g.info
    g.close()
    ok = e[0] < 1e-9 and e[1] < 1e-9 and e[2] < 1e-9 and e[3] < 1e-9 and e[4] < 1.0 and e[5] < 1.0
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} case {k}: {nx}x{ny} eps {eps:g} omega~{omega:.2g} do_w {int(do_w)} divn {int(divn)} flips {int(fu)}{int(fv)}{int(fw)} "
          f"nvis {nrow}x{nchan} -> wmode {info['wmode']} K {info['nderiv']} planes {info['nplanes']} W {info['W']} grid {info['nu']}x{info['nv']}  "
          f"vs restatement {e[0]:.1e} {e[1]:.1e}  halves {e[2]:.1e} repeat {e[3]:.1e}  vs DFT/eps {e[4]:.2f} {e[5]:.2f}", flush=True)
print(f"seed {seed}: {ncases} cases, {bad} bad; schemes chosen {modes}")
sys.exit(1 if bad else 0)
