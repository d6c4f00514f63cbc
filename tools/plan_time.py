#!/usr/bin/env python3
"""dev helper (GPU box): wall-clock of the plan-creation phases (verbosity = 1) for a config: C2 (default) or nrow,nchan,npix[,zscale]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.wgridder import Gridder

spec = sys.argv[1] if len(sys.argv) > 1 else "1250000,8,8192,0.001"
parts = spec.split(",")
nrow, nchan, npix = int(parts[0]), int(parts[1]), int(parts[2])
zscale = float(parts[3]) if len(parts) > 3 else 1e-3
c = synth.make_case(nrow, nchan, npix, zscale=zscale, seed=0, with_vis=False)
kw = dict(npix_x=npix, npix_y=npix, pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0, epsilon=1e-7, flip_u=False,
          flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False)
for it in range(3):
    t0 = time.perf_counter()
    g = Gridder(c["uvw"], c["freq"], c["mask"], verbosity=1 if it == 2 else 0, **kw)
    t1 = time.perf_counter()
    print(f"plan {it}: {1e3 * (t1 - t0):.1f} ms", g.info["nplanes"], g.info["nu"], flush=True)
    g.close()
