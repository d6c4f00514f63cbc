#!/usr/bin/env python3
"""dev helper (GPU box): w-plane count of the C5 geometry as a function of the antenna z-scale."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd.utils import synth  # noqa: E402
from pfb_imaging_amd.wgridder import Gridder  # noqa: E402

for zs in [float(v) for v in sys.argv[1:]] or [0.3, 0.6, 0.9]:
    c = synth.make_case(250_000, 8, 16384, zscale=zs, seed=5000, f0=856e6, with_vis=False)
    t = time.time()
    g = Gridder(c["uvw"], c["freq"], c["mask"], npix_x=16384, npix_y=16384, pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0,
                center_y=0.0, epsilon=1e-7, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False, sigma_min=1.1,
                sigma_max=3.0)
    i = g.info
    print(f"zscale {zs}: nplanes {i['nplanes']} wmode {i['wmode']} W {i['W']} sigma {i['sigma']:.3f} grid {i['nu']}x{i['nv']} "
          f"plan {time.time() - t:.1f}s cell {c['cell']:.3e}", flush=True)
    g.close()
