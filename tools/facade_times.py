#!/usr/bin/env python3
"""dev helper (GPU box): wall-clock of the reference-signature facades of the gridder path at a mid size, next to the device apply
they wrap -- to spot image-sized numpy passes on the host.   python tools/facade_times.py [npix] [nrow]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd import _lib  # noqa: E402
from pfb_imaging_amd.operators import gridder as G  # noqa: E402
from pfb_imaging_amd.operators.hessian import hessian_slice  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402

npix = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nrow = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
c = synth.make_case(nrow, 8, npix, zscale=1e-3, seed=0)
cell = c["cell"]


def t(label, f, n=3):
    f()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        out = f()
        best = min(best, time.perf_counter() - t0)
    print(f"{label:64s} {best * 1e3:9.1f} ms", flush=True)
    return out


x = c["x"]
beam = np.ones((1, npix, npix))
wgt3, vis3 = c["wgt"][None], c["vis"][None]
t("hessian_slice (facade, beam=None)", lambda: hessian_slice(x, uvw=c["uvw"], weight=c["wgt"], vis_mask=c["mask"], freq=c["freq"], beam=None,
                                                               cell=cell, epsilon=1e-7, eta=0.1, wsum=1.0))
t("hessian_slice (facade, beam)", lambda: hessian_slice(x, uvw=c["uvw"], weight=c["wgt"], vis_mask=c["mask"], freq=c["freq"], beam=beam[0],
                                                          cell=cell, epsilon=1e-7, eta=0.1, wsum=1.0))
t("vis2im (double)", lambda: G.vis2im(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], npix, npix, cell, cell, 0.0, 0.0, 1e-7, "double", True,
                                      False, 1, 1.1, 3.0, True))
t("compute_residual_arrays", lambda: G.compute_residual_arrays(x[None], x[None], c["uvw"], c["freq"], wgt3, c["mask"], beam, cell))
part = {"UVW": c["uvw"], "FREQ": c["freq"], "WEIGHT": wgt3, "MASK": c["mask"], "BEAM": beam, "attrs": {"l0": 0.0, "m0": 0.0}}
t("residual_from_partitions (1 partition; builds its plan per call)", lambda: G.residual_from_partitions(x[None], [part], x[None], cell))
t("image_data_products_arrays (dirty + psf + residual, natural wgt)",
  lambda: G.image_data_products_arrays(c["uvw"], c["freq"], vis3, wgt3, c["mask"], npix, npix, 2 * npix, 2 * npix, cell, cell, model=x[None],
                                       beam=beam), n=2)
