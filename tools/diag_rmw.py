#!/usr/bin/env python3
"""dev helper (GPU box): the two forms of the fused second axis (LDS image row / read-modify-write) against each other AND
against the oracle restatement, output by output -- which of them moved when a comparison between the two fails."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PFBHIP_SCATTER", "rec")
from oracle import wgridder as owg  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402
from pfb_imaging_amd.wgridder import Gridder  # noqa: E402


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


c = synth.make_case(1500, 2, 1024, zscale=1.0, seed=1)
c["cell"] = c["cell"] * 60.0
kw = dict(npix_x=c["nx"], npix_y=c["ny"], pixsize_x=c["cell"], pixsize_y=c["cell"], epsilon=1e-7, flip_v=True, do_wgridding=True,
          divide_by_n=False)
beam = 0.5 + np.random.default_rng(3).random((c["nx"], c["ny"]))
res = {}
for env in ("1", "0"):
    os.environ["PFBHIP_FUSED_LDSROW"] = env
    g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
    g.set_weights(c["wgt"])
    res[env] = (g.vis2dirty(c["vis"], c["wgt"]), g.dirty2vis(c["x"]), g.hessian(c["x"], beam=beam, eta=0.3, wsum=7.0))
    if env == "1":
        o = owg.Plan(c["uvw"], c["freq"], c["mask"], c["nx"], c["ny"], c["cell"], c["cell"], 0.0, 0.0, 1e-7, False, True, False, True,
                     False, params=g.oracle_params())
        ref = (o.vis2dirty(c["vis"], c["wgt"]), o.dirty2vis(c["x"]),
               beam * o.vis2dirty(o.dirty2vis(beam * c["x"]), c["wgt"]) / 7.0 + 0.3 * c["x"])
        print("info", {k: g.info[k] for k in ("nu", "nv", "nplanes", "W", "wmode", "fft_mode")})
    g.close()
for i, name in enumerate(("vis2dirty", "dirty2vis", "hessian")):
    d = np.abs(res["1"][i] - res["0"][i])
    print(f"{name:10s} row-vs-rmw {rel(res['1'][i], res['0'][i]):.3e}  row-vs-oracle {rel(res['1'][i], ref[i]):.3e}  "
          f"rmw-vs-oracle {rel(res['0'][i], ref[i]):.3e}  max|diff| at {np.unravel_index(np.argmax(d), d.shape)} = {d.max():.3e}")
