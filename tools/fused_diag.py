#!/usr/bin/env python3
"""dev tool: compare fused / rocFFT plane transforms on the test cases"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.wgridder import Gridder
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for widen, zscale in ((8.0, 0.02), (60.0, 1.0), (1150.0, 0.002)):
    c = synth.make_case(2500, 2, 1024, zscale=zscale, seed=1)
    c["cell"] *= widen
    kw = dict(npix_x=c["nx"], npix_y=c["ny"], pixsize_x=c["cell"], pixsize_y=c["cell"], epsilon=1e-7, flip_v=True,
              do_wgridding=True, divide_by_n=False, force=(1.25, 15), verbosity=1)
    res = {}
    for name, env in (("own", {}), ("rocfft", {"PFBHIP_FUSED_FFT": "0", "PFBHIP_ROWFFT": "0"}),
                      ("fused+rocfftA", {"PFBHIP_ROWFFT": "0"}), ("unfused+ownA", {"PFBHIP_FUSED_FFT": "0"})):
        for k in ("PFBHIP_FUSED_FFT", "PFBHIP_ROWFFT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
        res[name] = (g.vis2dirty(c["vis"], c["wgt"]), g.dirty2vis(c["x"]))
        info = g.info
        g.close()
    print("case", widen, zscale, "planes", info["nplanes"], "wmode", info["wmode"])
    for name in res:
        d = np.abs(res[name][0] - res["rocfft"][0])
        print("  ", name, rel(res[name][0], res["rocfft"][0]), rel(res[name][1], res["rocfft"][1]),
              "max at", np.unravel_index(d.argmax(), d.shape), d.max() / np.abs(res["rocfft"][0]).max())

