"""dev helper: block (register-footprint) scatter against the diagonal-walk scatter -- run-to-run noise and difference,
narrow field (no edge amplification) and wide field."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_gridder import make, rel
from oracle import dft
from pfb_imaging_amd.wgridder import Gridder
for name, kwm in (("narrow", dict(nrow=1500, npix=256, widen=2.0, zscale=0.01)), ("wide", dict(nrow=1500, npix=1024, widen=60.0, zscale=1.0))):
    c = make(**kwm)
    kw = dict(npix_x=c["nx"], npix_y=c["ny"], pixsize_x=c["cell"], pixsize_y=c["cell"], epsilon=1e-7, flip_v=True, do_wgridding=True, divide_by_n=False)
    out = {}
    for mode in ("walk", "block"):
        os.environ["PFBHIP_SCATTER"] = mode
        g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
        d1 = g.vis2dirty(c["vis"], c["wgt"]); d2 = g.vis2dirty(c["vis"], c["wgt"])
        print(name, mode, "run-to-run", rel(d1, d2), "nplanes", g.info["nplanes"], "W", g.info["W"], "wmode", g.info["wmode"], flush=True)
        out[mode] = d1
        g.close()
    print(name, "walk vs block", rel(out["walk"], out["block"]), flush=True)
    if name == "narrow":
        ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], c["nx"], c["ny"], c["cell"], c["cell"], 0.0, 0.0, False, True, False, True, False)
        print(name, "vs DFT: walk", rel(out["walk"], ref), "block", rel(out["block"], ref), flush=True)
