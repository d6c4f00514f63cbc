#!/usr/bin/env python3
"""dev helper (GPU box): one fuzz geometry under PFBHIP_SCATTER = rec_es / block / walk against the DFT"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PFBHIP_SCATTER", "rec")
import test_gpu_fuzz as F  # noqa: E402
from oracle import dft  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402
from pfb_imaging_amd.wgridder import Gridder  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 36
p = F.CASES[k]
print(p)
c = synth.make_case(p["nrow"], p["nchan"], max(p["nx"], p["ny"]), zscale=p["zscale"], seed=p["seed"])
cell = min(c["cell"] * p["widen"], 0.4 / max(p["nx"], p["ny"]))
fu, fv, fw = p["flips"]
cx, cy = p["center"]
nx, ny = p["nx"], p["ny"]
kw = dict(npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell * 1.1, center_x=cx, center_y=cy, epsilon=p["eps"], flip_u=fu,
          flip_v=fv, flip_w=fw, do_wgridding=True, divide_by_n=p["divn"], force_wmode=0)
ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], nx, ny, cell, cell * 1.1, cx, cy, fu, fv, fw, True, p["divn"])
out = {}
for mode in ("rec_es", "block", "walk"):
    os.environ["PFBHIP_SCATTER"] = mode
    g = Gridder(c["uvw"], c["freq"], c["mask"], **kw)
    info = g.info
    out[mode] = g.vis2dirty(c["vis"], c["wgt"])
    g.close()
    print(mode, "scatter_mode", info["scatter_mode"], "W", info["W"], "planes", info["nplanes"], "nwork", info["nwork"], "nactive", info["nactive"],
          "rel vs dft", np.linalg.norm(out[mode] - ref) / np.linalg.norm(ref))
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print("rec_es vs block", rel(out["rec_es"], out["block"]), "block vs walk", rel(out["block"], out["walk"]), "rec_es vs walk", rel(out["rec_es"], out["walk"]))
