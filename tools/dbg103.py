import os, sys
import numpy as np
os.environ["PFB_FUZZ_SEED"] = "5"; os.environ["PFB_FUZZ_N"] = "150"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dft, wgridder as owg
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.wgridder import Gridder
from tests.test_gpu_fuzz import CASES
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
p = CASES[103]; print(p)
c = synth.make_case(p["nrow"], p["nchan"], max(p["nx"], p["ny"]), zscale=p["zscale"], seed=p["seed"])
cell = min(c["cell"] * p["widen"], 0.4 / max(p["nx"], p["ny"]))
fu, fv, fw = p["flips"]; cx, cy = p["center"]; nx, ny = p["nx"], p["ny"]
x = np.random.default_rng(p["seed"]).standard_normal((nx, ny))
for fw_ in (None, 0):
    g = Gridder(c["uvw"], c["freq"], c["mask"], npix_x=nx, npix_y=ny, pixsize_x=cell, pixsize_y=cell * 1.1, center_x=cx, center_y=cy,
                epsilon=p["eps"], flip_u=fu, flip_v=fv, flip_w=fw, do_wgridding=p["do_w"], divide_by_n=p["divn"], force_wmode=fw_ if fw_ is not None else p["wmode"])
    print({q: g.info[q] for q in ("nu", "nv", "nplanes", "W", "wmode", "sigma", "kernel_eps", "dw", "nshift")})
    args = (cell, cell * 1.1, cx, cy, fu, fv, fw, p["do_w"], p["divn"])
    o = owg.Plan(c["uvw"], c["freq"], c["mask"], nx, ny, cell, cell * 1.1, cx, cy, p["eps"], fu, fv, fw, p["do_w"], p["divn"], params=g.oracle_params())
    v = g.dirty2vis(x); refv = dft.dft_dirty2vis(c["uvw"], c["freq"], x, *args); refv[c["mask"] == 0] = 0
    d = g.vis2dirty(c["vis"], c["wgt"]); ref = dft.dft_vis2dirty(c["uvw"], c["freq"], c["vis"], c["wgt"], c["mask"], nx, ny, *args)
    print("  d2v gpu/dft %.2e  oracle/dft %.2e   v2d gpu/dft %.2e oracle/dft %.2e" % (rel(v, refv), rel(o.dirty2vis(x), refv), rel(d, ref), rel(o.vis2dirty(c["vis"], c["wgt"]), ref)))
    n = np.sqrt(1 - (cx + (np.arange(nx) - nx // 2) * cell)[:, None] ** 2 - (cy * (-1 if fv else 1) + (np.arange(ny) - ny // 2) * cell * 1.1)[None] ** 2)
    print("  n range", n.min(), n.max())
