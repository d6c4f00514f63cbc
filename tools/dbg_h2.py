import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PFBHIP_SCATTER", "rec")
from tests.test_gpu_gridder import make, gpu_plan, rel

c = make(nrow=2000, npix=64, widen=8.0, zscale=0.05)
rng = np.random.default_rng(11)
c["nx"], c["ny"] = 1100, 1000
c["cell"] = c["cell"] * 64.0 / 1100
c["x"] = rng.standard_normal((1100, 1000))
g, kw, mask = gpu_plan(c)
print({k: g.info[k] for k in ("nu", "nv", "nplanes", "nwork", "occ_rows", "used_cells", "fft_mode", "scatter_mode", "scatter_launches")})
g.set_weights(c["wgt"])
h1 = g.hessian(c["x"])
h2 = g.hessian(2.0 * c["x"])
h3 = g.hessian(c["x"])
print("rel(h2, 2 h1)", rel(h2, 2 * h1), "rel(h3, h1)", rel(h3, h1), np.abs(h1).max(), np.abs(h2).max())
comp = g.vis2dirty(g.dirty2vis(c["x"]), c["wgt"])
print("rel(h1, composition)", rel(h1, comp), "rel(h3, comp)", rel(h3, comp))
