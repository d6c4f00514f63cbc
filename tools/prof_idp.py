import os, sys, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfb_imaging_amd.operators import gridder as G
from pfb_imaging_amd.utils import synth
npix=4096; c=synth.make_case(250000,8,npix,zscale=1e-3,seed=0); cell=c["cell"]
x=c["x"]; beam=np.ones((1,npix,npix)); wgt3,vis3=c["wgt"][None],c["vis"][None]
f=lambda: G.image_data_products_arrays(c["uvw"], c["freq"], vis3, wgt3, c["mask"], npix, npix, 2*npix, 2*npix, cell, cell, model=x[None], beam=beam)
f()
pr=cProfile.Profile(); pr.enable(); f(); pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
