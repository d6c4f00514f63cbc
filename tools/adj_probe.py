import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from pfb_imaging_amd.utils import synth
from pfb_imaging_amd.wgridder import Gridder
KW = dict(center_x=0.0, center_y=0.0, epsilon=1e-7, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False, sigma_min=1.1, sigma_max=3.0)
c = synth.make_case(250_000, 8, 4096, zscale=3.0, seed=5000, f0=856e6)
npix = 4096
y = c["vis"] * c["mask"]
for env in ({}, {"PFBHIP_SCATTER": "walk"}, {"PFBHIP_SCATTER": "block"}, {"PFBHIP_FUSED_FFT": "0"}, {"PFBHIP_FUSED_FFT": "0", "PFBHIP_ROWFFT": "0"}, {"PFBHIP_TFFT": "0"}):
    for k in ("PFBHIP_SCATTER", "PFBHIP_FUSED_FFT", "PFBHIP_ROWFFT", "PFBHIP_TFFT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for wm in (0, 1):
        try:
            g = Gridder(c["uvw"], c["freq"], c["mask"], npix_x=npix, npix_y=npix, pixsize_x=c["cell"], pixsize_y=c["cell"], force_wmode=wm, **KW)
        except Exception as e:
            print(env, wm, "plan failed", str(e)[:60]); continue
        vis = g.dirty2vis(c["x"]); d = g.vis2dirty(y)
        lhs, rhs = np.vdot(vis, y).real, np.vdot(c["x"], d)
        print(env, "wmode", g.info["wmode"], "planes", g.info["nplanes"], "scatter", g.info["scatter_mode"], "fft", g.info["fft_mode"], abs(lhs-rhs)/max(abs(lhs),abs(rhs)), flush=True)
        g.close()
