#!/usr/bin/env python3
"""dev helper (GPU box): where a work item of the record scatter spends its cycles.
   PFBHIP_STAMP=1 python tools/stamp_scatter.py [config]"""
import ctypes as ct
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PFBHIP_STAMP", "1")
from pfb_imaging_amd import _lib  # noqa: E402
from pfb_imaging_amd._lib import DeviceArray  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402
from pfb_imaging_amd.wgridder import Gridder  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
case = synth.make_config(cfg, band=0)
nx, ny = case["nx"], case["ny"]
g = Gridder(case["uvw"], case["freq"], case["mask"], npix_x=nx, npix_y=ny, pixsize_x=case["cell"], pixsize_y=case["cell"],
            center_x=0.0, center_y=0.0, epsilon=1e-7, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True,
            divide_by_n=False, sigma_min=1.1, sigma_max=3.0)
g.set_weights(case["wgt"])
x = DeviceArray.from_host(case["x"])
out = DeviceArray((nx, ny), np.float64)
for _ in range(3):
    g.hessian_dev(x, out, eta=0.0, wsum=1.0)
n = ct.c_int64(0)
L = _lib.lib()
_lib.check(L.pfbhip_gridder_debug_stamps(g._h, None, ct.c_int64(0), ct.byref(n)))
buf = np.zeros((n.value, 8), dtype=np.uint64)
_lib.check(L.pfbhip_gridder_debug_stamps(g._h, buf.ctypes.data_as(ct.c_void_p), ct.c_int64(n.value), ct.byref(n)))
b = buf.astype(np.float64)
b = b[b[:, 4] > 0]
tot = b[:, 0] + b[:, 1] + b[:, 2] + b[:, 3]
print(f"{len(b)} items, scatter_mode {g.info['scatter_mode']}; cycles summed over items (wave 0's view):")
names = ["prologue", "vis loop (wave 0)", "barrier wait (wave 0)", "tile flush"] if os.environ["PFBHIP_STAMP"] == "1" else ["tile load", "rounds (wave 0)", "-", "-"]
for i, name in enumerate(names):
    print(f"  {name:24s} {b[:, i].sum() / tot.sum() * 100:5.1f} %   mean {b[:, i].mean():9.0f} cycles")
print(f"  last wave loop mean {b[:, 5].mean():9.0f}, wait / total {b[:, 6].mean():9.0f}")
nw = 12 if (os.environ["PFBHIP_STAMP"] == "1" or g.info["wmode"] == 2) else 16
print(f"  vis per item mean {b[:, 4].mean():.0f}; cycles per visibility of a wave's share: "
      f"{(b[:, 1] / np.maximum(b[:, 4] / nw, 1)).mean():.0f} (mean over items), "
      f"{b[:, 1].sum() / (b[:, 4].sum() / nw):.0f} (weighted)")
for lo, hi in [(1, 64), (64, 256), (256, 1024), (1024, 4097)]:
    m = (b[:, 4] >= lo) & (b[:, 4] < hi)
    if m.any():
        print(f"  items with {lo:4d}..{hi:4d} vis: {m.sum():6d}  share of cycles {tot[m].sum() / tot.sum() * 100:5.1f} %  "
              f"prologue {b[m, 0].mean():7.0f} loop {b[m, 1].mean():8.0f} wait {b[m, 2].mean():7.0f} flush {b[m, 3].mean():7.0f}  "
              f"cycles/vis/wave {(b[m, 1].sum() / (b[m, 4].sum() / nw)):.0f}")
g.close()
