#!/usr/bin/env python3
"""dev helper (GPU box): TWO bands of BASELINE config C2 on ONE GPU -- their exact Hessian applies issued from two host threads over the
handles' own streams (what BandWorkerPool does with its local bands) against one band after the other.  The stages of an apply are
bound by different units (gather / scatter: f64 issue; row transforms: LDS + HBM), so two applies in flight fill each other's gaps.
   python tools/bench_two_bands.py [applies per band] [config]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_imaging_amd import _lib  # noqa: E402
from pfb_imaging_amd._lib import DeviceArray  # noqa: E402
from pfb_imaging_amd.utils import synth  # noqa: E402
from pfb_imaging_amd.wgridder import Gridder  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"
bands = []
for b in range(2):
    c = synth.make_config(cfg, band=b)
    nx, ny = c["nx"], c["ny"]
    g = Gridder(c["uvw"], c["freq"], c["mask"], npix_x=nx, npix_y=ny, pixsize_x=c["cell"], pixsize_y=c["cell"], center_x=0.0, center_y=0.0,
                epsilon=1e-7, flip_u=False, flip_v=True, flip_w=False, do_wgridding=True, divide_by_n=False, sigma_min=1.1, sigma_max=3.0)
    g.set_weights(c["wgt"])
    bands.append((g, DeviceArray.from_host(c["x"]), DeviceArray((nx, ny), np.float64), float(c["wgt"][c["mask"] != 0].sum())))


def run(band, k):
    g, x, out, wsum = band
    for _ in range(k):
        g.hessian_dev(x, out, eta=0.0, wsum=wsum)


for band in bands:
    run(band, 3)
_lib.check(_lib.lib().pfbhip_synchronize())
t0 = time.perf_counter()
for band in bands:
    run(band, K)
_lib.check(_lib.lib().pfbhip_synchronize())
t_seq = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(band, K)) for band in bands]
for t in th:
    t.start()
for t in th:
    t.join()
_lib.check(_lib.lib().pfbhip_synchronize())
t_par = time.perf_counter() - t0
nact = sum(b[0].nactive for b in bands)
print(f"{cfg}: 2 bands x {K} applies on one GPU: one after the other {t_seq / (2 * K) * 1e3:.3f} ms per apply "
      f"({2 * K * nact / t_seq / 1e6:.0f} Mvis/s), two host threads {t_par / (2 * K) * 1e3:.3f} ms per apply "
      f"({2 * K * nact / t_par / 1e6:.0f} Mvis/s per GPU)")
for b in bands:
    b[0].close()
