"""Synthetic inputs of BASELINE.md section 2 / SURVEY.md section 8(d) (numpy only; used by
bench.py, the tests and __graft_entry__.smoke()).

Scaled-up version of the reference's own synthetic arrays
(/root/reference/tests/test_hessian_approx.py:97-102: antenna-difference
baselines with z scaled 1e-3; /root/reference/tests/test_weighting.py:89:
``wgt = exp(N(0,1))``).
"""

import numpy as np

SPEED_OF_LIGHT = 299792458.0

CONFIGS = {
    # name: (nrow, nchan, npix, zscale)
    "C1": (50_000, 2, 1024, 1e-3),
    "C2": (1_250_000, 8, 8192, 1e-3),
    "C4": (1_250_000, 8, 4096, 1e-3),
    # antenna z-scale 1.4: with the C5 cell (16384^2 pixels over the full field) the plan then needs 64 ES-kernel w-planes
    # at epsilon 1e-7 (tools/calib_c5.py on the GPU box: planes = 15 + 35 zscale) -- the count BASELINE.json names
    "C5": (12_500_000, 8, 16384, 1.4),
}


def make_uvw(nrow, rng, zscale=1e-3, nant=64):
    """Antenna-difference baselines rotated over 'time' to reach nrow rows."""
    ant = 10e3 * rng.standard_normal((nant, 3))
    ant[:, 2] *= zscale
    a1, a2 = np.triu_indices(nant, 1)
    base = ant[a1] - ant[a2]
    nbl = base.shape[0]
    ntime = -(-nrow // nbl)
    ang = np.linspace(0.0, np.pi, ntime, endpoint=False)
    out = np.empty((ntime * nbl, 3))
    for t, a in enumerate(ang):
        ca, sa = np.cos(a), np.sin(a)
        blk = out[t * nbl:(t + 1) * nbl]
        blk[:, 0] = ca * base[:, 0] - sa * base[:, 1]
        blk[:, 1] = sa * base[:, 0] + ca * base[:, 1]
        blk[:, 2] = base[:, 2]
    return np.ascontiguousarray(out[:nrow])


def make_case(nrow, nchan, npix, zscale=1e-3, seed=0, f0=1.284e9, with_vis=True):
    """Returns dict(uvw, freq, vis, wgt, mask, cell, nx, ny, x)."""
    rng = np.random.default_rng(seed)
    uvw = make_uvw(nrow, rng, zscale)
    freq = np.linspace(0.9, 1.1, nchan) * f0 if nchan > 1 else np.array([f0])
    uvmax = np.sqrt((uvw[:, :2] ** 2).sum(axis=1)).max()
    cell = 1.0 / (2 * uvmax * freq.max() / SPEED_OF_LIGHT) / 2.0
    out = dict(uvw=uvw, freq=freq, cell=cell, nx=npix, ny=npix)
    if with_vis:
        out["vis"] = (rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))) / np.sqrt(2.0)
    out["wgt"] = np.exp(rng.standard_normal((nrow, nchan)))
    out["mask"] = (rng.random((nrow, nchan)) > 0.05).astype(np.uint8)
    out["x"] = rng.standard_normal((npix, npix))
    return out


def make_config(name, band=0, **kw):
    nrow, nchan, npix, zs = CONFIGS[name]
    cfg = int(name[1:])
    f0 = 856e6 + (1712e6 - 856e6) * (band / 8.0)
    return make_case(nrow, nchan, npix, zs, seed=1000 * cfg + band, f0=f0, **kw)
