#!/usr/bin/env python3
"""Reference-RUN fixtures: outputs of the reference's own code, executed in the build container.

The hot path's arithmetic lives in ducc0 / numba, neither importable here, so the reference's modules cannot be
imported as modules.  A handful of its functions, however, are plain numpy / scipy and carry no decorator.  This
script reads the files under /root/reference AT GENERATION TIME, takes those function definitions out of the parsed
module (``ast``) and executes them AS THEY STAND:

  * no stub modules: the module-level ``import`` statements of the file are executed one by one and the ones this
    container cannot satisfy (ducc0, numba, jax, ray, dask, numexpr, xarray, pfb_imaging ...) are simply skipped, so
    the names they would bind do not exist; a function that touched one would raise NameError, none taken here does;
  * only UNDECORATED top-level functions are taken (asserted), nothing is edited;
  * no reference source text is stored in the repository: only inputs and outputs go into ``ref_pins.npz``.

What is pinned (reference file:line -> what the oracle / product has to reproduce):

  tests/test_hessian_approx.py:44-67     explicit_wdegridder  -> oracle/dft.py (flip_v convention, divide_by_n)
  tests/test_hessian_approx.py:23-41     explicit_degridder   -> oracle/dft.py (no flips; both ``negate_w``)
  operators/gridder.py:23-34             wgridder_conventions -> operators/gridder.py
  prox/prox_21m.py:5-27, 64-71           prox_21m, dual_update -> oracle/psi.py, prox.py
  utils/weighting.py:212-254             filter_extreme_counts, box_sum_counts -> oracle/weighting.py, utils/weighting.py
  utils/misc.py:968-975                  taperf               -> oracle/fftconv.py, operators/hessian.py
  opt/power_method.py:95-147             power_method         -> oracle/fftconv.py, opt.py (host loop)
  opt/primal_dual.py:66-163              primal_dual (legacy) -> opt.py (host loop)
  utils/weighting.py:471-505             reduce_counts        -> (semantic check of the band/time grouping used by the tests)

NOT pinned by this (and it cannot be here): ducc0's floating-point output (wheel absent), the numba kernels
(_compute_counts, counts_to_weights, the DWT: decorated, numba absent), PyWavelets' filter tables.

Run from the repo root in the build container:  python tests/golden/make_ref_pins.py
It also rewrites conventions_two_sources.npz so that its expected visibilities are the REFERENCE's explicit formula.
"""

import ast
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def take(relpath, names):
    """Namespace holding the undecorated top-level functions ``names`` of a reference file, executed unmodified."""
    path = os.path.join(REF, relpath)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    ns = {"__name__": "refpin_" + os.path.basename(relpath)[:-3]}
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            try:
                exec(compile(ast.Module([node], []), path, "exec"), ns)
            except Exception:  # module absent here (ducc0, numba, jax, ...): the names stay unbound
                pass
    found = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            assert not node.decorator_list, f"{relpath}:{node.name} is decorated"
            exec(compile(ast.Module([node], []), path, "exec"), ns)
            found[node.name] = (node.lineno, node.end_lineno)
    missing = set(names) - set(found)
    assert not missing, f"{relpath}: {missing} not found"
    return ns, found


def main():
    out = {}
    cites = []

    def cite(rel, found):
        for k, (a, b) in sorted(found.items()):
            cites.append(f"{rel}:{a}-{b} {k}")

    # ---- conventions ----------------------------------------------------------------------------------------
    g_ns, f = take("src/pfb_imaging/operators/gridder.py", ["wgridder_conventions"])
    cite("src/pfb_imaging/operators/gridder.py", f)
    wgc = g_ns["wgridder_conventions"]
    lm = np.array([(0.0, 0.0), (0.1, -0.17), (0.2, 0.5), (-0.1, 0.2), (-0.15, -0.2)])
    out["conv_lm"] = lm
    out["conv_out"] = np.array([[float(v) for v in wgc(l0, m0)] for l0, m0 in lm])

    # ---- the measurement equation ---------------------------------------------------------------------------
    t_ns, f = take("tests/test_hessian_approx.py", ["explicit_degridder", "explicit_wdegridder"])
    cite("tests/test_hessian_approx.py", f)
    t_ns["wgridder_conventions"] = wgc  # the name explicit_wdegridder resolves at module scope in the reference
    ewd, ed = t_ns["explicit_wdegridder"], t_ns["explicit_degridder"]

    # (i) the inputs of the reference's own test_wgridder_conventions / test_gridder_conventions (:70-185)
    np.random.seed(42)
    npix, num_ants, num_freqs = 1024, 100, 2
    pixsize = 0.5 * np.pi / 180 / 3600.0
    a1, a2 = np.asarray(list(itertools.combinations(range(num_ants), 2))).T
    ant = 10e3 * np.random.normal(size=(num_ants, 3))
    ant[:, 2] *= 0.001
    uvw = ant[a1] - ant[a2]
    freqs = np.linspace(700e6, 2000e6, num_freqs)
    vis_w, vis_c0, vis_c1 = [], [], []
    for l0, m0 in lm:
        def lmn_w(xi, yi):  # test_wgridder_conventions' pixel_to_lmn (:145-149)
            l_c = -l0 + (-npix / 2 + xi) * pixsize
            m_c = m0 + (-npix / 2 + yi) * pixsize
            return np.asarray([l_c, m_c, np.sqrt(1.0 - l_c**2 - m_c**2)])

        def lmn_c(xi, yi):  # test_gridder_conventions' pixel_to_lmn (:88-92)
            l_c = l0 + (-npix / 2 + xi) * (-pixsize)
            m_c = m0 + (-npix / 2 + yi) * (-pixsize)
            return np.asarray([l_c, m_c, np.sqrt(1.0 - l_c**2 - m_c**2)])

        pts = [(npix // 2, npix // 2), (npix // 4, npix // 4)]
        vis_w.append(ewd(uvw, freqs, [lmn_w(*p) for p in pts], [1.0, 1.0]))
        vis_c0.append(ed(uvw, freqs, [lmn_c(*p) for p in pts], [1.0, 1.0], False, convention="casa"))
        vis_c1.append(ed(uvw, freqs, [lmn_c(*p) for p in pts], [1.0, 1.0], True, convention="casa"))
    np.savez_compressed(os.path.join(HERE, "conventions_two_sources.npz"), uvw=uvw, freq=freqs, npix=npix,
                        pixsize=pixsize, offsets=lm, vis=np.array(vis_w), vis_casa=np.array(vis_c0),
                        vis_casa_negw=np.array(vis_c1),
                        source="explicit_wdegridder / explicit_degridder of the reference, run by make_ref_pins.py")

    # (ii) a dense, well-conditioned case: every pixel of a small wide-field image, phases of a few hundred radians
    rng = np.random.default_rng(2024)
    nx, ny, nrow, nchan = 12, 10, 40, 3
    px, py = 2.0e-2, 1.5e-2
    d_uvw = rng.standard_normal((nrow, 3)) * np.array([4.0, 4.0, 2.0])
    d_freq = np.linspace(0.9e9, 1.1e9, nchan)
    d_img = rng.standard_normal((nx, ny))
    out.update(dense_uvw=d_uvw, dense_freq=d_freq, dense_img=d_img, dense_pix=np.array([px, py]))
    dense_w, dense_c = [], []
    for l0, m0 in lm[:3]:
        lmn_w, lmn_c = [], []
        for xi in range(nx):
            for yi in range(ny):
                l_c, m_c = -l0 + (-nx / 2 + xi) * px, m0 + (-ny / 2 + yi) * py
                lmn_w.append((l_c, m_c, np.sqrt(1.0 - l_c**2 - m_c**2)))
                l_c, m_c = l0 + (-nx / 2 + xi) * (-px), m0 + (-ny / 2 + yi) * (-py)
                lmn_c.append((l_c, m_c, np.sqrt(1.0 - l_c**2 - m_c**2)))
        dense_w.append(ewd(d_uvw, d_freq, lmn_w, d_img.ravel()))
        dense_c.append(ed(d_uvw, d_freq, lmn_c, d_img.ravel(), False, convention="casa"))
    out["dense_vis_w"] = np.array(dense_w)
    out["dense_vis_casa"] = np.array(dense_c)

    # ---- l21 prox and the allocating dual update --------------------------------------------------------------
    p_ns, f = take("src/pfb_imaging/prox/prox_21m.py", ["prox_21m", "dual_update"])
    cite("src/pfb_imaging/prox/prox_21m.py", f)
    prox_21m, dual_update = p_ns["prox_21m"], p_ns["dual_update"]
    for i, (nband, nbasis, nym, nxm, sigma) in enumerate([(1, 2, 7, 5, 0.3), (3, 4, 9, 11, 1.7), (6, 3, 8, 8, 1e-3)]):
        v = rng.standard_normal((nband, nbasis, nym, nxm))
        v[:, 0, 0, :] = 0.0  # zero band sums: the ratio[mask] branch
        w = np.abs(rng.standard_normal((nbasis, nym, nxm))) + 0.1
        out[f"prox{i}_v"], out[f"prox{i}_w"], out[f"prox{i}_sigma"] = v, w, sigma
        out[f"prox{i}_out"] = prox_21m(v, sigma, weight=w)
        out[f"prox{i}_out_w1"] = prox_21m(v, sigma)
    nband, nbasis, nym, nxm = 3, 2, 6, 5
    du_q = np.linalg.qr(rng.standard_normal((nym, nym)))[0]

    def psih(x, vout):  # analysis: basis 0 = the image itself, basis 1 = an orthogonal mix of its rows
        vout[:, 0] = x
        vout[:, 1] = np.einsum("ij,bjk->bik", du_q, x)

    du_v = rng.standard_normal((nband, nbasis, nym, nxm))
    du_x = rng.standard_normal((nband, nym, nxm))
    du_w = np.abs(rng.standard_normal((nbasis, nym, nxm))) + 0.1
    out.update(du_q=du_q, du_v=du_v, du_x=du_x, du_w=du_w, du_lam=0.8, du_sigma=1.3)
    out["du_out"] = dual_update(du_v, du_x, psih, 0.8, sigma=1.3, weight=du_w)

    # ---- counts filters ---------------------------------------------------------------------------------------
    w_ns, f = take("src/pfb_imaging/utils/weighting.py", ["filter_extreme_counts", "box_sum_counts", "reduce_counts"])
    cite("src/pfb_imaging/utils/weighting.py", f)
    counts = rng.gamma(0.5, 4.0, size=(2, 23, 17))
    counts[rng.random(counts.shape) < 0.4] = 0.0
    out["cnt_in"] = counts
    for lvl in (10.0, 2.0, 0.0):
        out[f"cnt_filter_{lvl}"] = w_ns["filter_extreme_counts"](counts.copy(), level=lvl)
    for s in (0, 1, 2, 5):
        out[f"cnt_box_{s}"] = w_ns["box_sum_counts"](counts.copy(), s)
    grids = {(b, t): rng.random((1, 4, 3)) for b in range(3) for t in range(2)}
    out["rc_in"] = np.array([[grids[(b, t)] for t in range(2)] for b in range(3)])
    for grouping in ("per-band-time", "mfs", "per-band", "per-time"):
        red = w_ns["reduce_counts"](grids, grouping)
        out[f"rc_{grouping}"] = np.array([[red[(b, t)] for t in range(2)] for b in range(3)])

    # ---- taper ------------------------------------------------------------------------------------------------
    m_ns, f = take("src/pfb_imaging/utils/misc.py", ["taperf"])
    cite("src/pfb_imaging/utils/misc.py", f)
    for shape, width in (((32, 48), 8), ((17, 9), 3), ((64, 64), 32)):
        out[f"taper_{shape[0]}_{shape[1]}_{width}"] = m_ns["taperf"](shape, width)

    # ---- power method (host loop) -----------------------------------------------------------------------------
    pm_ns, f = take("src/pfb_imaging/opt/power_method.py", ["power_method"])
    cite("src/pfb_imaging/opt/power_method.py", f)
    n1, n2 = 9, 7
    qa = rng.standard_normal((n1, n1))
    qb = rng.standard_normal((n2, n2))
    pm_a, pm_b = qa @ qa.T + np.eye(n1), qb @ qb.T + np.eye(n2)
    pm_b0 = rng.standard_normal((n1, n2))
    beta, bvec = pm_ns["power_method"](lambda x: pm_a @ x @ pm_b, (n1, n2), b0=pm_b0.copy(), tol=1e-10, maxit=400,
                                       verbosity=0)
    out.update(pm_a=pm_a, pm_b=pm_b, pm_b0=pm_b0, pm_beta=float(beta), pm_vec=bvec)

    # ---- legacy primal-dual loop ------------------------------------------------------------------------------
    pd_ns, f = take("src/pfb_imaging/opt/primal_dual.py", ["primal_dual"])
    cite("src/pfb_imaging/opt/primal_dual.py", f)
    nband, nym, nxm = 2, 6, 5
    pd_q = np.linalg.qr(rng.standard_normal((nym, nym)))[0]
    pd_h = 0.5 + rng.random((nband, nym, nxm))  # a diagonal, positive "Hessian"
    pd_truth = np.maximum(rng.standard_normal((nband, nym, nxm)), 0.0)
    pd_b = pd_h * pd_truth
    pd_w = np.abs(rng.standard_normal((2, nym, nxm))) + 0.1

    def pd_psih(x):
        return np.stack([x, np.einsum("ij,bjk->bik", pd_q, x)], axis=1)

    def pd_psi(v):
        return v[:, 0] + np.einsum("ji,bjk->bik", pd_q, v[:, 1])

    for pos in (0, 1, 2):
        x, v = pd_ns["primal_dual"](np.zeros((nband, nym, nxm)), np.zeros((nband, 2, nym, nxm)), 0.05, pd_psi, pd_psih,
                                    float(pd_h.max()), lambda a, s: prox_21m(a, s, weight=pd_w),
                                    lambda x: pd_h * x - pd_b, nu=2.0, tol=1e-9, maxit=60, minit=10, positivity=pos,
                                    verbosity=0)
        out[f"pd_x_{pos}"], out[f"pd_v_{pos}"] = x, v
    out.update(pd_q=pd_q, pd_h=pd_h, pd_b=pd_b, pd_w=pd_w, pd_lam=0.05)

    out["cites"] = np.array(cites)
    np.savez_compressed(os.path.join(HERE, "ref_pins.npz"), **out)
    print("\n".join(cites))
    for fn in ("ref_pins.npz", "conventions_two_sources.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("make_ref_pins.py needs /root/reference (build container only); the committed .npz files travel instead")
    main()
