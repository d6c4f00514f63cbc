"""GPU parity against REFERENCE-RUN fixtures (tests/golden/ref_pins.npz, conventions_two_sources.npz): what the
reference's own undecorated numpy / scipy functions returned in the build container (tests/golden/make_ref_pins.py),
compared with the HIP path through the C-ABI.  No oracle in between."""

import numpy as np
import pytest

from pfb_imaging_amd import prox, wgridder
from pfb_imaging_amd.operators import gridder as pg
from pfb_imaging_amd.utils import weighting as pw

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pins(golden_dir):
    return np.load(f"{golden_dir}/ref_pins.npz")


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


@pytest.mark.parametrize("eps", [1e-5, 1e-8, 1e-10])
def test_dense_widefield_vs_explicit_wdegridder(pins, eps):
    """dirty2vis on every pixel of a small wide-field image == the reference's explicit_wdegridder
    (tests/test_hessian_approx.py:44-67) to the requested epsilon (relative l2), for three phase centres under
    wgridder_conventions, and vis2dirty is its adjoint to rounding."""
    px, py = pins["dense_pix"]
    img, uvw, freq = pins["dense_img"], pins["dense_uvw"], pins["dense_freq"]
    for k, (l0, m0) in enumerate(pins["conv_lm"][:3]):
        fu, fv, fw, x0, y0 = pg.wgridder_conventions(l0, m0)
        kw = dict(uvw=uvw, freq=freq, pixsize_x=px, pixsize_y=py, center_x=x0, center_y=y0, epsilon=eps, flip_u=fu,
                  flip_v=fv, flip_w=fw, do_wgridding=True, divide_by_n=True)
        v = wgridder.dirty2vis(dirty=img, **kw)
        assert rel(v, pins["dense_vis_w"][k]) < eps
        y = pins["dense_vis_w"][k]
        d = wgridder.vis2dirty(vis=y, npix_x=img.shape[0], npix_y=img.shape[1], **kw)
        lhs, rhs = np.vdot(v, y).real, np.vdot(img, d)
        assert abs(lhs - rhs) < 1e-9 * abs(lhs)


@pytest.mark.parametrize("eps", [1e-6, 1e-9])
def test_dense_casa_convention_vs_explicit_degridder(pins, eps):
    """ducc0's default convention (no flips) == explicit_degridder(..., convention="casa")
    (tests/test_hessian_approx.py:23-41, used by test_gridder_conventions :70-125)."""
    px, py = pins["dense_pix"]
    for k, (l0, m0) in enumerate(pins["conv_lm"][:3]):
        v = wgridder.dirty2vis(uvw=pins["dense_uvw"], freq=pins["dense_freq"], dirty=pins["dense_img"], pixsize_x=px,
                               pixsize_y=py, center_x=-l0, center_y=-m0, epsilon=eps, flip_v=False, do_wgridding=True,
                               divide_by_n=True)
        assert rel(v, pins["dense_vis_casa"][k]) < eps


def test_casa_two_point_sources(golden_dir):
    """test_gridder_conventions (:70-125) on the reference-run vectors: atol 1e-4 at epsilon 1e-6, flip_v=False."""
    gold = np.load(f"{golden_dir}/conventions_two_sources.npz")
    npix, pix = int(gold["npix"]), float(gold["pixsize"])
    dirty = np.zeros((npix, npix))
    dirty[npix // 2, npix // 2] = 1.0
    dirty[npix // 4, npix // 4] = 1.0
    for k, (l0, m0) in enumerate(gold["offsets"]):
        v = wgridder.dirty2vis(uvw=gold["uvw"], freq=gold["freq"], dirty=dirty, pixsize_x=pix, pixsize_y=pix,
                               center_x=-l0, center_y=-m0, epsilon=1e-6, do_wgridding=True, flip_v=False,
                               divide_by_n=True)
        np.testing.assert_allclose(v.real, gold["vis_casa"][k].real, atol=1e-4)
        np.testing.assert_allclose(v.imag, gold["vis_casa"][k].imag, atol=1e-4)
        assert rel(v, gold["vis_casa"][k]) < 1e-6


def test_prox_21m_and_dual_update(pins):
    """pfbhip_prox_21m / pfbhip_dual_update == the reference's prox_21m (prox/prox_21m.py:5-27) and dual_update (:64-71)."""
    for i in range(3):
        v, w, s = pins[f"prox{i}_v"], pins[f"prox{i}_w"], float(pins[f"prox{i}_sigma"])
        assert np.abs(prox.prox_21m(v, s, weight=w) - pins[f"prox{i}_out"]).max() < 1e-13
        assert np.abs(prox.prox_21m(v, s) - pins[f"prox{i}_out_w1"]).max() < 1e-13
    q, x = pins["du_q"], pins["du_x"]
    coeffs = np.ascontiguousarray(np.stack([x, np.einsum("ij,bjk->bik", q, x)], axis=1))
    prox.dual_update_numba_fast(pins["du_v"], coeffs, float(pins["du_lam"]), sigma=float(pins["du_sigma"]),
                                weight=pins["du_w"])
    assert np.abs(coeffs - pins["du_out"]).max() < 1e-12


def test_counts_filters(pins):
    """pfbhip_filter_extreme_counts / pfbhip_box_sum_counts == utils/weighting.py:212-254 run by the reference."""
    for lvl in (10.0, 2.0, 0.0):
        got = pw.filter_extreme_counts(pins["cnt_in"].copy(), level=lvl)
        assert np.abs(got - pins[f"cnt_filter_{lvl}"]).max() < 1e-12
    for s in (0, 1, 2, 5):
        got = pw.box_sum_counts(pins["cnt_in"].copy(), s)
        assert np.abs(got - pins[f"cnt_box_{s}"]).max() < 1e-12 * max(1.0, np.abs(pins[f"cnt_box_{s}"]).max())
