"""CPU tests of the band loader (pfb_imaging_amd.store): an in-memory store in the three shapes the loader accepts
(plain mapping, objects with ``.values``, zarr-like arrays that decode into a caller buffer)."""

import numpy as np


class ZarrLike:
    """Stands in for a zarr array: chunked decode into ``out`` (what zarr's get_basic_selection(out=...) does)."""

    def __init__(self, a):
        self._a, self.shape, self.dtype = a, a.shape, a.dtype
        self.decoded_into = None

    def get_basic_selection(self, sel, out=None):
        assert sel is Ellipsis and out is not None
        for i in range(0, self._a.shape[0], 7):  # "chunks"
            out[i:i + 7] = self._a[i:i + 7]
        self.decoded_into = out
        return out

    def __getitem__(self, k):
        return self._a[k]


class XrLike:
    def __init__(self, a):
        self.values, self.shape, self.dtype = a, a.shape, a.dtype


def _store(rng, wrap):
    nrow, nchan, nx, ny, nxp = 50, 3, 16, 12, 32
    parts = {}
    for i, name in enumerate(("part0", "part1")):
        psfhat = rng.standard_normal((1, nxp, nxp // 2 + 1)) + 1j * rng.standard_normal((1, nxp, nxp // 2 + 1))
        parts[name] = {"arrays": {"UVW": wrap(rng.standard_normal((nrow, 3))), "WEIGHT": wrap(rng.random((1, nrow, nchan))),
                                  "MASK": wrap((rng.random((nrow, nchan)) > 0.2).astype(np.uint8)),
                                  "FREQ": wrap(np.linspace(1e9, 1.1e9, nchan)), "BEAM": wrap(rng.random((1, nx, ny))),
                                  "PSFHAT": wrap(psfhat)},
                       "attrs": {"wsum": np.array([3.5 + i]), "l0": 0.01 * i, "m0": -0.02}}
    return {"band3": {"arrays": {"DIRTY": wrap(rng.standard_normal((1, nx, ny)))}, "attrs": {}, "children": parts}}


def test_load_band_from_in_memory_stores(monkeypatch):
    monkeypatch.setenv("PFBHIP_PINNED_RESULTS", "0")  # no GPU here: the staging buffers are ordinary numpy arrays
    from pfb_imaging_amd import store as st
    from pfb_imaging_amd.operators.gridder import _attr, _field

    for wrap in (lambda a: a, XrLike, ZarrLike):
        raw = _store(np.random.default_rng(0), lambda a: a)
        s = _store(np.random.default_rng(0), wrap)
        dirty, parts, hess = st.load_band(s, "band3")
        assert np.array_equal(dirty, raw["band3"]["arrays"]["DIRTY"]) and dirty.dtype == np.float64
        assert len(parts) == len(hess) == 2
        for k, name in enumerate(("part0", "part1")):
            ra = raw["band3"]["children"][name]["arrays"]
            for f in st.GRID_FIELDS:
                assert np.array_equal(_field(parts[k], f), ra[f])
            assert parts[k]["MASK"].dtype == np.uint8 and parts[k]["UVW"].flags.c_contiguous
            assert _attr(parts[k], "l0") == 0.01 * k and _attr(parts[k], "m0") == -0.02
            assert hess[k]["psfhat"].dtype == np.float64 and np.allclose(hess[k]["psfhat"], np.abs(ra["PSFHAT"]))
            assert hess[k]["beam"] is parts[k]["BEAM"] and hess[k]["wsum"][0] == 3.5 + k
        if wrap is ZarrLike:  # decoded chunk by chunk straight into the staging buffer: no intermediate array
            z = s["band3"]["children"]["part0"]["arrays"]["UVW"]
            assert z.decoded_into is parts[0]["UVW"]


def test_read_pinned_looks_through_xarray_lazy_wrappers(monkeypatch):
    """An xarray variable opened from zarr keeps MemoryCachedArray(CopyOnWriteArray(LazilyIndexedArray(ZarrArrayWrapper)))
    in ``_data`` (xarray.core.indexing / xarray.backends.zarr): read_pinned decodes the zarr array behind them straight
    into the staging buffer, and stops at a value-changing (CF-decoding) wrapper, which it lets xarray materialise."""
    monkeypatch.setenv("PFBHIP_PINNED_RESULTS", "0")
    from pfb_imaging_amd import store as st

    def wrapper(name, inner, attr="array"):
        obj = type(name, (), {})()
        if attr == "get_array":
            obj.get_array = lambda: inner
        else:
            setattr(obj, attr, inner)
        return obj

    a = np.random.default_rng(1).standard_normal((23, 5))
    z = ZarrLike(a)
    data = wrapper("MemoryCachedArray", wrapper("CopyOnWriteArray", wrapper("LazilyIndexedArray",
                                                                            wrapper("ZarrArrayWrapper", z, "get_array"))))

    class Var:
        def __init__(self, d):
            self._data, self.shape, self.dtype, self.values = d, a.shape, a.dtype, a

    out = st.read_pinned(Var(data))
    assert z.decoded_into is out and np.array_equal(out, a)
    # a decoding wrapper in the chain: not looked through (its values differ from the stored ones)
    z2 = ZarrLike(a * 0 + 7.0)
    decoded = wrapper("MemoryCachedArray", wrapper("_ElementwiseFunctionArray", wrapper("LazilyIndexedArray", z2)))
    out2 = st.read_pinned(Var(decoded))
    assert z2.decoded_into is None and np.array_equal(out2, a)
    # a pending selection (shape mismatch): not looked through either
    z3 = ZarrLike(np.zeros((40, 5)))
    out3 = st.read_pinned(Var(wrapper("LazilyIndexedArray", z3)))
    assert z3.decoded_into is None and np.array_equal(out3, a)
