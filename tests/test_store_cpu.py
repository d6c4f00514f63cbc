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


# ---------------------------------------------------------------------------------------------------------
# Real bytes: a zarr-v2 DIRECTORY store written by hand (JSON metadata + chunk files, per the v2 spec) in the layout
# the reference writes (core/imager.py:138-194; tests/test_deconv.py:235-290 builds the same tree with xarray) and
# read back through the built-in reader -- no zarr / xarray on either side.
# ---------------------------------------------------------------------------------------------------------

def _write_zarr_array(path, a, chunks, compressor=None, sep=".", order="C", skip=(), dims=None, fill_value=0.0):
    import json
    import os
    import zlib

    os.makedirs(path)
    meta = {"zarr_format": 2, "shape": list(a.shape), "chunks": list(chunks), "dtype": a.dtype.str, "compressor": compressor,
            "fill_value": fill_value, "order": order, "filters": None}
    if sep != ".":
        meta["dimension_separator"] = sep
    with open(os.path.join(path, ".zarray"), "w") as fh:
        json.dump(meta, fh)
    with open(os.path.join(path, ".zattrs"), "w") as fh:
        json.dump({"_ARRAY_DIMENSIONS": list(dims or [f"d{i}" for i in range(a.ndim)])}, fh)
    grid = [-(-n // c) for n, c in zip(a.shape, chunks)]
    for idx in np.ndindex(*grid):
        if idx in skip:
            continue
        block = np.full(chunks, fill_value, dtype=a.dtype)  # edge chunks are stored at full chunk size
        sl = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(idx, chunks, a.shape))
        part = a[sl]
        block[tuple(slice(0, s) for s in part.shape)] = part
        raw = block.tobytes(order=order)
        if compressor is not None:
            raw = zlib.compress(raw, compressor.get("level", 1))
        fn = os.path.join(path, sep.join(str(i) for i in idx))
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        with open(fn, "wb") as fh:
            fh.write(raw)


def _write_group(path, attrs=None):
    import json
    import os

    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, ".zgroup"), "w") as fh:
        json.dump({"zarr_format": 2}, fh)
    with open(os.path.join(path, ".zattrs"), "w") as fh:
        json.dump(attrs or {}, fh)


def test_load_band_from_a_zarr_v2_directory_store(tmp_path, monkeypatch):
    import json
    import os

    monkeypatch.setenv("PFBHIP_PINNED_RESULTS", "0")
    from pfb_imaging_amd import store as st

    rng = np.random.default_rng(5)
    nrow, nchan, nx, ny, nxp = 53, 3, 16, 12, 32
    root = str(tmp_path / "run.dt")
    _write_group(root, {"product": "I"})
    band = os.path.join(root, "band0001_time0000")
    _write_group(band, {"bandid": 1, "timeid": 0, "cell_rad": 2.5e-6})
    dirty = rng.standard_normal((1, nx, ny))
    _write_zarr_array(os.path.join(band, "DIRTY"), dirty, (1, 5, ny), dims=("corr", "x", "y"))      # chunked along x: general path
    # a coordinate xarray stores with an object-dtype filter: present in the store, never opened by the loader
    os.makedirs(os.path.join(band, "corr"))
    with open(os.path.join(band, "corr", ".zarray"), "w") as fh:
        json.dump({"zarr_format": 2, "shape": [1], "chunks": [1], "dtype": "|O", "compressor": None, "fill_value": None,
                   "order": "C", "filters": [{"id": "vlen-utf8"}]}, fh)
    raw = {}
    variants = [dict(chunks_row=20, compressor=None, sep="."), dict(chunks_row=53, compressor={"id": "zlib", "level": 1}, sep="/")]
    for k, var in enumerate(variants):
        part = os.path.join(band, f"part{k:04d}")
        _write_group(part, {"wsum": [3.5 + k], "l0": 0.01 * k, "m0": -0.02, "msid": 0, "field_name": "f0"})
        cr, comp, sep = var["chunks_row"], var["compressor"], var["sep"]
        arrs = {"UVW": rng.standard_normal((nrow, 3)), "WEIGHT": rng.random((1, nrow, nchan)),
                "MASK": (rng.random((nrow, nchan)) > 0.2).astype(np.uint8), "FREQ": np.linspace(1e9, 1.1e9, nchan),
                "BEAM": rng.random((1, nx, ny)),
                "PSFHAT": rng.standard_normal((1, nxp, nxp // 2 + 1)) + 1j * rng.standard_normal((1, nxp, nxp // 2 + 1))}
        raw[k] = arrs
        _write_zarr_array(os.path.join(part, "UVW"), arrs["UVW"], (cr, 3), comp, sep, dims=("row", "three"))     # raw: direct read
        _write_zarr_array(os.path.join(part, "WEIGHT"), arrs["WEIGHT"], (1, cr, nchan), comp, sep, dims=("corr", "row", "chan"))
        _write_zarr_array(os.path.join(part, "MASK"), arrs["MASK"], (cr, nchan), comp, sep, dims=("row", "chan"), fill_value=0)
        _write_zarr_array(os.path.join(part, "FREQ"), arrs["FREQ"], (nchan,), comp, sep, dims=("chan",))
        _write_zarr_array(os.path.join(part, "BEAM"), arrs["BEAM"], (1, 7, 5), comp, sep, order="F" if k else "C",
                          dims=("corr", "x", "y"))
        _write_zarr_array(os.path.join(part, "PSFHAT"), arrs["PSFHAT"], (1, 10, nxp // 2 + 1), comp, sep,
                          dims=("corr", "x_psf", "yo2"))
    for source in (root, "file://" + root, st.open_store(root)):
        got_dirty, parts, hess = st.load_band(source, "band0001_time0000")
        assert np.array_equal(got_dirty, dirty)
        assert len(parts) == len(hess) == 2
        for k in range(2):
            for f in st.GRID_FIELDS:
                assert np.array_equal(parts[k][f], raw[k][f]), (k, f)
                assert parts[k][f].flags.c_contiguous
            assert parts[k]["MASK"].dtype == np.uint8 and parts[k]["l0"] == 0.01 * k and parts[k]["m0"] == -0.02
            assert np.allclose(hess[k]["psfhat"], np.abs(raw[k]["PSFHAT"]), rtol=1e-15) and hess[k]["psfhat"].dtype == np.float64
            assert hess[k]["wsum"][0] == 3.5 + k and hess[k]["beam"] is parts[k]["BEAM"]
    # a chunk that was never written reads as fill_value; a short chunk file is an error, not silence
    a = rng.standard_normal((9, 4))
    _write_zarr_array(str(tmp_path / "holes"), a, (4, 4), skip={(1, 0)}, fill_value="NaN")
    got = st.DirArray(str(tmp_path / "holes")).get_basic_selection(Ellipsis)
    assert np.array_equal(got[:4], a[:4]) and np.isnan(got[4:8]).all() and np.array_equal(got[8:], a[8:])
    with open(tmp_path / "holes" / "0.0", "wb") as fh:
        fh.write(b"\0" * 16)
    try:
        st.DirArray(str(tmp_path / "holes")).get_basic_selection(Ellipsis)
        raise AssertionError("short chunk accepted")
    except IOError:
        pass
    # unsupported compressor: a clear error naming it
    _write_zarr_array(str(tmp_path / "blosc"), a, (9, 4))
    meta = json.load(open(tmp_path / "blosc" / ".zarray"))
    meta["compressor"] = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1}
    json.dump(meta, open(tmp_path / "blosc" / ".zarray", "w"))
    try:
        st.DirArray(str(tmp_path / "blosc"))
        raise AssertionError("blosc accepted without numcodecs")
    except NotImplementedError as e:
        assert "blosc" in str(e)


def test_unwrap_refuses_a_pending_same_shape_selection(monkeypatch):
    """LazilyIndexedArray with a key that keeps the shape (a reversal) must not be looked through."""
    monkeypatch.setenv("PFBHIP_PINNED_RESULTS", "0")
    from pfb_imaging_amd import store as st

    a = np.arange(30.0).reshape(6, 5)
    z = ZarrLike(a)

    class Key:
        def __init__(self, *t):
            self.tuple = t

    def lazy(key):
        obj = type("LazilyIndexedArray", (), {})()
        obj.array, obj.key = z, key
        return obj

    class Var:
        def __init__(self, d, vals):
            self._data, self.shape, self.dtype, self.values = d, vals.shape, vals.dtype, vals

    out = st.read_pinned(Var(lazy(Key(slice(None), slice(None))), a))
    assert z.decoded_into is out
    z.decoded_into = None
    rev = a[::-1]
    out = st.read_pinned(Var(lazy(Key(slice(None, None, -1), slice(None))), rev))
    assert z.decoded_into is None and np.array_equal(out, rev)
    out = st.read_pinned(Var(lazy(Key(np.array([1, 0, 2, 3, 4, 5]), slice(None))), a[[1, 0, 2, 3, 4, 5]]))
    assert z.decoded_into is None and np.array_equal(out, a[[1, 0, 2, 3, 4, 5]])
