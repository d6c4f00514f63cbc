"""Band loader for the deconvolution store: the worker-side read of a band's inputs
(/root/reference/src/pfb_imaging/operators/band_worker.py:61-106) staged for the GPU.

The reference's ``load_band`` opens the ``.dt`` zarr store as an xarray DataTree and materialises, per band node, ``DIRTY``
and, per partition child, ``UVW / WEIGHT / MASK / FREQ / BEAM`` (gridding inputs) and ``PSFHAT / BEAM / wsum`` (Hessian
inputs).  Here every array lands in a page-locked host buffer (`_lib.result_empty`) -- zarr chunks are decoded straight
into it, see read_pinned -- so the plan constructors and ``set_weights`` upload them at the PCIe rate, and ``abs(PSFHAT)`` (the form the
Hessians consume, band_worker.py:89-95) is formed in place in that buffer.

A *store* is anything that maps node names to nodes; a *node* offers its arrays by name and its children:

* an xarray DataTree (``xr.open_datatree(url, engine="zarr", chunks=None)``), as in the reference;
* a zarr group hierarchy;
* :class:`DirStore` -- this module's own reader of a zarr-v2 DIRECTORY store (the ``.dt`` the reference writes with
  ``Dataset.to_zarr(store, group="bandNNNN_timeNNNN[/partNNNN]")``, core/imager.py:138-194): ``.zgroup`` / ``.zarray`` /
  ``.zattrs`` JSON and chunk files, no zarr / xarray / numcodecs import -- chunks are read from the file straight into the
  page-locked buffer when they are stored raw (``compressor: null``), through zlib when deflated;
* a plain nested mapping -- the in-memory store the tests use::

      {"band0": {"arrays": {"DIRTY": ...}, "attrs": {...},
                 "children": {"part0": {"arrays": {"UVW": ..., ...}, "attrs": {"wsum": ..., "l0": ..., "m0": ...}}}}}
"""

import json
import os
import zlib

import numpy as np

from . import _lib

GRID_FIELDS = ("UVW", "WEIGHT", "MASK", "FREQ", "BEAM")


class DirArray:
    """One array of a zarr-v2 directory store (https://zarr-specs.readthedocs.io v2: ``.zarray`` metadata + one file per
    chunk, named by the chunk's grid indices joined with ``dimension_separator``; edge chunks are stored at full chunk
    size).  Offers what ``read_pinned`` needs of a zarr array: ``shape``, ``dtype``, ``get_basic_selection(..., out=)``."""

    def __init__(self, path):
        self.path = path
        with open(os.path.join(path, ".zarray")) as fh:
            meta = json.load(fh)
        if meta.get("zarr_format") != 2:
            raise ValueError(f"{path}: zarr_format {meta.get('zarr_format')!r} (only v2 directory stores are read here)")
        if meta.get("filters"):
            raise NotImplementedError(f"{path}: filters {meta['filters']!r} are not supported by the built-in reader")
        self.shape = tuple(int(n) for n in meta["shape"])
        self.chunks = tuple(int(n) for n in meta["chunks"]) if self.shape else ()
        self.dtype = np.dtype(meta["dtype"])
        self.order = meta.get("order", "C")
        self.fill_value = meta.get("fill_value")
        self.sep = meta.get("dimension_separator", ".")
        comp = meta.get("compressor")
        self.codec = None if comp is None else comp.get("id")
        if self.codec not in (None, "zlib", "gzip"):
            try:
                import numcodecs  # noqa: F401  (only if the environment has it; never required)

                self._numcodec = numcodecs.get_codec(comp)
            except ImportError:
                raise NotImplementedError(f"{path}: compressor {self.codec!r} needs numcodecs; the built-in reader decodes "
                                          "raw ('compressor': null), zlib and gzip chunks") from None
        attrs_path = os.path.join(path, ".zattrs")
        self.attrs = {}
        if os.path.exists(attrs_path):
            with open(attrs_path) as fh:
                self.attrs = json.load(fh)

    @property
    def ndim(self):
        return len(self.shape)

    def _fill(self):
        fv = self.fill_value
        if fv is None:
            return 0
        if isinstance(fv, str):  # "NaN" / "Infinity" / "-Infinity" (and base64 for structured types, not read here)
            return {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}[fv]
        return fv

    def _chunk_file(self, idx):
        return os.path.join(self.path, self.sep.join(str(i) for i in idx) if idx else "0")

    def _decode(self, raw):
        if self.codec is None:
            return raw
        if self.codec == "zlib":
            return zlib.decompress(raw)
        if self.codec == "gzip":
            return zlib.decompress(raw, 16 + zlib.MAX_WBITS)
        return self._numcodec.decode(raw)

    def get_basic_selection(self, sel=Ellipsis, out=None):
        if sel is not Ellipsis:
            raise NotImplementedError("DirArray reads whole arrays")
        if out is None:
            out = np.empty(self.shape, self.dtype)
        if tuple(out.shape) != self.shape or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise TypeError("out must be a C-contiguous array of the stored shape and dtype")
        if not self.shape:  # 0-d: a single chunk "0"
            raw = self._decode(open(self._chunk_file(()), "rb").read())
            out[...] = np.frombuffer(raw, self.dtype, 1)[0]
            return out
        grid = [-(-n // c) for n, c in zip(self.shape, self.chunks)]
        # raw C-order chunks that span every trailing axis are byte ranges of the destination: read the file into it
        direct = self.codec is None and self.order == "C" and self.chunks[1:] == self.shape[1:]
        flat = out.reshape(-1).view(np.uint8) if direct else None
        row_bytes = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        for idx in np.ndindex(*grid):
            lo = [i * c for i, c in zip(idx, self.chunks)]
            hi = [min(a + c, n) for a, c, n in zip(lo, self.chunks, self.shape)]
            dst = tuple(slice(a, b) for a, b in zip(lo, hi))
            fn = self._chunk_file(idx)
            if not os.path.exists(fn):  # a missing chunk is all fill_value
                out[dst] = self._fill()
                continue
            if direct:
                want = (hi[0] - lo[0]) * row_bytes
                with open(fn, "rb", buffering=0) as fh:
                    view, got = memoryview(flat[lo[0] * row_bytes:lo[0] * row_bytes + want]), 0
                    while got < want:
                        k = fh.readinto(view[got:])
                        if not k:
                            raise IOError(f"{fn}: chunk is {got} bytes, expected at least {want}")
                        got += k
                continue
            with open(fn, "rb") as fh:
                raw = self._decode(fh.read())
            chunk = np.frombuffer(raw, self.dtype).reshape(self.chunks, order=self.order)
            out[dst] = chunk[tuple(slice(0, b - a) for a, b in zip(lo, hi))]
        return out

    def __getitem__(self, key):
        return self.get_basic_selection(Ellipsis)[key]


class DirNode:
    """One group of a zarr-v2 directory store: arrays by name, attributes (``.zattrs``), child groups."""

    def __init__(self, path):
        if not os.path.exists(os.path.join(path, ".zgroup")):
            raise KeyError(f"{path} is not a zarr group (.zgroup missing)")
        self.path = path
        self._arrays = self._children = None

    def _scan(self):
        if self._arrays is None:
            self._arrays, self._children = {}, {}
            for name in sorted(os.listdir(self.path)):
                p = os.path.join(self.path, name)
                if os.path.isdir(p):
                    if os.path.exists(os.path.join(p, ".zarray")):
                        self._arrays[name] = p
                    elif os.path.exists(os.path.join(p, ".zgroup")):
                        self._children[name] = p

    @property
    def arrays(self):
        self._scan()
        return _LazyMap(self._arrays, DirArray)

    @property
    def children(self):
        self._scan()
        return _LazyMap(self._children, DirNode)

    @property
    def attrs(self):
        p = os.path.join(self.path, ".zattrs")
        if not os.path.exists(p):
            return {}
        with open(p) as fh:
            return json.load(fh)

    def __getitem__(self, name):  # "part0000" or "a/b"
        node = self
        for piece in name.strip("/").split("/"):
            node = node.children[piece]
        return node


class _LazyMap(dict):
    """name -> path, opened on access."""

    def __init__(self, paths, factory):
        super().__init__(paths)
        self._factory = factory

    def __getitem__(self, k):
        return self._factory(dict.__getitem__(self, k))


class DirStore(DirNode):
    """The root group of a zarr-v2 directory store, e.g. ``DirStore("/data/run.dt")["band0000_time0000"]``."""


def open_store(url):
    """A store for ``load_band`` from a path / ``file://`` url of a zarr-v2 directory store."""
    path = url[len("file://"):] if isinstance(url, str) and url.startswith("file://") else url
    return DirStore(os.fspath(path))


def _node_arrays(node):
    if isinstance(node, DirNode):
        return node.arrays
    if isinstance(node, dict):
        return node.get("arrays", {})
    ds = getattr(node, "ds", node)            # DataTree node -> its Dataset
    return ds


def _node_attrs(node):
    if isinstance(node, DirNode):
        return dict(node.attrs)
    if isinstance(node, dict):
        return dict(node.get("attrs", {}))
    return dict(getattr(getattr(node, "ds", node), "attrs", {}))


def _node_children(node):
    if isinstance(node, DirNode):
        return node.children
    if isinstance(node, dict):
        return node.get("children", {})
    ch = getattr(node, "children", None)
    if ch is not None:
        return ch
    return {k: v for k, v in getattr(node, "groups", lambda: [])()}  # zarr group


_PURE_INDEXING_WRAPPERS = ("LazilyIndexedArray", "CopyOnWriteArray", "MemoryCachedArray", "ZarrArrayWrapper")


def _identity_key(key, shape):
    """True when a lazy-indexing wrapper's pending ``key`` (xarray ``ExplicitIndexer``: ``.tuple`` of slices / arrays /
    ints) selects everything in order: every entry a full forward slice.  ``None`` (no key attribute) counts as identity."""
    if key is None:
        return True
    entries = getattr(key, "tuple", key)
    try:
        entries = tuple(entries)
    except TypeError:
        return False
    if len(entries) != len(shape):
        return False
    for k, n in zip(entries, shape):
        if not isinstance(k, slice) or k.indices(n) != (0, n, 1):
            return False
    return True


def _unwrap_zarr(data, shape):
    """The zarr array behind an xarray variable's ``_data``, or None.  xarray keeps a lazily indexed stack of wrappers there
    (``MemoryCachedArray(CopyOnWriteArray(LazilyIndexedArray(ZarrArrayWrapper)))``); only wrappers that index without
    changing values are looked through -- a CF-decoding wrapper (scale / offset / mask / endianness) ends the search -- and
    the array found must have the variable's full shape and no wrapper on the way may carry a pending selection
    (``LazilyIndexedArray.key`` must be the identity: a same-shape reordering would otherwise be lost)."""
    obj = data
    for _ in range(8):
        if obj is None:
            return None
        if hasattr(obj, "get_basic_selection"):
            return obj if tuple(getattr(obj, "shape", ())) == tuple(shape) else None
        if type(obj).__name__ not in _PURE_INDEXING_WRAPPERS:
            return None
        if not _identity_key(getattr(obj, "key", None), shape):
            return None  # a pending lazy selection (even one that keeps the shape, e.g. a reversal): let xarray apply it
        nxt = getattr(obj, "array", None)
        if nxt is None:
            get = getattr(obj, "get_array", None)
            if get is not None:
                try:
                    nxt = get()
                except Exception:
                    nxt = None
            if nxt is None:
                nxt = getattr(obj, "_array", None)
        obj = nxt
    return None


def read_pinned(src, dtype=None):
    """Decode ``src`` (zarr array, xarray variable / DataArray, numpy array, anything with ``shape`` / ``dtype`` and
    ``__getitem__``) into a page-locked buffer and return it as a numpy array.  zarr arrays -- given directly or found behind
    an xarray variable's lazy-indexing wrappers (see _unwrap_zarr) -- decode chunk by chunk directly into the buffer
    (``get_basic_selection(out=...)``); every other source (in-memory xarray variables, CF-decoded ones, numpy arrays) is
    materialised by its owner and copied once."""
    var = getattr(src, "variable", src)
    shape = tuple(src.shape)
    zarr_like = src if hasattr(src, "get_basic_selection") else _unwrap_zarr(getattr(var, "_data", None), shape)
    dt = np.dtype(src.dtype if dtype is None else dtype)
    out = _lib.result_empty(shape, dt)
    if zarr_like is not None and np.dtype(zarr_like.dtype) == dt:
        try:
            zarr_like.get_basic_selection(Ellipsis, out=out)
            return out
        except TypeError:
            pass
    vals = getattr(src, "values", None)
    out[...] = src[...] if vals is None else vals
    return out


def load_band(store, node_name):
    """``(dirty, parts, hess_parts)`` of one band, every array page-locked:

    * ``dirty``      ``(corr, nx, ny)`` float64
    * ``parts``      list of dicts ``UVW, WEIGHT, MASK, FREQ, BEAM`` + the partition's attributes (``l0``, ``m0``, ...),
                     the form ``operators.gridder.PartitionResidual`` / ``_BandWorkerImpl.set_band`` take
    * ``hess_parts`` list of dicts ``psfhat = |PSFHAT|`` (real), ``beam``, ``wsum`` for ``HessianTree``
    """
    if isinstance(store, (str, os.PathLike)):
        store = open_store(store)
    band = store[node_name]
    dirty = read_pinned(_node_arrays(band)["DIRTY"], np.float64)
    parts, hess_parts = [], []
    children = _node_children(band)
    for cname in sorted(children):
        child = children[cname]
        arrs, attrs = _node_arrays(child), _node_attrs(child)
        part = {"UVW": read_pinned(arrs["UVW"], np.float64), "WEIGHT": read_pinned(arrs["WEIGHT"], np.float64),
                "MASK": read_pinned(arrs["MASK"], np.uint8), "FREQ": read_pinned(arrs["FREQ"], np.float64),
                "BEAM": read_pinned(arrs["BEAM"], np.float64), "attrs": attrs}
        part.update({k: v for k, v in attrs.items() if k not in part})
        parts.append(part)
        psfhat = read_pinned(arrs["PSFHAT"])                    # complex, as stored
        absbuf = _lib.result_empty(psfhat.shape, np.float64)
        np.abs(psfhat, out=absbuf)                              # the Hessians take the magnitude (band_worker.py:89-95)
        del psfhat
        hess_parts.append({"psfhat": absbuf, "beam": part["BEAM"], "wsum": np.asarray(attrs["wsum"], dtype=np.float64)})
    return dirty, parts, hess_parts
