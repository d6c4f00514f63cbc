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
* a plain nested mapping -- the in-memory store the tests use::

      {"band0": {"arrays": {"DIRTY": ...}, "attrs": {...},
                 "children": {"part0": {"arrays": {"UVW": ..., ...}, "attrs": {"wsum": ..., "l0": ..., "m0": ...}}}}}
"""

import numpy as np

from . import _lib

GRID_FIELDS = ("UVW", "WEIGHT", "MASK", "FREQ", "BEAM")


def _node_arrays(node):
    if isinstance(node, dict):
        return node.get("arrays", {})
    ds = getattr(node, "ds", node)            # DataTree node -> its Dataset
    return ds


def _node_attrs(node):
    if isinstance(node, dict):
        return dict(node.get("attrs", {}))
    return dict(getattr(getattr(node, "ds", node), "attrs", {}))


def _node_children(node):
    if isinstance(node, dict):
        return node.get("children", {})
    ch = getattr(node, "children", None)
    if ch is not None:
        return ch
    return {k: v for k, v in getattr(node, "groups", lambda: [])()}  # zarr group


_PURE_INDEXING_WRAPPERS = ("LazilyIndexedArray", "CopyOnWriteArray", "MemoryCachedArray", "ZarrArrayWrapper")


def _unwrap_zarr(data, shape):
    """The zarr array behind an xarray variable's ``_data``, or None.  xarray keeps a lazily indexed stack of wrappers there
    (``MemoryCachedArray(CopyOnWriteArray(LazilyIndexedArray(ZarrArrayWrapper)))``); only wrappers that index without
    changing values are looked through -- a CF-decoding wrapper (scale / offset / mask / endianness) ends the search -- and
    the array found must have the variable's full shape (no pending selection)."""
    obj = data
    for _ in range(8):
        if obj is None:
            return None
        if hasattr(obj, "get_basic_selection"):
            return obj if tuple(getattr(obj, "shape", ())) == tuple(shape) else None
        if type(obj).__name__ not in _PURE_INDEXING_WRAPPERS:
            return None
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
