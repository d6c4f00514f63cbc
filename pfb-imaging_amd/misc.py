"""``ducc0.misc`` replacements the reference calls around the hot path
(/root/reference/src/pfb_imaging/operators/band_worker.py:50-52, operators/hessian.py:293-295)."""

import numpy as np

from ._lib import cint, lib


def resize_thread_pool(nthreads):
    """The compute pool is the GPU; the value is only remembered so callers need no change."""
    lib().pfbhip_resize_thread_pool(cint(int(nthreads)))


def thread_pool_size():
    return int(lib().pfbhip_thread_pool_size())


def empty_noncritical(shape, dtype):
    """ducc0 pads host arrays to dodge cache-critical strides; irrelevant for device-side work."""
    return np.empty(shape, dtype=dtype)
