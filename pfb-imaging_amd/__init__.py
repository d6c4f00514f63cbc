"""pfb-imaging_amd -- MI355X-native measurement operator for pfb-imaging.

Drop-in for the hot path of ratt-ru/pfb-imaging: the w-stacking gridder / degridder and the
2-D FFTs that form the Hessian ``R^H W R`` and its PSF-convolution approximation.  Host code
is Python + numpy calling hand-written HIP (gfx950) through the C-ABI of ``include/pfbhip.h``
(``libpfbhip.so``, loaded with ctypes).  No CPU fallback exists: every operator raises if the
library cannot be loaded or no GPU is present.

Layout (mirrors the reference modules it replaces; citations are under /root/reference):
    wgridder            ducc0.wgridder.experimental.vis2dirty / dirty2vis   (operators/gridder.py:9-11)
    fft                 ducc0.fft.r2c / c2r / good_size                      (operators/psf.py:5)
    misc                ducc0.misc.resize_thread_pool / thread_pool_size / empty_noncritical
    operators.gridder   wgridder_conventions, vis2im, im2vis, residual_from_partitions, grid_partition
    operators.hessian   hessian_slice, hessian_psf_slice, hess_direct(_slice), HessPSF, HessianTree, HessTreeRay
    operators.psf       psf_convolve_slice / cube / fscube
    operators.band_worker  BandWorkerPool (one band per GPU)
    utils.weighting     _compute_counts, counts_to_weights
    parallel            band -> GPU map and the RCCL band reduce
"""

__version__ = "0.1.0"

from ._lib import device_count, last_error, lib  # noqa: F401
