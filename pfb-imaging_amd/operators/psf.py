"""``psf_convolve_slice / cube / fscube`` on the GPU.

Same signatures and in-place contract as /root/reference/src/pfb_imaging/operators/psf.py:8-96:
``xout`` is filled in place and returned; ``xpad`` / ``xhat`` are the caller's scratch buffers
(accepted for signature compatibility; the transform scratch lives on the device, so they are
left untouched).  ``psfhat`` may be complex or real; an identical array is uploaded once.
"""


from ..psfconv import cached_plan, cached_psf_slot


def psf_convolve_slice(xpad, xhat, xout, psfhat, lastsize, x, nthreads=1):
    nx, ny = x.shape
    nx_psf = psfhat.shape[-2]
    plan = cached_plan(nx, ny, nx_psf, lastsize)
    slot = cached_psf_slot(plan, psfhat)
    plan.apply(x, slot, out=xout)
    return xout


def psf_convolve_cube(xpad, xhat, xout, psfhat, lastsize, x, nthreads=1):
    nband, nx, ny = x.shape
    nx_psf = psfhat.shape[-2]
    plan = cached_plan(nx, ny, nx_psf, lastsize)
    for b in range(nband):
        slot = cached_psf_slot(plan, psfhat[b])
        plan.apply(x[b], slot, out=xout[b])
    return xout


def psf_convolve_fscube(xpad, xhat, xout, psfhat, lastsize, x, nthreads=1):
    nband, ncorr, nx, ny = x.shape
    nx_psf = psfhat.shape[-2]
    plan = cached_plan(nx, ny, nx_psf, lastsize)
    for b in range(nband):
        for c in range(ncorr):
            slot = cached_psf_slot(plan, psfhat[b, c])
            plan.apply(x[b, c], slot, out=xout[b, c])
    return xout
