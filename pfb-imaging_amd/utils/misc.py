"""Image-geometry helpers of ``pfb_imaging.utils.misc`` that sit on the hot path's boundary."""

import logging

import numpy as np

from ..fft import good_size

LIGHTSPEED = 299792458.0


def _even_good_size(n):
    """Smallest even FFT-friendly size >= n (the reference bumps odd sizes by one and re-rounds until even)."""
    n = good_size(int(n))
    while n % 2:
        n = good_size(n + 1)
    return n


def set_image_size(max_blength, max_freq, field_of_view, super_resolution_factor, cell_size=None, nx=None, ny=None,
                   psf_oversize=2.0, log=None):
    """Cell size and image / PSF dimensions of an imaging run (/root/reference/src/pfb_imaging/utils/misc.py:888-953).

    ``max_blength`` [m] and ``max_freq`` [Hz] give the Nyquist cell ``cell_n = 1 / (2 uv_max)`` with
    ``uv_max = max_blength * max_freq / c``.  Without ``cell_size`` [arcsec] the cell is ``cell_n / super_resolution_factor``;
    without ``nx`` the image covers ``field_of_view`` [deg] with an even, FFT-friendly pixel count.  The PSF grid is
    ``psf_oversize`` times the image (128 pixels when ``psf_oversize`` is falsy), also even and FFT-friendly.

    Returns ``(nx, ny, nx_psf, ny_psf, cell_n, cell_rad, cell_deg)``; odd ``nx`` / ``ny`` raise ``NotImplementedError``.
    """
    if log is None:
        log = logging.getLogger(__name__)
    cell_n = 1.0 / (2.0 * max_blength * max_freq / LIGHTSPEED)
    arcsec = np.pi / (180.0 * 3600.0)
    if cell_size is None:
        cell_rad = cell_n / super_resolution_factor
        cell_size = cell_rad / arcsec
        log.info(f"Cell size set to {cell_size} arcseconds")
    else:
        cell_rad = cell_size * arcsec
        srf = cell_n / cell_rad
        if srf < 1:
            log.info(f"Warning - requested cell size of {cell_size} arcseconds could be sub-Nyquist.")
        log.info(f"Super resolution factor = {srf}")
    cell_deg = np.rad2deg(cell_rad)
    if nx is None:
        nx = ny = _even_good_size(int(field_of_view * 3600 / cell_size))
    else:
        ny = nx if ny is None else ny
        if nx % 2 or ny % 2:
            log.error("Only even number of pixels currently supported")
            raise NotImplementedError("Only even number of pixels currently supported")
        log.info(f"Field of view is ({nx * cell_deg:.3e},{ny * cell_deg:.3e}) degrees")
    nx_psf = _even_good_size(int(psf_oversize * nx)) if psf_oversize else 128
    ny_psf = _even_good_size(int(psf_oversize * ny)) if psf_oversize else 128
    return nx, ny, nx_psf, ny_psf, cell_n, cell_rad, cell_deg
