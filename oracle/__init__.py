"""CPU oracle for the pfb-imaging measurement-operator hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``pfb-imaging_amd/`` (the product) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the timed CPU
baseline.

Parity pin: the reference's arithmetic for this path is in the third-party
wheel ``ducc0`` (locked 0.41.0, /root/reference/uv.lock:1119-1120), which is
absent from /root/reference and not installable here.  Against ducc0's exact
floating-point output this oracle is **parity unpinned**; it is pinned against
the direct-DFT definition and analytic identities used by the reference's own
tests (/root/reference/tests/test_hessian_approx.py:44-67,128-231).

Modules
    dft        exact measurement equation (C, OpenMP), pixel / visibility subsets
    wgridder   ES-kernel w-stacking restatement (C scatter/gather + scipy FFT)
    fftconv    numpy restatement of psf_convolve_* / HessPSF / HessianTree
    weighting  uv-cell index map, counts, Briggs weights
"""
