"""CPU oracle for the pfb-imaging measurement-operator hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``pfb-imaging_amd/`` (the product) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the timed CPU
baseline.

Parity pin (round 4): PINNED to reference-run fixtures wherever the reference's code can run in the build container --
``tests/golden/make_ref_pins.py`` executes the reference's own undecorated numpy / scipy functions (``explicit_wdegridder``,
``explicit_degridder``, ``wgridder_conventions``, ``prox_21m``, ``dual_update``, ``filter_extreme_counts``, ``box_sum_counts``,
``taperf``, ``power_method``, the legacy ``primal_dual``) and ``tests/test_oracle.py::test_oracle_vs_reference_run_fixtures`` holds
``dft``, ``psi``, ``weighting`` and ``fftconv`` to what they returned.  UNPINNED, and it cannot be otherwise here: the
floating-point output of the third-party wheel ``ducc0`` (locked 0.41.0, /root/reference/uv.lock:1119-1120; absent from
/root/reference and not installable) -- ``wgridder`` is held to the (pinned) direct-DFT definition at the requested epsilon,
ducc0's documented contract --, the numba kernels (``_compute_counts``, ``counts_to_weights``, the DWT) and PyWavelets' filter
tables (``psi``'s dictionary: perfect reconstruction, adjointness and hand-worked Haar values only).

Modules
    dft        exact measurement equation (C, OpenMP), pixel / visibility subsets
    wgridder   ES-kernel w-stacking restatement (C scatter/gather + scipy FFT): ES-kernel planes, polynomial planes, and the
               one-plane scheme with differentiated gridding kernels (wmode 0 / 1 / 2)
    psi        wavelet dictionary, l21 prox / dual update, positivity
    fftconv    numpy restatement of psf_convolve_* / HessPSF / HessianTree
    weighting  uv-cell index map, counts, Briggs weights
"""
