/*
 * pfbhip.h -- C-ABI of libpfbhip.so, the MI355X (gfx950) measurement operator
 * for pfb-imaging.
 *
 * Every entry point replaces one of the third-party native calls the reference
 * makes on its hot path (the reference is 100 % Python; its FFI for this path
 * is the ducc0 / numba call boundary).  Citations are file:line under
 * /root/reference.  All functions return 0 on success and a non-zero status
 * on failure; pfbhip_last_error() then returns a thread-local message.  No
 * C++ exception crosses this boundary.  Pointers named *_host are caller-owned
 * host memory (numpy arrays, possibly read-only); pointers named *_dev are
 * device memory obtained from pfbhip_malloc.  Complex arrays are interleaved
 * (re, im) doubles.  All arrays are C-contiguous.
 *
 * Threading: a handle is single-caller (like the reference's operators, which
 * share scratch: src/pfb_imaging/operators/hessian.py:481-485); different
 * handles may be used from different host threads.  Each handle owns a HIP
 * stream; calls are synchronous with respect to the host unless stated.
 */
#ifndef PFBHIP_H
#define PFBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFBHIP_OK 0
#define PFBHIP_ERR_INVALID 1  /* bad argument / shape (-> ValueError)   */
#define PFBHIP_ERR_RUNTIME 2  /* HIP / rocFFT / RCCL failure (-> RuntimeError) */

/* ---- runtime ------------------------------------------------------- */
const char *pfbhip_last_error(void);
int pfbhip_device_count(int *count);
int pfbhip_set_device(int device);
int pfbhip_get_device(int *device);
int pfbhip_device_name(char *buf, size_t buflen);
int pfbhip_mem_info(size_t *free_bytes, size_t *total_bytes);
/* Device blocks released by destroyed handles are kept for the next handle of the same sizes (hipMalloc costs ~45 ms per GB
 * here; a plan is created per band and major cycle, as the reference creates its ducc0 plans per call,
 * src/pfb_imaging/operators/gridder.py:590-613).  Reports the bytes currently cached (before the flush) and, with flush != 0,
 * returns them to the driver.  PFBHIP_DEVCACHE_MB bounds the cache (default 131072, 0 = off); it empties itself when an
 * allocation fails. */
int pfbhip_device_cache(size_t *cached_bytes, int flush);
/* replaces ducc0.misc.resize_thread_pool / thread_pool_size (src/pfb_imaging/operators/band_worker.py:50-52):
 * the "pool" is the GPU; kept so callers need no change. */
int pfbhip_resize_thread_pool(int nthreads);
int pfbhip_thread_pool_size(void);
/* replaces ducc0.fft.good_size (src/pfb_imaging/utils/misc.py:921-951):
 * smallest 2-3-5-7-11-smooth (real=0) or 2-3-5-smooth (real=1) integer >= n. */
int64_t pfbhip_good_size(int64_t n, int real);

/* ---- device memory (for device-resident callers: on-device CG, bench) ---- */
/* 64-bit content hash of a whole host buffer (multi-threaded, memory-bandwidth bound).  The Python layer keys its
 * plan caches on it: the reference's ducc0 calls are stateless (operators/gridder.py:590-613), so a cached plan is
 * reused only for byte-identical uvw / freq / mask / weights / psfhat. */
uint64_t pfbhip_hash64(const void *data_host, size_t nbytes);
/* Page-locked host buffers for result arrays (device-to-host copies into pinned memory run at the PCIe rate; into
 * pageable memory the runtime stages them at a fraction of it). */
int pfbhip_host_alloc(void **ptr_host, size_t bytes);
int pfbhip_host_free(void *ptr_host);
int pfbhip_malloc(void **ptr_dev, size_t bytes);
int pfbhip_free(void *ptr_dev);
int pfbhip_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int pfbhip_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int pfbhip_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes);
int pfbhip_memset(void *dst_dev, int value, size_t bytes);
int pfbhip_synchronize(void);

/* ---- w-stacking gridder / degridder --------------------------------- */
/*
 * Replaces ducc0.wgridder.experimental.vis2dirty / dirty2vis as called at
 *   src/pfb_imaging/operators/hessian.py:50-89      (hessian_slice)
 *   src/pfb_imaging/operators/gridder.py:78,128,590-613,633-656,852-910,972-1016,1067-1117
 * A handle binds what is constant over a run -- geometry, uvw, freq, mask --
 * (the reference pins exactly these per band: operators/band_worker.py:61-106)
 * and holds the tile-sorted visibility index, kernel choice, correction image,
 * rocFFT plans and device scratch.
 */
typedef struct pfbhip_gridder pfbhip_gridder;

typedef struct pfbhip_gridder_params {
    int64_t nrow, nchan;
    int64_t nx, ny;               /* npix_x, npix_y                          */
    double pixsize_x, pixsize_y;  /* radians                                 */
    double center_x, center_y;    /* as ducc0: x0 = -l0, y0 = -m0 on the pfb path (gridder.py:23-34) */
    double epsilon;
    double sigma_min, sigma_max;  /* oversampling bounds (gridder.py:609-610) */
    int32_t flip_u, flip_v, flip_w;
    int32_t do_wgridding;
    int32_t divide_by_n;
    int32_t verbosity;
    /* 0 / 0.0 = automatic.  Tests pin the kernel row / w-plane scheme with these. */
    int32_t force_W;
    int32_t force_wmode; /* 0 auto, 1 ES-kernel w-planes, 2 polynomial (Chebyshev-node) w-planes, 3 one plane (differentiated kernels) */
    double force_sigma;
} pfbhip_gridder_params;

typedef struct pfbhip_gridder_info {
    int64_t nu, nv;        /* oversampled uv-grid                             */
    int64_t nplanes;       /* w-planes                                        */
    int64_t nactive;       /* unmasked visibilities                           */
    int64_t ntiles, nwork; /* uv tiles, (tile, chunk) work items              */
    int32_t W, tile;       /* kernel support, tile edge (cells)               */
    double beta, sigma;    /* ES kernel exp(beta (sqrt(1-x^2) - 1)), design oversampling */
    double wmin, dw;       /* plane p sits at w = wmin + p dw (wavelengths)   */
    double nshift, lshift, mshift;
    double kernel_eps;     /* tabulated 1-D image-edge error of the chosen row  */
    /* w-plane scheme.  wmode 0: W-wide ES kernel over equispaced planes wmin + p dw (classic
     * w-stacking; each visibility touches W planes).  wmode 1: the w range [wcenter - whalf,
     * wcenter + whalf] is interpolated by a degree-(nplanes-1) polynomial through Chebyshev
     * nodes (each visibility touches all planes with Lagrange weights; no w-correction in the
     * image).  wmode 2: see nderiv below.  The cheapest admissible scheme is chosen per plan. */
    int32_t wmode;
    int32_t occ_rows;      /* rows of the uv-plane that hold visibilities (only these are cleared / transformed) */
    double wcenter, whalf;
    size_t device_bytes;   /* device memory held by the handle                */
    /* plane transform: bit 0 = hand-written row FFT on the first axis (else rocFFT); bit 1 = second axis fused
     * with pad / crop / w-screen (sizes {1,3,5,7,9,15} x 2^a in 1024..16384); bit 2 = second axis on the hand-written
     * FFT with separate pad / crop kernels (the doubled sizes 20480, 24576); neither bit 1 nor 2: rocFFT; bit 3 = first
     * axis with the crop / pad + transpose folded into the transform (no separate transpose kernels) */
    int32_t fft_mode;
    int32_t screen_poly;   /* coefficients of the n-1 polynomial of the fused w-screen (0: closed form) */
    /* scatter kernel: 2 = record-driven register-footprint form (k_grid_rec), 1 = register-footprint form (k_grid_blk:
     * visibilities sorted by tile and 4 x 4-cell block, LDS atomics only when the block changes), 0 = diagonal-walk form
     * (k_grid_mp: LDS atomics per tap) */
    int32_t scatter_mode;
    /* launches of the scatter per pass over the planes: 4 = one per tile colour (tile-row / tile-column parity), whose
     * tile flush is a plain read-add-write because no two tiles of a colour overlap; 1 = one launch, atomic flush */
    int32_t scatter_launches;
    /* cells of one uv-plane the scatter / gather can touch (tiles with visibilities + halo); the first-axis transforms and
     * the Hessian's plane clear move only these of the occupied rows (0: not computed, every cell of the occupied rows) */
    int64_t used_cells;
    /* w-screen of the fused second axis, passes (launches of <= 4 planes) per form: composite cos / sin polynomials of the whole
     * phase (small w x field), separable form (per-plane column table x row factor x residual polynomials), and the rest
     * (n - 1 polynomial + sincos per pixel and plane) */
    int32_t screen_composite, screen_separable;
    /* wmode 2 (round 4): ONE uv-plane at wcenter; the rest of the w-term, exp(-2 pi i (w - wcenter)(n - 1 + nshift)), is
     * interpolated in s = l^2 + m^2 through nderiv Chebyshev nodes of [0, smax] and carried by the gridding kernel of each
     * visibility: multiplication by s^k in the image = the 2k-th derivatives of the kernel on the uv-plane (nderiv kernel
     * functions per axis; phase centre on axis only).  Replaces the nderiv planes wmode 1 would use. */
    int32_t nderiv;
    double smax;
    /* Hessian applies replayed from a captured hipGraph so far (opt-in: PFBHIP_GRAPH=1; measured at parity with eager
     * launches even at C1's size, see gridder.hip) */
    int64_t graph_replays;
    /* edge (cells) of the blocks the register-footprint scatters anchor their frame on: 4 (the tile sort's 4 x 4-cell blocks), or 2
     * for the one-plane scatter at W = 14, 15 (16 x 16-cell frame on 4 x 16 lanes; the sort key then carries the 2 x 2 block) */
    int32_t scatter_block;
    int32_t reserved0;
} pfbhip_gridder_info;

int pfbhip_gridder_create(const pfbhip_gridder_params *params, const double *uvw_host /* (nrow,3) */,
                          const double *freq_host /* (nchan) */, const uint8_t *mask_host /* (nrow,nchan) or NULL */,
                          pfbhip_gridder **out);
int pfbhip_gridder_destroy(pfbhip_gridder *g);
int pfbhip_gridder_get_info(const pfbhip_gridder *g, pfbhip_gridder_info *info);

/* w (wavelengths) of each of the nplanes planes. */
int pfbhip_gridder_get_planes(const pfbhip_gridder *g, double *w_host /* [nplanes] */);

/* The bit-exact uv-cell / tile / plane map, for parity tests: per visibility
 * (row-major (nrow,nchan)) first-tap indices iu0, iv0, first plane p0, the
 * Hermitian-fold flag, and `order`, the nactive visibility indices in tile-sorted
 * order.  Any output pointer may be NULL. */
int pfbhip_gridder_get_binmap(pfbhip_gridder *g, int32_t *iu0_host, int32_t *iv0_host, int32_t *p0_host,
                              uint8_t *flip_host, int64_t *order_host);

/* vis2dirty: dirty (nx,ny) = R^H (wgt * mask * vis).  wgt_host may be NULL (ones). */
int pfbhip_gridder_vis2dirty(pfbhip_gridder *g, const double *vis_host /* (nrow,nchan,2) */,
                             const double *wgt_host /* (nrow,nchan) or NULL */, double *dirty_host /* (nx,ny) */);
/* dirty2vis: vis (nrow,nchan) = wgt * mask * R dirty. */
int pfbhip_gridder_dirty2vis(pfbhip_gridder *g, const double *dirty_host, const double *wgt_host,
                             double *vis_host /* (nrow,nchan,2) */);
/* One pre-FFT w-plane of the uv-grid (nu,nv,2), for intermediate parity tests. */
int pfbhip_gridder_grid_plane(pfbhip_gridder *g, const double *vis_host, const double *wgt_host, int64_t plane,
                              double *grid_host);

/*
 * Fused exact Hessian  out = beam * R^H W R (beam * x) / wsum + eta * x
 * (src/pfb_imaging/operators/hessian.py:15-100).  Weights are bound once with
 * set_weights (they are constant across CG iterations, opt/pcg.py:519-538);
 * model visibilities never leave the device.  beam may be NULL; wsum <= 0
 * means "do not normalise"; eta == 0 skips the Tikhonov term.
 */
int pfbhip_gridder_set_weights(pfbhip_gridder *g, const double *wgt_host /* (nrow,nchan) or NULL */);
int pfbhip_gridder_hessian(pfbhip_gridder *g, const double *x_host, const double *beam_host, double eta, double wsum,
                           double *out_host);
int pfbhip_gridder_hessian_dev(pfbhip_gridder *g, const double *x_dev, const double *beam_dev, double eta,
                               double wsum, double *out_dev);
/* The exact residual of one partition with every image resident in HBM: out = acc - R^H W R (beam * model)
 * (src/pfb_imaging/operators/gridder.py:962-1016: dirty2vis of beam * model, vis2dirty with the imaging weights, subtracted
 * from the dirty image; band_worker.py:167-182 calls it per band).  Weights as bound by set_weights; beam may be NULL; out may
 * be acc (chains over the partitions of a band) but not model. */
int pfbhip_gridder_residual_dev(pfbhip_gridder *g, const double *model_dev, const double *beam_dev, const double *acc_dev,
                                double *out_dev);
/* Device-resident single directions (bench / on-device solvers).  vis_sorted_dev
 * holds nactive complex values in the handle's tile-sorted order. */
/* vis2dirty with the image left in HBM (row-sharded single band: the partial images are summed over xGMI before one download) */
int pfbhip_gridder_vis2dirty_dev(pfbhip_gridder *g, const double *vis_host, const double *wgt_host, double *dirty_dev);
int pfbhip_gridder_degrid_dev(pfbhip_gridder *g, const double *dirty_dev, double *vis_sorted_dev);
int pfbhip_gridder_grid_dev(pfbhip_gridder *g, const double *vis_sorted_dev, double *dirty_dev);

/* Per-stage device timing (HIP events on the handle's stream).  Stages:
 * 0 grid (scatter kernel)  1 degrid (gather kernel)  2 fft_rows (plain row-FFT passes: first axis, and the
 * second axis on the rocFFT fallback)  3 pad (B -> A transpose; + pad/w-screen kernel on the fallback)
 * 4 crop (A -> B transpose; + crop/w-screen kernel on the fallback)  5 other (clears, image transposes,
 * scaling)  6 fft_crop (fused second-axis inverse FFT + crop + w-screen)  7 pad_fft (fused pad + w-screen +
 * second-axis forward FFT).
 * ms[s] = accumulated milliseconds, calls[s] = timed regions, since the last reset. */
#define PFBHIP_NSTAGES 8
int pfbhip_gridder_profile(pfbhip_gridder *g, int enable);
int pfbhip_gridder_profile_get(pfbhip_gridder *g, double *ms /* [PFBHIP_NSTAGES] */, int64_t *calls, int reset);
/* Diagnostic (plans created with PFBHIP_STAMP=1 in the environment): in-kernel phase stamps (shader cycles) of the last
 * record-scatter pass, 8 words per colour work item: prologue, wave 0 visibility loop, wave 0 barrier wait, tile flush,
 * visibilities, last wave's loop, last wave's wait, tile.  *nitems = 0 when stamping is off. */
int pfbhip_gridder_debug_stamps(pfbhip_gridder *g, unsigned long long *out_host, int64_t capacity_items, int64_t *nitems);

/* ---- FFT (replaces ducc0.fft.r2c / c2r) ---------------------------- */
/* r2c(forward=True, inorm=0) and c2r(forward=False, inorm=2, lastsize) over the last two axes of a
 * (nbatch, n0, n1) real / (nbatch, n0, n1/2+1) complex array
 * (src/pfb_imaging/operators/psf.py:20-32, operators/fft.py:10,39, operators/gridder.py:659,912). */
int pfbhip_r2c_2d(const double *in_host, int64_t nbatch, int64_t n0, int64_t n1, double *out_host);
/* r2c of ifftshift(x) over both axes, EVEN lengths only: fft2d / fft_cube of the reference (operators/fft.py:9-40, PSF -> PSFHAT);
 * the shift is the checkerboard sign (-1)^(k0 + k1) on the spectrum, applied on the device. */
int pfbhip_r2c_2d_centred(const double *in_host, int64_t nbatch, int64_t n0, int64_t n1, double *out_host);
int pfbhip_c2r_2d(const double *in_host, int64_t nbatch, int64_t n0, int64_t n1 /* lastsize */, double *out_host);

/* Hand-written batched row FFT (the second-axis pass of the plane transform), exposed for tests and
 * benchmarks: in-place transform of (nrows, n) complex doubles, n = m 2^a (m in 1,3,5,7,9,15), 1024 <= n <= 16384, or 20480 / 24576 / 32768.
 * ms_out (may be NULL) receives the device time per transform when reps > 1. */
int pfbhip_debug_rowfft(double *data_host, int64_t n, int64_t nrows, int inverse, int reps, double *ms_out);

/* ---- PSF-convolution operator family -------------------------------- */
/*
 * One plan serves psf_convolve_slice/cube/fscube (operators/psf.py:8-96),
 * hessian_psf_slice / hess_direct_slice (operators/hessian.py:103-248),
 * HessPSF.dot (:313-349) and HessianTree.dot (:487-518):
 *     out[b] = post[b] * crop( irfft2( rfft2( pad(pre[b] * x[b]) ) * f(psfhat[b]) ) ) * scale + eta[b] * x[b]
 * with real or complex psfhat resident on the device.
 */
typedef struct pfbhip_psfconv pfbhip_psfconv;
int pfbhip_psfconv_create(int64_t nx, int64_t ny, int64_t nx_psf, int64_t ny_psf, pfbhip_psfconv **out);
int pfbhip_psfconv_destroy(pfbhip_psfconv *p);
/* Bind slot `slot` (0 <= slot < nslots, grown on demand) to a Fourier-domain PSF:
 * is_complex = 0: (nx_psf, ny_psf/2+1) doubles; 1: interleaved complex. */
int pfbhip_psfconv_set_psfhat(pfbhip_psfconv *p, int64_t slot, const double *psfhat_host, int is_complex);
/* Bind slot `slot` to an image-plane multiplier (beam / taper), (nx,ny); NULL unbinds. */
int pfbhip_psfconv_set_beam(pfbhip_psfconv *p, int64_t slot, const double *beam_host);
/*
 * mode 0: multiply by psfhat[psf_slot]
 * mode 1: multiply by (psfhat + shift)          (hess_direct, forward)
 * mode 2: divide   by (psfhat + shift)          (hess_direct, backward)
 * beam_slot < 0: no image-plane multiplier.  accumulate != 0: out += result.
 */
int pfbhip_psfconv_apply(pfbhip_psfconv *p, const double *x_host, int64_t psf_slot, int64_t beam_slot, int mode,
                         double shift, double scale, double eta, int accumulate, double *out_host);
int pfbhip_psfconv_apply_dev(pfbhip_psfconv *p, const double *x_dev, int64_t psf_slot, int64_t beam_slot, int mode,
                             double shift, double scale, double eta, int accumulate, double *out_dev);
/* HessPSF.idot's direct estimate (operators/hessian.py:369-400): mode 2 with the taper in taper_slot, then -- beam_slot >= 0 --
 * x /= beam^2 where x > 0 and beam > min_beam, on the device (the reference divides on the host).  raw_host (may be NULL)
 * receives the estimate before the beam division, out_host after it. */
int pfbhip_psfconv_direct(pfbhip_psfconv *p, const double *x_host, int64_t psf_slot, int64_t taper_slot, double shift,
                          int64_t beam_slot, double min_beam, double *raw_host, double *out_host);

/* Single-precision host arrays: the reference's precision="single" (vis2im / im2vis, operators/gridder.py:58-100:
 * complex64 visibilities, float32 weights and images) with double-precision accumulation (ducc0's
 * double_precision_accumulation=True, the reference's default, core/grid.py:50-52).  Values cross PCIe as float / complex64
 * -- half the bytes of the double entry points, which is what bounds the host-array surface -- and are widened / narrowed on
 * the device; every device buffer and every sum stays double. */
int pfbhip_gridder_vis2dirty_sp(pfbhip_gridder *g, const float *vis_host /* (nrow,nchan,2) */,
                                const float *wgt_host /* (nrow,nchan) or NULL */, float *dirty_host /* (nx,ny) */);
int pfbhip_gridder_dirty2vis_sp(pfbhip_gridder *g, const float *dirty_host, const float *wgt_host, float *vis_host);
int pfbhip_gridder_set_weights_sp(pfbhip_gridder *g, const float *wgt_host);
int pfbhip_gridder_hessian_sp(pfbhip_gridder *g, const float *x_host, const float *beam_host /* or NULL */, double eta,
                              double wsum, float *out_host);

/* ---- on-device conjugate gradients ------------------------------------- */
/*
 * Whole-solve entry points: every CG vector stays in HBM, one host call per solve
 * (replaces the per-iteration host loop of pcg_numba, src/pfb_imaging/opt/pcg.py:88-199,
 * as used by HessTreeRay.cg -> band_worker.py:124-140 and pcg_dds, opt/pcg.py:519-540).
 * Same stopping rule: eps = ||x - xp|| / ||x|| <= tol and k >= minit, or k == maxit, or 5 stalls.
 * x_host holds x0 on entry when has_x0 != 0 (else zeros are used) and the solution on exit.
 */
typedef struct pfbhip_cg_info {
    int32_t iters;
    int32_t status; /* 0 converged, 1 maxit, 2 stalled, 3 zero initial residual */
    double eps;     /* last ||x - xp|| / ||x|| */
    double phi;     /* (r.r) / (r0.r0) */
} pfbhip_cg_info;
/* A = (scale) * sum_k beam[beam_slots[k]] * PSF[psf_slots[k]] (*) (beam * .) + eta * I   (HessianTree / HessPSF band) */
int pfbhip_psfconv_cg(pfbhip_psfconv *p, int64_t nparts, const int64_t *psf_slots, const int64_t *beam_slots,
                      double scale, double eta, const double *rhs_host, double *x_host, int has_x0, double tol,
                      int maxit, int minit, pfbhip_cg_info *info);
/* A = beam * R^H W R (beam * .) / wsum + eta * I   (exact Hessian, weights bound by set_weights) */
int pfbhip_gridder_cg(pfbhip_gridder *g, const double *beam_host, double eta, double wsum, const double *rhs_host,
                      double *x_host, int has_x0, double tol, int maxit, int minit, pfbhip_cg_info *info);
/* The same solve with rhs / x / beam resident in HBM (bench, band workers that keep their images on the device). */
int pfbhip_gridder_cg_dev(pfbhip_gridder *g, const double *beam_dev, double eta, double wsum, const double *rhs_dev,
                          double *x_dev, int has_x0, double tol, int maxit, int minit, pfbhip_cg_info *info);

/* ---- power method on the device (spectral norm of a Hessian) ------------
 * Replaces power_method / power_method_numba (src/pfb_imaging/opt/power_method.py:40-148) as called on
 * precond.dot / hess.dot for `hess_norm` (core/sara.py:200-209, deconv/pfb.py:118-126) -- and the per-actor
 * form power_method_dist (:178-208) when a communicator is passed.  b_host: start vector in (any non-zero norm; the
 * reference draws randn), normalised last iterate out.  Stops when |beta - beta_prev| / beta_prev <= tol or after
 * maxit iterations (status 1). */
typedef struct pfbhip_pm_info {
    int32_t iters;
    int32_t status; /* 0 converged, 1 maxit reached */
    double eps;
    double beta; /* Rayleigh quotient (bp . A bp) / (bp . bp) of the last iteration */
} pfbhip_pm_info;
typedef struct pfbhip_comm pfbhip_comm; /* RCCL communicator, see below */
/* A = blockdiag_b( scale[b] sum_p beam_p (PSF_p * (beam_p .)) + eta[b] I ) on a cube (nband, nx, ny); band b owns
 * nparts[b] consecutive entries of psf_slots / beam_slots of its plan pcs[b] (as in pfbhip_primal_dual).
 * comm != NULL: the arrays hold this rank's LOCAL bands, the three dots are all-reduced. */
int pfbhip_psfconv_power_method(pfbhip_psfconv *const *pcs, int64_t nband, const int64_t *nparts, const int64_t *psf_slots,
                                const int64_t *beam_slots, const double *scale, const double *eta, double *b_host, double tol,
                                int maxit, pfbhip_comm *comm, pfbhip_pm_info *info);
/* A = beam * R^H W R (beam * .) / wsum + eta * I (exact Hessian, weights bound by set_weights), image (nx, ny) */
int pfbhip_gridder_power_method(pfbhip_gridder *g, const double *beam_host, double eta, double wsum, double *b_host,
                                double tol, int maxit, pfbhip_pm_info *info);

/* ---- uv-cell counts / Briggs weights (utils/weighting.py:81-208) ------ */
/* cell index (u_idx*ny+v_idx, -1 if masked / out of bounds): the bit-exact index map. */
int pfbhip_uvcell_index(const double *uvw_host, const double *freq_host, const uint8_t *mask_host, int64_t nrow,
                        int64_t nchan, int64_t nx, int64_t ny, double cell_x, double cell_y, double usign,
                        double vsign, int64_t *cell_host);
/* _compute_counts: counts (ncorr,nx,ny) += wgt (ncorr,nrow,nchan) scattered by the index map. */
int pfbhip_compute_counts(const double *uvw_host, const double *freq_host, const uint8_t *mask_host,
                          const double *wgt_host, int64_t ncorr, int64_t nrow, int64_t nchan, int64_t nx, int64_t ny,
                          double cell_x, double cell_y, double usign, double vsign, double *counts_host);
/* gather-divide half of counts_to_weights: wgt /= counts[cell] where counts > 0. */
int pfbhip_counts_divide(const double *uvw_host, const double *freq_host, const uint8_t *mask_host,
                         const double *counts_host, int64_t ncorr, int64_t nrow, int64_t nchan, int64_t nx,
                         int64_t ny, double cell_x, double cell_y, double usign, double vsign, double *wgt_host);

/* box_sum_counts (utils/weighting.py:229-254): (2 s + 1)^2 box sum with zero padding, per correlation plane. */
int pfbhip_box_sum_counts(const double *counts_host, int64_t ncorr, int64_t nx, int64_t ny, int64_t npix_super,
                          double *out_host);
/* filter_extreme_counts (utils/weighting.py:212-226): positive counts below median(positive)/level are
 * raised to that value, in place; the median is returned through median_out (may be NULL). */
int pfbhip_filter_extreme_counts(double *counts_host, int64_t n, double level, double *median_out);
/* The imaging-weight chain of image_data_products (operators/gridder.py:534-576) as ONE device pipeline:
 * _compute_counts on the (nx, ny) padded uv-grid -> filter_extreme_counts(level; <= 0 skips) -> box_sum_counts(npix_super)
 * -> Briggs scaling counts * (5 * 10^-robust)^2 * sum(c)/sum(c^2) + 1 (robust > -2) -> weight /= counts[cell].
 * wgt_host (ncorr, nrow, nchan) is updated in place (left untouched when every count is zero); the final counts
 * come back through counts_out_host (ncorr, nx, ny) when it is not NULL.  One upload of uvw / freq / mask / weights. */
int pfbhip_imaging_weights(const double *uvw_host, const double *freq_host, const uint8_t *mask_host, double *wgt_host,
                           int64_t ncorr, int64_t nrow, int64_t nchan, int64_t nx, int64_t ny, double cell_x, double cell_y,
                           double usign, double vsign, double robust, double filter_level, int64_t npix_super,
                           double *counts_out_host);

/* ---- wavelet dictionary Psi and the l21 / positivity proxes ---------------- */
/*
 * One band's dictionary: replaces PsiBandNocopyt (src/pfb_imaging/operators/psi.py:415-535; multi-level 2-D
 * DWT of wavelets/wavelets.py:216-343 on the 1-D kernels of wavelets/convolutions.py:5-327, zero-padding
 * mode, packed x-first layout of psi.py:23-142).  bases[i]: 0 = "self" (identity), N = "dbN" (1..8).
 * x is (nx, ny), alpha is (nbasis, nxmax, nymax) with (nxmax, nymax) from pfbhip_psi_shape; nx, ny even.
 * dot = image -> coefficients (every element of alpha is written), hdot = coefficients -> image, summed
 * over the bases.  transposed != 0 on the host entries: alpha is (nbasis, nymax, nxmax), the layout of the
 * reference's older Psi (psi.py:218-345).
 */
typedef struct pfbhip_psi pfbhip_psi;
int pfbhip_psi_create(int64_t nx, int64_t ny, int32_t nbasis, const int32_t *bases, int32_t nlevel, pfbhip_psi **out);
int pfbhip_psi_destroy(pfbhip_psi *p);
int pfbhip_psi_shape(const pfbhip_psi *p, int64_t *nxmax, int64_t *nymax);
int pfbhip_psi_dot(pfbhip_psi *p, const double *x_host, double *alpha_host, int transposed);
int pfbhip_psi_hdot(pfbhip_psi *p, const double *alpha_host, double *x_host, int transposed);
int pfbhip_psi_dot_dev(pfbhip_psi *p, const double *x_dev, double *alpha_dev);
int pfbhip_psi_hdot_dev(pfbhip_psi *p, const double *alpha_dev, double *x_dev);

/* dual_update_numba_fast (src/pfb_imaging/prox/prox_21m.py:105-135): v <- vtilde min(1, lam w / |sum_band vtilde|),
 * vtilde = vp + sigma v, arrays (nband, n), weight (n), in place on v.  With bands spread over ranks the
 * update is two device phases around one all-reduce of the band sum (pfbhip_comm_allreduce_sum):
 * vtilde_sum (v <- vtilde, sum over the LOCAL bands) and scale (given the sum over ALL bands). */
int pfbhip_dual_update(const double *vp_host, double *v_host, int64_t nband, int64_t n, double lam, double sigma,
                       const double *weight_host);
int pfbhip_l21_vtilde_sum_dev(const double *vp_dev, double *v_dev, int64_t nband, int64_t n, double sigma, double *sum_dev);
int pfbhip_l21_scale_dev(double *v_dev, int64_t nband, int64_t n, double lam, const double *weight_dev, const double *sum_dev);
/* prox_21m (prox_21m.py:5-26): out = v max(|s| - sigma w, 0) / |s|, s = band sum; weight may be NULL (= 1). */
int pfbhip_prox_21m(const double *v_host, int64_t nband, int64_t n, double sigma, const double *weight_host, double *out_host);
/* positivity (prox/positivity.py:12-33): mode 1 clamps negatives, mode 2 zeroes a pixel in all bands where any
 * band is <= 0; x is (nband, n), in place. */
int pfbhip_positivity(double *x_host, int64_t nband, int64_t n, int mode);
int pfbhip_positivity_dev(double *x_dev, int64_t nband, int64_t n, int mode);

/* Primal-dual backward step with every cube resident on the device: PrimalDual.solve
 * (src/pfb_imaging/opt/primal_dual.py:406-448) with the l21 regulariser over `psi` (prox/l21.py:15-50) and
 * grad(x) = -H (xtilde - x) / gamma (deconv/pfb.py:158-161), H_b(x) = scale[b] sum_p beam_p (PSF_p * (beam_p x)) +
 * eta[b] x from the slots of band b's plan pcs[b] (one plan for all bands -- HessPSF -- or one per band --
 * HessTreeRay; band b owns nparts[b] consecutive entries of psf_slots / beam_slots, as in pfbhip_psfconv_cg).  x (nband, nx, ny): initial iterate in, solution out; v (nband, nbasis, nxmax, nymax):
 * warm-started dual in / out; weight (nbasis, nxmax, nymax); positivity 0 | 1 | 2 (prox/positivity.py).
 * Stops when ||x - xp|| / ||x|| < tol or after maxit iterations; info->iters is the last loop index.
 * comm == NULL: all nband bands are on this device.  comm != NULL (one process per GPU, every rank calls
 * collectively): the arrays hold this rank's nband LOCAL bands; the band sum of the dual update, the
 * "any band <= 0" test of positivity mode 2 and the convergence norms are completed with all-reduces. */
#define PFBHIP_PD_NSTAGES 5
typedef struct pfbhip_pd_info {
    int32_t iters;
    int32_t status; /* 0 converged, 1 maxit reached */
    double eps;
    double loop_ms; /* wall time of the iteration loop alone (cubes resident in HBM: uploads / downloads excluded) */
    /* device time (HIP events on the loop's stream) and number of bracketed launches per stage, summed over the iterations of
     * runs with maxit <= 64 (zeros otherwise): 0 Psi^H analysis of all local bands, 1 l21 dual update, 2 Psi synthesis (one
     * band per launch), 3 PSF-approximate Hessian (one partition apply per launch), 4 primal step + positivity + norms */
    double stage_ms[PFBHIP_PD_NSTAGES];
    int64_t stage_calls[PFBHIP_PD_NSTAGES];
} pfbhip_pd_info;
int pfbhip_primal_dual(pfbhip_psi *psi, pfbhip_psfconv *const *pcs /* [nband] */, int64_t nband, const int64_t *nparts, const int64_t *psf_slots,
                       const int64_t *beam_slots, const double *scale, const double *eta, const double *xtilde_host, double gamma,
                       double *x_host, double *v_host, const double *weight_host, double lam, double sigma, double tau,
                       int positivity, double tol, int maxit, pfbhip_comm *comm, pfbhip_pd_info *info);

/* ---- band reduce over xGMI (RCCL) ------------------------------------ */
/*
 * Replaces the driver-side band sums of the reference
 * (src/pfb_imaging/core/grid.py:430-446, core/deconv.py:320-321): one process
 * per GPU, sum(dirty/residual) to root.  The 128-byte unique id is produced on
 * rank 0 and carried to the other ranks by the caller's launcher (torchrun /
 * any store).
 */
#define PFBHIP_UNIQUE_ID_BYTES 128
int pfbhip_comm_unique_id(uint8_t *id /* [PFBHIP_UNIQUE_ID_BYTES] */);
int pfbhip_comm_create(const uint8_t *id, int nranks, int rank, pfbhip_comm **out);
int pfbhip_comm_destroy(pfbhip_comm *c);
int pfbhip_comm_reduce_sum(pfbhip_comm *c, const double *send_dev, double *recv_dev, int64_t count, int root);
int pfbhip_comm_allreduce_sum(pfbhip_comm *c, const double *send_dev, double *recv_dev, int64_t count);
/* All-gather of equal blocks (recv = nranks blocks of `count` doubles, block r from rank r): the band pool's cube-level
 * methods when every rank owns the same number of bands (operators/band_worker.py:239-308). */
int pfbhip_comm_allgather(pfbhip_comm *c, const double *send_dev, double *recv_dev, int64_t count);
/* Host-array forms on the communicator's persistent device staging buffers (no allocation per call). */
int pfbhip_comm_allreduce_sum_host(pfbhip_comm *c, double *inout_host, int64_t count);
int pfbhip_comm_reduce_sum_host(pfbhip_comm *c, const double *send_host, double *recv_host, int64_t count, int root);
int pfbhip_comm_allgather_host(pfbhip_comm *c, const double *send_host, double *recv_host, int64_t count);
int pfbhip_comm_barrier(pfbhip_comm *c);

#ifdef __cplusplus
}
#endif
#endif /* PFBHIP_H */
