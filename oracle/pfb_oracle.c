/*
 * pfb_oracle.c -- CPU ORACLE for the measurement-operator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pfb-imaging_amd/ may import, link
 * or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU
 * baseline -- never as the product.
 *
 * PARITY PIN: the reference's arithmetic for this path lives in the
 * third-party wheel ducc0 (pinned 0.41.0, /root/reference/uv.lock:1119-1120)
 * whose source is not under /root/reference and which is not installable
 * here.  Against ducc0's exact floating-point output this oracle is "parity
 * unpinned"; it is pinned against the *definition* the reference's own tests
 * use (direct DFT, /root/reference/tests/test_hessian_approx.py:44-67,128-185)
 * and against the reference's analytic identities (see tests/).
 *
 * Contents
 *   1. direct-DFT vis2dirty / dirty2vis (the measurement equation, exact)
 *   2. ES-kernel w-stacking gridder/degridder restatement (the algorithm the
 *      reference reaches through ducc0.wgridder.experimental.vis2dirty /
 *      dirty2vis, call sites /root/reference/src/pfb_imaging/operators/
 *      hessian.py:50-89, gridder.py:590-613): per-visibility position map,
 *      per-plane scatter (grid) and gather (degrid).  FFTs and image-domain
 *      screens are done by the Python wrapper (oracle/wgridder.py).
 *   3. uv-cell counts / Briggs weights restated from
 *      /root/reference/src/pfb_imaging/utils/weighting.py:81-140,143-208.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -ffp-contract=off).
 * -ffp-contract=off matters: the uv-cell / tile index map must be bit-exact
 * against the HIP kernels, which compute it with the same un-fused sequence
 * of IEEE double operations.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PFBO_C 299792458.0

/* ------------------------------------------------------------------ */
/* 1. direct DFT                                                       */
/* ------------------------------------------------------------------ */

/* n - 1 for direction cosines (l, m); accurate for small l, m.
 * For l^2+m^2 > 1 follows the convention of the reference's gridder
 * (n = -sqrt(r2 - 1)). */
static inline double nm1_of(double l, double m)
{
    double r2 = l * l + m * m;
    if (r2 <= 1.0) {
        double s = sqrt(1.0 - r2);
        return -r2 / (1.0 + s);
    }
    return -sqrt(r2 - 1.0) - 1.0;
}

static inline void cexp2pi(double cycles, double *c, double *s)
{
    double r = cycles - rint(cycles);
    double a = 6.283185307179586476925286766559 * r;
    *c = cos(a);
    *s = sin(a);
}

/*
 * dirty[k] = sum_{r,c} mask*wgt * Re( vis * exp(+2 pi i fc (u l + v m - w (n-1))) ) [/ n]
 * evaluated at the npixsel pixels (ix[k], iy[k]).
 *   l = lshift + (ix - nx/2) * px ; m = mshift + (iy - ny/2) * py   (integer nx/2: pixel nx/2 is the phase centre)
 *   (u, v, w) already carry the flip signs su, sv, sw.
 * Formula: /root/reference/tests/test_hessian_approx.py:44-67 (adjoint of it).
 */
void pfbo_dft_vis2dirty(int64_t nrow, int64_t nchan, const double *uvw, const double *freq,
                        const double *vis /* (nrow,nchan,2) */, const double *wgt /* nullable */,
                        const uint8_t *mask /* nullable */, double su, double sv, double sw,
                        int64_t nx, int64_t ny, double px, double py, double lshift, double mshift,
                        int do_w, int divide_by_n, int64_t npixsel, const int64_t *ix, const int64_t *iy,
                        double *out)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < npixsel; ++k) {
        double l = lshift + (double)(ix[k] - nx / 2) * px;
        double m = mshift + (double)(iy[k] - ny / 2) * py;
        double nm1 = do_w ? nm1_of(l, m) : 0.0;
        double acc = 0.0, comp = 0.0;
        for (int64_t r = 0; r < nrow; ++r) {
            double u = uvw[3 * r] * su, v = uvw[3 * r + 1] * sv, w = uvw[3 * r + 2] * sw;
            double ph0 = u * l + v * m - w * nm1;
            for (int64_t c = 0; c < nchan; ++c) {
                int64_t i = r * nchan + c;
                if (mask && !mask[i]) continue;
                double wg = wgt ? wgt[i] : 1.0;
                if (wg == 0.0) continue;
                double cs, sn;
                cexp2pi(ph0 * (freq[c] / PFBO_C), &cs, &sn);
                double term = wg * (vis[2 * i] * cs - vis[2 * i + 1] * sn);
                /* Kahan */
                double y = term - comp, t = acc + y;
                comp = (t - acc) - y;
                acc = t;
            }
        }
        if (divide_by_n) acc /= (nm1 + 1.0);
        out[k] = acc;
    }
}

/*
 * vis[k] = sum_pix dirty * exp(-2 pi i fc (u l + v m - w (n-1))) [/ n]
 * evaluated for the nsel visibilities (row[k], chan[k]).
 */
void pfbo_dft_dirty2vis(int64_t nsel, const int64_t *row, const int64_t *chan, const double *uvw,
                        const double *freq, double su, double sv, double sw, int64_t nx, int64_t ny,
                        double px, double py, double lshift, double mshift, int do_w, int divide_by_n,
                        const double *dirty, double *out /* (nsel,2) */)
{
    /* list the non-zero pixels once */
    int64_t npix = nx * ny, nnz = 0;
    for (int64_t i = 0; i < npix; ++i) nnz += (dirty[i] != 0.0);
    double *pl = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1) * 4);
    int64_t j = 0;
    for (int64_t i = 0; i < npix; ++i) {
        if (dirty[i] == 0.0) continue;
        int64_t a = i / ny, b = i % ny;
        double l = lshift + (double)(a - nx / 2) * px;
        double m = mshift + (double)(b - ny / 2) * py;
        double nm1 = do_w ? nm1_of(l, m) : 0.0;
        double val = dirty[i];
        if (divide_by_n) val /= (nm1 + 1.0);
        pl[4 * j] = l; pl[4 * j + 1] = m; pl[4 * j + 2] = nm1; pl[4 * j + 3] = val;
        ++j;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < nsel; ++k) {
        int64_t r = row[k];
        double fc = freq[chan[k]] / PFBO_C;
        double u = uvw[3 * r] * su * fc, v = uvw[3 * r + 1] * sv * fc, w = uvw[3 * r + 2] * sw * fc;
        double re = 0.0, im = 0.0;
        for (int64_t q = 0; q < nnz; ++q) {
            double cs, sn;
            cexp2pi(u * pl[4 * q] + v * pl[4 * q + 1] - w * pl[4 * q + 2], &cs, &sn);
            re += pl[4 * q + 3] * cs;
            im -= pl[4 * q + 3] * sn;
        }
        out[2 * k] = re;
        out[2 * k + 1] = im;
    }
    free(pl);
}

/* ------------------------------------------------------------------ */
/* 2. ES-kernel w-stacking gridder restatement                         */
/* ------------------------------------------------------------------ */

static inline double es_kernel(double x, double beta)
{
    double t = 1.0 - x * x;
    return (t >= 0.0) ? exp(beta * (sqrt(t) - 1.0)) : 0.0;
}

/* Kernel value of tap `a` for the visibility whose first tap is i0 at grid coordinate p:
 * either the exact ES kernel (ktab == NULL) or its piecewise polynomial of degree D in
 * z = 2 f - 1, f = p + 1 - W/2 - i0 in [0,1)  (the form the product's kernels evaluate). */
static inline double tap_kernel(int a, int i0, double p, int W, double beta, const double *ktab, int D)
{
    if (!ktab) return es_kernel(((double)(i0 + a) - p) * (2.0 / (double)W), beta);
    double z = 2.0 * ((p + (1.0 - 0.5 * (double)W)) - (double)i0) - 1.0;
    const double *c = ktab + (size_t)a * (size_t)(D + 1);
    double v = c[D];
    for (int k = D - 1; k >= 0; --k) v = fma(v, z, c[k]);
    return v;
}

/*
 * Per-visibility position map.  THE bit-exact contract shared with the HIP
 * kernels (pfb-imaging_amd/csrc/vismap.hpp): every statement below is one
 * IEEE-754 double operation, no fused multiply-add.
 *
 *   fc      = freq[c] / c0                      (caller passes fc[])
 *   (u,v,w) = uvw * (su,sv,sw) * fc
 *   flip    = do_w && w < 0  -> (u,v,w) = -(u,v,w)     [Hermitian fold]
 *   xu      = u * px ; fu = xu - floor(xu) ; pu = fu * nu
 *   iu0     = (int) floor(pu + (1 - W/2))       first of W taps, may be < 0
 *   pw      = (w - wmin) * xdw ; p0 = (int) floor(pw + (1 - W/2))
 *             (kernel w-planes: wmin = first plane, xdw = 1/dw;
 *              polynomial w-planes: wmin = centre of the w range, xdw = 1/half-range, so
 *              pw = s in [-1,1] is the interpolation abscissa and p0 is unused)
 *
 * active[i] = mask ? mask[i] != 0 : 1
 */
void pfbo_vismap(int64_t nrow, int64_t nchan, const double *uvw, const double *fc, const uint8_t *mask,
                 double su, double sv, double sw, double px, double py, int64_t nu, int64_t nv, int W,
                 int do_w, double wmin, double xdw, double *pu, double *pv, double *pw, double *uvw_l /* (n,3) lambda, post-flip */,
                 uint8_t *flip, int32_t *iu0, int32_t *iv0, int32_t *p0)
{
    const double shift = 1.0 - 0.5 * (double)W;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nrow; ++r) {
        double ur = uvw[3 * r] * su, vr = uvw[3 * r + 1] * sv, wr = uvw[3 * r + 2] * sw;
        for (int64_t c = 0; c < nchan; ++c) {
            int64_t i = r * nchan + c;
            double u = ur * fc[c], v = vr * fc[c], w = wr * fc[c];
            uint8_t fl = 0;
            if (do_w && w < 0.0) { u = -u; v = -v; w = -w; fl = 1; }
            double xu = u * px, xv = v * py;
            double fu = xu - floor(xu), fv = xv - floor(xv);
            double ppu = fu * (double)nu, ppv = fv * (double)nv;
            pu[i] = ppu; pv[i] = ppv;
            iu0[i] = (int32_t)floor(ppu + shift);
            iv0[i] = (int32_t)floor(ppv + shift);
            if (do_w) {
                double ppw = (w - wmin) * xdw;
                pw[i] = ppw;
                p0[i] = (int32_t)floor(ppw + shift);
            } else {
                pw[i] = 0.0;
                p0[i] = 0;
            }
            flip[i] = fl;
            uvw_l[3 * i] = u; uvw_l[3 * i + 1] = v; uvw_l[3 * i + 2] = w;
            (void)mask;
        }
    }
}

static inline int64_t wrapi(int64_t i, int64_t n)
{
    i %= n;
    return i < 0 ? i + n : i;
}

/*
 * Scatter the (already weighted / phased / conjugated) visibilities sval onto
 * w-plane `plane` of the oversampled grid (nu, nv) complex, v contiguous.
 * `order` lists the active visibilities sorted by tile (tile = (iu0w/T)*ntv +
 * iv0w/T), `tstart` has ntiles+1 offsets into it.
 */
void pfbo_grid_plane(int64_t ntiles, const int64_t *tstart, const int64_t *order, const double *pu,
                     const double *pv, const int32_t *iu0, const int32_t *iv0, const double *kwv /* per-vis plane weight, 0 = not on this plane */,
                     const double *sval /* (n,2) */, int W, double beta, const double *ktab /* (W,D+1) or NULL */, int D,
                     int64_t nu, int64_t nv, int T, double *grid /* (nu,nv,2) zeroed */)
{
    const int L = T + W - 1;
    const int64_t ntv = (nv + T - 1) / T;
#pragma omp parallel
    {
        double *loc = (double *)malloc(sizeof(double) * (size_t)L * L * 2);
        double ku[32], kv[32];
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < ntiles; ++t) {
            if (tstart[t + 1] == tstart[t]) continue;
            int64_t bu = (t / ntv) * T, bv = (t % ntv) * T;
            memset(loc, 0, sizeof(double) * (size_t)L * L * 2);
            int any = 0;
            for (int64_t q = tstart[t]; q < tstart[t + 1]; ++q) {
                int64_t i = order[q];
                double kw = kwv[i];
                if (kw == 0.0) continue;
                any = 1;
                int64_t lu = wrapi(iu0[i], nu) - bu, lv = wrapi(iv0[i], nv) - bv;
                for (int a = 0; a < W; ++a) {
                    ku[a] = tap_kernel(a, iu0[i], pu[i], W, beta, ktab, D);
                    kv[a] = tap_kernel(a, iv0[i], pv[i], W, beta, ktab, D);
                }
                double vr = sval[2 * i] * kw, vi = sval[2 * i + 1] * kw;
                for (int a = 0; a < W; ++a) {
                    double ar = vr * ku[a], ai = vi * ku[a];
                    double *row = loc + ((size_t)(lu + a) * L + (size_t)lv) * 2;
                    for (int b = 0; b < W; ++b) {
                        row[2 * b] += ar * kv[b];
                        row[2 * b + 1] += ai * kv[b];
                    }
                }
            }
            if (!any) continue;
            for (int a = 0; a < L; ++a) {
                int64_t gu = wrapi(bu + a, nu);
                for (int b = 0; b < L; ++b) {
                    double re = loc[((size_t)a * L + b) * 2], im = loc[((size_t)a * L + b) * 2 + 1];
                    if (re == 0.0 && im == 0.0) continue;
                    int64_t gv = wrapi(bv + b, nv);
                    double *g = grid + ((size_t)gu * (size_t)nv + (size_t)gv) * 2;
#pragma omp atomic
                    g[0] += re;
#pragma omp atomic
                    g[1] += im;
                }
            }
        }
        free(loc);
    }
}

/* Gather from one (already FFT'd) w-plane; acc (n,2) += kwv * sum_taps. */
void pfbo_degrid_plane(int64_t ntiles, const int64_t *tstart, const int64_t *order, const double *pu,
                       const double *pv, const int32_t *iu0, const int32_t *iv0, const double *kwv, int W,
                       double beta, const double *ktab, int D, int64_t nu, int64_t nv, int T, const double *grid,
                       double *acc /* (n,2) */)
{
    const int L = T + W - 1;
    const int64_t ntv = (nv + T - 1) / T;
#pragma omp parallel
    {
        double *loc = (double *)malloc(sizeof(double) * (size_t)L * L * 2);
        double ku[32], kv[32];
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < ntiles; ++t) {
            if (tstart[t + 1] == tstart[t]) continue;
            int64_t bu = (t / ntv) * T, bv = (t % ntv) * T;
            for (int a = 0; a < L; ++a) {
                int64_t gu = wrapi(bu + a, nu);
                for (int b = 0; b < L; ++b) {
                    int64_t gv = wrapi(bv + b, nv);
                    const double *g = grid + ((size_t)gu * (size_t)nv + (size_t)gv) * 2;
                    loc[((size_t)a * L + b) * 2] = g[0];
                    loc[((size_t)a * L + b) * 2 + 1] = g[1];
                }
            }
            for (int64_t q = tstart[t]; q < tstart[t + 1]; ++q) {
                int64_t i = order[q];
                double kw = kwv[i];
                if (kw == 0.0) continue;
                int64_t lu = wrapi(iu0[i], nu) - bu, lv = wrapi(iv0[i], nv) - bv;
                for (int a = 0; a < W; ++a) {
                    ku[a] = tap_kernel(a, iu0[i], pu[i], W, beta, ktab, D);
                    kv[a] = tap_kernel(a, iv0[i], pv[i], W, beta, ktab, D);
                }
                double sr = 0.0, si = 0.0;
                for (int a = 0; a < W; ++a) {
                    const double *row = loc + ((size_t)(lu + a) * L + (size_t)lv) * 2;
                    double tr = 0.0, ti = 0.0;
                    for (int b = 0; b < W; ++b) {
                        tr += row[2 * b] * kv[b];
                        ti += row[2 * b + 1] * kv[b];
                    }
                    sr += tr * ku[a];
                    si += ti * ku[a];
                }
                acc[2 * i] += sr * kw;
                acc[2 * i + 1] += si * kw;
            }
        }
        free(loc);
    }
}

/* ------------------------------------------------------------------ */
/* 3. uv-cell counts and Briggs weights                                */
/* ------------------------------------------------------------------ */

/*
 * uv-cell index map, restated from
 * /root/reference/src/pfb_imaging/utils/weighting.py:81-140.
 * Returns the flat cell index u_idx*ny+v_idx per visibility, or -1 when
 * masked / out of bounds, and (optionally) accumulates counts (ncorr,nx,ny).
 * Every floating-point statement is a single IEEE operation in the same
 * order as the reference (division by u_cell, not multiplication by 1/u_cell).
 */
void pfbo_uvcell_index(int64_t nrow, int64_t nchan, const double *uvw, const double *freq,
                       const uint8_t *mask, int64_t nx, int64_t ny, double cell_x, double cell_y,
                       double usign, double vsign, int64_t *cell /* (nrow,nchan) */)
{
    double u_cell = 1.0 / ((double)nx * cell_x);
    double umax = fabs(1.0 / cell_x / 2.0);
    double v_cell = 1.0 / ((double)ny * cell_y);
    double vmax = fabs(1.0 / cell_y / 2.0);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nrow; ++r) {
        for (int64_t f = 0; f < nchan; ++f) {
            int64_t i = r * nchan + f;
            if (!mask[i]) { cell[i] = -1; continue; }
            double nf = freq[f] / PFBO_C;
            double u = uvw[3 * r] * nf * usign;
            double v = uvw[3 * r + 1] * nf * vsign;
            if (v < 0.0) { u = -u; v = -v; }
            double ug = (u + umax) / u_cell;
            double vg = (v + vmax) / v_cell;
            double fu = floor(ug), fv = floor(vg);
            if (!(fu >= 0.0) || !(fu < (double)nx) || !(fv >= 0.0) || !(fv < (double)ny)) { cell[i] = -1; continue; }
            cell[i] = (int64_t)fu * ny + (int64_t)fv;
        }
    }
}

void pfbo_counts_accumulate(int64_t ncorr, int64_t nrow, int64_t nchan, const int64_t *cell,
                            const double *wgt /* (ncorr,nrow,nchan) */, int64_t nxy, double *counts /* (ncorr,nxy) zeroed */)
{
    /* serial over visibilities in row-major order: the same summation order
     * as the reference with ngrid=1 (weighting.py:105-138) */
    for (int64_t c = 0; c < ncorr; ++c)
        for (int64_t i = 0; i < nrow * nchan; ++i)
            if (cell[i] >= 0) counts[c * nxy + cell[i]] += wgt[c * nrow * nchan + i];
}

/* weighting.py:186-206: w /= counts[cell] where counts > 0 */
void pfbo_counts_divide(int64_t ncorr, int64_t nrow, int64_t nchan, const int64_t *cell,
                        const double *counts, int64_t nxy, double *wgt)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nrow * nchan; ++i) {
        if (cell[i] < 0) continue;
        for (int64_t c = 0; c < ncorr; ++c) {
            double cv = counts[c * nxy + cell[i]];
            if (cv > 0.0) wgt[c * nrow * nchan + i] /= cv;
        }
    }
}

/* weighting.py:161-176: serial sums for the Briggs factor, per correlation */
void pfbo_briggs_sums(int64_t ncorr, int64_t nxy, const double *counts, double *num /* sum c^2 */, double *den /* sum c */)
{
    for (int64_t c = 0; c < ncorr; ++c) {
        double a = 0.0, b = 0.0;
        for (int64_t i = 0; i < nxy; ++i) {
            double cv = counts[c * nxy + i];
            a += cv * cv;
            b += cv;
        }
        num[c] = a;
        den[c] = b;
    }
}

int pfbo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void pfbo_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* -------------------------------------------------------------------------------------------------------
 * wmode 2: ONE uv-plane, the w-term of every visibility folded into its gridding kernel.
 *
 *   exp(-2 pi i w t(s)) = exp(-2 pi i wc t) * E(dw; s),  s = l^2 + m^2 (phase centre on axis),  dw = w - wc,
 *   E(dw; s) ~ sum_k C_k(dw) (s / smax)^k                (Chebyshev interpolation in s at K nodes)
 *   multiplication by l^2 in the image  <->  -(nu px / (pi W))^2 d^2/dx^2 of the u-kernel phi(x)
 *
 * so a visibility's footprint is  sum_k C_k D^k[phi(x) phi(y)],  D^k = sum_r binom(k, r) a_r(x) b_{k-r}(y), with
 * a_r / b_r the 2r-th derivatives of the kernel polynomial scaled by (-alpha_u / smax)^r / (-alpha_v / smax)^r
 * (tables dtab_u / dtab_v, (K, W, D + 1), built by oracle/wgridder.py).  cw (n, K) complex are the C_k of every
 * visibility (the gridding direction; the gather conjugates them).
 * ------------------------------------------------------------------------------------------------------- */
static inline void wd_taps(int K, int W, int D, const double *dtab, int i0, double p, double *out /* (K, 32) */)
{
    double z = 2.0 * ((p + (1.0 - 0.5 * (double)W)) - (double)i0) - 1.0;
    for (int k = 0; k < K; ++k)
        for (int a = 0; a < W; ++a) {
            const double *c = dtab + ((size_t)k * (size_t)W + (size_t)a) * (size_t)(D + 1);
            double v = c[D];
            for (int q = D - 1; q >= 0; --q) v = fma(v, z, c[q]);
            out[k * 32 + a] = v;
        }
}

static const double wd_binom[4][4] = {{1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 1, 0}, {1, 3, 3, 1}};

void pfbo_grid_plane_wd(int64_t ntiles, const int64_t *tstart, const int64_t *order, const double *pu, const double *pv,
                        const int32_t *iu0, const int32_t *iv0, const double *cw /* (n, K, 2) */,
                        const double *sval /* (n, 2) */, int K, int W, const double *dtab_u, const double *dtab_v, int D,
                        int64_t nu, int64_t nv, int T, double *grid /* (nu, nv, 2) zeroed */)
{
    const int L = T + W - 1;
    const int64_t ntv = (nv + T - 1) / T;
#pragma omp parallel
    {
        double *loc = (double *)malloc(sizeof(double) * (size_t)L * L * 2);
        double ak[4 * 32], bk[4 * 32], sr[4][32], si[4][32];
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < ntiles; ++t) {
            if (tstart[t + 1] == tstart[t]) continue;
            int64_t bu = (t / ntv) * T, bv = (t % ntv) * T;
            memset(loc, 0, sizeof(double) * (size_t)L * L * 2);
            for (int64_t q = tstart[t]; q < tstart[t + 1]; ++q) {
                int64_t i = order[q];
                int64_t lu = wrapi(iu0[i], nu) - bu, lv = wrapi(iv0[i], nv) - bv;
                wd_taps(K, W, D, dtab_u, iu0[i], pu[i], ak);
                wd_taps(K, W, D, dtab_v, iv0[i], pv[i], bk);
                /* P_k = val * C_k;  S_r(col) = sum_m binom(r + m, r) P_{r+m} b_m(col) */
                double pr[4], pi[4];
                for (int k = 0; k < K; ++k) {
                    double cr = cw[((size_t)i * K + k) * 2], ci = cw[((size_t)i * K + k) * 2 + 1];
                    pr[k] = sval[2 * i] * cr - sval[2 * i + 1] * ci;
                    pi[k] = sval[2 * i] * ci + sval[2 * i + 1] * cr;
                }
                for (int r = 0; r < K; ++r)
                    for (int b = 0; b < W; ++b) {
                        double xr = 0.0, xi = 0.0;
                        for (int m = 0; r + m < K; ++m) {
                            xr += wd_binom[r + m][r] * pr[r + m] * bk[m * 32 + b];
                            xi += wd_binom[r + m][r] * pi[r + m] * bk[m * 32 + b];
                        }
                        sr[r][b] = xr;
                        si[r][b] = xi;
                    }
                for (int a = 0; a < W; ++a) {
                    double *row = loc + ((size_t)(lu + a) * L + (size_t)lv) * 2;
                    for (int b = 0; b < W; ++b) {
                        double xr = 0.0, xi = 0.0;
                        for (int r = 0; r < K; ++r) {
                            xr += ak[r * 32 + a] * sr[r][b];
                            xi += ak[r * 32 + a] * si[r][b];
                        }
                        row[2 * b] += xr;
                        row[2 * b + 1] += xi;
                    }
                }
            }
            for (int a = 0; a < L; ++a) {
                int64_t gu = wrapi(bu + a, nu);
                for (int b = 0; b < L; ++b) {
                    double re = loc[((size_t)a * L + b) * 2], im = loc[((size_t)a * L + b) * 2 + 1];
                    if (re == 0.0 && im == 0.0) continue;
                    int64_t gv = wrapi(bv + b, nv);
                    double *g = grid + ((size_t)gu * (size_t)nv + (size_t)gv) * 2;
#pragma omp atomic
                    g[0] += re;
#pragma omp atomic
                    g[1] += im;
                }
            }
        }
        free(loc);
    }
}

/* Gather: acc (n, 2) += sum_cells grid(cell) * conj(kernel of the visibility)(cell)  (the kernel's a_r, b_r are real:
 * only the C_k are conjugated). */
void pfbo_degrid_plane_wd(int64_t ntiles, const int64_t *tstart, const int64_t *order, const double *pu, const double *pv,
                          const int32_t *iu0, const int32_t *iv0, const double *cw, int K, int W, const double *dtab_u,
                          const double *dtab_v, int D, int64_t nu, int64_t nv, int T, const double *grid, double *acc)
{
    const int L = T + W - 1;
    const int64_t ntv = (nv + T - 1) / T;
#pragma omp parallel
    {
        double *loc = (double *)malloc(sizeof(double) * (size_t)L * L * 2);
        double ak[4 * 32], bk[4 * 32];
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < ntiles; ++t) {
            if (tstart[t + 1] == tstart[t]) continue;
            int64_t bu = (t / ntv) * T, bv = (t % ntv) * T;
            for (int a = 0; a < L; ++a) {
                int64_t gu = wrapi(bu + a, nu);
                for (int b = 0; b < L; ++b) {
                    int64_t gv = wrapi(bv + b, nv);
                    const double *g = grid + ((size_t)gu * (size_t)nv + (size_t)gv) * 2;
                    loc[((size_t)a * L + b) * 2] = g[0];
                    loc[((size_t)a * L + b) * 2 + 1] = g[1];
                }
            }
            for (int64_t q = tstart[t]; q < tstart[t + 1]; ++q) {
                int64_t i = order[q];
                int64_t lu = wrapi(iu0[i], nu) - bu, lv = wrapi(iv0[i], nv) - bv;
                wd_taps(K, W, D, dtab_u, iu0[i], pu[i], ak);
                wd_taps(K, W, D, dtab_v, iv0[i], pv[i], bk);
                /* T_r(col) = sum_rows a_r(row) grid(row, col);  D_k = sum_col sum_{r <= k} binom(k, r) b_{k-r}(col) T_r(col) */
                double dr[4] = {0, 0, 0, 0}, di[4] = {0, 0, 0, 0};
                for (int b = 0; b < W; ++b) {
                    double tr[4] = {0, 0, 0, 0}, ti[4] = {0, 0, 0, 0};
                    for (int a = 0; a < W; ++a) {
                        const double *cell = loc + ((size_t)(lu + a) * L + (size_t)(lv + b)) * 2;
                        for (int r = 0; r < K; ++r) {
                            tr[r] += ak[r * 32 + a] * cell[0];
                            ti[r] += ak[r * 32 + a] * cell[1];
                        }
                    }
                    for (int k = 0; k < K; ++k)
                        for (int r = 0; r <= k; ++r) {
                            dr[k] += wd_binom[k][r] * bk[(k - r) * 32 + b] * tr[r];
                            di[k] += wd_binom[k][r] * bk[(k - r) * 32 + b] * ti[r];
                        }
                }
                double xr = 0.0, xi = 0.0;
                for (int k = 0; k < K; ++k) {
                    double cr = cw[((size_t)i * K + k) * 2], ci = -cw[((size_t)i * K + k) * 2 + 1];
                    xr += dr[k] * cr - di[k] * ci;
                    xi += dr[k] * ci + di[k] * cr;
                }
                acc[2 * i] += xr;
                acc[2 * i + 1] += xi;
            }
        }
        free(loc);
    }
}
