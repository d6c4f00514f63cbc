// psi.hip -- wavelet dictionary Psi (multi-level 2-D Daubechies DWT, zero-padding mode, packed x-first
// coefficient layout) and the fused l21 dual update / positivity of the primal-dual iteration.
// Replaces PsiBandNocopyt.dot / hdot (/root/reference/src/pfb_imaging/operators/psi.py:466-535) built on
// wavelets/wavelets.py:216-343 and wavelets/convolutions.py:5-327, dual_update_numba_fast
// (prox/prox_21m.py:105-135), prox_21m (prox_21m.py:5-26) and positivity (prox/positivity.py:12-33).
//
// Every kernel is a short FIR pass (2..16 taps), HBM-bound: one thread per output element, lo and hi
// bands computed together from the same inputs, reads served by L1/L2 after the first touch.
//   analysis  level: rows (axis 1) -> cbuff (n0, 2 s1) ; columns (axis 0) -> packed block (2 s0, 2 s1)
//   synthesis level: columns (axis 0) -> cbuff (n0out, 2 s1) ; rows (axis 1) -> image (n0out, n1out)
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>

#include <memory>
#include <vector>

#include "common.hpp"
#include "pipeline_api.hpp"

namespace pfbhip {

constexpr int MAXTAP = 16;
static const double kDb[8][MAXTAP] = {
#include "wavelet_filters.inc"
};

struct Filt {
    int L;
    double lo[MAXTAP], hi[MAXTAP];
};

// decomposition (dec_lo = h reversed, dec_hi[k] = (-1)^(k+1) h[k]) / reconstruction (rec_lo = h,
// rec_hi[k] = (-1)^k h[L-1-k]) filter banks of dbN in PyWavelets' convention
static void make_filters(int N, Filt *dec, Filt *rec)
{
    const int L = 2 * N;
    dec->L = rec->L = L;
    for (int k = 0; k < MAXTAP; ++k) dec->lo[k] = dec->hi[k] = rec->lo[k] = rec->hi[k] = 0.0;
    const double *h = kDb[N - 1];
    for (int k = 0; k < L; ++k) {
        rec->lo[k] = h[k];
        rec->hi[k] = (k & 1 ? -1.0 : 1.0) * h[L - 1 - k];
    }
    for (int k = 0; k < L; ++k) {
        dec->lo[k] = rec->lo[L - 1 - k];
        dec->hi[k] = rec->hi[L - 1 - k];
    }
}

static inline dim3 grid2(int64_t ncols, int64_t nrows) { return dim3(uint32_t(ceil_div(ncols, 256)), uint32_t(nrows)); }

// out[i][o] = sum_j lo[j] in[i][2o+1-j], out[i][s1+o] likewise with hi  (o < s1; in zero outside [0, n1))
__global__ void __launch_bounds__(256) k_dwt_rows(const double *__restrict__ in, size_t ldi, int n1, Filt f, int s1,
                                                   double *__restrict__ out, size_t ldo)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= s1) return;
    const double *row = in + size_t(blockIdx.y) * ldi;
    double a = 0.0, d = 0.0;
    for (int j = 0; j < f.L; ++j) {
        const int p = 2 * o + 1 - j;
        if (p >= 0 && p < n1) {
            const double v = row[p];
            a += f.lo[j] * v;
            d += f.hi[j] * v;
        }
    }
    double *orow = out + size_t(blockIdx.y) * ldo;
    orow[o] = a;
    orow[s1 + o] = d;
}

// out[o][c] = sum_j lo[j] in[2o+1-j][c], out[s0+o][c] with hi  (o < s0, c < ncol)
__global__ void __launch_bounds__(256) k_dwt_cols(const double *__restrict__ in, size_t ldi, int n0, Filt f, int s0, int ncol,
                                                   double *__restrict__ out, size_t ldo)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y;
    if (c >= ncol) return;
    double a = 0.0, d = 0.0;
    for (int j = 0; j < f.L; ++j) {
        const int p = 2 * o + 1 - j;
        if (p >= 0 && p < n0) {
            const double v = in[size_t(p) * ldi + size_t(c)];
            a += f.lo[j] * v;
            d += f.hi[j] * v;
        }
    }
    out[size_t(o) * ldo + size_t(c)] = a;
    out[size_t(s0 + o) * ldo + size_t(c)] = d;
}

// cb[t][c] = sum_q lo[2q+p] blk[m+h-1-q][c] + hi[2q+p] blk[s0+m+h-1-q][c],  t = 2m+p < n0out, c < ncol.
// The LL quadrant (rows < s0, columns < sll) is read from `ll` (the reconstruction of the level below)
// when ll != NULL, so that the caller's coefficient array is never written.
// One thread produces BOTH output rows 2m and 2m + 1 of its column from the same h input pairs (blockIdx.y = m).
__global__ void __launch_bounds__(256) k_idwt_cols(const double *__restrict__ blk, size_t ldb, int s0, int ncol, Filt f,
                                                    const double *__restrict__ ll, size_t ldl, int sll, int n0out,
                                                    double *__restrict__ cb, size_t ldc)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (c >= ncol) return;
    const int h = f.L / 2;
    const bool from_ll = ll != nullptr && c < sll;
    double a0 = 0.0, a1 = 0.0;
    for (int q = 0; q < h; ++q) {
        const int r = m + h - 1 - q;
        const double lo = from_ll ? ll[size_t(r) * ldl + size_t(c)] : blk[size_t(r) * ldb + size_t(c)];
        const double hi = blk[size_t(s0 + r) * ldb + size_t(c)];
        a0 += f.lo[2 * q] * lo + f.hi[2 * q] * hi;
        a1 += f.lo[2 * q + 1] * lo + f.hi[2 * q + 1] * hi;
    }
    cb[size_t(2 * m) * ldc + size_t(c)] = a0;
    if (2 * m + 1 < n0out) cb[size_t(2 * m + 1) * ldc + size_t(c)] = a1;
}

// img[i][u] (=|+=) sum_q lo[2q+p] cb[i][m+h-1-q] + hi[2q+p] cb[i][s1+m+h-1-q],  u = 2m+p < n1out
// One thread produces the output pair u = 2m, 2m + 1 (one 16-byte store when the row pitch allows it).
// `extra` (ld = lde; NULL: none) is added on top when not accumulating: the identity basis' slice of the cube, which then
// needs no pass of its own over the image.
__global__ void __launch_bounds__(256) k_idwt_rows(const double *__restrict__ cb, size_t ldc, int s1, Filt f, int n1out,
                                                    int accumulate, double *__restrict__ img, size_t ldi,
                                                    const double *__restrict__ extra, size_t lde)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (2 * m >= n1out) return;
    const double *row = cb + size_t(blockIdx.y) * ldc;
    const int h = f.L / 2;
    double a0 = 0.0, a1 = 0.0;
    for (int q = 0; q < h; ++q) {
        const int r = m + h - 1 - q;
        const double lo = row[r], hi = row[s1 + r];
        a0 += f.lo[2 * q] * lo + f.hi[2 * q] * hi;
        a1 += f.lo[2 * q + 1] * lo + f.hi[2 * q + 1] * hi;
    }
    double *o = img + size_t(blockIdx.y) * ldi + size_t(2 * m);
    if (accumulate) {
        o[0] += a0;
        if (2 * m + 1 < n1out) o[1] += a1;
    } else {
        if (extra != nullptr) {
            const double *x = extra + size_t(blockIdx.y) * lde + size_t(2 * m);
            a0 += x[0];
            if (2 * m + 1 < n1out) a1 += x[1];
        }
        o[0] = a0;
        if (2 * m + 1 < n1out) o[1] = a1;
    }
}

// dst (rows x cols, ldd) (=|+=) src (lds)
__global__ void __launch_bounds__(256) k_copy2d(const double *__restrict__ src, size_t lds, int cols, int accumulate,
                                                 double *__restrict__ dst, size_t ldd)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const double v = src[size_t(blockIdx.y) * lds + size_t(c)];
    double *o = dst + size_t(blockIdx.y) * ldd + size_t(c);
    *o = accumulate ? *o + v : v;
}

// Zero the rectangles {slice, row0, nrows, col0, ncols} of a coefficient cube (slices of nxmax x nymax doubles): the cells
// of alpha no transform writes (margins of the packed Mallat layout, padding up to the largest basis).
struct ZeroRect {
    int slice, row0, nrows, col0, ncols;
};
__global__ void __launch_bounds__(256) k_zero_rects(const ZeroRect *__restrict__ rects, double *__restrict__ alpha, size_t slice_stride,
                                                     size_t ld)
{
    const ZeroRect r = rects[blockIdx.x];
    double *base = alpha + size_t(r.slice) * slice_stride + size_t(r.row0) * ld + size_t(r.col0);
    const int64_t n = int64_t(r.nrows) * r.ncols;
    for (int64_t i = threadIdx.x; i < n; i += 256) base[size_t(i / r.ncols) * ld + size_t(i % r.ncols)] = 0.0;
}

// out (cols, rows) = in (rows, cols)^T
__global__ void __launch_bounds__(256) k_transpose_f64_any(const double *__restrict__ in, int rows, int cols,
                                                            double *__restrict__ out)
{
    __shared__ double tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + int(threadIdx.x);
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[size_t(r) * size_t(cols) + size_t(c)];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + int(threadIdx.x);
        if (r < rows && c < cols) out[size_t(c) * size_t(rows) + size_t(r)] = tile[threadIdx.x][j];
    }
}

// ---- l21 dual update (prox_21m.py:105-135) and friends ------------------------------------------------
// phase 1: v <- vp + sigma v for every local band, sum[i] = sum over the local bands
__global__ void __launch_bounds__(256) k_l21_vtilde(const double *__restrict__ vp, double *__restrict__ v, int nband, int64_t n,
                                                     double sigma, double *__restrict__ sum)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < nband; ++b) {
        const double vt = vp[size_t(b) * size_t(n) + size_t(i)] + sigma * v[size_t(b) * size_t(n) + size_t(i)];
        v[size_t(b) * size_t(n) + size_t(i)] = vt;
        s += vt;
    }
    sum[i] = s;
}
// phase 2: v *= lam w / |sum| where |sum| > lam w   (sum = the band sum over ALL bands)
__global__ void __launch_bounds__(256) k_l21_scale(double *__restrict__ v, int nband, int64_t n, double lam,
                                                    const double *__restrict__ weight, const double *__restrict__ sum)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    const double a = fabs(sum[i]);
    const double thr = lam * weight[i];
    if (a > thr) {
        const double sc = thr / a;
        for (int b = 0; b < nband; ++b) v[size_t(b) * size_t(n) + size_t(i)] *= sc;
    }
}
// all bands local: both phases and the dual extrapolation in ONE pass over the cubes.
// a = Psi^H xp on entry; on exit a = v_new (the updated dual) and ext = 2 v_new - vp.
__global__ void __launch_bounds__(256) k_l21_fused(const double *__restrict__ vp, double *__restrict__ a, double *__restrict__ ext,
                                                    int nband, int64_t n, double lam, double sigma,
                                                    const double *__restrict__ weight)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < nband; ++b) s += vp[size_t(b) * size_t(n) + size_t(i)] + sigma * a[size_t(b) * size_t(n) + size_t(i)];
    const double as = fabs(s), thr = lam * weight[i];
    const double sc = as > thr ? thr / as : 1.0;
    for (int b = 0; b < nband; ++b) {
        const size_t o = size_t(b) * size_t(n) + size_t(i);
        const double p = vp[o];
        const double vn = (p + sigma * a[o]) * sc;
        a[o] = vn;
        ext[o] = 2.0 * vn - p;
    }
}

// bands spread over ranks: (1) the LOCAL band sum of vtilde, (2) -- after the all-reduce -- the update itself
__global__ void __launch_bounds__(256) k_l21_localsum(const double *__restrict__ vp, const double *__restrict__ a, int nband,
                                                       int64_t n, double sigma, double *__restrict__ sum)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < nband; ++b) s += vp[size_t(b) * size_t(n) + size_t(i)] + sigma * a[size_t(b) * size_t(n) + size_t(i)];
    sum[i] = s;
}
__global__ void __launch_bounds__(256) k_l21_apply(const double *__restrict__ vp, double *__restrict__ a, double *__restrict__ ext,
                                                    int nband, int64_t n, double lam, double sigma,
                                                    const double *__restrict__ weight, const double *__restrict__ sum)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    const double as = fabs(sum[i]), thr = lam * weight[i];
    const double sc = as > thr ? thr / as : 1.0;
    for (int b = 0; b < nband; ++b) {
        const size_t o = size_t(b) * size_t(n) + size_t(i);
        const double p = vp[o];
        const double vn = (p + sigma * a[o]) * sc;
        a[o] = vn;
        ext[o] = 2.0 * vn - p;
    }
}
// positivity mode 2 across ranks: (1) bad[i] = 1 if any LOCAL band is <= 0, (2) zero where the all-reduced count > 0
__global__ void __launch_bounds__(256) k_pos_flag(const double *__restrict__ x, int nband, int64_t n, double *__restrict__ bad)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    bool b0 = false;
    for (int b = 0; b < nband; ++b) b0 = b0 || x[size_t(b) * size_t(n) + size_t(i)] <= 0.0;
    bad[i] = b0 ? 1.0 : 0.0;
}
__global__ void __launch_bounds__(256) k_pos_zero(double *__restrict__ x, int nband, int64_t n, const double *__restrict__ bad)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    if (bad[i] > 0.0)
        for (int b = 0; b < nband; ++b) x[size_t(b) * size_t(n) + size_t(i)] = 0.0;
}

// prox_{sigma ||.||_21}(v): v * max(|s| - sigma w, 0) / |s|, s = band sum (prox_21m.py:5-26)
__global__ void __launch_bounds__(256) k_prox21(const double *__restrict__ v, int nband, int64_t n, double sigma,
                                                const double *__restrict__ weight, double *__restrict__ out)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < nband; ++b) s += v[size_t(b) * size_t(n) + size_t(i)];
    const double a = fabs(s);
    const double w = weight != nullptr ? weight[i] : 1.0;
    const double ratio = a > 0.0 ? fmax(a - sigma * w, 0.0) / a : 0.0;
    for (int b = 0; b < nband; ++b) out[size_t(b) * size_t(n) + size_t(i)] = v[size_t(b) * size_t(n) + size_t(i)] * ratio;
}
// mode 1: clamp negatives; mode 2: zero a pixel in all bands where any band is <= 0 (positivity.py:12-33)
__global__ void __launch_bounds__(256) k_positivity(double *__restrict__ x, int nband, int64_t n, int mode)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n) return;
    if (mode == 1) {
        for (int b = 0; b < nband; ++b)
            if (x[size_t(b) * size_t(n) + size_t(i)] < 0.0) x[size_t(b) * size_t(n) + size_t(i)] = 0.0;
    } else {
        bool bad = false;
        for (int b = 0; b < nband; ++b) bad = bad || x[size_t(b) * size_t(n) + size_t(i)] <= 0.0;
        if (bad)
            for (int b = 0; b < nband; ++b) x[size_t(b) * size_t(n) + size_t(i)] = 0.0;
    }
}

}  // namespace pfbhip

using namespace pfbhip;

// One band's dictionary: bookkeeping of operators/psi.py:23-142 + device scratch.
struct pfbhip_psi {
    int64_t nx = 0, ny = 0, nxmax = 0, nymax = 0;
    int nbasis = 0, nlevel = 0;
    std::vector<int> bases;  // 0 = self, N = dbN
    struct Wav {
        Filt dec, rec;
        std::vector<int64_t> sx, sy, spx, spy, hx, hy;  // per level: coeff sizes, signal sizes, block end indices
        int64_t ntotx = 0, ntoty = 0;
    };
    std::vector<Wav> wav;  // one per basis (unused for self)
    hipStream_t stream = nullptr;
    DevBuf<double> cbuff, img, d_x, d_alpha, d_alphaT;
    DevBuf<ZeroRect> d_zero;  // what dot() has to zero itself: everything else of alpha is written by a transform
    int n_zero = 0;
    ~pfbhip_psi()
    {
        if (stream) (void)hipStreamDestroy(stream);
    }
    size_t cube() const { return size_t(nbasis) * size_t(nxmax) * size_t(nymax); }

    // alpha (nbasis, nxmax, nymax) <- x (nx, ny); every element of alpha is written
    void dot(const double *x, double *alpha)
    {
        if (n_zero > 0)
            hipLaunchKernelGGL(k_zero_rects, dim3(uint32_t(n_zero)), dim3(256), 0, stream, d_zero.p, alpha, size_t(nxmax) * size_t(nymax),
                               size_t(nymax));
        for (int b = 0; b < nbasis; ++b) {
            double *ab = alpha + size_t(b) * size_t(nxmax) * size_t(nymax);
            if (bases[size_t(b)] == 0) {
                hipLaunchKernelGGL(k_copy2d, grid2(ny, nx), dim3(256), 0, stream, x, size_t(ny), int(ny), 0, ab, size_t(nymax));
                continue;
            }
            const Wav &w = wav[size_t(b)];
            const double *in = x;
            size_t ldi = size_t(ny);
            int64_t n0 = nx, n1 = ny;
            for (int i = 0; i < nlevel; ++i) {
                const int64_t sx = w.sx[size_t(i)], sy = w.sy[size_t(i)];
                const int64_t lx = w.hx[size_t(i)] - 2 * sx, ly = w.hy[size_t(i)] - 2 * sy;
                const size_t ldc = size_t(2 * sy);
                hipLaunchKernelGGL(k_dwt_rows, grid2(sy, n0), dim3(256), 0, stream, in, ldi, int(n1), w.dec, int(sy), cbuff.p,
                                   ldc);
                double *blk = ab + size_t(lx) * size_t(nymax) + size_t(ly);
                hipLaunchKernelGGL(k_dwt_cols, grid2(2 * sy, sx), dim3(256), 0, stream, cbuff.p, ldc, int(n0), w.dec, int(sx),
                                   int(2 * sy), blk, size_t(nymax));
                in = blk;  // LL quadrant: top-left (sx, sy) of the block
                ldi = size_t(nymax);
                n0 = sx;
                n1 = sy;
            }
        }
        PFB_HIP(hipGetLastError());
    }

    // x (nx, ny) <- sum over bases of the synthesis of alpha; alpha is not modified
    void hdot(const double *alpha, double *x)
    {
        bool first = true;
        // the identity basis rides on the first wavelet basis' last row pass (if there is a wavelet basis)
        const double *ident = nullptr;
        bool any_wavelet = false;
        for (int b = 0; b < nbasis; ++b) {
            if (bases[size_t(b)] == 0 && ident == nullptr) ident = alpha + size_t(b) * size_t(nxmax) * size_t(nymax);
            any_wavelet = any_wavelet || bases[size_t(b)] != 0;
        }
        const double *ident_used = any_wavelet ? ident : nullptr;
        // The ride only works on a pass that OVERWRITES x (first == true: o = a + extra).  With a second identity basis in front
        // of the first wavelet basis (bases = self, self, db1) that pass accumulates instead, and the slice is added by a copy
        // kernel after the loop.
        bool ident_added = ident_used == nullptr;
        for (int b = 0; b < nbasis; ++b) {
            const double *ab = alpha + size_t(b) * size_t(nxmax) * size_t(nymax);
            if (bases[size_t(b)] == 0) {
                if (ab == ident_used) continue;  // (added by the first wavelet basis below)
                hipLaunchKernelGGL(k_copy2d, grid2(ny, nx), dim3(256), 0, stream, ab, size_t(nymax), int(ny), first ? 0 : 1, x,
                                   size_t(ny));
                first = false;
                continue;
            }
            const Wav &w = wav[size_t(b)];
            for (int i = nlevel - 1; i >= 0; --i) {
                const int64_t sx = w.sx[size_t(i)], sy = w.sy[size_t(i)];
                const int64_t lx = w.hx[size_t(i)] - 2 * sx, ly = w.hy[size_t(i)] - 2 * sy;
                const int64_t nxo = w.spx[size_t(i)], nyo = w.spy[size_t(i)];
                const double *blk = ab + size_t(lx) * size_t(nymax) + size_t(ly);
                const bool deepest = i == nlevel - 1;
                const size_t ldc = size_t(2 * sy);
                // the level below left its reconstruction in img (ld = ny); it replaces this level's LL quadrant
                hipLaunchKernelGGL(k_idwt_cols, grid2(2 * sy, (nxo + 1) / 2), dim3(256), 0, stream, blk, size_t(nymax), int(sx), int(2 * sy),
                                   w.rec, deepest ? static_cast<const double *>(nullptr) : img2(), size_t(ny), int(sy), int(nxo),
                                   cbuff.p, ldc);
                if (i > 0) {
                    // img and img2 alternate so that a level never reads the buffer it writes
                    swap_img();
                    hipLaunchKernelGGL(k_idwt_rows, grid2((nyo + 1) / 2, nxo), dim3(256), 0, stream, cbuff.p, ldc, int(sy), w.rec, int(nyo), 0,
                                       img2(), size_t(ny), static_cast<const double *>(nullptr), size_t(0));
                } else {
                    // (nxo == nx, nyo == ny at level 0: the pass covers the image)
                    const double *extra = (first && !ident_added) ? ident_used : nullptr;
                    hipLaunchKernelGGL(k_idwt_rows, grid2((nyo + 1) / 2, nxo), dim3(256), 0, stream, cbuff.p, ldc, int(sy), w.rec, int(nyo),
                                       first ? 0 : 1, x, size_t(ny), extra, size_t(nymax));
                    if (extra != nullptr) ident_added = true;
                }
            }
            first = false;
        }
        if (!ident_added) {
            hipLaunchKernelGGL(k_copy2d, grid2(ny, nx), dim3(256), 0, stream, ident_used, size_t(nymax), int(ny), first ? 0 : 1, x,
                               size_t(ny));
            first = false;
        }
        PFB_HIP(hipGetLastError());
    }
    // reconstruction scratch: one buffer suffices (k_idwt_cols of level i finishes reading it before
    // k_idwt_rows of level i writes it -- same stream), kept behind an accessor for clarity
    double *img2() { return img.p; }
    void swap_img() {}
};

namespace pfbhip {
hipStream_t psi_swap_stream(pfbhip_psi *p, hipStream_t st)
{
    hipStream_t prev = p->stream;
    p->stream = st;
    return prev;
}
void psi_geometry(const pfbhip_psi *p, int64_t *nx, int64_t *ny, int *nbasis, int64_t *nxmax, int64_t *nymax)
{
    *nx = p->nx;
    *ny = p->ny;
    *nbasis = p->nbasis;
    *nxmax = p->nxmax;
    *nymax = p->nymax;
}
void psi_dot_async(pfbhip_psi *p, const double *x_dev, double *alpha_dev) { p->dot(x_dev, alpha_dev); }
void l21_vtilde_async(const double *vp_dev, double *v_dev, int64_t nband, int64_t n, double sigma, double *sum_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_l21_vtilde, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, vp_dev, v_dev, int(nband), n, sigma, sum_dev);
    PFB_HIP(hipGetLastError());
}
void l21_scale_async(double *v_dev, int64_t nband, int64_t n, double lam, const double *weight_dev, const double *sum_dev,
                     hipStream_t st)
{
    hipLaunchKernelGGL(k_l21_scale, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, v_dev, int(nband), n, lam, weight_dev,
                       sum_dev);
    PFB_HIP(hipGetLastError());
}
void l21_fused_async(const double *vp_dev, double *a_dev, double *ext_dev, int64_t nband, int64_t n, double lam, double sigma,
                     const double *weight_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_l21_fused, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, vp_dev, a_dev, ext_dev, int(nband), n, lam,
                       sigma, weight_dev);
    PFB_HIP(hipGetLastError());
}
void l21_localsum_async(const double *vp_dev, const double *a_dev, int64_t nband, int64_t n, double sigma, double *sum_dev,
                        hipStream_t st)
{
    hipLaunchKernelGGL(k_l21_localsum, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, vp_dev, a_dev, int(nband), n, sigma,
                       sum_dev);
    PFB_HIP(hipGetLastError());
}
void l21_apply_async(const double *vp_dev, double *a_dev, double *ext_dev, int64_t nband, int64_t n, double lam, double sigma,
                     const double *weight_dev, const double *sum_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_l21_apply, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, vp_dev, a_dev, ext_dev, int(nband), n, lam,
                       sigma, weight_dev, sum_dev);
    PFB_HIP(hipGetLastError());
}
void positivity_flag_async(const double *x_dev, int64_t nband, int64_t n, double *bad_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_pos_flag, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, x_dev, int(nband), n, bad_dev);
    PFB_HIP(hipGetLastError());
}
void positivity_zero_async(double *x_dev, int64_t nband, int64_t n, const double *bad_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_pos_zero, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, x_dev, int(nband), n, bad_dev);
    PFB_HIP(hipGetLastError());
}
void positivity_async(double *x_dev, int64_t nband, int64_t n, int mode, hipStream_t st)
{
    hipLaunchKernelGGL(k_positivity, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, x_dev, int(nband), n, mode);
    PFB_HIP(hipGetLastError());
}
void psi_hdot_async(pfbhip_psi *p, const double *alpha_dev, double *x_dev) { p->hdot(alpha_dev, x_dev); }
}  // namespace pfbhip

extern "C" {

int pfbhip_psi_create(int64_t nx, int64_t ny, int32_t nbasis, const int32_t *bases, int32_t nlevel, pfbhip_psi **out)
{
    return guarded([&] {
        PFB_REQUIRE(out && bases && nbasis >= 1 && nlevel >= 1, "bad arguments");
        PFB_REQUIRE(nx >= 2 && ny >= 2 && nx % 2 == 0 && ny % 2 == 0, "image sizes must be even (%lld, %lld)", (long long)nx,
                    (long long)ny);
        std::unique_ptr<pfbhip_psi> p(new pfbhip_psi);
        p->nx = nx;
        p->ny = ny;
        p->nbasis = nbasis;
        p->nlevel = nlevel;
        p->bases.assign(bases, bases + nbasis);
        p->wav.resize(size_t(nbasis));
        int64_t nxmax = nx, nymax = ny, symax = 0;
        for (int b = 0; b < nbasis; ++b) {
            const int N = bases[b];
            PFB_REQUIRE(N >= 0 && N <= 8, "basis %d: only 'self' (0) and db1..db8 are supported", N);
            if (N == 0) continue;
            auto &w = p->wav[size_t(b)];
            make_filters(N, &w.dec, &w.rec);
            const int64_t L = 2 * N;
            // operators/psi.py:73-113
            // pywt.dwt_max_level(min(nx, ny), wavelet) = floor(log2(n / (L - 1)))  (psi.py:66-68)
            int max_level = 0;
            for (int64_t n = std::min(nx, ny) / (L - 1); n >= 2; n /= 2) ++max_level;
            PFB_REQUIRE(nlevel <= max_level, "The requested decomposition level %d is not possible for db%d on (%lld, %lld)", nlevel,
                        N, (long long)nx, (long long)ny);
            int64_t cx = 0, cy = 0, n_x = nx, n_y = ny;
            for (int k = 0; k < nlevel; ++k) {
                cx = (n_x + L - 1) / 2;
                cy = (n_y + L - 1) / 2;
                w.sx.push_back(cx);
                w.sy.push_back(cy);
                w.spx.push_back(2 * cx - L + 2);
                w.spy.push_back(2 * cy - L + 2);
                w.ntotx += cx;
                w.ntoty += cy;
                symax = std::max(symax, cy);
                n_x = cx + cx % 2;
                n_y = cy + cy % 2;
            }
            w.ntotx += cx;
            w.ntoty += cy;
            nxmax = std::max(nxmax, w.ntotx);
            nymax = std::max(nymax, w.ntoty);
            w.hx.assign(size_t(nlevel), 0);
            w.hy.assign(size_t(nlevel), 0);
            int64_t lowx = 2 * w.sx.back(), lowy = 2 * w.sy.back();
            w.hx[size_t(nlevel - 1)] = lowx;
            w.hy[size_t(nlevel - 1)] = lowy;
            for (int k = nlevel - 2; k >= 0; --k) {
                w.hx[size_t(k)] = lowx + w.sx[size_t(k)];
                w.hy[size_t(k)] = lowy + w.sy[size_t(k)];
                lowx += w.sx[size_t(k)];
                lowy += w.sy[size_t(k)];
            }
        }
        p->nxmax = nxmax;
        p->nymax = nymax;
        {   // complement of the written rectangles per slice, on the grid of their boundaries; long rectangles are cut into
            // slabs of 64 rows so that a workgroup's share stays small
            std::vector<ZeroRect> zr;
            for (int b = 0; b < nbasis; ++b) {
                struct R { int64_t x0, x1, y0, y1; };
                std::vector<R> written;
                if (bases[b] == 0) {
                    written.push_back({0, nx, 0, ny});
                } else {
                    const auto &w = p->wav[size_t(b)];
                    for (int k = 0; k < nlevel; ++k)
                        written.push_back({w.hx[size_t(k)] - 2 * w.sx[size_t(k)], w.hx[size_t(k)], w.hy[size_t(k)] - 2 * w.sy[size_t(k)],
                                           w.hy[size_t(k)]});
                }
                std::vector<int64_t> xs{0, nxmax}, ys{0, nymax};
                for (auto &r : written) {
                    xs.push_back(r.x0); xs.push_back(r.x1); ys.push_back(r.y0); ys.push_back(r.y1);
                }
                std::sort(xs.begin(), xs.end()); xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
                std::sort(ys.begin(), ys.end()); ys.erase(std::unique(ys.begin(), ys.end()), ys.end());
                for (size_t i = 0; i + 1 < xs.size(); ++i)
                    for (size_t j = 0; j + 1 < ys.size(); ++j) {
                        if (xs[i] < 0 || xs[i + 1] > nxmax || ys[j] < 0 || ys[j + 1] > nymax) continue;
                        bool covered = false;
                        for (auto &r : written)
                            covered = covered || (xs[i] >= r.x0 && xs[i + 1] <= r.x1 && ys[j] >= r.y0 && ys[j + 1] <= r.y1);
                        if (covered) continue;
                        for (int64_t r0 = xs[i]; r0 < xs[i + 1]; r0 += 64)
                            zr.push_back({b, int(r0), int(std::min<int64_t>(64, xs[i + 1] - r0)), int(ys[j]), int(ys[j + 1] - ys[j])});
                    }
            }
            p->n_zero = int(zr.size());
            if (!zr.empty()) {
                p->d_zero.alloc(zr.size());
                PFB_HIP(hipMemcpy(p->d_zero.p, zr.data(), zr.size() * sizeof(ZeroRect), hipMemcpyHostToDevice));
            }
        }
        PFB_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        p->cbuff.alloc(size_t(nxmax) * size_t(2 * std::max<int64_t>(symax, 1)));
        p->img.alloc(size_t(nx) * size_t(ny));
        *out = p.release();
    });
}

int pfbhip_psi_destroy(pfbhip_psi *p)
{
    return guarded([&] { delete p; });
}

int pfbhip_psi_shape(const pfbhip_psi *p, int64_t *nxmax, int64_t *nymax)
{
    return guarded([&] {
        PFB_REQUIRE(p && nxmax && nymax, "NULL argument");
        *nxmax = p->nxmax;
        *nymax = p->nymax;
    });
}

int pfbhip_psi_dot_dev(pfbhip_psi *p, const double *x_dev, double *alpha_dev)
{
    return guarded([&] {
        PFB_REQUIRE(p && x_dev && alpha_dev, "NULL argument");
        p->dot(x_dev, alpha_dev);
        PFB_HIP(hipStreamSynchronize(p->stream));
    });
}

int pfbhip_psi_hdot_dev(pfbhip_psi *p, const double *alpha_dev, double *x_dev)
{
    return guarded([&] {
        PFB_REQUIRE(p && x_dev && alpha_dev, "NULL argument");
        p->hdot(alpha_dev, x_dev);
        PFB_HIP(hipStreamSynchronize(p->stream));
    });
}

// transposed != 0: alpha_host is (nbasis, nymax, nxmax), the layout of the reference's older Psi (psi.py:273-345)
int pfbhip_psi_dot(pfbhip_psi *p, const double *x_host, double *alpha_host, int transposed)
{
    return guarded([&] {
        PFB_REQUIRE(p && x_host && alpha_host, "NULL argument");
        const size_t n = size_t(p->nx) * size_t(p->ny), plane = size_t(p->nxmax) * size_t(p->nymax);
        p->d_x.ensure(n);
        p->d_alpha.ensure(p->cube());
        PFB_HIP(hipMemcpyAsync(p->d_x.p, x_host, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
        p->dot(p->d_x.p, p->d_alpha.p);
        const double *src = p->d_alpha.p;
        if (transposed) {
            p->d_alphaT.ensure(p->cube());
            for (int b = 0; b < p->nbasis; ++b)
                hipLaunchKernelGGL(k_transpose_f64_any, dim3(uint32_t(ceil_div(p->nymax, 32)), uint32_t(ceil_div(p->nxmax, 32))),
                                   dim3(32, 8), 0, p->stream, p->d_alpha.p + size_t(b) * plane, int(p->nxmax), int(p->nymax),
                                   p->d_alphaT.p + size_t(b) * plane);
            PFB_HIP(hipGetLastError());
            src = p->d_alphaT.p;
        }
        PFB_HIP(hipMemcpyAsync(alpha_host, src, p->cube() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        PFB_HIP(hipStreamSynchronize(p->stream));
    });
}

int pfbhip_psi_hdot(pfbhip_psi *p, const double *alpha_host, double *x_host, int transposed)
{
    return guarded([&] {
        PFB_REQUIRE(p && x_host && alpha_host, "NULL argument");
        const size_t n = size_t(p->nx) * size_t(p->ny), plane = size_t(p->nxmax) * size_t(p->nymax);
        p->d_x.ensure(n);
        p->d_alpha.ensure(p->cube());
        const double *src = p->d_alpha.p;
        if (transposed) {
            p->d_alphaT.ensure(p->cube());
            PFB_HIP(hipMemcpyAsync(p->d_alphaT.p, alpha_host, p->cube() * sizeof(double), hipMemcpyHostToDevice, p->stream));
            for (int b = 0; b < p->nbasis; ++b)
                hipLaunchKernelGGL(k_transpose_f64_any, dim3(uint32_t(ceil_div(p->nxmax, 32)), uint32_t(ceil_div(p->nymax, 32))),
                                   dim3(32, 8), 0, p->stream, p->d_alphaT.p + size_t(b) * plane, int(p->nymax), int(p->nxmax),
                                   p->d_alpha.p + size_t(b) * plane);
            PFB_HIP(hipGetLastError());
        } else {
            PFB_HIP(hipMemcpyAsync(p->d_alpha.p, alpha_host, p->cube() * sizeof(double), hipMemcpyHostToDevice, p->stream));
        }
        p->hdot(src, p->d_x.p);
        PFB_HIP(hipMemcpyAsync(x_host, p->d_x.p, n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        PFB_HIP(hipStreamSynchronize(p->stream));
    });
}

// ---- l21 dual update / prox / positivity on device arrays (default stream) ---------------------------

int pfbhip_l21_vtilde_sum_dev(const double *vp_dev, double *v_dev, int64_t nband, int64_t n, double sigma, double *sum_dev)
{
    return guarded([&] {
        PFB_REQUIRE(vp_dev && v_dev && sum_dev && nband >= 1 && n >= 0, "bad arguments");
        if (n == 0) return;
        hipLaunchKernelGGL(k_l21_vtilde, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, nullptr, vp_dev, v_dev, int(nband), n,
                           sigma, sum_dev);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipStreamSynchronize(nullptr));
    });
}

int pfbhip_l21_scale_dev(double *v_dev, int64_t nband, int64_t n, double lam, const double *weight_dev, const double *sum_dev)
{
    return guarded([&] {
        PFB_REQUIRE(v_dev && weight_dev && sum_dev && nband >= 1 && n >= 0, "bad arguments");
        if (n == 0) return;
        hipLaunchKernelGGL(k_l21_scale, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, nullptr, v_dev, int(nband), n, lam,
                           weight_dev, sum_dev);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipStreamSynchronize(nullptr));
    });
}

// host arrays: vp, v (nband, n), weight (n); v <- dual update in place (all bands on this device)
int pfbhip_dual_update(const double *vp_host, double *v_host, int64_t nband, int64_t n, double lam, double sigma,
                       const double *weight_host)
{
    return guarded([&] {
        PFB_REQUIRE(vp_host && v_host && weight_host && nband >= 1 && n >= 0, "bad arguments");
        if (n == 0) return;
        const size_t tot = size_t(nband) * size_t(n);
        DevBuf<double> vp(tot), v(tot), w{size_t(n)}, s{size_t(n)};
        PFB_HIP(hipMemcpy(vp.p, vp_host, tot * sizeof(double), hipMemcpyHostToDevice));
        PFB_HIP(hipMemcpy(v.p, v_host, tot * sizeof(double), hipMemcpyHostToDevice));
        PFB_HIP(hipMemcpy(w.p, weight_host, size_t(n) * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_l21_vtilde, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, nullptr, vp.p, v.p, int(nband), n, sigma,
                           s.p);
        hipLaunchKernelGGL(k_l21_scale, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, nullptr, v.p, int(nband), n, lam, w.p, s.p);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipMemcpy(v_host, v.p, tot * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int pfbhip_prox_21m(const double *v_host, int64_t nband, int64_t n, double sigma, const double *weight_host, double *out_host)
{
    return guarded([&] {
        PFB_REQUIRE(v_host && out_host && nband >= 1 && n >= 0, "bad arguments");
        if (n == 0) return;
        const size_t tot = size_t(nband) * size_t(n);
        DevBuf<double> v(tot), o(tot), w;
        PFB_HIP(hipMemcpy(v.p, v_host, tot * sizeof(double), hipMemcpyHostToDevice));
        if (weight_host) {
            w.alloc(size_t(n));
            PFB_HIP(hipMemcpy(w.p, weight_host, size_t(n) * sizeof(double), hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(k_prox21, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, nullptr, v.p, int(nband), n, sigma, w.p, o.p);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipMemcpy(out_host, o.p, tot * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int pfbhip_positivity(double *x_host, int64_t nband, int64_t n, int mode)
{
    return guarded([&] {
        PFB_REQUIRE(x_host && nband >= 1 && n >= 0 && (mode == 1 || mode == 2), "bad arguments");
        if (n == 0) return;
        const size_t tot = size_t(nband) * size_t(n);
        DevBuf<double> x(tot);
        PFB_HIP(hipMemcpy(x.p, x_host, tot * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_positivity, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, nullptr, x.p, int(nband), n, mode);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipMemcpy(x_host, x.p, tot * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int pfbhip_positivity_dev(double *x_dev, int64_t nband, int64_t n, int mode)
{
    return guarded([&] {
        PFB_REQUIRE(x_dev && nband >= 1 && n >= 0 && (mode == 1 || mode == 2), "bad arguments");
        if (n == 0) return;
        hipLaunchKernelGGL(k_positivity, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, nullptr, x_dev, int(nband), n, mode);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipStreamSynchronize(nullptr));
    });
}

}  // extern "C"
