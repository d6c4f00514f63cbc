// rowfft_api.hpp -- host-callable launchers of the hand-written row FFT (defined in rowfft.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rowfft.hpp"

namespace pfbhip {

// Host-side owner of a plan: radix schedule + the device twiddle table.
struct RowFFT {
    RowFFTPlan pl;
    double2 *d_tw = nullptr;
    size_t d_tw_bytes = 0;
    bool ok = false;
    bool init(int64_t N);  // false if N is not supported
    void release();
    ~RowFFT() { release(); }
};

// pitch: complex elements between consecutive rows (0: dense, = the row length)
void rowfft_plain(const RowFFTPlan &pl, double2 *data_dev, int nrows, bool inverse, hipStream_t stream, size_t pitch = 0);

// First-axis transforms of the occupied rows of A (nu, nv = pl.N) with the crop / pad + transpose folded into the store /
// load: rowmap_dev[b] is the row of workgroup b (see rowfft.hip: the 8 rows of a 128-byte line of B share an XCD); nu here is the
// row pitch of B, apitch that of A (complex elements).
//   a2b: B[y][u] = IFFT_v(A[u][:])[wrap(y - ny/2)]   b2a: A[u][:] = FFT_v(v -> B[y(v)][u], 0 outside the image)
// nplanes planes per launch, astride / bstride elements apart in A / B; colruns_dev[u >> 5] = the (at most two) column runs
// [x, y) and [z, w) of A that rows of tile row u >> 5 use (loads outside them read as zero, stores outside them are dropped)
void rowfft_a2b(const RowFFTPlan &pl, const double2 *A_dev, double2 *B_dev, const int *rowmap_dev, int nrows, int nu, int ny,
                size_t apitch, int nplanes, size_t astride, size_t bstride, const int4 *colruns_dev, hipStream_t stream);
// tpitch > 0: B_dev is Bt[u][y] (tpitch elements per row u, FusedGeom::tpitch): contiguous loads, no transpose here
void rowfft_b2a(const RowFFTPlan &pl, const double2 *B_dev, double2 *A_dev, const int *rowmap_dev, int nrows, int nu, int ny,
                size_t apitch, int tpitch, int nplanes, size_t astride, size_t bstride, const int4 *colruns_dev, hipStream_t stream);

// Geometry of the second-axis (u) pass of the gridder's plane transform.
constexpr int FUSED_MAXPOLY = 20;
constexpr int FUSED_TSHORT = 8;  // coefficient count up to which the short Horner chain serves (fields up to ~10 degrees)
struct FusedGeom {
    int nx, ny, nu;
    int bpitch;  // complex elements between consecutive rows of B (>= nu; padded off the power-of-two pitch, see gridder.hip)
    int tpitch;  // > 0: the fused pad kernel stores TRANSPOSED, Bt[u][y] with this many elements per row u (0: B[y][u])
    double px, py, lshift, mshift, nshift;
    // n - 1 = sqrt(1 - r2) - 1 as a polynomial in s = r2 * za + zb in [-1, 1]: npoly coefficients, highest first, RIGHT-ALIGNED
    // in pc (pc[FUSED_MAXPOLY + 1 - npoly ..], zeros in front: the kernels run fixed-length Horner chains over the last
    // FUSED_TSHORT or all FUSED_MAXPOLY + 1 entries, see fg_t); npoly = 0: evaluate the square root (fields wider than 45
    // degrees; the fused kernels are not used then).  Filled by fused_geom_fit().
    int npoly = 0;
    double za = 0.0, zb = 0.0;
    double pc[FUSED_MAXPOLY + 1] = {};
};
// Fit the polynomial of g.pc over the field of view of g (absolute error <= 4e-16 max|n - 1|), or leave npoly = 0.
void fused_geom_fit(FusedGeom &g);
// Image-side element-wise steps folded into the fused kernels (the plan's accT layout is the caller's image
// layout).  FusedPrep: the degrid input row is x * corr [* beam] read straight from the caller's image instead
// of a prepared copy.  FusedFinal: the LAST grid launch writes out = sum * corr [* beam] * scale + eta * x
// instead of the raw accumulator.
struct FusedPrep {
    const double *x = nullptr, *corr = nullptr, *beam = nullptr;  // x == NULL: disabled (read the prepared accT)
};
struct FusedFinal {
    const double *corr = nullptr, *beam = nullptr, *x = nullptr;  // corr == NULL: disabled (write accT)
    double scale = 1.0, eta = 0.0;
    double *out = nullptr;
};
constexpr int FUSED_MAXPLANES = 4;
constexpr int FUSED_SCMAX = 8;  // most coefficients a composite / residual screen polynomial may have (FusedPlanes::nsc of them are evaluated)
struct FusedPlanes {
    int kp;
    double w[FUSED_MAXPLANES];  // w of each plane (wavelengths)
    // Narrow fields / small w: cos and sin of the w-screen phase 2 pi w_k (n - 1 + nshift) are themselves low-degree
    // polynomials in s = r2 * za + zb (the abscissa of FusedGeom's n - 1 polynomial): FUSED_SCMAX coefficients each, highest
    // first (leading zeros), fitted to a Chebyshev tail of 1e-14 and verified to 1e-13 by fused_planes_fit(); nsc = 4, 6 or
    // FUSED_SCMAX = how many of them the kernels evaluate (the rest are the leading zeros); nsc = 0: evaluate n - 1, then
    // sincos (the general path).
    int nsc = 0;
    double cs[FUSED_MAXPLANES][FUSED_SCMAX] = {};
    double sn[FUSED_MAXPLANES][FUSED_SCMAX] = {};
    // Larger phases (ES-kernel plane stacks; round 3): SEPARABLE form.  With n - 1 = -z/2 + R(z), z = l^2 + m^2, R = O(z^2 / 8),
    //   exp(2 pi i w (n - 1 + nshift)) = rho(y) * tau[ix] * exp(2 pi i w R(z)),
    // rho = cis(2 pi w (nshift - m^2 / 2)) one sincos per row and plane, tau[ix] = cis(-pi w l(ix)^2) a per-plane TABLE over
    // the image columns (fused_screen_table; nx entries per plane, L2-resident, the same for every row) and the residual a
    // short polynomial pair in s again (cs / sn / nsc above, fitted to w R(z)).  sep = 1: cs / sn hold the residual and
    // tau points at the table of plane 0 of this group (plane k at tau + k * nx).
    int sep = 0;
    const double2 *tau = nullptr;
};
// Fills pl.nsc / cs / sn for the planes pl.w[0..kp) over the field of view of g (g.npoly > 0 required), or leaves nsc = 0.
// residual = false: the whole screen (pl.sep = 0); true: the residual of the separable form (pl.sep = 1; the caller sets tau).
void fused_planes_fit(const FusedGeom &g, FusedPlanes &pl, bool residual = false);
// tau[p * g.nx + ix] = cis(-pi w[p] l(ix)^2) for nplanes planes (device arrays)
void fused_screen_table(const FusedGeom &g, const double *w_dev, int nplanes, double2 *tau_dev, hipStream_t stream);

// grid side: for every image row y and every plane k < kp: inverse row FFT of B_k[y][:] (blocks of 32
// columns that are not occupied are taken as zero without being read), then
// accT[y][x] (+)= Re( out[wrap(x - nx/2)] * exp(-2 pi i w_k t(x, y)) ).  first: plane 0 overwrites accT.
void fused_fft_crop(const RowFFT &f, const FusedGeom &g, const uint8_t *occ_dev, const double2 *B_dev, size_t bstride,
                    const FusedPlanes &pl, int do_w, bool first, double *accT_dev, const FusedFinal &fin, hipStream_t stream);
// degrid side: for every image row y and plane k: B_k[y][wrap(x - nx/2)] = dcT[y][x] exp(+2 pi i w_k t), 0
// elsewhere, forward row FFT, and only the occupied 32-column blocks of the result are written.
// (prep.x != NULL only where fused_pad_takes_prep(): the kernel then keeps the prepared row in LDS; elsewhere -- no room for
// the row, doubled shapes -- the caller passes a prepared image)
bool fused_pad_takes_prep(const RowFFT &f, const FusedGeom &g);
// doubled shapes whose fused kernels park the waiting half transform in LDS (20480 points): fused by default; the others
// (24576, 32768: the half stays in registers and spills) only with PFBHIP_FUSED_DOUBLED=1
bool fused_doubled_stashes(const RowFFT &f);
void fused_pad_fft(const RowFFT &f, const FusedGeom &g, const uint8_t *occ_dev, const double *dcT_dev, const FusedPrep &prep,
                   const FusedPlanes &pl, int do_w, double2 *B_dev, size_t bstride, hipStream_t stream);

#if defined(__HIPCC__)
// ---- w-screen helpers shared by the fused kernels (rowfft.hip) and the separate pad / crop kernels (gridder.hip) ----
// n - 1 + nshift at image pixel (ix, iy): polynomial in r2 where the field is narrow enough for
// fused_geom_fit(), the numerically stable closed form otherwise
// Round 3: the polynomial used to be a run-time loop over g.pc -- one dependent scalar load from the kernel-argument segment
// per coefficient and pixel (C5's fused second axis spent a third of its time there).  The chains are now of fixed length with
// compile-time indices, so the coefficients are loop-invariant scalars.
template <int NC>
__device__ __forceinline__ double fg_t_horner(const FusedGeom &g, double sv)
{
    double acc = g.pc[FUSED_MAXPOLY + 1 - NC];
#pragma unroll
    for (int k = FUSED_MAXPOLY + 2 - NC; k <= FUSED_MAXPOLY; ++k) acc = acc * sv + g.pc[k];
    return acc;
}
// (g.npoly > 0: what the fused kernels require)
__device__ __forceinline__ double fg_t_poly(const FusedGeom &g, int ix, int iy)
{
    const double l = g.lshift + double(ix - g.nx / 2) * g.px;
    const double m = g.mshift + double(iy - g.ny / 2) * g.py;
    const double sv = (l * l + m * m) * g.za + g.zb;
    return (g.npoly <= FUSED_TSHORT ? fg_t_horner<FUSED_TSHORT>(g, sv) : fg_t_horner<FUSED_MAXPOLY + 1>(g, sv)) + g.nshift;
}
__device__ __forceinline__ double fg_t(const FusedGeom &g, int ix, int iy)
{
    if (g.npoly > 0) return fg_t_poly(g, ix, iy);
    const double l = g.lshift + double(ix - g.nx / 2) * g.px;
    const double m = g.mshift + double(iy - g.ny / 2) * g.py;
    const double r2 = l * l + m * m;
    if (r2 <= 1.0) return -r2 / (1.0 + sqrt(1.0 - r2)) + g.nshift;
    return -sqrt(r2 - 1.0) - 1.0 + g.nshift;
}

// sin(2 pi r) and cos(2 pi r) for |r| <= 1/2 (the caller has removed the integer part): one fold to
// |q| <= 1/4, then the Taylor polynomials on |x| <= pi/2 (truncation < 2e-17).  About half the
// instructions of sincospi(), which repeats the range reduction and handles specials.
__device__ __forceinline__ void fg_sincos2pi(double r, double &sn, double &cs)
{
    const bool fold = fabs(r) > 0.25;
    const double q = fold ? copysign(0.5, r) - r : r;
    const double x = q * 6.283185307179586476925286766559;
    const double x2 = x * x;
    double ps = 1.0 / 51090942171709440000.0;   // 21!
    ps = 1.0 / 121645100408832000.0 - ps * x2;  // 19!
    ps = 1.0 / 355687428096000.0 - ps * x2;     // 17!
    ps = 1.0 / 1307674368000.0 - ps * x2;       // 15!
    ps = 1.0 / 6227020800.0 - ps * x2;          // 13!
    ps = 1.0 / 39916800.0 - ps * x2;            // 11!
    ps = 1.0 / 362880.0 - ps * x2;              // 9!
    ps = 1.0 / 5040.0 - ps * x2;                // 7!
    ps = 1.0 / 120.0 - ps * x2;                 // 5!
    ps = 1.0 / 6.0 - ps * x2;                   // 3!
    sn = x - x * x2 * ps;
    double pc = 1.0 / 1124000727777607680000.0;   // 22!
    pc = 1.0 / 2432902008176640000.0 - pc * x2;   // 20!
    pc = 1.0 / 6402373705728000.0 - pc * x2;      // 18!
    pc = 1.0 / 20922789888000.0 - pc * x2;        // 16!
    pc = 1.0 / 87178291200.0 - pc * x2;           // 14!
    pc = 1.0 / 479001600.0 - pc * x2;             // 12!
    pc = 1.0 / 3628800.0 - pc * x2;               // 10!
    pc = 1.0 / 40320.0 - pc * x2;                 // 8!
    pc = 1.0 / 720.0 - pc * x2;                   // 6!
    pc = 1.0 / 24.0 - pc * x2;                    // 4!
    pc = 0.5 - pc * x2;                           // 2!
    const double c0 = 1.0 - x2 * pc;
    cs = fold ? -c0 : c0;
}

// cos / sin of the screen phase at pixel (ix, iy) from a plane's composite polynomials (FusedPlanes::nsc > 0; the caller
// holds the plane's coefficients in registers)
// nsc (wave-uniform: 4, 6 or FUSED_SCMAX) = coefficients that are not leading zeros (FusedPlanes::nsc): the Horner chains start
// there (round 3: C2's screens need 5-6 coefficients at the 1e-13 the fit is held to, not the 8 the kernels used to evaluate)
template <int N>
__device__ __forceinline__ void fg_screen_horner(const double (&cc)[FUSED_SCMAX], const double (&ss)[FUSED_SCMAX], double sv, double &sn,
                                                 double &cs)
{
    double c = cc[FUSED_SCMAX - N], s = ss[FUSED_SCMAX - N];
#pragma unroll
    for (int j = FUSED_SCMAX - N + 1; j < FUSED_SCMAX; ++j) {
        c = c * sv + cc[j];
        s = s * sv + ss[j];
    }
    cs = c;
    sn = s;
}
__device__ __forceinline__ void fg_screen_poly(const FusedGeom &g, const double (&cc)[FUSED_SCMAX], const double (&ss)[FUSED_SCMAX],
                                               int ix, int iy, double &sn, double &cs, int nsc = FUSED_SCMAX)
{
    const double l = g.lshift + double(ix - g.nx / 2) * g.px;
    const double m = g.mshift + double(iy - g.ny / 2) * g.py;
    const double sv = (l * l + m * m) * g.za + g.zb;
    if (nsc <= 4) fg_screen_horner<4>(cc, ss, sv, sn, cs);
    else if (nsc <= 6) fg_screen_horner<6>(cc, ss, sv, sn, cs);
    else fg_screen_horner<FUSED_SCMAX>(cc, ss, sv, sn, cs);
}

// separable form (FusedPlanes::sep): the row factor rho of plane w at image row iy, and the screen (sn, cs) of a pixel from the
// residual polynomials, the table value and rho
__device__ __forceinline__ void fg_row_factor(const FusedGeom &g, double w, int iy, double &rs, double &rc)
{
    const double m = g.mshift + double(iy - g.ny / 2) * g.py;
    double ph = w * (g.nshift - 0.5 * (m * m));
    ph -= rint(ph);
    fg_sincos2pi(ph, rs, rc);
}
__device__ __forceinline__ void fg_screen_sep(const FusedGeom &g, const double (&cc)[FUSED_SCMAX], const double (&ss)[FUSED_SCMAX],
                                              int ix, int iy, double2 tau, double rs, double rc, double &sn, double &cs, int nsc)
{
    double s, c;
    fg_screen_poly(g, cc, ss, ix, iy, s, c, nsc);
    const double tc = rc * tau.x - rs * tau.y, ts = rc * tau.y + rs * tau.x;
    cs = tc * c - ts * s;
    sn = tc * s + ts * c;
}

// image column of uv-column u (-1: u lies in the zero padding)
__device__ __forceinline__ int fg_ix(const FusedGeom &g, int u)
{
    const int hx = g.nx / 2;
    if (u < g.nx - hx) return u + hx;
    if (u >= g.nu - hx) return u - (g.nu - hx);
    return -1;
}

#endif  // __HIPCC__

}  // namespace pfbhip
