// rowfft_api.hpp -- host-callable launchers of the hand-written row FFT (defined in rowfft.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rowfft.hpp"

namespace pfbhip {

// Host-side owner of a plan: radix schedule + the device twiddle table.
struct RowFFT {
    RowFFTPlan pl;
    double2 *d_tw = nullptr;
    bool ok = false;
    bool init(int64_t N);  // false if N is not supported
    void release();
    ~RowFFT() { release(); }
};

void rowfft_plain(const RowFFTPlan &pl, double2 *data_dev, int nrows, bool inverse, hipStream_t stream);

// Geometry of the second-axis (u) pass of the gridder's plane transform.
constexpr int FUSED_MAXPOLY = 20;
struct FusedGeom {
    int nx, ny, nu;
    double px, py, lshift, mshift, nshift;
    // n - 1 = sqrt(1 - r2) - 1 as a polynomial in s = r2 * za + zb in [-1, 1] (npoly coefficients, highest
    // first); npoly = 0: evaluate the square root (wide fields).  Filled by fused_geom_fit().
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
struct FusedPlanes {
    int kp;
    double w[FUSED_MAXPLANES];  // w of each plane (wavelengths)
};

// grid side: for every image row y and every plane k < kp: inverse row FFT of B_k[y][:] (blocks of 32
// columns that are not occupied are taken as zero without being read), then
// accT[y][x] (+)= Re( out[wrap(x - nx/2)] * exp(-2 pi i w_k t(x, y)) ).  first: plane 0 overwrites accT.
void fused_fft_crop(const RowFFT &f, const FusedGeom &g, const uint8_t *occ_dev, const double2 *B_dev, size_t bstride,
                    const FusedPlanes &pl, int do_w, bool first, double *accT_dev, const FusedFinal &fin, hipStream_t stream);
// degrid side: for every image row y and plane k: B_k[y][wrap(x - nx/2)] = dcT[y][x] exp(+2 pi i w_k t), 0
// elsewhere, forward row FFT, and only the occupied 32-column blocks of the result are written.
void fused_pad_fft(const RowFFT &f, const FusedGeom &g, const uint8_t *occ_dev, const double *dcT_dev, const FusedPrep &prep,
                   const FusedPlanes &pl, int do_w, double2 *B_dev, size_t bstride, hipStream_t stream);

}  // namespace pfbhip
