// eskernel.hpp -- ES gridding kernel table and Fourier-transform helpers (host side).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "common.hpp"

namespace pfbhip {

struct KernelRow {
    int W;
    double sigma, beta;
    double eps;      // 1-D aliasing error, L2 average over the image
    double eps_max;  // 1-D aliasing error at the image edge (its maximum); like eps an RMS over the sub-cell position
    double eps_sup;  // ... for the WORST sub-cell position: the bound that holds for a single visibility (admissibility test)
};

const KernelRow *kernel_table(size_t *n);

// psi(v) = W int_0^1 phi(s) cos(pi W v s) ds : Fourier transform of the width-W kernel.
struct KernelFT {
    int W;
    double beta;
    std::vector<double> s, pw;
    KernelFT(int W, double beta);
    double operator()(double v) const;
    // 1/psi((i - npix/2)/ngrid), i = 0..npix-1
    std::vector<double> correction_1d(int64_t npix, int64_t ngrid) const;
    // Chebyshev coefficients c_j of 1/psi(z) = sum_j c_j T_j(2 (z/zmax)^2 - 1), |z| <= zmax
    std::vector<double> inverse_cheb(double zmax, double tol = 1e-15) const;
};

int64_t grid_size(int64_t npix, double sigma);

// Piecewise-polynomial form of the kernel for the device: for tap a (0 <= a < W) and sub-cell
// offset f in [0,1) (first tap index i0 = floor(p + 1 - W/2), f = p + 1 - W/2 - i0)
//     phi_a(f) = phi((a + 1 - W/2 - f) * 2 / W)  ~=  sum_k c[a][k] z^k ,  z = 2 f - 1 ,
// degree D = 12 (Chebyshev interpolation, converted to monomials in long double).
// Returns c as a row-major (W, D+1) table and the measured max abs error (kernel peak = 1).
constexpr int kernel_poly_degree(int) { return 12; }
std::vector<double> kernel_poly_table(int W, double beta, double *max_err);

}  // namespace pfbhip
