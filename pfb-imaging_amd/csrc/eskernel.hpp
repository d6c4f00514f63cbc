// eskernel.hpp -- ES gridding kernel table and Fourier-transform helpers (host side).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "common.hpp"

namespace pfbhip {

struct KernelRow {
    int W;
    double sigma, beta, eps;
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

}  // namespace pfbhip
