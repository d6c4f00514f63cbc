// eskernel.cpp -- ES gridding-kernel table, its Fourier transform (the image-domain
// grid correction) and the host-side numerics that go with it.
//
// Kernel form: exp(beta (sqrt(1-x^2) - 1)), the expression the reference writes at
// /root/reference/src/pfb_imaging/utils/weighting.py:25-35.  The (W, sigma, beta, eps)
// table is this build's own (tools/make_kernel_table.py); ducc0's table is not available.
#include "eskernel.hpp"

#include <algorithm>
#include <utility>
#include <cmath>

namespace pfbhip {

static const KernelRow k_table[] = {
#include "es_kernel_table.inc"
};

const KernelRow *kernel_table(size_t *n)
{
    *n = sizeof(k_table) / sizeof(k_table[0]);
    return k_table;
}

void gauss_legendre(int n, std::vector<double> &x, std::vector<double> &w)
{
    x.assign(n, 0.0);
    w.assign(n, 0.0);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(pi * (i + 0.75) / (n + 0.5)), pp = 0.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 0; j < n; ++j) {
                double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            double z1 = z;
            z = z1 - p1 / pp;
            if (std::fabs(z - z1) < 1e-16) break;
        }
        x[i] = -z;
        x[n - 1 - i] = z;
        w[i] = w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

KernelFT::KernelFT(int W_, double beta_) : W(W_), beta(beta_)
{
    // (the nodes are the same for every kernel row: the plan's row loop constructs ~200 of these)
    static const std::pair<std::vector<double>, std::vector<double>> gl = [] {
        std::pair<std::vector<double>, std::vector<double>> p;
        gauss_legendre(96, p.first, p.second);
        return p;
    }();
    const std::vector<double> &gx = gl.first, &gw = gl.second;
    s.resize(gx.size());
    pw.resize(gx.size());
    for (size_t i = 0; i < gx.size(); ++i) {
        s[i] = 0.5 * (gx[i] + 1.0);
        pw[i] = 0.5 * gw[i] * std::exp(beta * (std::sqrt(1.0 - s[i] * s[i]) - 1.0));
    }
}

double KernelFT::operator()(double v) const
{
    const double pi = 3.14159265358979323846;
    double acc = 0.0;
    for (size_t i = 0; i < s.size(); ++i) acc += pw[i] * std::cos(pi * W * s[i] * v);
    return W * acc;
}

std::vector<double> KernelFT::correction_1d(int64_t npix, int64_t ngrid) const
{
    // psi is even: one evaluation (96 cosines) per |i - npix / 2|
    std::vector<double> cf(npix);
    const int64_t h = npix / 2;
    for (int64_t k = 0; k <= h; ++k) {
        const double v = 1.0 / (*this)(double(k) / double(ngrid));
        if (h - k >= 0) cf[size_t(h - k)] = v;
        if (h + k < npix) cf[size_t(h + k)] = v;
    }
    return cf;
}

// Chebyshev fit of 1/psi(z) on |z| <= zmax as a polynomial in y = 2 (z/zmax)^2 - 1.
std::vector<double> KernelFT::inverse_cheb(double zmax, double tol) const
{
    const double pi = 3.14159265358979323846;
    for (int deg = 8; deg <= 96; deg += 8) {
        int n = deg + 1;
        std::vector<double> f(n), c(n);
        for (int k = 0; k < n; ++k) {
            double y = std::cos(pi * (k + 0.5) / n);
            double z = zmax * std::sqrt(0.5 * (y + 1.0));
            f[k] = 1.0 / (*this)(z);
        }
        for (int j = 0; j < n; ++j) {
            double sum = 0.0;
            for (int k = 0; k < n; ++k) sum += f[k] * std::cos(pi * j * (k + 0.5) / n);
            c[j] = 2.0 * sum / n;
        }
        c[0] *= 0.5;
        double tail = std::max(std::fabs(c[n - 1]), std::max(std::fabs(c[n - 2]), std::fabs(c[n - 3])));
        if (tail <= tol * std::fabs(c[0]) || deg == 96) {
            // trim negligible trailing coefficients
            while (c.size() > 1 && std::fabs(c.back()) <= 0.1 * tol * std::fabs(c[0])) c.pop_back();
            return c;
        }
    }
    return {};
}

std::vector<double> kernel_poly_table(int W, double beta, double *max_err)
{
    const int D = kernel_poly_degree(W), n = D + 1;
    const long double pi = 3.141592653589793238462643383279502884L;
    auto phi = [&](long double x) -> long double {
        long double t = 1.0L - x * x;
        return t >= 0.0L ? expl((long double)beta * (sqrtl(t) - 1.0L)) : 0.0L;
    };
    std::vector<double> tab(size_t(W) * n);
    double worst = 0.0;
    for (int a = 0; a < W; ++a) {
        auto f_of_z = [&](long double z) {
            long double f = 0.5L * (z + 1.0L);
            return phi(((long double)a + 1.0L - 0.5L * W - f) * 2.0L / W);
        };
        // Chebyshev coefficients on z in [-1,1]
        std::vector<long double> fv(n), cc(n);
        for (int k = 0; k < n; ++k) fv[k] = f_of_z(cosl(pi * (k + 0.5L) / n));
        for (int j = 0; j < n; ++j) {
            long double sum = 0.0L;
            for (int k = 0; k < n; ++k) sum += fv[k] * cosl(pi * j * (k + 0.5L) / n);
            cc[j] = 2.0L * sum / n;
        }
        cc[0] *= 0.5L;
        // to monomials: T_0 = 1, T_1 = z, T_{j+1} = 2 z T_j - T_{j-1}
        std::vector<long double> mono(n, 0.0L), t0(n, 0.0L), t1(n, 0.0L), t2(n, 0.0L);
        t0[0] = 1.0L;
        t1[1] = 1.0L;
        for (int k = 0; k < n; ++k) mono[k] += cc[0] * t0[k];
        if (n > 1)
            for (int k = 0; k < n; ++k) mono[k] += cc[1] * t1[k];
        for (int j = 2; j < n; ++j) {
            for (int k = 0; k < n; ++k) t2[k] = (k > 0 ? 2.0L * t1[k - 1] : 0.0L) - t0[k];
            for (int k = 0; k < n; ++k) mono[k] += cc[j] * t2[k];
            t0 = t1;
            t1 = t2;
        }
        for (int k = 0; k < n; ++k) tab[size_t(a) * n + k] = (double)mono[k];
        // measured error of the double-precision Horner form
        for (int q = 0; q <= 400; ++q) {
            double z = -1.0 + 2.0 * q / 400.0;
            double v = tab[size_t(a) * n + D];
            for (int k = D - 1; k >= 0; --k) v = v * z + tab[size_t(a) * n + k];
            worst = std::max(worst, std::fabs(v - (double)f_of_z((long double)z)));
        }
    }
    if (max_err) *max_err = worst;
    return tab;
}

int64_t good_size_2357(int64_t n)
{
    if (n < 1) n = 1;
    for (;; ++n) {
        int64_t m = n;
        for (int p : {2, 3, 5, 7})
            while (m % p == 0) m /= p;
        if (m == 1) return n;
    }
}

int64_t good_size(int64_t n, bool real)
{
    if (n < 1) n = 1;
    for (;; ++n) {
        int64_t m = n;
        for (int p : {2, 3, 5})
            while (m % p == 0) m /= p;
        if (!real)
            for (int p : {7, 11})
                while (m % p == 0) m /= p;
        if (m == 1) return n;
    }
}

int64_t grid_size(int64_t npix, double sigma)
{
    int64_t half = (int64_t)std::ceil(0.5 * sigma * double(npix) - 1e-9);
    return std::max<int64_t>(2 * good_size_2357(half), 32);
}

}  // namespace pfbhip
