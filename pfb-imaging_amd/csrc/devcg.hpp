// devcg.hpp -- conjugate gradients with every vector resident in HBM.
//
// Mirrors pcg_numba without preconditioner (/root/reference/src/pfb_imaging/opt/pcg.py:88-199,
// fused kernels :23-85): r = A x0 - b; p = -r; per iteration alpha = (r.r)/(p.Ap),
// x += alpha p, r += alpha Ap, beta = (r.r)_new/(r.r)_old, p = beta p - r; stop when
// eps = ||x - xp|| / ||x|| <= tol (and k >= minit), or k == maxit, or 5 stalls
// (|eps_prev - eps| < 1e-3 tol).  ||x - xp|| is alpha ||p|| analytically; ||x||^2 is floored
// at 1e-12 like _nb_norm_diff.  Reductions are two-stage (per-block partials, summed on the
// host in a fixed order), so results are run-to-run reproducible.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "common.hpp"

// the kernels below are internal to each translation unit that includes this header; not every unit uses all of them
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunneeded-internal-declaration"
#pragma clang diagnostic ignored "-Wunused-function"

namespace pfbhip {

constexpr int CG_BLOCKS = 1024;
constexpr int CG_THREADS = 256;

template <int NS>
__device__ __forceinline__ void block_reduce_store(double (&v)[NS], double *partials)
{
    __shared__ double sm[NS][CG_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double t = v[s];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) sm[s][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double t = 0.0;
        for (int w = 0; w < CG_THREADS / 64; ++w) t += sm[threadIdx.x][w];
        partials[size_t(threadIdx.x) * CG_BLOCKS + blockIdx.x] = t;
    }
}

// partials: [0] = a.b, [1] = c.d
static __global__ void __launch_bounds__(CG_THREADS) k_cg_dot2(int64_t n, const double *a, const double *b,
                                                                const double *c, const double *d, double *partials)
{
    double v[2] = {0.0, 0.0};
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < n; i += int64_t(CG_BLOCKS) * CG_THREADS) {
        v[0] += a[i] * b[i];
        v[1] += c[i] * d[i];
    }
    block_reduce_store<2>(v, partials);
}

// r = ax - b ; p = -r ; partials [0] = r.r, [1] = any(r != 0)
static __global__ void __launch_bounds__(CG_THREADS) k_cg_init(int64_t n, const double *ax, const double *b, double *r,
                                                                double *p, double *partials)
{
    double v[2] = {0.0, 0.0};
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < n; i += int64_t(CG_BLOCKS) * CG_THREADS) {
        double ri = ax[i] - b[i];
        r[i] = ri;
        p[i] = -ri;
        v[0] += ri * ri;
        v[1] += (ri != 0.0) ? 1.0 : 0.0;
    }
    block_reduce_store<2>(v, partials);
}

// x += alpha p ; r += alpha ap ; partials [0] = r.r, [1] = p.p, [2] = x.x
static __global__ void __launch_bounds__(CG_THREADS) k_cg_update(int64_t n, double alpha, const double *p,
                                                                  const double *ap, double *x, double *r,
                                                                  double *partials)
{
    double v[3] = {0.0, 0.0, 0.0};
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < n; i += int64_t(CG_BLOCKS) * CG_THREADS) {
        double pi = p[i];
        double xi = x[i] + alpha * pi;
        double ri = r[i] + alpha * ap[i];
        x[i] = xi;
        r[i] = ri;
        v[0] += ri * ri;
        v[1] += pi * pi;
        v[2] += xi * xi;
    }
    block_reduce_store<3>(v, partials);
}

static __global__ void __launch_bounds__(CG_THREADS) k_cg_newp(int64_t n, double beta, const double *r, double *p)
{
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < n; i += int64_t(CG_BLOCKS) * CG_THREADS)
        p[i] = beta * p[i] - r[i];
}

struct DevCG {
    int64_t n;
    hipStream_t stream;
    DevBuf<double> r, p, ap, partials;
    std::vector<double> host;
    DevCG(int64_t n_, hipStream_t st) : n(n_), stream(st), r(size_t(n_)), p(size_t(n_)), ap(size_t(n_)),
                                         partials(size_t(3) * CG_BLOCKS), host(size_t(3) * CG_BLOCKS)
    {
    }
    void fetch(int ns, double *out)
    {
        PFB_HIP(hipMemcpyAsync(host.data(), partials.p, size_t(ns) * CG_BLOCKS * sizeof(double), hipMemcpyDeviceToHost,
                               stream));
        PFB_HIP(hipStreamSynchronize(stream));
        for (int s = 0; s < ns; ++s) {
            double t = 0.0;
            for (int b = 0; b < CG_BLOCKS; ++b) t += host[size_t(s) * CG_BLOCKS + b];
            out[s] = t;
        }
    }
    // aop(in_dev, out_dev) must enqueue on `stream`.  x_dev holds x0 on entry, the solution on exit.
    template <class Op>
    void solve(Op &&aop, const double *b_dev, double *x_dev, double tol, int maxit, int minit, pfbhip_cg_info *info)
    {
        double s[3];
        aop(x_dev, ap.p);
        hipLaunchKernelGGL(k_cg_init, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, ap.p, b_dev, r.p, p.p, partials.p);
        PFB_HIP(hipGetLastError());
        fetch(2, s);
        double rnorm = s[0];
        int k = 0, stall = 0, status = 0;
        double eps = 1.0, phi0 = (std::isnan(rnorm) || rnorm == 0.0) ? 1.0 : rnorm;
        if (s[1] == 0.0) {
            status = 3;  // initial residual is zero
        } else {
            while ((eps > tol || k < minit) && k < maxit && stall < 5) {
                aop(p.p, ap.p);
                hipLaunchKernelGGL(k_cg_dot2, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, p.p, ap.p, p.p, p.p,
                                   partials.p);
                PFB_HIP(hipGetLastError());
                fetch(1, s);
                double alpha = rnorm / s[0];
                hipLaunchKernelGGL(k_cg_update, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, alpha, p.p, ap.p, x_dev,
                                   r.p, partials.p);
                PFB_HIP(hipGetLastError());
                fetch(3, s);
                double rnorm_next = s[0];
                double beta = rnorm_next / rnorm;
                hipLaunchKernelGGL(k_cg_newp, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, beta, r.p, p.p);
                PFB_HIP(hipGetLastError());
                rnorm = rnorm_next;
                ++k;
                double epsp = eps;
                eps = std::sqrt(alpha * alpha * s[1] / std::max(s[2], 1e-12));
                if (std::fabs(epsp - eps) < 1e-3 * tol) ++stall;
            }
            status = k >= maxit ? 1 : (stall >= 5 ? 2 : 0);
        }
        PFB_HIP(hipStreamSynchronize(stream));
        if (info) {
            info->iters = k;
            info->status = status;
            info->eps = eps;
            info->phi = rnorm / phi0;
        }
    }
};

// ---- power method ---------------------------------------------------------------------------------------
// Mirrors power_method_numba (/root/reference/src/pfb_imaging/opt/power_method.py:40-93): b <- b0 / ||b0||; per
// iteration b = A bp, beta = (bp.b) / (bp.bp), b /= ||b||, eps = |beta - beta_prev| / beta_prev, bp <- b; stops
// when eps <= tol or after maxit iterations.  One reduction pass (three dots) and one scaling pass per iteration
// instead of the reference's norm + vdot pair + normalise + copy; one host round trip (three scalars).

// partials [0] = b.b, [1] = bp.b, [2] = bp.bp
static __global__ void __launch_bounds__(CG_THREADS) k_pm_dots(int64_t n, const double *bp, const double *b, double *partials)
{
    double v[3] = {0.0, 0.0, 0.0};
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < n; i += int64_t(CG_BLOCKS) * CG_THREADS) {
        const double bi = b[i], pi = bp[i];
        v[0] += bi * bi;
        v[1] += pi * bi;
        v[2] += pi * pi;
    }
    block_reduce_store<3>(v, partials);
}
static __global__ void __launch_bounds__(CG_THREADS) k_pm_scale(int64_t n, double s, const double *b, double *bp)
{
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < n; i += int64_t(CG_BLOCKS) * CG_THREADS) bp[i] = b[i] * s;
}

struct DevPower {
    int64_t n;
    hipStream_t stream;
    DevBuf<double> b, partials;
    std::vector<double> host;
    DevPower(int64_t n_, hipStream_t st) : n(n_), stream(st), b(size_t(n_)), partials(size_t(3) * CG_BLOCKS), host(size_t(3) * CG_BLOCKS) {}
    // sums over this device; `allreduce(s)` (3 doubles, in place) completes them when the vector spans ranks
    template <class Reduce>
    void dots(const double *bp, const double *bv, double *s, Reduce &&allreduce)
    {
        hipLaunchKernelGGL(k_pm_dots, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, bp, bv, partials.p);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipMemcpyAsync(host.data(), partials.p, host.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
        PFB_HIP(hipStreamSynchronize(stream));
        for (int k = 0; k < 3; ++k) {
            double t = 0.0;
            for (int i = 0; i < CG_BLOCKS; ++i) t += host[size_t(k) * CG_BLOCKS + i];
            s[k] = t;
        }
        allreduce(s);
    }
    // bp_dev: b0 on entry (any non-zero norm), the normalised last iterate on exit.  aop(in, out) enqueues on `stream`.
    template <class Op, class Reduce>
    void run(Op &&aop, Reduce &&allreduce, double *bp_dev, double tol, int maxit, pfbhip_pm_info *info)
    {
        double s[3];
        dots(bp_dev, bp_dev, s, allreduce);
        PFB_REQUIRE(s[0] > 0.0 && std::isfinite(s[0]), "the power method needs a non-zero, finite start vector");
        hipLaunchKernelGGL(k_pm_scale, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, 1.0 / std::sqrt(s[0]), bp_dev, bp_dev);
        double beta = 1.0, eps = 1.0;
        int k = 0;
        while (eps > tol && k < maxit) {
            aop(bp_dev, b.p);
            dots(bp_dev, b.p, s, allreduce);
            const double betap = beta;
            beta = s[1] / s[2];
            hipLaunchKernelGGL(k_pm_scale, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, 1.0 / std::sqrt(s[0]), b.p, bp_dev);
            PFB_HIP(hipGetLastError());
            eps = std::fabs(beta - betap) / betap;
            ++k;
        }
        PFB_HIP(hipStreamSynchronize(stream));
        if (info) {
            info->iters = k;
            info->status = (k == maxit && eps > tol) ? 1 : 0;
            info->eps = eps;
            info->beta = beta;
        }
    }
};

}  // namespace pfbhip

#pragma clang diagnostic pop
