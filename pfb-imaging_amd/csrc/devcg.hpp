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

// The CG scalars live on the device (scal: [0] alpha, [1] beta, [2] r.r of the current residual, [3] p.p, [4] x.x): the
// loop enqueues a whole iteration -- operator, dot, alpha, update, beta, new direction -- without waiting for the host,
// and reads the three numbers of the stopping rule back once per iteration, behind the last kernel.  (Two host round
// trips per iteration before; on a busy host each one cost the GPU milliseconds of idle time.)
// One block: sum of the CG_BLOCKS partials of ns quantities, in a fixed order.
template <int NS>
__device__ __forceinline__ void cg_sum_partials(const double *partials, double (&out)[NS])
{
    __shared__ double sm[NS][CG_THREADS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double t = 0.0;
        for (int i = threadIdx.x; i < CG_BLOCKS; i += CG_THREADS) t += partials[size_t(s) * CG_BLOCKS + i];
        sm[s][threadIdx.x] = t;
    }
    __syncthreads();
    for (int o = CG_THREADS / 2; o > 0; o >>= 1) {
        if (int(threadIdx.x) < o)
#pragma unroll
            for (int s = 0; s < NS; ++s) sm[s][threadIdx.x] += sm[s][threadIdx.x + o];
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) out[s] = sm[s][0];
}
// alpha = r.r / p.Ap
static __global__ void __launch_bounds__(CG_THREADS) k_cg_alpha(const double *partials, double *scal)
{
    double s[1];
    cg_sum_partials<1>(partials, s);
    if (threadIdx.x == 0) scal[0] = scal[2] / s[0];
}
// beta = r'.r' / r.r ; r.r <- r'.r' ; p.p and x.x for the stopping rule
static __global__ void __launch_bounds__(CG_THREADS) k_cg_beta(const double *partials, double *scal)
{
    double s[3];
    cg_sum_partials<3>(partials, s);
    if (threadIdx.x == 0) {
        scal[1] = s[0] / scal[2];
        scal[2] = s[0];
        scal[3] = s[1];
        scal[4] = s[2];
    }
}

// x += alpha p ; r += alpha ap ; partials [0] = r.r, [1] = p.p, [2] = x.x
static __global__ void __launch_bounds__(CG_THREADS) k_cg_update(int64_t n, const double *scal, const double *p,
                                                                  const double *ap, double *x, double *r,
                                                                  double *partials)
{
    const double alpha = scal[0];
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

static __global__ void __launch_bounds__(CG_THREADS) k_cg_newp(int64_t n, const double *scal, const double *r, double *p)
{
    const double beta = scal[1];
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < n; i += int64_t(CG_BLOCKS) * CG_THREADS)
        p[i] = beta * p[i] - r[i];
}

struct DevCG {
    int64_t n;
    hipStream_t stream;
    DevBuf<double> r, p, ap, partials, scal;
    std::vector<double> host;
    double *hscal = nullptr;  // pinned: the per-iteration read-back
    DevCG(int64_t n_, hipStream_t st) : n(n_), stream(st), r(size_t(n_)), p(size_t(n_)), ap(size_t(n_)),
                                         partials(size_t(3) * CG_BLOCKS), scal(8), host(size_t(3) * CG_BLOCKS)
    {
        PFB_HIP(hipHostMalloc(reinterpret_cast<void **>(&hscal), 8 * sizeof(double), hipHostMallocDefault));
    }
    ~DevCG()
    {
        if (hscal) (void)hipHostFree(hscal);
    }
    DevCG(const DevCG &) = delete;
    DevCG &operator=(const DevCG &) = delete;
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
            hscal[2] = rnorm;
            PFB_HIP(hipMemcpyAsync(scal.p + 2, hscal + 2, sizeof(double), hipMemcpyHostToDevice, stream));
            while ((eps > tol || k < minit) && k < maxit && stall < 5) {
                aop(p.p, ap.p);
                hipLaunchKernelGGL(k_cg_dot2, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, p.p, ap.p, p.p, p.p,
                                   partials.p);
                hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(CG_THREADS), 0, stream, partials.p, scal.p);
                hipLaunchKernelGGL(k_cg_update, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, scal.p, p.p, ap.p, x_dev,
                                   r.p, partials.p);
                hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(CG_THREADS), 0, stream, partials.p, scal.p);
                // (the new direction is formed before the host has looked at the stopping rule: harmless if the loop ends)
                hipLaunchKernelGGL(k_cg_newp, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, stream, n, scal.p, r.p, p.p);
                PFB_HIP(hipGetLastError());
                PFB_HIP(hipMemcpyAsync(hscal, scal.p, 5 * sizeof(double), hipMemcpyDeviceToHost, stream));
                PFB_HIP(hipStreamSynchronize(stream));
                const double alpha = hscal[0];
                rnorm = hscal[2];
                ++k;
                double epsp = eps;
                eps = std::sqrt(alpha * alpha * hscal[3] / std::max(hscal[4], 1e-12));
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
