// pd.hip -- the primal-dual (backward) step of the SARA minor cycle with every cube resident in HBM.
//
// Mirrors PrimalDual.solve (/root/reference/src/pfb_imaging/opt/primal_dual.py:406-448; legacy loop :230-262)
// with the l21 regulariser over the wavelet dictionary (prox/l21.py:15-50, fused dual update
// prox/prox_21m.py:105-135) and the gradient of the forward-backward splitting,
// grad(x) = -H (xtilde - x) / gamma (deconv/pfb.py:158-161, core/sara.py:288-289), H the PSF-approximate
// Hessian of HessianTree / HessPSF (operators/hessian.py:326-348, 439-522):
//     v     <- Psi^H xp ;  v <- dual update(vp, v) ;  vp <- 2 v - vp
//     xout  <- Psi vp + grad(xp) ;  x <- xp - tau xout ;  x <- positivity(x)
//     eps   =  ||x - xp|| / max(||x||, 1e-6) (1 if x == 0) ;  stop if eps < tol ;  xp <- x, vp <- v
// One host round trip per iteration (the three scalars of eps) instead of ~10 cubes.
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>

#include <cmath>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <chrono>

#include "common.hpp"
#include "devcg.hpp"
#include "pipeline_api.hpp"

// a failed C-ABI call inside the driver (its message is already in pfbhip_last_error())
#define PFB_CHECK_STATUS(call)                                                    \
    do {                                                                          \
        if ((call) != 0) throw std::runtime_error(std::string(pfbhip_last_error())); \
    } while (0)

namespace pfbhip {

__global__ void __launch_bounds__(256) k_pd_diff(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ d,
                                                 int64_t n)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i < n) d[i] = a[i] - b[i];
}
__global__ void __launch_bounds__(256) k_pd_primal(double *__restrict__ x, const double *__restrict__ xp,
                                                   const double *__restrict__ xout, double tau, int64_t n)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i < n) x[i] = xp[i] - tau * xout[i];
}
// Primal step, positivity and the three norms in ONE pass over the images (all bands of a pixel in the same thread):
// x = xp - tau xout ; mode 1: clamp negatives, mode 2: zero the pixel in every band where any band is <= 0
// (positivity.py:12-33) ; partials [0] = |x - xp|^2, [1] = |x|^2, [2] = #nonzero(x).  nband <= PD_MAXB.
constexpr int PD_MAXB = 16;
static __global__ void __launch_bounds__(CG_THREADS) k_pd_step(int64_t npix, int nband, double *__restrict__ x,
                                                                const double *__restrict__ xp, const double *__restrict__ xout,
                                                                double tau, int mode, double *partials)
{
    double v[3] = {0.0, 0.0, 0.0};
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < npix; i += int64_t(CG_BLOCKS) * CG_THREADS) {
        double xs[PD_MAXB], ps[PD_MAXB];
        bool bad = false;
        for (int b = 0; b < nband; ++b) {
            const size_t o = size_t(b) * size_t(npix) + size_t(i);
            ps[b] = xp[o];
            xs[b] = ps[b] - tau * xout[o];
            bad = bad || xs[b] <= 0.0;
        }
        for (int b = 0; b < nband; ++b) {
            double xi = xs[b];
            if (mode == 1 && xi < 0.0) xi = 0.0;
            if (mode == 2 && bad) xi = 0.0;
            x[size_t(b) * size_t(npix) + size_t(i)] = xi;
            const double d = xi - ps[b];
            v[0] += d * d;
            v[1] += xi * xi;
            v[2] += (xi != 0.0) ? 1.0 : 0.0;
        }
    }
    block_reduce_store<3>(v, partials);
}

// partials [0] = |x - xp|^2, [1] = |x|^2, [2] = #nonzero(x)
static __global__ void __launch_bounds__(CG_THREADS) k_pd_norms(int64_t n, const double *x, const double *xp, double *partials)
{
    double v[3] = {0.0, 0.0, 0.0};
    for (int64_t i = blockIdx.x * int64_t(CG_THREADS) + threadIdx.x; i < n; i += int64_t(CG_BLOCKS) * CG_THREADS) {
        const double xi = x[i], d = xi - xp[i];
        v[0] += d * d;
        v[1] += xi * xi;
        v[2] += (xi != 0.0) ? 1.0 : 0.0;
    }
    block_reduce_store<3>(v, partials);
}

}  // namespace pfbhip

using namespace pfbhip;

extern "C" {

int pfbhip_primal_dual(pfbhip_psi *psi, pfbhip_psfconv *const *pcs, int64_t nband, const int64_t *nparts, const int64_t *psf_slots,
                       const int64_t *beam_slots, const double *scale, const double *eta, const double *xtilde_host, double gamma,
                       double *x_host, double *v_host, const double *weight_host, double lam, double sigma, double tau,
                       int positivity, double tol, int maxit, pfbhip_comm *comm, pfbhip_pd_info *info)
{
    return guarded([&] {
        PFB_REQUIRE(psi && pcs && nparts && psf_slots && beam_slots && scale && eta && xtilde_host && x_host && v_host &&
                        weight_host && nband >= 1 && maxit >= 1,
                    "bad arguments");
        PFB_REQUIRE(positivity >= 0 && positivity <= 2, "positivity mode %d", positivity);
        PFB_REQUIRE(gamma != 0.0, "gamma must be non-zero");
        int64_t nx, ny, nxmax, nymax, px, py;
        int nbasis;
        psi_geometry(psi, &nx, &ny, &nbasis, &nxmax, &nymax);
        for (int64_t b = 0; b < nband; ++b) {
            PFB_REQUIRE(pcs[b] != nullptr, "band %lld has no PSF plan", (long long)b);
            psfconv_geometry(pcs[b], &px, &py);
            PFB_REQUIRE(px == nx && py == ny, "Psi is (%lld, %lld) but the PSF plan of band %lld is (%lld, %lld)", (long long)nx,
                        (long long)ny, (long long)b, (long long)px, (long long)py);
        }
        const size_t npix = size_t(nx) * size_t(ny), cube = size_t(nbasis) * size_t(nxmax) * size_t(nymax);
        const size_t nimg = size_t(nband) * npix, ncoef = size_t(nband) * cube;
        // one stream for everything: the first plan's; Psi and the other plans are switched to it for the call
        hipStream_t st = psfconv_stream(pcs[0]);
        struct Restore {
            pfbhip_psi *p;
            hipStream_t prev;
            std::vector<std::pair<pfbhip_psfconv *, hipStream_t>> plans;
            ~Restore()
            {
                (void)psi_swap_stream(p, prev);
                for (auto it = plans.rbegin(); it != plans.rend(); ++it) {
                    try {
                        (void)psfconv_swap_stream(it->first, it->second);
                    } catch (...) {
                    }
                }
            }
        } restore{psi, psi_swap_stream(psi, st), {}};
        for (int64_t b = 1; b < nband; ++b) {
            bool seen = pcs[b] == pcs[0];
            for (auto &pr : restore.plans) seen = seen || pr.first == pcs[b];
            if (!seen) restore.plans.emplace_back(pcs[b], psfconv_swap_stream(pcs[b], st));
        }

        // Buffer rotation instead of copies: xa / xb alternate as (x, xp); va / vb alternate as (dual, previous
        // dual); vext holds the extrapolated dual 2 v - vp of the current iteration.
        DevBuf<double> xa(nimg), xb(nimg), xout(nimg), xt(nimg), d(npix), va(ncoef), vb(ncoef), vext(ncoef), w(cube);
        DevBuf<double> sum(comm != nullptr ? cube : 0);
        DevBuf<double> partials(3 * size_t(CG_BLOCKS));
        std::vector<double> hpart(3 * size_t(CG_BLOCKS));
        double *xp = xa.p, *x = xb.p, *vp = va.p, *v = vb.p;
        PFB_HIP(hipMemcpyAsync(xp, x_host, nimg * sizeof(double), hipMemcpyHostToDevice, st));
        PFB_HIP(hipMemcpyAsync(xt.p, xtilde_host, nimg * sizeof(double), hipMemcpyHostToDevice, st));
        PFB_HIP(hipMemcpyAsync(vp, v_host, ncoef * sizeof(double), hipMemcpyHostToDevice, st));
        PFB_HIP(hipMemcpyAsync(w.p, weight_host, cube * sizeof(double), hipMemcpyHostToDevice, st));
        std::vector<int64_t> off(size_t(nband) + 1, 0);
        for (int64_t b = 0; b < nband; ++b) {
            PFB_REQUIRE(nparts[b] >= 1, "band %lld has no partitions", (long long)b);
            off[size_t(b) + 1] = off[size_t(b)] + nparts[b];
        }
        auto blocks = [](size_t n) { return dim3(uint32_t(ceil_div(int64_t(n), 256))); };
        // Stage clocks (HIP events on this stream, read back after the loop): Psi^H analysis, dual update, Psi synthesis, the
        // PSF-approximate Hessian applies, primal step + norms -- what bench.py's C4 roofline is computed from.
        struct StageClock {
            hipStream_t st;
            std::vector<hipEvent_t> ev;
            std::vector<int> stage;
            void begin(int s)
            {
                hipEvent_t a, b;
                PFB_HIP(hipEventCreate(&a));
                PFB_HIP(hipEventCreate(&b));
                PFB_HIP(hipEventRecord(a, st));
                ev.push_back(a);
                ev.push_back(b);
                stage.push_back(s);
            }
            void end() { PFB_HIP(hipEventRecord(ev.back(), st)); }
            ~StageClock()
            {
                for (auto e : ev) (void)hipEventDestroy(e);
            }
        } clk{st, {}, {}};
        const bool timed = info != nullptr && maxit <= 64;  // (bounded number of events: short, benchmark-style runs only)
        auto tick = [&](int s) {
            if (timed) clk.begin(s);
        };
        auto tock = [&]() {
            if (timed) clk.end();
        };
        double eps = 1.0;
        int k = 0, status = 1;
        const auto t_loop0 = std::chrono::steady_clock::now();  // (the stream is idle here: the uploads above were synchronised)
        PFB_HIP(hipStreamSynchronize(st));
        for (; k < maxit; ++k) {
            tick(0);
            for (int64_t b = 0; b < nband; ++b) psi_dot_async(psi, xp + size_t(b) * npix, v + size_t(b) * cube);
            tock();
            // v <- dual update(vp, Psi^H xp) ; vext <- 2 v - vp
            tick(1);
            if (comm == nullptr) {
                l21_fused_async(vp, v, vext.p, nband, int64_t(cube), lam, sigma, w.p, st);  // one pass over the cubes
            } else {
                // the bands of this rank only: the band sum of vtilde is completed with ONE all-reduce per iteration
                l21_localsum_async(vp, v, nband, int64_t(cube), sigma, sum.p, st);
                PFB_HIP(hipStreamSynchronize(st));
                PFB_CHECK_STATUS(pfbhip_comm_allreduce_sum(comm, sum.p, sum.p, int64_t(cube)));
                l21_apply_async(vp, v, vext.p, nband, int64_t(cube), lam, sigma, w.p, sum.p, st);
            }
            tock();
            for (int64_t b = 0; b < nband; ++b) {
                double *xo = xout.p + size_t(b) * npix;
                tick(2);
                psi_hdot_async(psi, vext.p + size_t(b) * cube, xo);
                tock();
                hipLaunchKernelGGL(k_pd_diff, blocks(npix), dim3(256), 0, st, xt.p + size_t(b) * npix, xp + size_t(b) * npix, d.p,
                                   int64_t(npix));
                for (int64_t q = off[size_t(b)]; q < off[size_t(b) + 1]; ++q) {
                    tick(3);
                    psfconv_apply_async(pcs[b], d.p, psf_slots[q], beam_slots[q], 0, 0.0, -scale[b] / gamma,
                                        q == off[size_t(b)] ? -eta[b] / gamma : 0.0, 1, xo);
                    tock();
                }
            }
            tick(4);
            const bool one_pass = !(positivity == 2 && comm != nullptr) && nband <= PD_MAXB;
            if (one_pass) {
                hipLaunchKernelGGL(k_pd_step, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, st, int64_t(npix), int(nband), x, xp, xout.p, tau,
                                   positivity, partials.p);
            } else {
                hipLaunchKernelGGL(k_pd_primal, blocks(nimg), dim3(256), 0, st, x, xp, xout.p, tau, int64_t(nimg));
                if (positivity == 2 && comm != nullptr) {  // "any band <= 0" spans the ranks
                    positivity_flag_async(x, nband, int64_t(npix), d.p, st);
                    PFB_HIP(hipStreamSynchronize(st));
                    PFB_CHECK_STATUS(pfbhip_comm_allreduce_sum(comm, d.p, d.p, int64_t(npix)));
                    positivity_zero_async(x, nband, int64_t(npix), d.p, st);
                } else if (positivity) {
                    positivity_async(x, nband, int64_t(npix), positivity, st);
                }
                hipLaunchKernelGGL(k_pd_norms, dim3(CG_BLOCKS), dim3(CG_THREADS), 0, st, int64_t(nimg), x, xp, partials.p);
            }
            tock();
            PFB_HIP(hipGetLastError());
            PFB_HIP(hipMemcpyAsync(hpart.data(), partials.p, hpart.size() * sizeof(double), hipMemcpyDeviceToHost, st));
            PFB_HIP(hipStreamSynchronize(st));
            double num = 0.0, den = 0.0, nnz = 0.0;
            for (int i = 0; i < CG_BLOCKS; ++i) {
                num += hpart[size_t(i)];
                den += hpart[size_t(CG_BLOCKS) + size_t(i)];
                nnz += hpart[2 * size_t(CG_BLOCKS) + size_t(i)];
            }
            if (comm != nullptr) {  // the norms are over ALL bands
                const double loc[3] = {num, den, nnz};
                double tot[3];
                PFB_HIP(hipMemcpyAsync(partials.p, loc, sizeof loc, hipMemcpyHostToDevice, st));
                PFB_HIP(hipStreamSynchronize(st));
                PFB_CHECK_STATUS(pfbhip_comm_allreduce_sum(comm, partials.p, partials.p, 3));
                PFB_HIP(hipMemcpy(tot, partials.p, sizeof tot, hipMemcpyDeviceToHost));
                num = tot[0];
                den = tot[1];
                nnz = tot[2];
            }
            eps = nnz > 0.0 ? std::sqrt(num / std::max(den, 1e-12)) : 1.0;  // _nb_norm_diff, primal_dual.py:40-52, 429
            if (eps < tol) {
                status = 0;
                break;
            }
            std::swap(x, xp);  // xp <- x
            std::swap(v, vp);  // vp <- v
        }
        // (every iteration ends with a stream synchronisation: the wall clock brackets exactly the device work of the loop)
        const double loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count();
        // after a break x / v hold the last iterate; after maxit the final swap moved them to xp / vp
        if (status != 0) {
            std::swap(x, xp);
            std::swap(v, vp);
        }
        PFB_HIP(hipMemcpyAsync(x_host, x, nimg * sizeof(double), hipMemcpyDeviceToHost, st));
        PFB_HIP(hipMemcpyAsync(v_host, v, ncoef * sizeof(double), hipMemcpyDeviceToHost, st));
        PFB_HIP(hipStreamSynchronize(st));
        if (info) {
            info->iters = status == 0 ? k : maxit - 1;  // the reference reports the loop index k
            info->status = status;
            info->eps = eps;
            info->loop_ms = loop_ms;
            for (int q = 0; q < PFBHIP_PD_NSTAGES; ++q) {
                info->stage_ms[q] = 0.0;
                info->stage_calls[q] = 0;
            }
            for (size_t i = 0; i < clk.stage.size(); ++i) {
                float ms = 0.f;
                PFB_HIP(hipEventElapsedTime(&ms, clk.ev[2 * i], clk.ev[2 * i + 1]));
                info->stage_ms[clk.stage[i]] += double(ms);
                info->stage_calls[clk.stage[i]] += 1;
            }
        }
    });
}

int pfbhip_psfconv_power_method(pfbhip_psfconv *const *pcs, int64_t nband, const int64_t *nparts, const int64_t *psf_slots,
                                const int64_t *beam_slots, const double *scale, const double *eta, double *b_host, double tol,
                                int maxit, pfbhip_comm *comm, pfbhip_pm_info *info)
{
    return guarded([&] {
        PFB_REQUIRE(pcs && nparts && psf_slots && beam_slots && scale && eta && b_host && nband >= 1 && maxit >= 0, "bad arguments");
        int64_t nx = 0, ny = 0, px, py;
        for (int64_t b = 0; b < nband; ++b) {
            PFB_REQUIRE(pcs[b] != nullptr, "band %lld has no PSF plan", (long long)b);
            psfconv_geometry(pcs[b], &px, &py);
            if (b == 0) nx = px, ny = py;
            PFB_REQUIRE(px == nx && py == ny, "band %lld is (%lld, %lld), band 0 is (%lld, %lld)", (long long)b, (long long)px,
                        (long long)py, (long long)nx, (long long)ny);
        }
        const size_t npix = size_t(nx) * size_t(ny), nimg = size_t(nband) * npix;
        hipStream_t st = psfconv_stream(pcs[0]);
        struct Restore {
            std::vector<std::pair<pfbhip_psfconv *, hipStream_t>> plans;
            ~Restore()
            {
                for (auto it = plans.rbegin(); it != plans.rend(); ++it) {
                    try {
                        (void)psfconv_swap_stream(it->first, it->second);
                    } catch (...) {
                    }
                }
            }
        } restore;
        for (int64_t b = 1; b < nband; ++b) {
            bool seen = pcs[b] == pcs[0];
            for (auto &pr : restore.plans) seen = seen || pr.first == pcs[b];
            if (!seen) restore.plans.emplace_back(pcs[b], psfconv_swap_stream(pcs[b], st));
        }
        std::vector<int64_t> off(size_t(nband) + 1, 0);
        for (int64_t b = 0; b < nband; ++b) {
            PFB_REQUIRE(nparts[b] >= 1, "band %lld has no partitions", (long long)b);
            off[size_t(b) + 1] = off[size_t(b)] + nparts[b];
        }
        DevBuf<double> bp(nimg), red(comm != nullptr ? 3 : 0);
        PFB_HIP(hipMemcpyAsync(bp.p, b_host, nimg * sizeof(double), hipMemcpyHostToDevice, st));
        DevPower pm(int64_t(nimg), st);
        auto aop = [&](const double *in, double *out) {
            for (int64_t b = 0; b < nband; ++b)
                for (int64_t q = off[size_t(b)]; q < off[size_t(b) + 1]; ++q)
                    psfconv_apply_async(pcs[b], in + size_t(b) * npix, psf_slots[q], beam_slots[q], 0, 0.0, scale[b],
                                        q == off[size_t(b)] ? eta[b] : 0.0, q > off[size_t(b)], out + size_t(b) * npix);
        };
        auto allreduce = [&](double *s) {
            if (comm == nullptr) return;
            PFB_HIP(hipMemcpyAsync(red.p, s, 3 * sizeof(double), hipMemcpyHostToDevice, st));
            PFB_HIP(hipStreamSynchronize(st));
            PFB_CHECK_STATUS(pfbhip_comm_allreduce_sum(comm, red.p, red.p, 3));
            PFB_HIP(hipMemcpy(s, red.p, 3 * sizeof(double), hipMemcpyDeviceToHost));
        };
        pm.run(aop, allreduce, bp.p, tol, maxit, info);
        PFB_HIP(hipMemcpyAsync(b_host, bp.p, nimg * sizeof(double), hipMemcpyDeviceToHost, st));
        PFB_HIP(hipStreamSynchronize(st));
    });
}


}  // extern "C"
