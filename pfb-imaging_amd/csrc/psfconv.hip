// psfconv.hip -- FFT (r2c / c2r) and the PSF-convolution operator family on MI355X.
//
// Replaces ducc0.fft.r2c / c2r as called by the reference at
//   /root/reference/src/pfb_imaging/operators/psf.py:8-96        psf_convolve_slice/cube/fscube
//   /root/reference/src/pfb_imaging/operators/hessian.py:103-248 hessian_psf_slice, hess_direct(_slice)
//   /root/reference/src/pfb_imaging/operators/hessian.py:313-349 HessPSF.dot
//   /root/reference/src/pfb_imaging/operators/hessian.py:487-518 HessianTree.dot
//   /root/reference/src/pfb_imaging/operators/gridder.py:659,912 PSFHAT = r2c(ifftshift(psf))
//
// One application  out = post * crop(irfft2(rfft2(pad(pre * x)) * f(psfhat))) * scale + eta * x
// is five passes: fused pad*beam (writes the whole padded plane once, so no separate memset),
// rocFFT real forward, fused spectral multiply (with the 1/N of inorm=2 folded in), rocFFT real
// inverse, fused crop*beam*scale + eta*x (+ accumulate).  PSFs and beams stay resident on the device.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "psffft_api.hpp"
#include "pipeline_api.hpp"
#include "devcg.hpp"

namespace pfbhip {

#define PFB_ROCFFT(expr)                                                                              \
    do {                                                                                              \
        rocfft_status _s = (expr);                                                                    \
        if (_s != rocfft_status_success)                                                              \
            throw std::runtime_error(pfbhip::strprintf("%s failed: rocfft status %d (%s:%d)", #expr, \
                                                       int(_s), __FILE__, __LINE__));                 \
    } while (0)

void rocfft_setup_once();

static inline dim3 blocks1d(int64_t n, int t = 256) { return dim3(uint32_t(std::max<int64_t>(ceil_div(n, t), 1))); }

// xpad (nxp, nyp): [0:nx, 0:ny] = x * beam, 0 elsewhere
__global__ void k_psf_pad(const double *x, const double *beam, int nx, int ny, int nxp, int nyp, double *xpad)
{
    int iy = blockIdx.x * blockDim.x + threadIdx.x;
    int ix = blockIdx.y;
    if (iy >= nyp) return;
    double v = 0.0;
    if (ix < nx && iy < ny) {
        size_t o = size_t(ix) * ny + iy;
        v = x[o];
        if (beam) v *= beam[o];
    }
    xpad[size_t(ix) * nyp + iy] = v;
}

// xhat *= f(psfhat) * norm ;  mode 0: psf, 1: psf + shift, 2: 1 / (psf + shift)
__global__ void k_psf_mul(double2 *xhat, const double *psf, int is_complex, int mode, double shift, double norm,
                          int64_t n)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    double2 v = xhat[i];
    double pr, pi = 0.0;
    if (is_complex) {
        pr = psf[2 * i];
        pi = psf[2 * i + 1];
    } else {
        pr = psf[i];
    }
    if (mode != 0) pr += shift;
    double2 r;
    if (mode == 2) {
        // v / (pr + i pi)
        double d = pr * pr + pi * pi;
        r.x = (v.x * pr + v.y * pi) / d;
        r.y = (v.y * pr - v.x * pi) / d;
    } else {
        r.x = v.x * pr - v.y * pi;
        r.y = v.x * pi + v.y * pr;
    }
    r.x *= norm;
    r.y *= norm;
    xhat[i] = r;
}

// out (nx, ny) = [out +] beam * xpad[0:nx, 0:ny] * scale + eta * x
__global__ void k_psf_crop(const double *xpad, const double *beam, const double *x, int nx, int ny, int nyp,
                           double scale, double eta, int accumulate, double *out)
{
    int iy = blockIdx.x * blockDim.x + threadIdx.x;
    int ix = blockIdx.y;
    if (iy >= ny) return;
    size_t o = size_t(ix) * ny + iy;
    double v = xpad[size_t(ix) * nyp + iy];
    if (beam) v *= beam[o];
    v *= scale;
    if (eta != 0.0) v += eta * x[o];
    out[o] = accumulate ? out[o] + v : v;
}

struct RealFFT2D {
    int64_t n0 = 0, n1 = 0;
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    DevBuf<char> work;
    hipStream_t stream = nullptr;
    void create(int64_t n0_, int64_t n1_, hipStream_t st)
    {
        n0 = n0_;
        n1 = n1_;
        stream = st;
        rocfft_setup_once();
        size_t lengths[2] = {size_t(n1), size_t(n0)};
        rocfft_status rst = rocfft_status_success;
        // (rocFFT allocates inside plan creation: on failure the cache of released blocks gives way, once)
        if (!retry_after_cache_flush([&] {
                if (fwd == nullptr)
                    rst = rocfft_plan_create(&fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                            rocfft_precision_double, 2, lengths, 1, nullptr);
                if (rst == rocfft_status_success && inv == nullptr)
                    rst = rocfft_plan_create(&inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                            rocfft_precision_double, 2, lengths, 1, nullptr);
                return rst == rocfft_status_success;
            }))
            PFB_ROCFFT(rst);
        size_t a = 0, b = 0;
        PFB_ROCFFT(rocfft_plan_get_work_buffer_size(fwd, &a));
        PFB_ROCFFT(rocfft_plan_get_work_buffer_size(inv, &b));
        PFB_ROCFFT(rocfft_execution_info_create(&info));
        if (std::max(a, b)) {
            work.alloc(std::max(a, b));
            PFB_ROCFFT(rocfft_execution_info_set_work_buffer(info, work.p, work.n));
        }
        PFB_ROCFFT(rocfft_execution_info_set_stream(info, st));
    }
    void r2c(double *in, double2 *out)
    {
        void *i[1] = {in}, *o[1] = {out};
        PFB_ROCFFT(rocfft_execute(fwd, i, o, info));
    }
    // destroys `in` (like ducc0's allow_overwriting_input=True, psf.py:31)
    void c2r(double2 *in, double *out)
    {
        void *i[1] = {in}, *o[1] = {out};
        PFB_ROCFFT(rocfft_execute(inv, i, o, info));
    }
    ~RealFFT2D()
    {
        if (fwd) rocfft_plan_destroy(fwd);
        if (inv) rocfft_plan_destroy(inv);
        if (info) rocfft_execution_info_destroy(info);
    }
};

}  // namespace pfbhip

using namespace pfbhip;

struct pfbhip_psfconv {
    int64_t nx, ny, nxp, nyp, nyo2;
    hipStream_t stream = nullptr;
    RealFFT2D fft;    // rocFFT 2-D r2c / c2r: the fallback for padded sizes the row-FFT pipeline does not take
    PsfFFT own;       // three pruned row passes on the hand-written FFT (power-of-two padded sizes)
    DevBuf<double> xpad, d_x, d_out;
    DevBuf<double2> xhat;
    struct Slot {
        DevBuf<double> data;
        bool is_complex = false;
        bool bound = false;
    };
    std::vector<std::unique_ptr<Slot>> psf, beam;
    ~pfbhip_psfconv()
    {
        if (stream) (void)hipStreamDestroy(stream);
    }
    Slot &slot(std::vector<std::unique_ptr<Slot>> &v, int64_t i)
    {
        PFB_REQUIRE(i >= 0 && i < 65536, "slot %lld out of range", (long long)i);
        while (int64_t(v.size()) <= i) v.emplace_back(new Slot);
        return *v[size_t(i)];
    }
    void apply(const double *x_dev, int64_t psf_slot, int64_t beam_slot, int mode, double shift, double scale,
               double eta, int accumulate, double *out_dev)
    {
        PFB_REQUIRE(psf_slot >= 0 && psf_slot < int64_t(psf.size()) && psf[size_t(psf_slot)]->bound,
                    "psfhat slot %lld is not bound", (long long)psf_slot);
        PFB_REQUIRE(mode >= 0 && mode <= 2, "bad mode %d", mode);
        const double *bm = nullptr;
        if (beam_slot >= 0) {
            PFB_REQUIRE(beam_slot < int64_t(beam.size()) && beam[size_t(beam_slot)]->bound,
                        "beam slot %lld is not bound", (long long)beam_slot);
            bm = beam[size_t(beam_slot)]->data.p;
        }
        Slot &ps = *psf[size_t(psf_slot)];
        if (own.ok) {  // slot data is stored transposed, (nyo2, nxp)
            own.apply(x_dev, bm, ps.data.p, ps.is_complex, mode, shift, scale, eta, accumulate, out_dev, stream);
            return;
        }
        dim3 blk(256);
        hipLaunchKernelGGL(k_psf_pad, dim3(uint32_t(ceil_div(nyp, 256)), uint32_t(nxp)), blk, 0, stream, x_dev, bm,
                           int(nx), int(ny), int(nxp), int(nyp), xpad.p);
        PFB_HIP(hipGetLastError());
        fft.r2c(xpad.p, xhat.p);
        const int64_t nh = nxp * nyo2;
        hipLaunchKernelGGL(k_psf_mul, blocks1d(nh), blk, 0, stream, xhat.p, ps.data.p, int(ps.is_complex), mode, shift,
                           1.0 / (double(nxp) * double(nyp)), nh);
        PFB_HIP(hipGetLastError());
        fft.c2r(xhat.p, xpad.p);
        hipLaunchKernelGGL(k_psf_crop, dim3(uint32_t(ceil_div(ny, 256)), uint32_t(nx)), blk, 0, stream, xpad.p, bm,
                           x_dev, int(nx), int(ny), int(nyp), scale, eta, accumulate, out_dev);
        PFB_HIP(hipGetLastError());
    }
};

namespace pfbhip {
hipStream_t psfconv_stream(pfbhip_psfconv *p) { return p->stream; }
hipStream_t psfconv_swap_stream(pfbhip_psfconv *p, hipStream_t st)
{
    hipStream_t prev = p->stream;
    p->stream = st;
    p->fft.stream = st;
    if (p->fft.info) PFB_ROCFFT(rocfft_execution_info_set_stream(p->fft.info, st));
    return prev;
}
void psfconv_geometry(const pfbhip_psfconv *p, int64_t *nx, int64_t *ny)
{
    *nx = p->nx;
    *ny = p->ny;
}
void psfconv_apply_async(pfbhip_psfconv *p, const double *x_dev, int64_t psf_slot, int64_t beam_slot, int mode, double shift,
                         double scale, double eta, int accumulate, double *out_dev)
{
    p->apply(x_dev, psf_slot, beam_slot, mode, shift, scale, eta, accumulate, out_dev);
}
}  // namespace pfbhip

extern "C" {

int pfbhip_psfconv_create(int64_t nx, int64_t ny, int64_t nx_psf, int64_t ny_psf, pfbhip_psfconv **out)
{
    return guarded([&] {
        PFB_REQUIRE(out, "NULL argument");
        PFB_REQUIRE(nx >= 1 && ny >= 1 && nx_psf >= nx && ny_psf >= ny, "bad PSF-convolution geometry (%lld,%lld)->(%lld,%lld)",
                    (long long)nx, (long long)ny, (long long)nx_psf, (long long)ny_psf);
        std::unique_ptr<pfbhip_psfconv> p(new pfbhip_psfconv);
        p->nx = nx;
        p->ny = ny;
        p->nxp = nx_psf;
        p->nyp = ny_psf;
        p->nyo2 = ny_psf / 2 + 1;
        PFB_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        if (!p->own.init(nx, ny, nx_psf, ny_psf)) {
            p->fft.create(nx_psf, ny_psf, p->stream);
            p->xpad.alloc(size_t(nx_psf) * size_t(ny_psf));
            p->xhat.alloc(size_t(nx_psf) * size_t(p->nyo2));
        }
        *out = p.release();
    });
}

int pfbhip_psfconv_destroy(pfbhip_psfconv *p)
{
    return guarded([&] { delete p; });
}

int pfbhip_psfconv_set_psfhat(pfbhip_psfconv *p, int64_t slot, const double *psfhat_host, int is_complex)
{
    return guarded([&] {
        PFB_REQUIRE(p && psfhat_host, "NULL argument");
        auto &s = p->slot(p->psf, slot);
        size_t n = size_t(p->nxp) * size_t(p->nyo2) * (is_complex ? 2 : 1);
        s.data.ensure(n);
        if (p->own.ok) {  // keep (nyo2, nxp): the layout the per-frequency pass reads contiguously
            DevBuf<double> tmp(n);
            PFB_HIP(hipMemcpyAsync(tmp.p, psfhat_host, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
            p->own.transpose_psf(tmp.p, is_complex != 0, s.data.p, p->stream);
            PFB_HIP(hipStreamSynchronize(p->stream));
        } else {
            PFB_HIP(hipMemcpyAsync(s.data.p, psfhat_host, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
            PFB_HIP(hipStreamSynchronize(p->stream));
        }
        s.is_complex = is_complex != 0;
        s.bound = true;
    });
}

int pfbhip_psfconv_set_beam(pfbhip_psfconv *p, int64_t slot, const double *beam_host)
{
    return guarded([&] {
        PFB_REQUIRE(p, "NULL argument");
        auto &s = p->slot(p->beam, slot);
        if (!beam_host) {
            s.bound = false;
            return;
        }
        size_t n = size_t(p->nx) * size_t(p->ny);
        s.data.ensure(n);
        PFB_HIP(hipMemcpyAsync(s.data.p, beam_host, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
        PFB_HIP(hipStreamSynchronize(p->stream));
        s.bound = true;
    });
}

int pfbhip_psfconv_apply_dev(pfbhip_psfconv *p, const double *x_dev, int64_t psf_slot, int64_t beam_slot, int mode,
                             double shift, double scale, double eta, int accumulate, double *out_dev)
{
    return guarded([&] {
        PFB_REQUIRE(p && x_dev && out_dev, "NULL argument");
        p->apply(x_dev, psf_slot, beam_slot, mode, shift, scale, eta, accumulate, out_dev);
        PFB_HIP(hipStreamSynchronize(p->stream));
    });
}

int pfbhip_psfconv_apply(pfbhip_psfconv *p, const double *x_host, int64_t psf_slot, int64_t beam_slot, int mode,
                         double shift, double scale, double eta, int accumulate, double *out_host)
{
    return guarded([&] {
        PFB_REQUIRE(p && x_host && out_host, "NULL argument");
        size_t n = size_t(p->nx) * size_t(p->ny);
        p->d_x.ensure(n);
        p->d_out.ensure(n);
        PFB_HIP(hipMemcpyAsync(p->d_x.p, x_host, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
        if (accumulate)
            PFB_HIP(hipMemcpyAsync(p->d_out.p, out_host, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
        p->apply(p->d_x.p, psf_slot, beam_slot, mode, shift, scale, eta, accumulate, p->d_out.p);
        PFB_HIP(hipMemcpyAsync(out_host, p->d_out.p, n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        PFB_HIP(hipStreamSynchronize(p->stream));
    });
}

// HessPSF.idot's direct estimate (operators/hessian.py:369-387 of the reference): out = taper-weighted division by (psfhat + shift)
// (mode 2 with the taper as the image-plane multiplier), then -- where a beam is bound -- the beam division the reference does on
// the host, x /= beam^2 where x > 0 and beam > min_beam, on the device.  raw_host (may be NULL) receives the estimate BEFORE the
// division (the CG start vector), out_host the divided one.
__global__ void k_beam_divide(int64_t n, const double *__restrict__ beam, double min_beam, double *__restrict__ x)
{
    const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const double v = x[i], b = beam[i];
    if (v > 0.0 && b > min_beam) x[i] = v / (b * b);
}

int pfbhip_psfconv_direct(pfbhip_psfconv *p, const double *x_host, int64_t psf_slot, int64_t taper_slot, double shift,
                          int64_t beam_slot, double min_beam, double *raw_host, double *out_host)
{
    return guarded([&] {
        PFB_REQUIRE(p && x_host && out_host, "NULL argument");
        const size_t n = size_t(p->nx) * size_t(p->ny);
        p->d_x.ensure(n);
        p->d_out.ensure(n);
        PFB_HIP(hipMemcpyAsync(p->d_x.p, x_host, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
        p->apply(p->d_x.p, psf_slot, taper_slot, 2, shift, 1.0, 0.0, 0, p->d_out.p);
        if (raw_host) PFB_HIP(hipMemcpyAsync(raw_host, p->d_out.p, n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        if (beam_slot >= 0) {
            PFB_REQUIRE(beam_slot < int64_t(p->beam.size()) && p->beam[size_t(beam_slot)]->bound, "beam slot %lld is not bound",
                        (long long)beam_slot);
            hipLaunchKernelGGL(k_beam_divide, dim3(uint32_t(ceil_div(int64_t(n), 256))), dim3(256), 0, p->stream, int64_t(n),
                               p->beam[size_t(beam_slot)]->data.p, min_beam, p->d_out.p);
            PFB_HIP(hipGetLastError());
        }
        PFB_HIP(hipMemcpyAsync(out_host, p->d_out.p, n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        PFB_HIP(hipStreamSynchronize(p->stream));
    });
}

int pfbhip_psfconv_cg(pfbhip_psfconv *p, int64_t nparts, const int64_t *psf_slots, const int64_t *beam_slots,
                      double scale, double eta, const double *rhs_host, double *x_host, int has_x0, double tol,
                      int maxit, int minit, pfbhip_cg_info *info)
{
    return guarded([&] {
        PFB_REQUIRE(p && rhs_host && x_host && psf_slots && beam_slots && nparts >= 1, "bad arguments");
        hipStream_t st = p->stream;
        const size_t n = size_t(p->nx) * size_t(p->ny);
        DevBuf<double> b{n}, x{n};
        PFB_HIP(hipMemcpyAsync(b.p, rhs_host, n * sizeof(double), hipMemcpyHostToDevice, st));
        if (has_x0) PFB_HIP(hipMemcpyAsync(x.p, x_host, n * sizeof(double), hipMemcpyHostToDevice, st));
        else PFB_HIP(hipMemsetAsync(x.p, 0, n * sizeof(double), st));
        DevCG cg(int64_t(n), st);
        cg.solve(
            [&](const double *in, double *out) {
                for (int64_t k = 0; k < nparts; ++k)
                    p->apply(in, psf_slots[k], beam_slots[k], 0, 0.0, scale, k == 0 ? eta : 0.0, k > 0, out);
            },
            b.p, x.p, tol, maxit, minit, info);
        PFB_HIP(hipMemcpyAsync(x_host, x.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
        PFB_HIP(hipStreamSynchronize(st));
    });
}

// ---- stand-alone r2c / c2r (host arrays) ------------------------------------------------

// out[k0][k1] *= (-1)^(k0 + k1): the spectrum of ifftshift(x) from the spectrum of x when both lengths are even
__global__ void k_checker_sign(double2 *a, int64_t n0, int64_t nh)
{
    const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i >= n0 * nh) return;
    const int64_t k0 = i / nh, k1 = i - k0 * nh;
    if ((k0 + k1) & 1) a[i] = make_double2(-a[i].x, -a[i].y);
}

static int r2c_2d_impl(const double *in_host, int64_t nbatch, int64_t n0, int64_t n1, double *out_host, bool centred);

int pfbhip_r2c_2d(const double *in_host, int64_t nbatch, int64_t n0, int64_t n1, double *out_host)
{
    return r2c_2d_impl(in_host, nbatch, n0, n1, out_host, false);
}

int pfbhip_r2c_2d_centred(const double *in_host, int64_t nbatch, int64_t n0, int64_t n1, double *out_host)
{
    return r2c_2d_impl(in_host, nbatch, n0, n1, out_host, true);
}

static int r2c_2d_impl(const double *in_host, int64_t nbatch, int64_t n0, int64_t n1, double *out_host, bool centred)
{
    return guarded([&] {
        PFB_REQUIRE(in_host && out_host && nbatch >= 0 && n0 >= 1 && n1 >= 1, "bad r2c arguments");
        PFB_REQUIRE(!centred || (n0 % 2 == 0 && n1 % 2 == 0), "the centred transform needs even lengths (%lld, %lld)", (long long)n0,
                    (long long)n1);
        hipStream_t st;
        PFB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        struct Guard {
            hipStream_t s;
            ~Guard() { (void)hipStreamDestroy(s); }
        } guard{st};
        RealFFT2D fft;
        fft.create(n0, n1, st);
        const size_t nr = size_t(n0) * size_t(n1), nc = size_t(n0) * size_t(n1 / 2 + 1);
        DevBuf<double> d_in(nr);
        DevBuf<double2> d_out(nc);
        for (int64_t b = 0; b < nbatch; ++b) {
            PFB_HIP(hipMemcpyAsync(d_in.p, in_host + size_t(b) * nr, nr * sizeof(double), hipMemcpyHostToDevice, st));
            fft.r2c(d_in.p, d_out.p);
            if (centred) {
                hipLaunchKernelGGL(k_checker_sign, dim3(uint32_t(ceil_div(int64_t(nc), 256))), dim3(256), 0, st, d_out.p, n0, n1 / 2 + 1);
                PFB_HIP(hipGetLastError());
            }
            PFB_HIP(hipMemcpyAsync(out_host + size_t(b) * nc * 2, d_out.p, nc * sizeof(double2), hipMemcpyDeviceToHost,
                                   st));
        }
        PFB_HIP(hipStreamSynchronize(st));
    });
}

__global__ void k_scale(double *a, double s, int64_t n)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i < n) a[i] *= s;
}

int pfbhip_c2r_2d(const double *in_host, int64_t nbatch, int64_t n0, int64_t n1, double *out_host)
{
    return guarded([&] {
        PFB_REQUIRE(in_host && out_host && nbatch >= 0 && n0 >= 1 && n1 >= 1, "bad c2r arguments");
        hipStream_t st;
        PFB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        struct Guard {
            hipStream_t s;
            ~Guard() { (void)hipStreamDestroy(s); }
        } guard{st};
        RealFFT2D fft;
        fft.create(n0, n1, st);
        const size_t nr = size_t(n0) * size_t(n1), nc = size_t(n0) * size_t(n1 / 2 + 1);
        DevBuf<double> d_out(nr);
        DevBuf<double2> d_in(nc);
        for (int64_t b = 0; b < nbatch; ++b) {
            PFB_HIP(hipMemcpyAsync(d_in.p, in_host + size_t(b) * nc * 2, nc * sizeof(double2), hipMemcpyHostToDevice,
                                   st));
            fft.c2r(d_in.p, d_out.p);
            hipLaunchKernelGGL(k_scale, blocks1d(int64_t(nr)), dim3(256), 0, st, d_out.p, 1.0 / double(nr), int64_t(nr));
            PFB_HIP(hipGetLastError());
            PFB_HIP(hipMemcpyAsync(out_host + size_t(b) * nr, d_out.p, nr * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        PFB_HIP(hipStreamSynchronize(st));
    });
}

}  // extern "C"
