// rowfft.hip -- kernels built on rowfft.hpp: plain batched row transform (debug / benchmark entry)
// and the fused second-axis passes of the gridder's plane transform.
// Compiled with FMA contraction ON (no bit-exact index arithmetic lives in this file).
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "common.hpp"
#include "rowfft.hpp"
#include "rowfft_api.hpp"

namespace pfbhip {

struct PlainLoad {
    const double2 *row;
    __device__ __forceinline__ double2 operator()(int i) const { return row[i]; }
};
struct PlainStore {
    double2 *row;
    __device__ __forceinline__ void operator()(int i, double2 v) const { row[i] = v; }
};

// MAXT bounds the workgroup size: it sets the register budget (512 threads -> 2 waves/SIMD -> 256
// VGPRs, 768 -> 168, 1024 -> 128); the 16-complex-per-thread row needs ~150 to stay out of scratch.
#ifndef RF_MINWAVES
#define RF_MINWAVES 1
#endif
template <int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT == 512 ? RF_MINWAVES : 1)) k_rowfft_plain(RowFFTPlan pl, double2 *data, int nrows, int inverse)
{
    extern __shared__ double rf_lds[];
    const int row = blockIdx.x;  // one row per workgroup
    if (row >= nrows) return;
    PlainLoad ld{data + size_t(row) * pl.N};
    PlainStore st{data + size_t(row) * pl.N};
    rf_row(pl, ld, st, inverse != 0, rf_lds);
}

bool RowFFT::init(int64_t N)
{
    release();
    if (!rowfft_make_plan(N, &pl)) return false;
    std::vector<double2> tw(static_cast<size_t>(N), make_double2(0.0, 0.0));
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int64_t k = 0; k < N; ++k) {
        long double a = -2.0L * pi * (long double)k / (long double)N;
        tw[size_t(k)] = make_double2(double(cosl(a)), double(sinl(a)));
    }
    PFB_HIP(hipMalloc(reinterpret_cast<void **>(&d_tw), size_t(N) * sizeof(double2)));
    PFB_HIP(hipMemcpy(d_tw, tw.data(), size_t(N) * sizeof(double2), hipMemcpyHostToDevice));
    pl.twiddle = d_tw;
    ok = true;
    return true;
}

void RowFFT::release()
{
    if (d_tw) (void)hipFree(d_tw);
    d_tw = nullptr;
    ok = false;
}

void rowfft_plain(const RowFFTPlan &pl, double2 *data_dev, int nrows, bool inverse, hipStream_t stream)
{
    const size_t lds = size_t(pl.N) * sizeof(double);
    static bool attr = false;
    if (!attr) {
        const int maxlds = 16384 * int(sizeof(double));
        PFB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rowfft_plain<512>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
        PFB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rowfft_plain<768>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
        PFB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rowfft_plain<1024>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
        attr = true;
    }
    const int nblk = nrows;
    const int inv = inverse ? 1 : 0;
    if (pl.T <= 512)
        hipLaunchKernelGGL(k_rowfft_plain<512>, dim3(nblk), dim3(pl.T), lds, stream, pl, data_dev, nrows, inv);
    else if (pl.T <= 768)
        hipLaunchKernelGGL(k_rowfft_plain<768>, dim3(nblk), dim3(pl.T), lds, stream, pl, data_dev, nrows, inv);
    else
        hipLaunchKernelGGL(k_rowfft_plain<1024>, dim3(nblk), dim3(pl.T), lds, stream, pl, data_dev, nrows, inv);
    PFB_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
// fused second-axis passes of the plane transform
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ double fg_nm1(double l, double m)
{
    double r2 = l * l + m * m;
    if (r2 <= 1.0) return -r2 / (1.0 + sqrt(1.0 - r2));
    return -sqrt(r2 - 1.0) - 1.0;
}
__device__ __forceinline__ double fg_t(const FusedGeom &g, int ix, int iy)
{
    double l = g.lshift + double(ix - g.nx / 2) * g.px;
    double m = g.mshift + double(iy - g.ny / 2) * g.py;
    return fg_nm1(l, m) + g.nshift;
}
// image column of uv-column u (-1: u lies in the zero padding)
__device__ __forceinline__ int fg_ix(const FusedGeom &g, int u)
{
    const int hx = g.nx / 2;
    if (u < g.nx - hx) return u + hx;
    if (u >= g.nu - hx) return u - (g.nu - hx);
    return -1;
}

struct OccLoad {
    const double2 *row;
    const uint8_t *occ;
    __device__ __forceinline__ double2 operator()(int u) const
    {
        return occ[u >> 5] ? row[u] : make_double2(0.0, 0.0);
    }
};

template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_fused_fft_crop(RowFFTPlan pl, FusedGeom g, const uint8_t *occ,
                                                          const double2 *B, size_t bstride, FusedPlanes planes,
                                                          int do_w, int first, double *accT)
{
    extern __shared__ double rf_lds[];
    const int y = blockIdx.x;
    double *arow = accT + size_t(y) * g.nx;
    for (int k = 0; k < planes.kp; ++k) {
        OccLoad ld{B + size_t(k) * bstride + size_t(y) * g.nu, occ};
        double re[RF_E], im[RF_E];
        int t;
        rf_row_compute(pl, ld, true, rf_lds, t, re, im);
        const double wk = planes.w[k];
        const bool overwrite = first && k == 0;
#pragma unroll
        for (int e = 0; e < RF_E; ++e) {
            const int u = rf_out_pos(pl, t, e);
            const int ix = fg_ix(g, u);
            if (ix >= 0) {
                double r = im[e];  // inverse transform: value = (im, re)
                if (do_w) {
                    double ph = wk * fg_t(g, ix, y);
                    ph -= rint(ph);
                    double s, c;
                    sincospi(2.0 * ph, &s, &c);
                    r = im[e] * c + re[e] * s;  // Re( (im + i re) * (c - i s) )
                }
                arow[ix] = overwrite ? r : arow[ix] + r;
            }
            if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the number of sincospi chains in flight
        }
    }
}

struct PadLoad {
    const double *drow;  // dcT row y
    FusedGeom g;
    int y, do_w;
    double wk;
    __device__ __forceinline__ double2 operator()(int u) const
    {
        const int ix = fg_ix(g, u);
        if (ix < 0) return make_double2(0.0, 0.0);
        const double val = drow[ix];
        if (!do_w) return make_double2(val, 0.0);
        double ph = wk * fg_t(g, ix, y);
        ph -= rint(ph);
        double s, c;
        sincospi(2.0 * ph, &s, &c);
        return make_double2(val * c, val * s);
    }
};

template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_fused_pad_fft(RowFFTPlan pl, FusedGeom g, const uint8_t *occ,
                                                         const double *dcT, FusedPlanes planes, int do_w, double2 *B,
                                                         size_t bstride)
{
    extern __shared__ double rf_lds[];
    const int y = blockIdx.x;
    for (int k = 0; k < planes.kp; ++k) {
        PadLoad ld{dcT + size_t(y) * g.nx, g, y, do_w, planes.w[k]};
        double re[RF_E], im[RF_E];
        int t;
        rf_row_compute(pl, ld, false, rf_lds, t, re, im);
        double2 *brow = B + size_t(k) * bstride + size_t(y) * g.nu;
#pragma unroll
        for (int e = 0; e < RF_E; ++e) {
            const int u = rf_out_pos(pl, t, e);
            if (occ[u >> 5]) brow[u] = make_double2(re[e], im[e]);
        }
    }
}

template <class K512, class K768, class K1024>
static void set_lds_attr(K512 a, K768 b, K1024 c)
{
    const int maxlds = 16384 * int(sizeof(double));
    PFB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(a), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    PFB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(b), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    PFB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(c), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
}

void fused_fft_crop(const RowFFT &f, const FusedGeom &g, const uint8_t *occ_dev, const double2 *B_dev, size_t bstride,
                    const FusedPlanes &planes, int do_w, bool first, double *accT_dev, hipStream_t stream)
{
    static bool attr = false;
    if (!attr) {
        set_lds_attr(&k_fused_fft_crop<512>, &k_fused_fft_crop<768>, &k_fused_fft_crop<1024>);
        attr = true;
    }
    const RowFFTPlan &pl = f.pl;
    const size_t lds = size_t(pl.N) * sizeof(double);
    dim3 grid(uint32_t(g.ny)), blk(uint32_t(pl.T));
    const int fi = first ? 1 : 0;
    if (pl.T <= 512)
        hipLaunchKernelGGL(k_fused_fft_crop<512>, grid, blk, lds, stream, pl, g, occ_dev, B_dev, bstride, planes, do_w, fi,
                           accT_dev);
    else if (pl.T <= 768)
        hipLaunchKernelGGL(k_fused_fft_crop<768>, grid, blk, lds, stream, pl, g, occ_dev, B_dev, bstride, planes, do_w, fi,
                           accT_dev);
    else
        hipLaunchKernelGGL(k_fused_fft_crop<1024>, grid, blk, lds, stream, pl, g, occ_dev, B_dev, bstride, planes, do_w,
                           fi, accT_dev);
    PFB_HIP(hipGetLastError());
}

void fused_pad_fft(const RowFFT &f, const FusedGeom &g, const uint8_t *occ_dev, const double *dcT_dev,
                   const FusedPlanes &planes, int do_w, double2 *B_dev, size_t bstride, hipStream_t stream)
{
    static bool attr = false;
    if (!attr) {
        set_lds_attr(&k_fused_pad_fft<512>, &k_fused_pad_fft<768>, &k_fused_pad_fft<1024>);
        attr = true;
    }
    const RowFFTPlan &pl = f.pl;
    const size_t lds = size_t(pl.N) * sizeof(double);
    dim3 grid(uint32_t(g.ny)), blk(uint32_t(pl.T));
    if (pl.T <= 512)
        hipLaunchKernelGGL(k_fused_pad_fft<512>, grid, blk, lds, stream, pl, g, occ_dev, dcT_dev, planes, do_w, B_dev,
                           bstride);
    else if (pl.T <= 768)
        hipLaunchKernelGGL(k_fused_pad_fft<768>, grid, blk, lds, stream, pl, g, occ_dev, dcT_dev, planes, do_w, B_dev,
                           bstride);
    else
        hipLaunchKernelGGL(k_fused_pad_fft<1024>, grid, blk, lds, stream, pl, g, occ_dev, dcT_dev, planes, do_w, B_dev,
                           bstride);
    PFB_HIP(hipGetLastError());
}

}  // namespace pfbhip

using namespace pfbhip;

extern "C" {

// Debug / benchmark entry: in-place batched row transform of (nrows, n) complex doubles on the host.
// Returns the device time of `reps` back-to-back transforms (ms) through *ms_out if not NULL.
int pfbhip_debug_rowfft(double *data_host, int64_t n, int64_t nrows, int inverse, int reps, double *ms_out)
{
    return guarded([&] {
        PFB_REQUIRE(data_host && nrows >= 1 && reps >= 1, "bad arguments");
        RowFFT plan;
        PFB_REQUIRE(plan.init(n), "row length %lld is not supported by the hand-written FFT", (long long)n);
        const RowFFTPlan &pl = plan.pl;
        const size_t tot = size_t(n) * size_t(nrows);
        DevBuf<double2> d(tot);
        PFB_HIP(hipMemcpy(d.p, data_host, tot * sizeof(double2), hipMemcpyHostToDevice));
        hipEvent_t a, b;
        PFB_HIP(hipEventCreate(&a));
        PFB_HIP(hipEventCreate(&b));
        if (reps > 1) {  // warm-up on a scratch copy so that the result stays a single transform
            DevBuf<double2> s(tot);
            PFB_HIP(hipMemcpy(s.p, d.p, tot * sizeof(double2), hipMemcpyDeviceToDevice));
            rowfft_plain(pl, s.p, int(nrows), inverse != 0, nullptr);
            PFB_HIP(hipEventRecord(a, nullptr));
            for (int r = 0; r < reps; ++r) rowfft_plain(pl, s.p, int(nrows), inverse != 0, nullptr);
            PFB_HIP(hipEventRecord(b, nullptr));
            PFB_HIP(hipEventSynchronize(b));
            float ms = 0;
            PFB_HIP(hipEventElapsedTime(&ms, a, b));
            if (ms_out) *ms_out = ms / reps;
        }
        rowfft_plain(pl, d.p, int(nrows), inverse != 0, nullptr);
        PFB_HIP(hipDeviceSynchronize());
        PFB_HIP(hipMemcpy(data_host, d.p, tot * sizeof(double2), hipMemcpyDeviceToHost));
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    });
}

}  // extern "C"
