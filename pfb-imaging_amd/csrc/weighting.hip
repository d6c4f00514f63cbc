// weighting.hip -- uv-cell index map, counts scatter and Briggs gather-divide on MI355X.
//
// Replaces the numba kernels _compute_counts / counts_to_weights of
// /root/reference/src/pfb_imaging/utils/weighting.py:81-140,143-208.
// The integer cell index is bit-exact against the CPU oracle (oracle/pfb_oracle.c:
// pfbo_uvcell_index): identical un-fused IEEE double sequence, compiled -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <vector>

#include "common.hpp"

namespace pfbhip {

constexpr double SPEED_OF_LIGHT = 299792458.0;

struct CellArgs {
    const double *uvw, *freq;
    const uint8_t *mask;
    int64_t nvis;
    int nchan;
    int64_t nx, ny;
    double u_cell, umax, v_cell, vmax, usign, vsign;
};

__device__ __forceinline__ int64_t uv_cell(const CellArgs &a, int64_t i)
{
    if (!a.mask[i]) return -1;
    int64_t r = i / a.nchan;
    int f = int(i - r * a.nchan);
    double nf = a.freq[f] / SPEED_OF_LIGHT;
    double u = a.uvw[3 * r] * nf;
    u = u * a.usign;
    double v = a.uvw[3 * r + 1] * nf;
    v = v * a.vsign;
    if (v < 0.0) {
        u = -u;
        v = -v;
    }
    double ug = (u + a.umax) / a.u_cell;
    double vg = (v + a.vmax) / a.v_cell;
    double fu = floor(ug), fv = floor(vg);
    if (!(fu >= 0.0) || !(fu < double(a.nx)) || !(fv >= 0.0) || !(fv < double(a.ny))) return -1;
    return int64_t(fu) * a.ny + int64_t(fv);
}

__global__ void k_uvcell(CellArgs a, int64_t *cell)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i < a.nvis) cell[i] = uv_cell(a, i);
}

__global__ void k_counts(CellArgs a, const double *wgt, int ncorr, double *counts)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i >= a.nvis) return;
    int64_t c = uv_cell(a, i);
    if (c < 0) return;
    for (int k = 0; k < ncorr; ++k) unsafeAtomicAdd(&counts[size_t(k) * a.nx * a.ny + c], wgt[size_t(k) * a.nvis + i]);
}

__global__ void k_counts_divide(CellArgs a, const double *counts, int ncorr, double *wgt)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i >= a.nvis) return;
    int64_t c = uv_cell(a, i);
    if (c < 0) return;
    for (int k = 0; k < ncorr; ++k) {
        double cv = counts[size_t(k) * a.nx * a.ny + c];
        if (cv > 0.0) wgt[size_t(k) * a.nvis + i] /= cv;
    }
}

struct CellSetup {
    DevBuf<double> uvw, freq;
    DevBuf<uint8_t> mask;
    CellArgs a;
    CellSetup(const double *uvw_h, const double *freq_h, const uint8_t *mask_h, int64_t nrow, int64_t nchan, int64_t nx,
              int64_t ny, double cell_x, double cell_y, double usign, double vsign)
    {
        PFB_REQUIRE(uvw_h && freq_h && mask_h, "NULL argument");
        PFB_REQUIRE(nrow >= 0 && nchan >= 1 && nx >= 1 && ny >= 1, "bad shapes");
        uvw.alloc(size_t(std::max<int64_t>(nrow, 1)) * 3);
        freq.alloc(size_t(nchan));
        mask.alloc(size_t(std::max<int64_t>(nrow * nchan, 1)));
        PFB_HIP(hipMemcpy(uvw.p, uvw_h, size_t(nrow) * 3 * sizeof(double), hipMemcpyHostToDevice));
        PFB_HIP(hipMemcpy(freq.p, freq_h, size_t(nchan) * sizeof(double), hipMemcpyHostToDevice));
        PFB_HIP(hipMemcpy(mask.p, mask_h, size_t(nrow * nchan), hipMemcpyHostToDevice));
        a.uvw = uvw.p;
        a.freq = freq.p;
        a.mask = mask.p;
        a.nvis = nrow * nchan;
        a.nchan = int(nchan);
        a.nx = nx;
        a.ny = ny;
        // same expressions as weighting.py:84-90
        a.u_cell = 1.0 / (double(nx) * cell_x);
        a.umax = std::fabs(1.0 / cell_x / 2.0);
        a.v_cell = 1.0 / (double(ny) * cell_y);
        a.vmax = std::fabs(1.0 / cell_y / 2.0);
        a.usign = usign;
        a.vsign = vsign;
    }
    dim3 grid() const { return dim3(uint32_t(std::max<int64_t>(ceil_div(a.nvis, 256), 1))); }
};

// box sum along one axis with zero padding: out[r][c] = sum_{|d| <= s} in[r][c + d]   (axis 1)
//                                              or  sum_{|d| <= s} in[r + d][c]   (axis 0)
__global__ void k_box_sum_axis(const double *in, int64_t nplanes, int nx, int ny, int s, int axis, double *out)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t per = int64_t(nx) * ny;
    if (i >= nplanes * per) return;
    const int64_t pl = i / per, o = i - pl * per;
    const int r = int(o / ny), c = int(o - int64_t(r) * ny);
    const double *p = in + pl * per;
    double acc = 0.0;
    if (axis == 1) {
        for (int d = -s; d <= s; ++d) {
            int cc = c + d;
            if (cc >= 0 && cc < ny) acc += p[int64_t(r) * ny + cc];
        }
    } else {
        for (int d = -s; d <= s; ++d) {
            int rr = r + d;
            if (rr >= 0 && rr < nx) acc += p[int64_t(rr) * ny + c];
        }
    }
    out[i] = acc;
}

struct IsPositive {
    __host__ __device__ bool operator()(const double &v) const { return v > 0.0; }
};

__global__ void k_floor_positive(double *a, int64_t n, double lowval)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i < n && a[i] > 0.0 && a[i] < lowval) a[i] = lowval;
}

// per-plane sum and sum of squares (Briggs normalisation): out[2 k] += sum, out[2 k + 1] += sum of squares
__global__ void k_sum_sumsq(const double *a, int64_t per, double *out)
{
    __shared__ double s1[256], s2[256];
    const int k = blockIdx.y;
    double t1 = 0.0, t2 = 0.0;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < per; i += int64_t(gridDim.x) * blockDim.x) {
        const double v = a[size_t(k) * per + i];
        t1 += v;
        t2 += v * v;
    }
    s1[threadIdx.x] = t1;
    s2[threadIdx.x] = t2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) {
            s1[threadIdx.x] += s1[threadIdx.x + s];
            s2[threadIdx.x] += s2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(&out[2 * k], s1[0]);
        unsafeAtomicAdd(&out[2 * k + 1], s2[0]);
    }
}

// counts = counts * ssq[k] + 1
__global__ void k_scale_plus_one(double *a, int64_t per, const double *ssq)
{
    const int k = blockIdx.y;
    const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i < per) a[size_t(k) * per + i] = a[size_t(k) * per + i] * ssq[k] + 1.0;
}

// positive entries below median(positive) / level are raised to it (device buffer, in place); returns the median
static double dev_filter_extreme(double *a, int64_t n, double level)
{
    PFB_REQUIRE(n < (int64_t(1) << 31), "counts grid too large");
    if (n == 0) return 0.0;
    DevBuf<double> pos{size_t(n)}, sorted{size_t(n)};
    DevBuf<int> d_num(1);
    size_t tb = 0;
    PFB_HIP(hipcub::DeviceSelect::If(nullptr, tb, a, pos.p, d_num.p, int(n), IsPositive()));
    DevBuf<char> tmp(tb);
    PFB_HIP(hipcub::DeviceSelect::If(tmp.p, tb, a, pos.p, d_num.p, int(n), IsPositive()));
    int npos = 0;
    PFB_HIP(hipMemcpy(&npos, d_num.p, sizeof(int), hipMemcpyDeviceToHost));
    if (npos == 0) return 0.0;
    size_t sb = 0;
    PFB_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, sb, pos.p, sorted.p, npos));
    DevBuf<char> stmp(sb);
    PFB_HIP(hipcub::DeviceRadixSort::SortKeys(stmp.p, sb, pos.p, sorted.p, npos));
    double mid[2] = {0.0, 0.0};
    const int lo = (npos - 1) / 2, hi = npos / 2;  // numpy.median: mean of the two middle values
    PFB_HIP(hipMemcpy(&mid[0], sorted.p + lo, sizeof(double), hipMemcpyDeviceToHost));
    PFB_HIP(hipMemcpy(&mid[1], sorted.p + hi, sizeof(double), hipMemcpyDeviceToHost));
    const double med = 0.5 * (mid[0] + mid[1]);
    hipLaunchKernelGGL(k_floor_positive, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, 0, a, n, med / level);
    PFB_HIP(hipGetLastError());
    return med;
}

}  // namespace pfbhip

using namespace pfbhip;

extern "C" {

// The imaging-weight chain of image_data_products (/root/reference/src/pfb_imaging/operators/gridder.py:534-576) as ONE
// device pipeline: counts of the (natural) weights on the padded uv-grid -> filter_extreme_counts -> box_sum_counts ->
// Briggs scaling (utils/weighting.py:161-176) -> per-visibility division.  The visibilities' coordinates, mask and weights
// are uploaded once, the counts never leave HBM unless counts_out_host is given; weights are updated in place.
int pfbhip_imaging_weights(const double *uvw_host, const double *freq_host, const uint8_t *mask_host, double *wgt_host,
                           int64_t ncorr, int64_t nrow, int64_t nchan, int64_t nx, int64_t ny, double cell_x, double cell_y,
                           double usign, double vsign, double robust, double filter_level, int64_t npix_super,
                           double *counts_out_host)
{
    return guarded([&] {
        PFB_REQUIRE(wgt_host && ncorr >= 1 && npix_super >= 0, "bad arguments");
        CellSetup s(uvw_host, freq_host, mask_host, nrow, nchan, nx, ny, cell_x, cell_y, usign, vsign);
        const int64_t per = nx * ny;
        const size_t ncnt = size_t(ncorr) * size_t(per);
        DevBuf<double> counts(ncnt), tmp(npix_super > 0 ? ncnt : 0);
        PFB_HIP(hipMemset(counts.p, 0, counts.bytes()));
        if (!s.a.nvis) {
            if (counts_out_host) PFB_HIP(hipMemcpy(counts_out_host, counts.p, counts.bytes(), hipMemcpyDeviceToHost));
            return;
        }
        DevBuf<double> wgt(size_t(ncorr) * size_t(s.a.nvis));
        PFB_HIP(hipMemcpy(wgt.p, wgt_host, wgt.bytes(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_counts, s.grid(), dim3(256), 0, 0, s.a, wgt.p, int(ncorr), counts.p);
        PFB_HIP(hipGetLastError());
        if (filter_level > 0.0) (void)dev_filter_extreme(counts.p, int64_t(ncnt), filter_level);
        if (npix_super > 0) {
            dim3 grid(uint32_t(ceil_div(int64_t(ncnt), 256)));
            hipLaunchKernelGGL(k_box_sum_axis, grid, dim3(256), 0, 0, counts.p, ncorr, int(nx), int(ny), int(npix_super), 1, tmp.p);
            hipLaunchKernelGGL(k_box_sum_axis, grid, dim3(256), 0, 0, tmp.p, ncorr, int(nx), int(ny), int(npix_super), 0, counts.p);
            PFB_HIP(hipGetLastError());
        }
        // Briggs: counts <- counts * (5 * 10^-R)^2 * sum(c) / sum(c^2) + 1 when R > -2; all-zero counts leave the weights alone
        DevBuf<double> sums(size_t(2 * ncorr));
        PFB_HIP(hipMemset(sums.p, 0, sums.bytes()));
        hipLaunchKernelGGL(k_sum_sumsq, dim3(uint32_t(std::min<int64_t>(ceil_div(per, 256), 1024)), uint32_t(ncorr)), dim3(256), 0, 0,
                           counts.p, per, sums.p);
        PFB_HIP(hipGetLastError());
        std::vector<double> hs(size_t(2 * ncorr));
        PFB_HIP(hipMemcpy(hs.data(), sums.p, sums.bytes(), hipMemcpyDeviceToHost));
        bool any = false;
        for (int64_t k = 0; k < ncorr; ++k) any = any || hs[size_t(2 * k + 1)] > 0.0;
        if (any) {
            if (robust > -2.0) {
                const double numsqrt = 5.0 * std::pow(10.0, -robust);
                std::vector<double> ssq(static_cast<size_t>(ncorr), 0.0);
                for (int64_t k = 0; k < ncorr; ++k)
                    ssq[size_t(k)] = hs[size_t(2 * k + 1)] > 0.0 ? numsqrt * numsqrt * hs[size_t(2 * k)] / hs[size_t(2 * k + 1)] : 0.0;
                DevBuf<double> d_ssq{size_t(ncorr)};
                PFB_HIP(hipMemcpy(d_ssq.p, ssq.data(), d_ssq.bytes(), hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_scale_plus_one, dim3(uint32_t(ceil_div(per, 256)), uint32_t(ncorr)), dim3(256), 0, 0, counts.p,
                                   per, d_ssq.p);
                PFB_HIP(hipGetLastError());
            }
            hipLaunchKernelGGL(k_counts_divide, s.grid(), dim3(256), 0, 0, s.a, counts.p, int(ncorr), wgt.p);
            PFB_HIP(hipGetLastError());
            PFB_HIP(hipMemcpy(wgt_host, wgt.p, wgt.bytes(), hipMemcpyDeviceToHost));
        }
        if (counts_out_host) PFB_HIP(hipMemcpy(counts_out_host, counts.p, counts.bytes(), hipMemcpyDeviceToHost));
    });
}

int pfbhip_uvcell_index(const double *uvw_host, const double *freq_host, const uint8_t *mask_host, int64_t nrow,
                        int64_t nchan, int64_t nx, int64_t ny, double cell_x, double cell_y, double usign, double vsign,
                        int64_t *cell_host)
{
    return guarded([&] {
        PFB_REQUIRE(cell_host, "NULL argument");
        CellSetup s(uvw_host, freq_host, mask_host, nrow, nchan, nx, ny, cell_x, cell_y, usign, vsign);
        if (!s.a.nvis) return;
        DevBuf<int64_t> cell(size_t(s.a.nvis));
        hipLaunchKernelGGL(k_uvcell, s.grid(), dim3(256), 0, 0, s.a, cell.p);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipMemcpy(cell_host, cell.p, cell.bytes(), hipMemcpyDeviceToHost));
    });
}

int pfbhip_compute_counts(const double *uvw_host, const double *freq_host, const uint8_t *mask_host,
                          const double *wgt_host, int64_t ncorr, int64_t nrow, int64_t nchan, int64_t nx, int64_t ny,
                          double cell_x, double cell_y, double usign, double vsign, double *counts_host)
{
    return guarded([&] {
        PFB_REQUIRE(wgt_host && counts_host && ncorr >= 1, "NULL argument");
        CellSetup s(uvw_host, freq_host, mask_host, nrow, nchan, nx, ny, cell_x, cell_y, usign, vsign);
        size_t ncnt = size_t(ncorr) * size_t(nx) * size_t(ny);
        DevBuf<double> counts(ncnt);
        PFB_HIP(hipMemcpy(counts.p, counts_host, counts.bytes(), hipMemcpyHostToDevice));
        if (s.a.nvis) {
            DevBuf<double> wgt(size_t(ncorr) * size_t(s.a.nvis));
            PFB_HIP(hipMemcpy(wgt.p, wgt_host, wgt.bytes(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_counts, s.grid(), dim3(256), 0, 0, s.a, wgt.p, int(ncorr), counts.p);
            PFB_HIP(hipGetLastError());
        }
        PFB_HIP(hipMemcpy(counts_host, counts.p, counts.bytes(), hipMemcpyDeviceToHost));
    });
}

int pfbhip_counts_divide(const double *uvw_host, const double *freq_host, const uint8_t *mask_host,
                         const double *counts_host, int64_t ncorr, int64_t nrow, int64_t nchan, int64_t nx, int64_t ny,
                         double cell_x, double cell_y, double usign, double vsign, double *wgt_host)
{
    return guarded([&] {
        PFB_REQUIRE(wgt_host && counts_host && ncorr >= 1, "NULL argument");
        CellSetup s(uvw_host, freq_host, mask_host, nrow, nchan, nx, ny, cell_x, cell_y, usign, vsign);
        if (!s.a.nvis) return;
        DevBuf<double> counts(size_t(ncorr) * size_t(nx) * size_t(ny)), wgt(size_t(ncorr) * size_t(s.a.nvis));
        PFB_HIP(hipMemcpy(counts.p, counts_host, counts.bytes(), hipMemcpyHostToDevice));
        PFB_HIP(hipMemcpy(wgt.p, wgt_host, wgt.bytes(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_counts_divide, s.grid(), dim3(256), 0, 0, s.a, counts.p, int(ncorr), wgt.p);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipMemcpy(wgt_host, wgt.p, wgt.bytes(), hipMemcpyDeviceToHost));
    });
}

int pfbhip_box_sum_counts(const double *counts_host, int64_t ncorr, int64_t nx, int64_t ny, int64_t npix_super,
                          double *out_host)
{
    return guarded([&] {
        PFB_REQUIRE(counts_host && out_host && ncorr >= 1 && nx >= 1 && ny >= 1 && npix_super >= 0, "bad arguments");
        const size_t n = size_t(ncorr) * size_t(nx) * size_t(ny);
        DevBuf<double> a(n), b(n);
        PFB_HIP(hipMemcpy(a.p, counts_host, n * sizeof(double), hipMemcpyHostToDevice));
        dim3 grid(uint32_t(ceil_div(int64_t(n), 256)));
        hipLaunchKernelGGL(k_box_sum_axis, grid, dim3(256), 0, 0, a.p, ncorr, int(nx), int(ny), int(npix_super), 1, b.p);
        hipLaunchKernelGGL(k_box_sum_axis, grid, dim3(256), 0, 0, b.p, ncorr, int(nx), int(ny), int(npix_super), 0, a.p);
        PFB_HIP(hipGetLastError());
        PFB_HIP(hipMemcpy(out_host, a.p, n * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int pfbhip_filter_extreme_counts(double *counts_host, int64_t n, double level, double *median_out)
{
    return guarded([&] {
        PFB_REQUIRE(counts_host && n >= 0 && level > 0.0, "bad arguments");
        if (median_out) *median_out = 0.0;
        if (n == 0) return;
        DevBuf<double> a{size_t(n)};
        PFB_HIP(hipMemcpy(a.p, counts_host, size_t(n) * sizeof(double), hipMemcpyHostToDevice));
        const double med = dev_filter_extreme(a.p, n, level);
        if (median_out) *median_out = med;
        PFB_HIP(hipMemcpy(counts_host, a.p, size_t(n) * sizeof(double), hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
