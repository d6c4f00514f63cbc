// gridder_wd.hip -- the one-plane w-scheme's scatter / gather (gridder_kernels_wd.hpp) in a translation unit of its own:
// 13 kernel supports x 3 term counts x 2 kernels compile next to gridder.hip, not behind it.
#include "gridder_kernels_wd.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

#include "common.hpp"

namespace pfbhip {

// Threads per workgroup: 256 -- one wave per SIMD, three workgroups per CU (see gridder_kernels_wd.hpp); PFBHIP_WD_SCATTER_THREADS /
// PFBHIP_WD_GATHER_THREADS override (measured on C2, ms per apply: scatter 768 x 1: 2.76, 384 x 2: 3.3, 256 x 3: 2.31; gather
// 768 x 1: 1.90, 384 x 2: 2.33, 256 x 3: 1.64).
static int env_threads(const char *name, int dflt, int cap)
{
    const char *e = std::getenv(name);
    if (e == nullptr) return dflt;
    const int v = std::atoi(e);
    return (v >= 256 && v <= cap && v % 64 == 0) ? v : dflt;  // (>= 256: the tile flush holds its cells in registers, sized for that)
}
// Launches of fewer work items than three per CU (C1: ~300) cannot fill the chip with 256-thread workgroups: those take one
// 768-thread workgroup per item.
static int scatter_threads_for(uint32_t nwork)
{
    static const int t = env_threads("PFBHIP_WD_SCATTER_THREADS", 0, wd_threads());
    return t > 0 ? t : (nwork < 768u ? wd_threads() : 256);
}
int wd_scatter_threads() { return wd_threads(); }  // (the largest: what the LDS limit is sized for)
static int wd_gather_threads_rt(int NJ, uint32_t nwork)
{
    static const int t = env_threads("PFBHIP_WD_GATHER_THREADS", 0, MP_THREADS);
    return std::min(t > 0 ? t : (nwork < 768u ? 768 : 256), wd_gather_threads(NJ));
}
size_t wd_scatter_lds_bytes(int W)
{
    size_t n = 0;
    switch (W) {
#define PFB_CASE(w) case w: n = wd_lds_doubles(w, wd_scatter_threads() / 64); break;
        PFB_CASE(4) PFB_CASE(5) PFB_CASE(6) PFB_CASE(7) PFB_CASE(8) PFB_CASE(9) PFB_CASE(10) PFB_CASE(11)
        PFB_CASE(12) PFB_CASE(13) PFB_CASE(14) PFB_CASE(15) PFB_CASE(16)
#undef PFB_CASE
        default: throw std::runtime_error("unsupported kernel support");
    }
    return n * sizeof(double);
}
size_t wd_gather_lds_bytes() { return size_t(RW_LS) * RW_LS * sizeof(double2); }

void wd_launch_coeffs(const WdArgs &wa, int64_t nactive, const double *pw, double2 *cw, hipStream_t st)
{
    const int64_t n = nactive + REC_PAD;
    hipLaunchKernelGGL(k_wd_coeffs, dim3(uint32_t(ceil_div(n, 256))), dim3(256), 0, st, wa, nactive, pw, cw);
    PFB_HIP(hipGetLastError());
}

void wd_launch_plane_values(int K, int64_t nactive, const double2 *cw, const double2 *sval, double2 *pval, hipStream_t st)
{
    if (nactive <= 0) return;
    hipLaunchKernelGGL(k_plane_values_wd, dim3(uint32_t(ceil_div(nactive, 256))), dim3(256), 0, st, K, nactive, cw, sval, pval);
    PFB_HIP(hipGetLastError());
}

template <int W, int NJ, int BC>
static void grid_wkb(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *pval, double2 *grid, hipStream_t st)
{
    allow_dynamic_lds(reinterpret_cast<const void *>(&k_grid_wd<W, NJ, BC>), 160 * 1024);
    const int threads = scatter_threads_for(ga.a.nwork);
    const size_t lds = wd_lds_doubles(W, threads / 64) * sizeof(double);
    hipLaunchKernelGGL((k_grid_wd<W, NJ, BC>), dim3(ga.a.nwork), dim3(threads), lds, st, ga, wa, rec, pval, grid);
}
template <int W, int NJ>
static void grid_wk(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *pval, double2 *grid, hipStream_t st)
{
    if constexpr (W == 14 || W == 15) {
        if (wa.bc == 2) return grid_wkb<W, NJ, 2>(ga, wa, rec, pval, grid, st);
    }
    if (wa.bc != 4) throw std::runtime_error("one-plane scatter: block edge 2 is built for W = 14, 15 only");
    grid_wkb<W, NJ, 4>(ga, wa, rec, pval, grid, st);
}
template <int W>
static void grid_w(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *pval, double2 *grid, hipStream_t st)
{
    switch (wa.K) {
        case 2: grid_wk<W, 2>(ga, wa, rec, pval, grid, st); break;
        case 3: grid_wk<W, 3>(ga, wa, rec, pval, grid, st); break;
        case 4: grid_wk<W, 4>(ga, wa, rec, pval, grid, st); break;
        default: throw std::runtime_error("one-plane w-scheme: 2..4 kernel functions");
    }
}
void wd_launch_grid(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *pval, double2 *grid, hipStream_t st)
{
    if (ga.a.nwork == 0) return;
    switch (wa.W) {
#define PFB_CASE(w) case w: grid_w<w>(ga, wa, rec, pval, grid, st); break;
        PFB_CASE(4) PFB_CASE(5) PFB_CASE(6) PFB_CASE(7) PFB_CASE(8) PFB_CASE(9) PFB_CASE(10) PFB_CASE(11)
        PFB_CASE(12) PFB_CASE(13) PFB_CASE(14) PFB_CASE(15) PFB_CASE(16)
#undef PFB_CASE
        default: throw std::runtime_error("unsupported kernel support");
    }
    PFB_HIP(hipGetLastError());
}

template <int W, int NJ>
static void degrid_wk(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *grid, double2 *sacc,
                      const double *swgt, double2 *pval_out, hipStream_t st)
{
    allow_dynamic_lds(reinterpret_cast<const void *>(&k_degrid_wd<W, NJ>), 160 * 1024);
    hipLaunchKernelGGL((k_degrid_wd<W, NJ>), dim3(ga.a.nwork), dim3(wd_gather_threads_rt(NJ, ga.a.nwork)), wd_gather_lds_bytes(), st, ga, wa, rec, grid, sacc,
                       swgt, pval_out);
}
template <int W>
static void degrid_w(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *grid, double2 *sacc,
                     const double *swgt, double2 *pval_out, hipStream_t st)
{
    switch (wa.K) {
        case 2: degrid_wk<W, 2>(ga, wa, rec, grid, sacc, swgt, pval_out, st); break;
        case 3: degrid_wk<W, 3>(ga, wa, rec, grid, sacc, swgt, pval_out, st); break;
        case 4: degrid_wk<W, 4>(ga, wa, rec, grid, sacc, swgt, pval_out, st); break;
        default: throw std::runtime_error("one-plane w-scheme: 2..4 kernel functions");
    }
}
void wd_launch_degrid(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *grid, double2 *sacc,
                      const double *swgt, double2 *pval_out, hipStream_t st)
{
    if (ga.a.nwork == 0) return;
    switch (wa.W) {
#define PFB_CASE(w) case w: degrid_w<w>(ga, wa, rec, grid, sacc, swgt, pval_out, st); break;
        PFB_CASE(4) PFB_CASE(5) PFB_CASE(6) PFB_CASE(7) PFB_CASE(8) PFB_CASE(9) PFB_CASE(10) PFB_CASE(11)
        PFB_CASE(12) PFB_CASE(13) PFB_CASE(14) PFB_CASE(15) PFB_CASE(16)
#undef PFB_CASE
        default: throw std::runtime_error("unsupported kernel support");
    }
    PFB_HIP(hipGetLastError());
}

}  // namespace pfbhip
