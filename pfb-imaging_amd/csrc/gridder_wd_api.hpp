// gridder_wd_api.hpp -- host entry points of the one-plane w-scheme's kernels (info.wmode == 2; gridder_kernels_wd.hpp,
// compiled in their own translation unit, gridder_wd.hip).
#pragma once
#include "gridder_kernels_mp.hpp"

namespace pfbhip {

constexpr int WD_MAX_K = 4;

struct WdArgs {
    int K;                      // kernel functions per axis (2..4)
    int W;
    int bc;                     // block edge of the scatter's register frame: 4, or 2 when the sort key carries 2 x 2-cell blocks (W = 14, 15)
    double whalf, nshift;
    double tq[WD_MAX_K];        // t(s_q) = n(s_q) - 1 + nshift at the K interpolation nodes in s
    double M[WD_MAX_K][WD_MAX_K];  // C_k = sum_q M[k][q] exp(-2 pi i dw t_q)
    double su[WD_MAX_K], sv[WD_MAX_K];  // (-alpha_u / smax)^k, (-alpha_v / smax)^k
    const double *dtab;         // (K, W, D + 1): 2k-th derivative (in x) of the kernel polynomial, unscaled
    const double2 *cw;          // (nactive + REC_PAD, K): C_k of every sorted visibility (gridding direction)
};

// the block edge the scatter of a plan of support W runs with, given whether its sort key carries 2 x 2-cell blocks
__host__ __device__ constexpr int wd_block_edge(int W, bool fine_key) { return (W == 14 || W == 15) && fine_key ? 2 : BLK_CELLS; }
// threads per workgroup of the scatter (the caller sizes the dynamic LDS with wd_scatter_lds_bytes)
int wd_scatter_threads();
size_t wd_scatter_lds_bytes(int W);
size_t wd_gather_lds_bytes();
// C_k of every sorted visibility (plan time)
void wd_launch_coeffs(const WdArgs &wa, int64_t nactive, const double *pw, double2 *cw, hipStream_t st);
// pval[j][k] = sval[j] C_k(j)  (scatter input outside Hessian applies)
void wd_launch_plane_values(int K, int64_t nactive, const double2 *cw, const double2 *sval, double2 *pval, hipStream_t st);
// one colour launch of the scatter / the gather of the work list in ga.a (W = 4..16, wa.K = 2..4)
void wd_launch_grid(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *pval, double2 *grid, hipStream_t st);
void wd_launch_degrid(const GroupArgs &ga, const WdArgs &wa, const VisRec *rec, const double2 *grid, double2 *sacc,
                      const double *swgt, double2 *pval_out, hipStream_t st);

}  // namespace pfbhip
