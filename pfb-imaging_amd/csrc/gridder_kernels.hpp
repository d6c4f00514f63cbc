// gridder_kernels.hpp -- HIP kernels of the w-stacking gridder/degridder (gfx950).
//
// Data layout in HBM (all double precision, the reference's hard-coded precision,
// /root/reference/src/pfb_imaging/operators/hessian.py:293-295):
//   grid      (nu, nv) interleaved complex, v contiguous      one w-plane of the oversampled uv-grid
//   image     (nx, ny) real, y contiguous                     accumulator / correction / beam images
//   records   SoA over the nactive unmasked visibilities in TILE-SORTED order:
//             pu[], pv[], pw[] grid coordinates; src[] original index | flip<<31
//   sval      nactive interleaved complex, tile-sorted         weighted visibilities (grid input / degrid output)
//   work      (tile, begin, end) chunks of <= CHUNK sorted visibilities of one tile
//   coef      (W, D+1) monomial coefficients of the kernel's piecewise polynomial (eskernel.hpp)
//
// The kernels themselves are in gridder_kernels_mp.hpp; this header holds the shared types/helpers.
#pragma once
#include <hip/hip_runtime.h>

#include "vismap.hpp"

namespace pfbhip {

constexpr int CHUNK = 4096;        // sorted visibilities per work item

struct WorkItem {
    uint32_t tile, begin, end, pad;
};

constexpr int MAX_POLY_PLANES = 24;

// degree 12 for every support: beyond ~10 the error is set by the kernel's square-root end-point singularity (the
// two outermost pieces), not by the degree -- 6e-14 at W = 16, checked against 0.25 x the row's epsilon at plan time
__host__ __device__ constexpr int kernel_poly_degree_c(int) { return 12; }

struct PlaneArgs {
    int nu, nv, ntv;
    int apitch;   // complex elements between consecutive rows of the uv-plane buffer (>= nv: padded off the power-of-two pitch)
    int do_w;
    int plane;
    int wmode;    // 0: ES kernel over equispaced planes, 1: Lagrange weights over Chebyshev nodes
    int nplanes;
    double coef;                    // wmode 1: prod_{m != plane} 1 / (s_plane - s_m)
    double nodes[MAX_POLY_PLANES];  // wmode 1: interpolation abscissae s_m in [-1, 1]
    const double *pu, *pv, *pw;     // tile-sorted records
    const double *ktab;             // (W, D+1) kernel polynomial coefficients
    const WorkItem *work;
    uint32_t nwork;
};

// value of lane (l -/+ 1) of the same 16-lane row (direction is irrelevant to the callers: they
// rotate the tap index alongside the data)
// (mov_dpp leaves the "old" operand undefined: every lane is written, and update_dpp(0, ...) costs a
// v_mov to materialise the zero in front of every rotation)
__device__ __forceinline__ int rot1_i32(int v) { return __builtin_amdgcn_mov_dpp(v, 0x121, 0xf, 0xf, false); }
__device__ __forceinline__ double rot1_f64(double v)
{
    int lo = rot1_i32(__double2loint(v)), hi = rot1_i32(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ double rotn_f64(double v)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x120 + N, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// LDS tile geometry shared by host (allocation) and device: rows of even stride; the branch-free
// diagonal walk of wide kernels (W >= 14) may touch row/column T+15 with zero contributions.
__host__ __device__ constexpr int tile_stride(int W)
{
    return W >= 14 ? TILE + 16 : (((TILE + W - 1) & 1) ? TILE + W : TILE + W - 1);
}
__host__ __device__ constexpr int tile_rows(int W) { return W >= 14 ? TILE + 16 : TILE + W - 1; }

__device__ __forceinline__ int wrap_once(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

template <int D>
__device__ __forceinline__ double horner(const double (&c)[D + 1], double z)
{
    double v = c[D];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) v = fma(v, z, c[k]);
    return v;
}

}  // namespace pfbhip
