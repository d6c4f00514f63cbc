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
// Mapping (both kernels): one workgroup per work item, the (TILE+W-1)^2 footprint of the tile
// in LDS.  A wavefront handles FOUR visibilities at a time, one per 16-lane DPP row.  Lane b of
// a row owns footprint column b: it evaluates the v-kernel of tap b and the u-kernel of tap b
// (two Horner chains on per-lane register coefficients -- no exp/sqrt), then walks the W x W
// footprint along wrapped diagonals: at step i it holds the u-kernel value of row (b+i) mod 16,
// obtained by rotating the row's u-values one lane per step with a DPP row_ror (a VALU move,
// no LDS traffic).  At every step the 16 lanes of a row touch 16 different rows AND columns;
// with an even LDS row stride that is bank-conflict free.
//   scatter (k_grid):   LDS f64 atomics (ds_add_f64) into two planes (re, im), then the tile is
//                       flushed to HBM with global f64 atomics (halo cells are shared by tiles);
//   gather  (k_degrid): the tile is loaded into LDS as interleaved complex (ds_read_b128 per
//                       tap), per-lane partial sums, 4-step DPP row reduction.
#pragma once
#include <hip/hip_runtime.h>

#include "vismap.hpp"

namespace pfbhip {

constexpr int CHUNK = 4096;        // sorted visibilities per work item
constexpr int GRID_THREADS = 256;  // 4 wavefronts per workgroup

struct WorkItem {
    uint32_t tile, begin, end, pad;
};

constexpr int MAX_POLY_PLANES = 24;

__host__ __device__ constexpr int kernel_poly_degree_c(int W) { return W + 6 > 20 ? 20 : (W + 6 < 12 ? 12 : W + 6); }

struct PlaneArgs {
    int nu, nv, ntv;
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
__device__ __forceinline__ int rot1_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false); }
__device__ __forceinline__ double rot1_f64(double v)
{
    int lo = rot1_i32(__double2loint(v)), hi = rot1_i32(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ double rotn_f64(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Work items are sorted by decreasing size and taken in block order: blocks are dealt
// round-robin over the 8 XCDs, so every XCD gets the same mix of heavy and light items.  (A
// contiguous run of tile-sorted items per XCD was measured 2x slower: the uv density peaks at
// the centre, so two XCDs got nearly all the work.)
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t) { return b; }

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

// w-plane weight of a visibility for plane a.plane (0 = does not touch the plane)
template <int W, int D>
__device__ __forceinline__ double plane_weight(const PlaneArgs &a, double pw, const double *wtab /* LDS (W, D+1) */)
{
    if (!a.do_w) return 1.0;
    if (a.wmode == 0) {
        const double shift = 1.0 - 0.5 * double(W);
        const double fl = floor(pw + shift);
        const int dp = a.plane - (int)fl;
        if (dp < 0 || dp >= W) return 0.0;
        const double z = 2.0 * ((pw + shift) - fl) - 1.0;
        const double *c = wtab + dp * (D + 1);
        double v = c[D];
#pragma unroll
        for (int k = D - 1; k >= 0; --k) v = fma(v, z, c[k]);
        return v;
    }
    double kw = a.coef;
    for (int m = 0; m < a.nplanes; ++m)
        if (m != a.plane) kw *= (pw - a.nodes[m]);
    return kw;
}

template <int W>
__global__ void __launch_bounds__(GRID_THREADS) k_grid(PlaneArgs a, const double2 *__restrict__ sval,
                                                        double2 *__restrict__ grid)
{
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int L = TILE + W - 1;
    constexpr int LS = tile_stride(W);  // even row stride: conflict-free diagonal walk
    constexpr int LL = tile_rows(W) * LS;
    extern __shared__ double lds[];
    double *lre = lds;
    double *lim = lds + LL;
    double *wtab = lds + 2 * LL;

    uint32_t item = xcd_swizzle(blockIdx.x, gridDim.x);
    if (item >= a.nwork) return;
    const WorkItem wi = a.work[item];
    for (int i = threadIdx.x; i < 2 * LL; i += GRID_THREADS) lds[i] = 0.0;
    for (int i = threadIdx.x; i < W * (D + 1); i += GRID_THREADS) wtab[i] = a.ktab[i];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane & 15, g = lane >> 4;
    double c[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) c[k] = b < W ? a.ktab[b * (D + 1) + k] : 0.0;
    __syncthreads();

    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    const double shift = 1.0 - 0.5 * double(W);

    // Records are streamed 4 visibilities (one per DPP row) at a time, one step ahead.
    const uint32_t stride = (GRID_THREADS / 64) * 4;  // visibilities per workgroup step
    uint32_t j = wi.begin + wave * 4 + g;
    bool valid = j < wi.end;
    double pu = valid ? a.pu[j] : 0.0, pv = valid ? a.pv[j] : 0.0, pw = (valid && a.do_w) ? a.pw[j] : 0.0;
    double2 val = valid ? sval[j] : make_double2(0.0, 0.0);
    for (uint32_t jb = wi.begin + wave * 4; jb < wi.end; jb += stride) {
        // prefetch the next step's records
        const uint32_t jn = j + stride;
        const bool nvalid = jn < wi.end;
        const double npu = nvalid ? a.pu[jn] : 0.0, npv = nvalid ? a.pv[jn] : 0.0;
        const double npw = (nvalid && a.do_w) ? a.pw[jn] : 0.0;
        const double2 nval = nvalid ? sval[jn] : make_double2(0.0, 0.0);
        {
            double kw = plane_weight<W, D>(a, pw, wtab);
            if (!valid) kw = 0.0;
            const double fu = floor(pu + shift), fv = floor(pv + shift);
            const double zu = 2.0 * ((pu + shift) - fu) - 1.0, zv = 2.0 * ((pv + shift) - fv) - 1.0;
            double ku = horner<D>(c, zu);  // 0 for lanes b >= W (zero coefficients)
            const double kv = horner<D>(c, zv) * kw;
            const int lu = wrap_once((int)fu, a.nu) - bu, lv = wrap_once((int)fv, a.nv) - bv;
            const double vr = val.x * kv, vi = val.y * kv;
            const int colbase = lu * LS + lv + b;
            int arow = b;
            if (kw != 0.0) {  // rows that do not touch this plane (or are past the end) sit out
                // Every lane adds at every step: taps outside the W x W footprint carry ku == 0 or
                // kv == 0 and land in the padding rows/columns of the tile (LR rows are allocated).
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int off = colbase + arow * LS;
                    if (W >= 14 || (arow < W && b < W)) {
                        unsafeAtomicAdd(&lre[off], vr * ku);
                        unsafeAtomicAdd(&lim[off], vi * ku);
                    }
                    ku = rot1_f64(ku);
                    arow = rot1_i32(arow);
                }
            }
        }
        j = jn;
        valid = nvalid;
        pu = npu;
        pv = npv;
        pw = npw;
        val = nval;
    }
    __syncthreads();
    double *gp = reinterpret_cast<double *>(grid);
    for (int i = threadIdx.x; i < L * L; i += GRID_THREADS) {
        const int la = i / L, lb = i - la * L;
        const double re = lre[la * LS + lb], im = lim[la * LS + lb];
        if (re != 0.0 || im != 0.0) {
            int gu = bu + la, gv = bv + lb;
            gu = gu >= a.nu ? gu % a.nu : gu;
            gv = gv >= a.nv ? gv % a.nv : gv;
            const size_t o = (size_t(gu) * size_t(a.nv) + size_t(gv)) * 2;
            unsafeAtomicAdd(&gp[o], re);
            unsafeAtomicAdd(&gp[o + 1], im);
        }
    }
}

template <int W>
__global__ void __launch_bounds__(GRID_THREADS) k_degrid(PlaneArgs a, const double2 *__restrict__ grid,
                                                          double2 *__restrict__ sacc)
{
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int L = TILE + W - 1;
    constexpr int LS = tile_stride(W);
    constexpr int LL = tile_rows(W) * LS;
    extern __shared__ double lds[];
    double2 *tile = reinterpret_cast<double2 *>(lds);
    double *wtab = lds + 2 * LL;

    uint32_t item = xcd_swizzle(blockIdx.x, gridDim.x);
    if (item >= a.nwork) return;
    const WorkItem wi = a.work[item];
    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    for (int i = threadIdx.x; i < L * L; i += GRID_THREADS) {
        const int la = i / L, lb = i - la * L;
        int gu = bu + la, gv = bv + lb;
        gu = gu >= a.nu ? gu % a.nu : gu;
        gv = gv >= a.nv ? gv % a.nv : gv;
        tile[la * LS + lb] = grid[size_t(gu) * size_t(a.nv) + size_t(gv)];
    }
    if (W >= 14)  // padding cells are read (and multiplied by 0): keep them finite
        for (int i = threadIdx.x; i < LL; i += GRID_THREADS) {
            const int la = i / LS, lb = i - la * LS;
            if (la >= L || lb >= L) tile[i] = make_double2(0.0, 0.0);
        }
    for (int i = threadIdx.x; i < W * (D + 1); i += GRID_THREADS) wtab[i] = a.ktab[i];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane & 15, g = lane >> 4;
    double c[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) c[k] = b < W ? a.ktab[b * (D + 1) + k] : 0.0;
    __syncthreads();

    const double shift = 1.0 - 0.5 * double(W);

    const uint32_t stride = (GRID_THREADS / 64) * 4;
    uint32_t j = wi.begin + wave * 4 + g;
    bool valid = j < wi.end;
    double pu = valid ? a.pu[j] : 0.0, pv = valid ? a.pv[j] : 0.0, pw = (valid && a.do_w) ? a.pw[j] : 0.0;
    for (uint32_t jb = wi.begin + wave * 4; jb < wi.end; jb += stride) {
        const uint32_t jn = j + stride;
        const bool nvalid = jn < wi.end;
        const double npu = nvalid ? a.pu[jn] : 0.0, npv = nvalid ? a.pv[jn] : 0.0;
        const double npw = (nvalid && a.do_w) ? a.pw[jn] : 0.0;
        {
            double kw = plane_weight<W, D>(a, pw, wtab);
            if (!valid) kw = 0.0;
            const double fu = floor(pu + shift), fv = floor(pv + shift);
            const double zu = 2.0 * ((pu + shift) - fu) - 1.0, zv = 2.0 * ((pv + shift) - fv) - 1.0;
            double ku = horner<D>(c, zu);
            const double kv = horner<D>(c, zv) * kw;
            const int lu = wrap_once((int)fu, a.nu) - bu, lv = wrap_once((int)fv, a.nv) - bv;
            const int colbase = lu * LS + lv + b;
            int arow = b;
            double sr = 0.0, si = 0.0;
            if (kw != 0.0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (W >= 14 || (arow < W && b < W)) {
                        const double2 gval = tile[colbase + arow * LS];
                        sr = fma(gval.x, ku, sr);
                        si = fma(gval.y, ku, si);
                    }
                    ku = rot1_f64(ku);
                    arow = rot1_i32(arow);
                }
            }
            sr *= kv;
            si *= kv;
            sr += rotn_f64<8>(sr);
            si += rotn_f64<8>(si);
            sr += rotn_f64<4>(sr);
            si += rotn_f64<4>(si);
            sr += rotn_f64<2>(sr);
            si += rotn_f64<2>(si);
            sr += rotn_f64<1>(sr);
            si += rotn_f64<1>(si);
            if (b == 0 && kw != 0.0) {
                double2 acc = sacc[j];
                acc.x += sr;
                acc.y += si;
                sacc[j] = acc;
            }
        }
        j = jn;
        valid = nvalid;
        pu = npu;
        pv = npv;
        pw = npw;
    }
}

}  // namespace pfbhip
