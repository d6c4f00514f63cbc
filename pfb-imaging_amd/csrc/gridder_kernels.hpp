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
//
// Scatter (k_grid): one workgroup per work item; the (TILE+W-1)^2 footprint of the tile
// lives in LDS as two planes (re, im); each wavefront reads 64 records coalesced, then walks
// them with v_readlane broadcasts; lanes < 32 evaluate the 2W one-dimensional kernel values
// once per visibility, the W^2 taps are spread over the 64 lanes (ds_bpermute broadcast of
// the 1-D values) and accumulated with LDS f64 atomics; the tile is flushed to HBM with
// global f64 atomics (halo cells are shared with neighbouring tiles).
// Gather (k_degrid): the same walk with the tile loaded into LDS and a wavefront reduction.
#pragma once
#include <hip/hip_runtime.h>

#include "vismap.hpp"

namespace pfbhip {

constexpr int CHUNK = 2048;        // sorted visibilities per work item
constexpr int GRID_THREADS = 256;  // 4 wavefronts per workgroup

struct WorkItem {
    uint32_t tile, begin, end, pad;
};

struct PlaneArgs {
    int nu, nv, ntv;
    int do_w;
    int plane;
    double beta;
    const double *pu, *pv, *pw;  // tile-sorted records
    const WorkItem *work;
    uint32_t nwork;
};

__device__ __forceinline__ double es_kernel(double x, double beta)
{
    double t = 1.0 - x * x;
    return t >= 0.0 ? exp(beta * (sqrt(t) - 1.0)) : 0.0;
}

__device__ __forceinline__ double readlane_f64(double v, int k)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

// Blocks b and b+8 share an XCD (round-robin dispatch): give every XCD a contiguous run of
// work items so neighbouring tiles (shared halo lines) meet in one L2.  Speed only.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t n)
{
    uint32_t per = (n + 7u) / 8u;
    uint32_t item = (b & 7u) * per + (b >> 3);
    return item;  // may be >= n: caller checks
}

template <int W>
__global__ void __launch_bounds__(GRID_THREADS) k_grid(PlaneArgs a, const double2 *__restrict__ sval,
                                                        double2 *__restrict__ grid)
{
    constexpr int L = TILE + W - 1;
    constexpr int LL = L * L;
    extern __shared__ double lds[];
    double *lre = lds;
    double *lim = lds + LL;

    uint32_t item = xcd_swizzle(blockIdx.x, gridDim.x);
    if (item >= a.nwork) return;
    const WorkItem wi = a.work[item];
    for (int i = threadIdx.x; i < 2 * LL; i += GRID_THREADS) lds[i] = 0.0;
    __syncthreads();

    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double xs = 2.0 / double(W), shift = 1.0 - 0.5 * double(W);
    const double dplane = double(a.plane);

    for (uint32_t base = wi.begin + wave * 64; base < wi.end; base += (GRID_THREADS / 64) * 64) {
        const uint32_t j = base + lane;
        const bool ok = j < wi.end;
        const double lpu = ok ? a.pu[j] : 0.0;
        const double lpv = ok ? a.pv[j] : 0.0;
        const double lpw = (ok && a.do_w) ? a.pw[j] : 0.0;
        const double2 lval = ok ? sval[j] : make_double2(0.0, 0.0);
        const int cnt = min(64u, wi.end - base);
        for (int k = 0; k < cnt; ++k) {
            const double cpu = readlane_f64(lpu, k), cpv = readlane_f64(lpv, k);
            double kw = 1.0;
            if (a.do_w) {
                const double cpw = readlane_f64(lpw, k);
                const int dp = a.plane - (int)floor(cpw + shift);
                if (dp < 0 || dp >= W) continue;  // wave-uniform
                kw = es_kernel((dplane - cpw) * xs, a.beta);
            }
            const int iu0 = (int)floor(cpu + shift), iv0 = (int)floor(cpv + shift);
            const int lu = wrap_index(iu0, a.nu) - bu, lv = wrap_index(iv0, a.nv) - bv;
            // lanes 0..W-1: u taps, lanes 16..16+W-1: v taps
            double kval = 0.0;
            {
                const int tap = lane & 15;
                if (lane < 32 && tap < W) {
                    const bool isv = (lane & 16) != 0;
                    const double x = (double((isv ? iv0 : iu0) + tap) - (isv ? cpv : cpu)) * xs;
                    kval = es_kernel(x, a.beta);
                }
            }
            const double vr = readlane_f64(lval.x, k) * kw, vi = readlane_f64(lval.y, k) * kw;
#pragma unroll
            for (int t0 = 0; t0 < W * W; t0 += 64) {
                const int t = t0 + lane;
                const int ta = t / W, tb = t - ta * W;
                const double ku = __shfl(kval, ta & 15);
                const double kv = __shfl(kval, 16 + (tb & 15));
                if (t < W * W) {
                    const double kk = ku * kv;
                    const int off = (lu + ta) * L + lv + tb;
                    unsafeAtomicAdd(&lre[off], vr * kk);
                    unsafeAtomicAdd(&lim[off], vi * kk);
                }
            }
        }
    }
    __syncthreads();
    double *g = reinterpret_cast<double *>(grid);
    for (int i = threadIdx.x; i < LL; i += GRID_THREADS) {
        const double re = lre[i], im = lim[i];
        if (re != 0.0 || im != 0.0) {
            const int la = i / L, lb = i - la * L;
            int gu = bu + la, gv = bv + lb;
            gu = gu >= a.nu ? gu % a.nu : gu;
            gv = gv >= a.nv ? gv % a.nv : gv;
            const size_t o = (size_t(gu) * size_t(a.nv) + size_t(gv)) * 2;
            unsafeAtomicAdd(&g[o], re);
            unsafeAtomicAdd(&g[o + 1], im);
        }
    }
}

template <int W>
__global__ void __launch_bounds__(GRID_THREADS) k_degrid(PlaneArgs a, const double2 *__restrict__ grid,
                                                          double2 *__restrict__ sacc)
{
    constexpr int L = TILE + W - 1;
    constexpr int LL = L * L;
    extern __shared__ double lds[];
    double2 *tile = reinterpret_cast<double2 *>(lds);

    uint32_t item = xcd_swizzle(blockIdx.x, gridDim.x);
    if (item >= a.nwork) return;
    const WorkItem wi = a.work[item];
    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    for (int i = threadIdx.x; i < LL; i += GRID_THREADS) {
        const int la = i / L, lb = i - la * L;
        int gu = bu + la, gv = bv + lb;
        gu = gu >= a.nu ? gu % a.nu : gu;
        gv = gv >= a.nv ? gv % a.nv : gv;
        tile[i] = grid[size_t(gu) * size_t(a.nv) + size_t(gv)];
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double xs = 2.0 / double(W), shift = 1.0 - 0.5 * double(W);
    const double dplane = double(a.plane);

    for (uint32_t base = wi.begin + wave * 64; base < wi.end; base += (GRID_THREADS / 64) * 64) {
        const uint32_t j = base + lane;
        const bool ok = j < wi.end;
        const double lpu = ok ? a.pu[j] : 0.0;
        const double lpv = ok ? a.pv[j] : 0.0;
        const double lpw = (ok && a.do_w) ? a.pw[j] : 0.0;
        const int cnt = min(64u, wi.end - base);
        double outr = 0.0, outi = 0.0;  // lane k keeps the result of visibility base+k
        for (int k = 0; k < cnt; ++k) {
            const double cpu = readlane_f64(lpu, k), cpv = readlane_f64(lpv, k);
            double kw = 1.0;
            if (a.do_w) {
                const double cpw = readlane_f64(lpw, k);
                const int dp = a.plane - (int)floor(cpw + shift);
                if (dp < 0 || dp >= W) continue;  // wave-uniform
                kw = es_kernel((dplane - cpw) * xs, a.beta);
            }
            const int iu0 = (int)floor(cpu + shift), iv0 = (int)floor(cpv + shift);
            const int lu = wrap_index(iu0, a.nu) - bu, lv = wrap_index(iv0, a.nv) - bv;
            double kval = 0.0;
            {
                const int tap = lane & 15;
                if (lane < 32 && tap < W) {
                    const bool isv = (lane & 16) != 0;
                    const double x = (double((isv ? iv0 : iu0) + tap) - (isv ? cpv : cpu)) * xs;
                    kval = es_kernel(x, a.beta);
                }
            }
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int t0 = 0; t0 < W * W; t0 += 64) {
                const int t = t0 + lane;
                const int ta = t / W, tb = t - ta * W;
                const double ku = __shfl(kval, ta & 15);
                const double kv = __shfl(kval, 16 + (tb & 15));
                if (t < W * W) {
                    const double kk = ku * kv;
                    const double2 gval = tile[(lu + ta) * L + lv + tb];
                    sr += gval.x * kk;
                    si += gval.y * kk;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sr += __shfl_xor(sr, o);
                si += __shfl_xor(si, o);
            }
            if (lane == k) {
                outr = sr * kw;
                outi = si * kw;
            }
        }
        if (ok) {
            double2 acc = sacc[j];
            acc.x += outr;
            acc.y += outi;
            sacc[j] = acc;
        }
    }
}

}  // namespace pfbhip
