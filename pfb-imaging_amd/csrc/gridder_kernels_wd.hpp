// gridder_kernels_wd.hpp -- scatter / gather of the ONE-PLANE w-scheme (info.wmode == 2, round 4).
//
// Narrow fields / near-coplanar arrays (BASELINE C2: omega = 2 pi (w range / 2) max|n - 1| ~ 0.01) used to interpolate the w-term
// through K = 3 Chebyshev planes in w: three uv-planes scattered, transformed, cropped, padded, transformed and gathered per
// apply, and the plane transforms were 57 % of it.  The same K-term accuracy is available on ONE plane:
//
//     exp(-2 pi i w t(s)) = exp(-2 pi i wc t(s)) * E(dw; s),      s = l^2 + m^2 (phase centre on axis), dw = w - wc
//     E(dw; s) ~ sum_k C_k(dw) (s / smax)^k                        (Chebyshev interpolation in s at K nodes of [0, smax])
//     multiplication by l^2 in the image  <->  -(nu px / (pi W))^2 d^2/dx^2 on the u-kernel phi(x)   (exact for the kernel's
//                                                                                                      own Fourier transform)
//
// so the first factor is the w-screen of a single plane at wc (the fused second-axis kernels apply it as before) and the
// second one goes INTO the gridding kernel of each visibility:
//
//     footprint(i, j) = sum_k C_k D^k[phi phi](i, j),   D^k[phi phi] = sum_r binom(k, r) a_r(i) b_{k-r}(j),
//     a_r = (-alpha_u / smax)^r phi^(2r)(x_i),  b_r = (-alpha_v / smax)^r phi^(2r)(y_j)
//         = sum_r a_r(i) S_r(j),                S_r(j) = sum_m binom(r + m, r) C_{r+m} b_m(j)            (K complex per column)
//
// A cell costs 2 K real FMAs per visibility -- what the K planes cost -- and everything that scaled with the plane count is
// paid once: LDS tile, accumulator registers, flush, plane clear, four row-FFT kernels.  The derivatives are those of the
// kernel's piecewise POLYNOMIAL (the function the device evaluates anyway); they enter at relative weight omega^k / k!, so
// their own accuracy needs are loose (1e-5 / 1e-3 of their size at C2).  Restated on the CPU in oracle/pfb_oracle.c
// (pfbo_grid_plane_wd / pfbo_degrid_plane_wd) and oracle/wgridder.py (Plan._init_wd); tools/proto_wderiv.py is the numpy
// prototype against the direct DFT.
#pragma once
#include "gridder_wd_api.hpp"

namespace pfbhip {

__host__ __device__ constexpr int wd_binom(int n, int r)
{
    return r == 0 || r == n ? 1 : (n == 2 ? 2 : (n == 3 ? 3 : 1));  // n <= 3
}

// plan time: C_k of every visibility; pw = (w - wcenter) / whalf
__global__ void k_wd_coeffs(WdArgs wa, int64_t nactive, const double *__restrict__ pw, double2 *__restrict__ cw)
{
    const int64_t j = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (j >= nactive + REC_PAD) return;
    double er[WD_MAX_K], ei[WD_MAX_K];
    const double dw = j < nactive ? pw[j] * wa.whalf : 0.0;
    for (int q = 0; q < wa.K; ++q) {
        double ph = dw * wa.tq[q];
        ph -= rint(ph);
        double s, c;
        sincospi(-2.0 * ph, &s, &c);
        er[q] = c;
        ei[q] = s;
    }
    for (int k = 0; k < wa.K; ++k) {
        double xr = 0.0, xi = 0.0;
        for (int q = 0; q < wa.K; ++q) {
            xr = fma(wa.M[k][q], er[q], xr);
            xi = fma(wa.M[k][q], ei[q], xi);
        }
        cw[size_t(j) * size_t(wa.K) + size_t(k)] = j < nactive ? make_double2(xr, xi) : make_double2(0.0, 0.0);
    }
}

// per apply (outside Hessian applies): pval[j][k] = sval[j] * C_k(j)
__global__ void k_plane_values_wd(int K, int64_t nactive, const double2 *__restrict__ cw, const double2 *__restrict__ sval,
                                  double2 *__restrict__ pval)
{
    const int64_t j = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (j >= nactive) return;
    const double2 v = sval[j];
    for (int k = 0; k < K; ++k) {
        const double2 c = cw[size_t(j) * size_t(K) + size_t(k)];
        pval[size_t(j) * size_t(K) + size_t(k)] = make_double2(v.x * c.x - v.y * c.y, v.x * c.y + v.y * c.x);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Scatter: k_grid_rec's structure (one visibility at a time per wave, its (W + 3)^2 block footprint in registers, records
// and values through scalar loads, three scratch lines round robin) with the K kernel functions.
//   scratch line (per wave, 3 of them): SU[22][4] then SV[22][4] doubles -- entry t + G holds (a_0 .. a_3)(tap t), zero guards of
//   G = 3 entries in front and behind; a lane reads whole entries (ds_read_b128 [+ b64 / b128]).
//   Workgroup size (round 4, measured on C2): 256 threads -- one wave per SIMD --, three workgroups per CU: the same 12 waves as
//   one 768-thread workgroup, but a workgroup's prologue (10 k cycles), its wait for the slowest wave and its tile flush overlap
//   the other two's visibility loops.  (384 threads x 2 was SLOWER than 768 x 1: six waves do not spread evenly over four
//   SIMDs, and at 3 waves per SIMD by registers the second workgroup does not fit beside the first.)  The LDS budget of three
//   workgroups is what sets the tile stride (the region's edge rounded up to odd) and the 22-entry scratch arrays.
//   kernel evaluation (round 4b): for a PAIR of visibilities at a time -- lanes 0..15 u / 16..31 v of the first, 32..47 u / 48..63 v of the
//   second; lane (axis, tap) runs the K Horner chains (degrees D, D - 2, D - 4 [, D - 6]) on its own coefficient registers and writes
//   its whole scratch entry.  Visibility s lives in line s mod 3; the pair (s + 1, s + 2) is evaluated under every second visibility.
constexpr int WD_ENT = 22;               // entries per scratch array: G + 16 taps + G
constexpr int WD_LINE = 2 * WD_ENT * 4;  // doubles per scratch line
constexpr int WD_NLINE = 3;
__host__ __device__ constexpr int wd_threads() { return 768; }  // register budget: 768 threads per CU (168 VGPRs), as 1 x 768 or 3 x 256
// tile stride: odd, >= the region's edge (an even stride puts two of a flush's three rows on the same banks)
__host__ __device__ constexpr int wd_stride(int W) { return (TILE + W - 1) | 1; }
__host__ __device__ constexpr size_t wd_lds_doubles(int W, int waves)
{
    return size_t(2) * blk_tile_rows(W) * wd_stride(W) + size_t(waves) * WD_NLINE * WD_LINE + 64;  // (+ slack for the idle lanes' reads)
}

// Block edge BC (the anchor of the register footprint) and lane layout.  The frame a wave holds is FP = W + BC - 1 cells a side.
//   FP <= 16 (round 4b): 16 columns x 4 row groups, rows 4 k + g, NR = ceil(FP / 4) <= 4 cells per lane -- every lane works, and
//   W = 15 (C2) costs 4 x 2K cell FMAs per visibility instead of 6 x 2K.  W <= 13 fits with the 4 x 4-cell blocks of the tile sort;
//   W = 14, 15 take 2 x 2-cell blocks (the sort key then carries the 2 x 2 block inside the 4 x 4 one: runs a quarter as long,
//   flushes of 8 instead of 12 LDS atomics).
//   FP > 16 (W = 16, or W = 14 / 15 when the finer sort key does not fit 32 bits): 20 columns x 3 row groups, rows 3 k + g.
__host__ __device__ constexpr bool wd_frame16(int W, int BC) { return W + BC - 1 <= 16; }
__host__ __device__ constexpr int wd_rows_per_lane(int W, int BC)
{
    return wd_frame16(W, BC) ? (W + BC - 1 + 3) / 4 : (W + BC - 1 + 2) / 3;
}

template <int W, int NJ, int BC>
__global__ void __launch_bounds__(wd_threads()) k_grid_wd(GroupArgs ga, WdArgs wa, const VisRec *__restrict__ rec,
                                                           const double2 *__restrict__ pval, double2 *__restrict__ grid)
{
    const PlaneArgs &a = ga.a;
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int LS = wd_stride(W);
    constexpr int LL = blk_tile_rows(W) * LS;
    constexpr int FP = W + BC - 1;
    constexpr bool F16 = wd_frame16(W, BC);
    constexpr int NCOL = F16 ? 16 : 20, NGRP = F16 ? 4 : 3;
    constexpr int NR = wd_rows_per_lane(W, BC);
    constexpr int G = BC - 1;
    constexpr bool SKIP = !F16 && (FP % 3) == 1;
    static_assert(BC == 2 || BC == 4, "block edge: 2 or 4 cells");
    static_assert(FP <= NCOL && NR * NGRP >= FP && G + 16 + G <= WD_ENT, "frame does not fit the lane layout");
    const int BLK_THREADS = int(blockDim.x);
    extern __shared__ double lds[];
    double *scr_all = lds + 2 * LL;

    const uint32_t item = blockIdx.x;
    if (item >= a.nwork) return;
    const bool stamp = ga.dbg != nullptr;
    const unsigned long long ts0 = stamp ? __builtin_readcyclecounter() : 0ull;
    const WorkItem wi = a.work[item];
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t n = wi.end - wi.begin;
    const uint32_t NW = uint32_t(BLK_THREADS) / 64;
    auto share_begin = [&](uint32_t w) {
        // (workgroups of one wave per SIMD: no age classes inside the workgroup, equal shares)
        if (NW % 3 != 0) return w >= NW ? n : uint32_t((uint64_t(n) * w) / NW);
        const uint32_t per = NW / 3, cls = min(w / max(per, 1u), 2u), in = w - cls * per;
        float f = 0.f;
        for (uint32_t q = 0; q < cls; ++q) f += ga.wshare[q];
        const uint32_t ncls = cls == 2 ? NW - 2 * per : per;
        f += ga.wshare[cls] * float(in) / float(max(ncls, 1u));
        return w >= NW ? n : min(uint32_t(f * float(n)), n);
    };
    const uint32_t j0 = wi.begin + share_begin(uint32_t(wave));
    const uint32_t j1 = wi.begin + share_begin(uint32_t(wave) + 1);
    const uint32_t nmine = j1 - j0;
    constexpr uint32_t pbytes = uint32_t(NJ) * 16u;
    const char *rbase = reinterpret_cast<const char *>(rec);
    const char *pbase = reinterpret_cast<const char *>(pval);
    // Kernel evaluation.  PAIR (round 4b; the 16 x 16 frame with K <= 3, where the registers allow it): for TWO visibilities at a time --
    // lanes 0..15 u / 16..31 v of the pair's first visibility, 32..47 u / 48..63 v of its second; lane (axis, b) runs the NJ Horner
    // chains of tap b on its own coefficient registers (scaled by (-alpha / smax)^k of the lane's axis): 3 D - 5 instructions per pair
    // at NJ = 3, every lane busy.  Otherwise one visibility at a time: round 1 -- lanes 0..15 a_0, 16..31 b_0, 32..47 a_1, 48..63 b_1
    // (degree D), round 2 (K >= 3) -- a_2, b_2, a_3, b_3 (degree D - 4): 2 D - 2 instructions per visibility.
    constexpr bool PAIR = F16 && NJ <= 3;
    constexpr int D2 = D - 4;
    const int role = lane >> 4, b = lane & 15;
    const int axis = role & 1, vsel = role >> 1;
    // z of the lane's axis: PAIR -- of visibility j0 + vsel, + 2 per pair; else of visibility j0, + 1 per visibility
    const char *zptr = rbase + (size_t(j0) + size_t(PAIR ? vsel : 0)) * 32 + (axis ? 8 : 0);
    double cA[PAIR ? D + 1 : 1], cB[PAIR ? D - 1 : 1], cC[PAIR && NJ > 2 ? D - 3 : 1];
    double c1[PAIR ? 1 : D + 1], c2[PAIR ? 1 : D2 + 1];
    if constexpr (PAIR) {
        const bool on = b < W;
        const double s0 = axis ? wa.sv[0] : wa.su[0], s1 = axis ? wa.sv[1] : wa.su[1];
#pragma unroll
        for (int q = 0; q <= D; ++q) cA[q] = on ? wa.dtab[(size_t(0) * W + b) * (D + 1) + q] * s0 : 0.0;
#pragma unroll
        for (int q = 0; q <= D - 2; ++q) cB[q] = on ? wa.dtab[(size_t(1) * W + b) * (D + 1) + q] * s1 : 0.0;
        if constexpr (NJ > 2) {
            const double s2 = axis ? wa.sv[2] : wa.su[2];
#pragma unroll
            for (int q = 0; q <= D - 4; ++q) cC[q] = on ? wa.dtab[(size_t(2) * W + b) * (D + 1) + q] * s2 : 0.0;
        }
    } else {
        const int k1 = role >> 1, k2 = 2 + (role >> 1);
        const double s1 = axis ? wa.sv[k1] : wa.su[k1];
        const bool on1 = b < W && k1 < NJ;
#pragma unroll
        for (int q = 0; q <= D; ++q) c1[q] = on1 ? wa.dtab[(size_t(k1) * W + b) * (D + 1) + q] * s1 : 0.0;
        const bool on2 = b < W && k2 < NJ;
        const double s2 = on2 ? (axis ? wa.sv[k2] : wa.su[k2]) : 0.0;
#pragma unroll
        for (int q = 0; q <= D2; ++q) c2[q] = on2 ? wa.dtab[(size_t(k2 < NJ ? k2 : 0) * W + b) * (D + 1) + q] * s2 : 0.0;
    }
    static_assert(REC_PAD >= 63 + 63 + 3, "the warm-up reads one entry per lane up to 63 + 63 past the current visibility (z: at most five pairs ahead)");
    auto touch = [&](uint32_t first) {
        const uint32_t jt = first + uint32_t(lane);  // (padded arrays: no clamp)
        const int t0 = *reinterpret_cast<const int *>(rbase + size_t(jt) * 32 + 16);
        const double t1 = *reinterpret_cast<const double *>(pbase + size_t(jt) * pbytes);
        return double(t0) + t1;
    };
    double warm = touch(j0);
    constexpr int ZSTEP = PAIR ? 64 : 32;
    double zq[3];  // z of the lane's axis for the next three pairs (slot = pair index mod 3) / visibilities
#pragma unroll
    for (int u = 0; u < 3; ++u) zq[u] = *reinterpret_cast<const double *>(zptr + u * ZSTEP);
    zptr += 3 * ZSTEP;

    for (int i = threadIdx.x; i < 2 * LL; i += BLK_THREADS) lds[i] = 0.0;
    for (int i = threadIdx.x; i < (BLK_THREADS / 64) * WD_NLINE * WD_LINE; i += BLK_THREADS) scr_all[i] = 0.0;

    char *scr = reinterpret_cast<char *>(scr_all + wave * WD_NLINE * WD_LINE);
    const int g = lane / NCOL, cc = lane - NCOL * g;
    const bool act = g < NGRP && cc < FP;
    // where this lane writes its kernel values: entry (b + G) of SU (u lanes) / SV (v lanes) -- the whole entry (a_0 .. a_{NJ-1}) --
    // of the line of ITS visibility.  Visibility s lives in line s mod 3; the pair (s + 1, s + 2) is evaluated while visibility s
    // (s odd within a block of six) is processed: wq[q] for s = 2 q + 1.
    char *const went = scr + ((axis ? WD_ENT : 0) + b + G) * 32;
    char *wq[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) wq[q] = went + ((2 * q + 2 + vsel) % 3) * (WD_LINE * 8);
    // (one visibility at a time: slot k = role >> 1 of the entry in round 1, k + 2 in round 2)
    char *const wptr1 = went + (role >> 1) * 8;
    char *const wptr2 = wptr1 + 16;
    const char *suptr = scr + g * 32;               // + offu + 32 NGRP k: entry of row NGRP k + g
    const char *svptr = scr + WD_ENT * 32 + cc * 32;  // + offv: entry of column cc
    __syncthreads();

    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;

    double are[NR], aim[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) are[k] = aim[k] = 0.0;

    char *const tile0 = reinterpret_cast<char *>(lds) + (g * LS + cc) * 8;
    auto flush = [&](int blk) {
        const int r0 = (blk >> 8) * BC, c0 = (blk & 255) * BC;
        char *base = tile0 + (r0 * LS + c0) * 8;
        if (act) {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                if (NGRP * k + g < FP) {
                    unsafeAtomicAdd(reinterpret_cast<double *>(base + (NGRP * k * LS) * 8), are[k]);
                    unsafeAtomicAdd(reinterpret_cast<double *>(base + (LL + NGRP * k * LS) * 8), aim[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) are[k] = aim[k] = 0.0;
    };

    // Horner in z^2 on the even / odd coefficients (two independent half chains: a single wave can issue them back to back)
    auto evens = [](const auto &c, auto deg, double zz, double z2) {
        constexpr int DG = decltype(deg)::value;
        double e = c[DG], o = c[DG - 1];
#pragma unroll
        for (int k = DG - 2; k >= 0; k -= 2) {
            e = fma(e, z2, c[k]);
            if (k >= 1) o = fma(o, z2, c[k - 1]);
        }
        return fma(o, zz, e);
    };
    static_assert((D & 1) == 0 && D >= 8, "even polynomial degrees assumed");
    auto eval_pair = [&](double z, char *wp) {
        if constexpr (PAIR) {
            const double z2 = z * z;
            const double f0 = evens(cA, std::integral_constant<int, D>{}, z, z2);
            const double f1 = evens(cB, std::integral_constant<int, D - 2>{}, z, z2);
            *reinterpret_cast<double2 *>(wp) = make_double2(f0, f1);
            if constexpr (NJ == 3) *reinterpret_cast<double *>(wp + 16) = evens(cC, std::integral_constant<int, D - 4>{}, z, z2);
        }
    };
    auto stage_a = [&](double z, int line) {
        if constexpr (!PAIR) {
            const double z2 = z * z;
            *reinterpret_cast<double *>(wptr1 + line * (WD_LINE * 8)) = evens(c1, std::integral_constant<int, D>{}, z, z2);
            if constexpr (NJ > 2)
                *reinterpret_cast<double *>(wptr2 + line * (WD_LINE * 8)) = evens(c2, std::integral_constant<int, D2>{}, z, z2);
        }
    };
    const unsigned long long ts1 = stamp ? __builtin_readcyclecounter() : 0ull;
    if (nmine > 0) {
        if constexpr (PAIR) {  // pair 0 -> lines 0, 1
            eval_pair(zq[0], wq[2]);
            zq[0] = *reinterpret_cast<const double *>(zptr);
            zptr += ZSTEP;
        } else
            stage_a(zq[0], 0);
    }
    int cur = -1;
    int4 kq[3];
    double2 pq[3][NJ];
    const char *const rwave = rbase + size_t(j0) * 32;
    const char *const pwave = pbase + size_t(j0) * size_t(pbytes);
    uint32_t roff = 16u, poff = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        kq[u] = *reinterpret_cast<const int4 *>(rwave + roff);
#pragma unroll
        for (int p = 0; p < NJ; ++p) pq[u][p] = *reinterpret_cast<const double2 *>(pwave + poff + p * 16);
        roff += 32u;
        poff += pbytes;
    }
    kq[2] = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NJ; ++p) pq[2][p] = make_double2(0.0, 0.0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        asm volatile("" ::"s"(kq[u].x), "s"(kq[u].y), "s"(kq[u].z));
#pragma unroll
        for (int p = 0; p < NJ; ++p) asm volatile("" ::"s"(pq[u][p].x), "s"(pq[u][p].y));
    }
    // block of a visibility: the record's 4 x 4 block, or the 2 x 2 block of its first-tap cell (key = (lu << 8) | lv)
    auto block_of = [](const int4 &r) { return BC == 4 ? r.x : ((r.w >> 1) & 0x0F0F); };
    if (nmine > 0) cur = block_of(kq[0]);
    constexpr uint32_t BS = PAIR ? 6u : 3u, WS = PAIR ? 60u : 63u;  // unrolled block (visibility s: line s mod 3, pair s / 2), warm-up stride
    for (uint32_t wb = 0; wb < nmine; wb += WS) {
        asm volatile("" ::"v"(warm));
        warm = touch(j0 + wb + WS);
        const uint32_t wend = min(wb + WS, nmine);
        for (uint32_t sb = wb; sb < wend; sb += BS) {
#pragma unroll
            for (int i6 = 0; i6 < int(BS); ++i6) {
                if (sb + uint32_t(i6) >= wend) break;
                const int u = i6 % 3, ld = (u + 2) % 3, nx = (u + 1) % 3;
                double znext = 0.0;
                if constexpr (!PAIR) {
                    znext = zq[nx];
                    zq[u] = *reinterpret_cast<const double *>(zptr);
                    zptr += ZSTEP;
                }
                const int4 rk = kq[u];
                const int bnow = block_of(rk);
                if (bnow != cur) {
                    flush(cur);
                    cur = bnow;
                }
                // scratch offsets from the first-tap cell (rk.w = (lu << 8) | lv): entry G - du / G - dv
                const int offu = (G - ((rk.w >> 8) & (BC - 1))) * 32, offv = (G - (rk.w & (BC - 1))) * 32;
                double B[NJ];
                {
                    const char *sv = svptr + u * (WD_LINE * 8) + offv;
                    const double2 b01 = *reinterpret_cast<const double2 *>(sv);
                    B[0] = b01.x;
                    B[1] = b01.y;
                    if constexpr (NJ == 3) B[2] = *reinterpret_cast<const double *>(sv + 16);
                    if constexpr (NJ == 4) {
                        const double2 b23 = *reinterpret_cast<const double2 *>(sv + 16);
                        B[2] = b23.x;
                        B[3] = b23.y;
                    }
                }
                const char *su = suptr + u * (WD_LINE * 8) + offu;
                double A[NR][NJ];
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const double2 a01 = *reinterpret_cast<const double2 *>(su + 32 * NGRP * k);
                    A[k][0] = a01.x;
                    A[k][1] = a01.y;
                    if constexpr (NJ == 3) A[k][2] = *reinterpret_cast<const double *>(su + 32 * NGRP * k + 16);
                    if constexpr (NJ == 4) {
                        const double2 a23 = *reinterpret_cast<const double2 *>(su + 32 * NGRP * k + 16);
                        A[k][2] = a23.x;
                        A[k][3] = a23.y;
                    }
                }
                if constexpr (PAIR) {
                    if (i6 & 1) {  // kernel values of the pair (s + 1, s + 2) -> their lines
                        const int q = i6 >> 1, slot = (q + 1) % 3;
                        eval_pair(zq[slot], wq[q]);
                        zq[slot] = *reinterpret_cast<const double *>(zptr);
                        zptr += ZSTEP;
                    }
                } else
                    stage_a(znext, nx);  // kernel values of visibility s + 1 -> line nx
#pragma unroll
                for (int k = 0; k < NR; ++k)
#pragma unroll
                    for (int r = 0; r < NJ; ++r) asm volatile("" : "+v"(A[k][r])::"memory");
#pragma unroll
                for (int r = 0; r < NJ; ++r) asm volatile("" : "+v"(B[r])::"memory");
                kq[ld] = *reinterpret_cast<const int4 *>(rwave + roff);
#pragma unroll
                for (int p = 0; p < NJ; ++p) pq[ld][p] = *reinterpret_cast<const double2 *>(pwave + poff + p * 16);
                roff += 32u;
                poff += pbytes;
                __builtin_amdgcn_sched_barrier(0);
                // S_r = sum_m binom(r + m, r) P_{r+m} b_m  (P = value x C_k: wave-uniform, scalar operands)
                double sr[NJ], si[NJ];
#pragma unroll
                for (int r = 0; r < NJ; ++r) {
                    sr[r] = pq[u][r].x * B[0];
                    si[r] = pq[u][r].y * B[0];
#pragma unroll
                    for (int m = 1; r + m < NJ; ++m) {
                        const double bm = double(wd_binom(r + m, r)) * B[m];
                        sr[r] = fma(pq[u][r + m].x, bm, sr[r]);
                        si[r] = fma(pq[u][r + m].y, bm, si[r]);
                    }
                }
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    bool on = true;
                    if (SKIP && k == 0) on = (rk.w & 0x300) != 0x300;
                    if (SKIP && k == NR - 1) on = (rk.w & 0x300) == 0x300;
                    if (on) {
#pragma unroll
                        for (int r = 0; r < NJ; ++r) {
                            are[k] = fma(A[k][r], sr[r], are[k]);
                            aim[k] = fma(A[k][r], si[r], aim[k]);
                        }
                    }
                }
                asm volatile("" ::"s"(rk.y), "s"(rk.z));
            }
        }
    }
    asm volatile("" ::"v"(warm));
    if (cur >= 0) flush(cur);
    const unsigned long long ts2 = stamp ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();
    const unsigned long long ts3 = stamp ? __builtin_readcyclecounter() : 0ull;
    blk_tile_to_grid<W, 1, LS, 256>(ga, wi, lds, bu, bv, grid);
    if (stamp) {  // PFBHIP_STAMP=1: phase stamps (tools/stamp_scatter.py), the layout of k_grid_rec
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts4 = __builtin_readcyclecounter();
        unsigned long long *d = ga.dbg + size_t(item) * 8;
        if (threadIdx.x == 0) {
            d[0] = ts1 - ts0;
            d[1] = ts2 - ts1;
            d[2] = ts3 - ts2;
            d[3] = ts4 - ts3;
            d[4] = n;
            d[7] = wi.tile;
        }
        if (threadIdx.x == uint32_t(BLK_THREADS) - 64) {
            d[5] = ts2 - ts1;
            d[6] = ts3 - ts2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Gather: k_degrid_rw's row walk (lane b on footprint column (b - lv) mod 16, the u-kernel value broadcast from the lane that
// evaluated it by v_fmac_f64_dpp row_newbcast) on ONE tile, with K u-kernel functions: T_r(col) = sum_i a_r(i) cell(i, col),
// then value = sum_col sum_r T_r(col) conj(S_r(col)), S_r from the C_k of the visibility (see the header comment).
// (rows I >= W carry zero kernel values -- the lanes b >= W hold zero coefficients --: the walk stops at W)
template <int W, int NJ, int I>
__device__ __forceinline__ void wd_steps(const char *base, const double (&ku)[NJ], double (&tr)[NJ], double (&ti)[NJ])
{
    if constexpr (I < W) {
        const double2 cell = *reinterpret_cast<const double2 *>(base + size_t(I) * RW_LS * 16);
#pragma unroll
        for (int r = 0; r < NJ; ++r) {
            fmac_row_bcast<I>(tr[r], ku[r], cell.x);
            fmac_row_bcast<I>(ti[r], ku[r], cell.y);
        }
        wd_steps<W, NJ, I + 1>(base, ku, tr, ti);
    }
}

// register budget of the gather: 768 threads per CU (168 VGPRs: the three / four coefficient sets of K >= 3 do not fit the 128 of
// 1024 threads), launched as three workgroups of 256 (one wave per SIMD each: a workgroup's tile load -- 19 % of an item's
// cycles -- overlaps the other two's rounds; C2: 1.90 -> 1.64 ms)
__host__ __device__ constexpr int wd_gather_threads(int) { return 768; }

template <int W, int NJ>
__global__ void __launch_bounds__(wd_gather_threads(NJ)) k_degrid_wd(GroupArgs ga, WdArgs wa, const VisRec *__restrict__ rec,
                                                           const double2 *__restrict__ grid, double2 *__restrict__ sacc,
                                                           const double *__restrict__ swgt, double2 *__restrict__ pval_out)
{
    const PlaneArgs &a = ga.a;
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int L = TILE + W - 1;
    constexpr int LL = RW_LS * RW_LS;
    extern __shared__ double lds[];
    double2 *tiles = reinterpret_cast<double2 *>(lds);

    const uint32_t item = blockIdx.x;
    if (item >= a.nwork) return;
    const bool stamp = ga.dbg != nullptr;
    const unsigned long long ts0 = stamp ? __builtin_readcyclecounter() : 0ull;
    const WorkItem wi = a.work[item];
    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    const int NT = int(blockDim.x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane & 15, g = lane >> 4;
    const uint32_t stride = uint32_t(NT / 64) * 4;
    const uint32_t jlast = wi.end - 1;
    uint32_t j = wi.begin + wave * 4 + g;
    bool valid = j < wi.end;
    auto load_z = [&](uint32_t jj) { return *reinterpret_cast<const double2 *>(rec + min(jj, jlast)); };
    auto load_key = [&](uint32_t jj) { return rec[min(jj, jlast)].key; };
    double2 z = load_z(j);
    int key = load_key(j);
    double2 cwv[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) cwv[k] = wa.cw[size_t(min(j, jlast)) * NJ + k];
    {
        for (int i0 = 0; i0 < LL; i0 += 4 * NT) {  // four loads per thread in flight per round
            double2 v[4];
            int idx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + int(threadIdx.x) + q * NT;
                idx[q] = i;
                const int ic = min(i, LL - 1);
                const int la = ic / RW_LS, lb = ic - la * RW_LS;
                int gu = bu + la, gv = bv + lb;
                gu = gu >= a.nu ? gu % a.nu : gu;
                gv = gv >= a.nv ? gv % a.nv : gv;
                const bool in = i < LL && la < L && lb < L;
                const double2 t = grid[size_t(gu) * size_t(a.apitch) + size_t(gv)];
                v[q] = in ? t : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (idx[q] < LL) tiles[idx[q]] = v[q];
        }
    }
    // per-lane coefficient registers of tap b: phi (degree D), phi'' (D - 2), phi'''' (D - 4), phi^(6) (D - 6), unscaled
    double c0[D + 1], c1[D - 1], c2[NJ > 2 ? D - 3 : 1], c3[NJ > 3 ? D - 5 : 1];
#pragma unroll
    for (int q = 0; q <= D; ++q) c0[q] = b < W ? wa.dtab[(size_t(0) * W + b) * (D + 1) + q] : 0.0;
#pragma unroll
    for (int q = 0; q <= D - 2; ++q) c1[q] = b < W ? wa.dtab[(size_t(1) * W + b) * (D + 1) + q] : 0.0;
    if constexpr (NJ > 2) {
#pragma unroll
        for (int q = 0; q <= D - 4; ++q) c2[q] = b < W ? wa.dtab[(size_t(2) * W + b) * (D + 1) + q] : 0.0;
    }
    if constexpr (NJ > 3) {
#pragma unroll
        for (int q = 0; q <= D - 6; ++q) c3[q] = b < W ? wa.dtab[(size_t(3) * W + b) * (D + 1) + q] : 0.0;
    }
    auto evens = [](const auto &c, auto deg, double zz) {  // Horner in z^2 on the even / odd coefficients, joined at the end
        constexpr int DG = decltype(deg)::value;
        const double z2 = zz * zz;
        double e = c[DG], o = c[DG - 1];
#pragma unroll
        for (int k = DG - 2; k >= 0; k -= 2) {
            e = fma(e, z2, c[k]);
            if (k >= 1) o = fma(o, z2, c[k - 1]);
        }
        return fma(o, zz, e);
    };
    auto kernel_values = [&](double zz, const double *scale, double (&out)[NJ]) {
        out[0] = evens(c0, std::integral_constant<int, D>{}, zz);
        out[1] = evens(c1, std::integral_constant<int, D - 2>{}, zz) * scale[1];
        if constexpr (NJ > 2) out[2] = evens(c2, std::integral_constant<int, D - 4>{}, zz) * scale[2];
        if constexpr (NJ > 3) out[3] = evens(c3, std::integral_constant<int, D - 6>{}, zz) * scale[3];
    };
    const int bsel = (b & 7) < NJ ? (b & 7) : 0;
    __syncthreads();
    const unsigned long long ts1 = stamp ? __builtin_readcyclecounter() : 0ull;

    const char *tbase = reinterpret_cast<const char *>(tiles);
    for (uint32_t jb = wi.begin + wave * 4; jb < wi.end; jb += stride) {
        const uint32_t jn = j + stride;
        const bool nvalid = jn < wi.end;
        const double2 nz = load_z(jn);
        const int nkey = load_key(jn);
        double2 ncw[NJ];
#pragma unroll
        for (int k = 0; k < NJ; ++k) ncw[k] = wa.cw[size_t(min(jn, jlast)) * NJ + k];
        {
            double ku[NJ], kvb[NJ];
            kernel_values(z.x, wa.su, ku);
            kernel_values(z.y, wa.sv, kvb);
            const int lu = key >> 8, lv = key & 255;
            const int cb = (b - lv) & 15;
            double B[NJ];
#pragma unroll
            for (int r = 0; r < NJ; ++r) B[r] = __shfl(kvb[r], (lane & ~15) + cb);
            const char *base = tbase + (lu * RW_LS + lv + cb) * 16;
            double tr[NJ], ti[NJ];
#pragma unroll
            for (int r = 0; r < NJ; ++r) tr[r] = ti[r] = 0.0;
#pragma unroll
            for (int r = 0; r < NJ; ++r) asm volatile("s_nop 1" : "+v"(ku[r]));  // VALU write -> DPP read needs 2 wait states
            wd_steps<W, NJ, 0>(base, ku, tr, ti);
            // Q_r = conj(S_r), S_r = sum_m binom(r + m, r) C_{r+m} b_m ; value += T_r Q_r
            double vr = 0.0, vi = 0.0;
#pragma unroll
            for (int r = 0; r < NJ; ++r) {
                double qr = cwv[r].x * B[0], qi = cwv[r].y * B[0];
#pragma unroll
                for (int m = 1; r + m < NJ; ++m) {
                    const double bm = double(wd_binom(r + m, r)) * B[m];
                    qr = fma(cwv[r + m].x, bm, qr);
                    qi = fma(cwv[r + m].y, bm, qi);
                }
                // (tr + i ti) (qr - i qi)
                vr = fma(tr[r], qr, vr);
                vr = fma(ti[r], qi, vr);
                vi = fma(ti[r], qr, vi);
                vi = fma(-tr[r], qi, vi);
            }
            const bool lo = b < 8;
            const double keep = lo ? vr : vi, give = lo ? vi : vr;
            const double tot = half_row_sum(keep + rotn_f64<8>(give));  // lanes 0..7: Re, lanes 8..15: Im
            const double oth = rotn_f64<8>(tot);                        // the other component
            if (valid) {
                if (pval_out != nullptr) {
                    if ((b & 7) < NJ) {
                        const double wj = swgt[j];
                        const double re = (lo ? tot : oth) * wj, im = (lo ? oth : tot) * wj;
                        double cr = cwv[0].x, ci = cwv[0].y;
#pragma unroll
                        for (int k = 1; k < NJ; ++k) {
                            cr = bsel == k ? cwv[k].x : cr;
                            ci = bsel == k ? cwv[k].y : ci;
                        }
                        double *o = reinterpret_cast<double *>(pval_out + size_t(j) * size_t(NJ) + size_t(bsel));
                        // ((v * swgt) * C_k): what k_plane_values_wd would form from the weighted model visibility
                        o[lo ? 0 : 1] = lo ? (re * cr - im * ci) : (re * ci + im * cr);
                    }
                } else if ((b & 7) == 0) {
                    double *o = reinterpret_cast<double *>(sacc + j);
                    o[lo ? 0 : 1] = tot;
                }
            }
        }
        j = jn;
        valid = nvalid;
        z = nz;
        key = nkey;
#pragma unroll
        for (int k = 0; k < NJ; ++k) cwv[k] = ncw[k];
    }
    if (stamp) {  // PFBHIP_STAMP=2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts2 = __builtin_readcyclecounter();
        unsigned long long *d = ga.dbg + size_t(item) * 8;
        if (threadIdx.x == 0) {
            d[0] = ts1 - ts0;
            d[1] = ts2 - ts1;
            d[2] = 0;
            d[3] = 0;
            d[4] = wi.end - wi.begin;
            d[7] = wi.tile;
        }
        if (threadIdx.x == uint32_t(NT) - 64) {
            d[5] = ts2 - ts1;
            d[6] = ts2 - ts0;
        }
    }
}

}  // namespace pfbhip
