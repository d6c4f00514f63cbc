// gridder_kernels_mp.hpp -- the scatter / gather kernels (multi-plane).
//
// Mapping: one workgroup per work item, the (TILE+W-1)^2 footprint of the tile in LDS.  A wavefront
// handles FOUR visibilities at a time, one per 16-lane DPP row.  Lane b of a row owns footprint column
// b: it evaluates the v-kernel of tap b and the u-kernel of tap b (two Horner chains on per-lane
// register coefficients -- no exp/sqrt), then walks the W x W footprint along wrapped diagonals: at
// step i it holds the u-kernel value of row (b+i) mod 16, obtained by rotating the row's u-values one
// lane per step with a DPP row_ror (a VALU move, no LDS traffic).  At every step the 16 lanes of a row
// touch 16 different rows AND columns; with an even LDS row stride that is bank-conflict free.
//   scatter (k_grid_mp):   LDS f64 atomics (ds_add_f64) into two planes (re, im) per w-plane, then the
//                          tile is flushed to HBM with global f64 atomics (halo cells are shared by tiles);
//   gather  (k_degrid_mp): the tile is loaded into LDS as interleaved complex (ds_read_b128 per tap),
//                          per-lane partial sums, 4-step DPP row reduction.
//
// A visibility's W x W footprint (u- and v-kernel values, LDS addresses) is the same on every
// w-plane it touches; only a scalar plane weight differs.  These kernels keep the tiles of up to
// KP = 4 consecutive planes in LDS at once (one 1024-thread workgroup per CU, ~150 KiB of the
// 160 KiB), evaluate the footprint once per visibility and scatter / gather it on all KP
// planes.  The per-visibility VALU work (record loads, two Horner chains, DPP rotations,
// addressing) is shared by the planes; what remains per plane is the LDS traffic itself
// (ds_add_f64 scatter, ds_read_b128 gather), which is the floor of this design.
//
// Plane weights: lane b (< kp) of every 16-lane row evaluates the weight of plane plane0 + b of
// its row's visibility (Lagrange basis polynomial at the visibility's abscissa, or the W-wide ES
// kernel in w); the KP values are then broadcast inside the row.
#pragma once
#include <type_traits>

#include "gridder_kernels.hpp"

namespace pfbhip {

constexpr int KP_MAX = 4;
constexpr int MP_THREADS = 1024;

struct GroupArgs {
    PlaneArgs a;             // a.plane = first plane of the group
    int kp;                  // planes in this group (1..KP_MAX)
    int kp_alloc;            // planes the LDS allocation holds (the plan's planes per pass)
    double coefk[KP_MAX];    // wmode 1: Lagrange denominators of the group's planes
    size_t plane_stride;     // complex elements between consecutive planes of the uv-grid buffer
    // k_grid_rec: relative share of a work item's visibilities given to the waves of age class 0 (the first NW/3 waves, the
    // oldest on their SIMDs), 1 and 2.  The SIMD arbitrates VALU issue by age, so equal shares finish 37K / 50K / 62K
    // cycles apart (stamps, C2) and the barrier waits for the youngest.
    float wshare[3];
    unsigned long long *dbg; // diagnostic build / PFBHIP_STAMP=1 only: 8 words per work item of in-kernel phase stamps (else NULL)
};

// weight of plane `plane` for abscissa/coordinate pw
template <int W, int D>
__device__ __forceinline__ double plane_weight_of(const PlaneArgs &a, int plane, double coef, double pw, const double *wtab)
{
    if (!a.do_w) return 1.0;
    if (a.wmode == 0) {
        const double shift = 1.0 - 0.5 * double(W);
        const double fl = floor(pw + shift);
        const int dp = plane - (int)fl;
        if (dp < 0 || dp >= W) return 0.0;
        const double z = 2.0 * ((pw + shift) - fl) - 1.0;
        const double *c = wtab + dp * (D + 1);
        double v = c[D];
#pragma unroll
        for (int k = D - 1; k >= 0; --k) v = fma(v, z, c[k]);
        return v;
    }
    double kw = coef;
    for (int m = 0; m < a.nplanes; ++m)
        if (m != plane) kw *= (pw - a.nodes[m]);
    return kw;
}

// value held by lane `k` of this lane's 16-lane row
__device__ __forceinline__ double row_bcast_f64(double v, int k)
{
    const int src = ((threadIdx.x & 63) & ~15) + k;
    return __shfl(v, src);
}

// KP (planes of this launch) is a template parameter: with a run-time count every plane's LDS access
// sits in its own basic block behind a branch, which serialises the gather (each ds_read_b128 was
// followed by s_waitcnt lgkmcnt(0)) and costs the scatter a branch per atomic pair.
template <int W, int KP>
__global__ void __launch_bounds__(MP_THREADS) k_grid_mp(GroupArgs ga, const double2 *__restrict__ sval,
                                                         double2 *__restrict__ grid)
{
    const PlaneArgs &a = ga.a;
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int L = TILE + W - 1;
    constexpr int LS = tile_stride(W);
    constexpr int LL = tile_rows(W) * LS;
    extern __shared__ double lds[];
    double *wtab = lds + 2 * ga.kp_alloc * LL;
    constexpr int kp = KP;

    uint32_t item = blockIdx.x;
    if (item >= a.nwork) return;
    const WorkItem wi = a.work[item];
    for (int i = threadIdx.x; i < 2 * kp * LL; i += MP_THREADS) lds[i] = 0.0;
    for (int i = threadIdx.x; i < W * (D + 1); i += MP_THREADS) wtab[i] = a.ktab[i];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane & 15, g = lane >> 4;
    double c[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) c[k] = b < W ? a.ktab[b * (D + 1) + k] : 0.0;
    const double mycoef = b < KP_MAX ? ga.coefk[b] : 0.0;
    __syncthreads();

    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    const double shift = 1.0 - 0.5 * double(W);

    const uint32_t stride = (MP_THREADS / 64) * 4;
    uint32_t j = wi.begin + wave * 4 + g;
    bool valid = j < wi.end;
    double pu = valid ? a.pu[j] : 0.0, pv = valid ? a.pv[j] : 0.0, pw = (valid && a.do_w) ? a.pw[j] : 0.0;
    double2 val = valid ? sval[j] : make_double2(0.0, 0.0);
    for (uint32_t jb = wi.begin + wave * 4; jb < wi.end; jb += stride) {
        const uint32_t jn = j + stride;
        const bool nvalid = jn < wi.end;
        const double npu = nvalid ? a.pu[jn] : 0.0, npv = nvalid ? a.pv[jn] : 0.0;
        const double npw = (nvalid && a.do_w) ? a.pw[jn] : 0.0;
        const double2 nval = nvalid ? sval[jn] : make_double2(0.0, 0.0);
        {
            double kwl = (b < kp && valid) ? plane_weight_of<W, D>(a, a.plane + b, mycoef, pw, wtab) : 0.0;
            const double fu = floor(pu + shift), fv = floor(pv + shift);
            const double zu = 2.0 * ((pu + shift) - fu) - 1.0, zv = 2.0 * ((pv + shift) - fv) - 1.0;
            double ku = horner<D>(c, zu);
            const double kv = horner<D>(c, zv);
            const int lu = wrap_once((int)fu, a.nu) - bu, lv = wrap_once((int)fv, a.nv) - bv;
            double vr[KP], vi[KP];
            bool touch = false;  // uniform over the 16-lane row
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const double kw = row_bcast_f64(kwl, k);
                touch = touch || (kw != 0.0);
                vr[k] = val.x * (kw * kv);
                vi[k] = val.y * (kw * kv);
            }
            const int colbase = lu * LS + lv + b;
            int arow = b;
            int rowoff = b * LS;  // arow * LS, rotated alongside (one DPP move instead of a multiply-add per step)
            if (touch) {  // rows whose visibility touches no plane of the group (or is past the end) sit out
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int off = colbase + rowoff;
                    if (W >= 14 || (arow < W && b < W)) {
#pragma unroll
                        for (int k = 0; k < KP; ++k) {
                            unsafeAtomicAdd(&lds[(2 * k) * LL + off], vr[k] * ku);
                            unsafeAtomicAdd(&lds[(2 * k + 1) * LL + off], vi[k] * ku);
                        }
                    }
                    ku = rot1_f64(ku);
                    rowoff = rot1_i32(rowoff);
                    if (W < 14) arow = rot1_i32(arow);
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
    for (int k = 0; k < kp; ++k) {
        double *gp = reinterpret_cast<double *>(grid + size_t(k) * ga.plane_stride);
        const double *lre = lds + (2 * k) * LL, *lim = lds + (2 * k + 1) * LL;
        for (int i = threadIdx.x; i < L * L; i += MP_THREADS) {
            const int la = i / L, lb = i - la * L;
            const double re = lre[la * LS + lb], im = lim[la * LS + lb];
            if (re != 0.0 || im != 0.0) {
                int gu = bu + la, gv = bv + lb;
                gu = gu >= a.nu ? gu % a.nu : gu;
                gv = gv >= a.nv ? gv % a.nv : gv;
                const size_t o = (size_t(gu) * size_t(a.apitch) + size_t(gv)) * 2;
                unsafeAtomicAdd(&gp[o], re);
                unsafeAtomicAdd(&gp[o + 1], im);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Scatter, register-footprint form (k_grid_blk).
//
// k_grid_mp above pays 2 W^2 LDS f64 atomics per visibility and plane, and the LDS atomic pipe (~12 cycles per
// wave instruction) is its floor.  Visibilities are sorted by (tile, 4 x 4-cell block of the footprint origin), so
// consecutive visibilities mostly share a block: their footprints all lie inside the (W + 3)^2 cells anchored at
// the block origin.  Here a wavefront takes ONE visibility at a time and keeps that (W + 3)^2 footprint of every
// plane in REGISTERS (lane = (row group g, column c), NR rows per lane); a visibility costs 1 + 2 KP FMAs per held
// cell and no atomic; the footprint goes to the LDS tile (ds_add_f64) only when the block changes.  The u / v kernel
// values are evaluated by lanes 0..15 / 16..31 (per-lane coefficient registers, as above) and handed to the cells
// through a per-wave LDS scratch line with zero guards, read at offset (cell - origin offset).
// Any visibility order is correct; the sort only sets the run length (C2: 14.6 visibilities per block).
// 12 waves (3 per SIMD) while the accumulators fit 168 VGPRs (KP <= 3), 8 waves otherwise: a third wave per SIMD overlaps one
// wave's flush (LDS pipe) with the others' FMAs
__host__ __device__ constexpr int blk_threads(int KP) { return KP <= 3 ? 768 : 512; }
constexpr int BLK_CELLS = 4;                       // block edge in grid cells
constexpr int BLK_SCRATCH = 80;                    // doubles per scratch line: SU[24], SV[24], 32 dump slots (lanes 32..63); two lines per wave
__host__ __device__ constexpr int blk_rows_per_lane(int W) { return (W + BLK_CELLS - 1 + 2) / 3; }
// LDS tile of the block kernel: TILE + W - 1 rows.  A flush instruction touches 3 adjacent rows x (W + 3) columns; a row
// stride of +-11 doubles mod 32 spreads them over the 64 banks two-deep (the minimum for 60 doubles); an even stride
// puts two of the three rows on the same banks.  The wide stride is used when the tiles of KP planes still fit.
__host__ __device__ constexpr int blk_tile_rows(int W) { return TILE + W - 1; }
__host__ __device__ constexpr size_t blk_fixed_doubles(int W, int waves) { return size_t(W) * (kernel_poly_degree_c(W) + 1) + size_t(waves) * 3 * BLK_SCRATCH; }  // (k_grid_rec uses three scratch lines per wave, k_grid_blk two)
__host__ __device__ constexpr int blk_stride(int W, int KP)
{
    const int L = TILE + W - 1;
    const int wide = L <= 43 ? 43 : 53;
    const size_t bytes = (size_t(2) * KP * L * wide + blk_fixed_doubles(W, 12)) * sizeof(double);
    return bytes <= size_t(160) * 1024 ? wide : ((L & 1) ? L + 1 : L);
}

// Tile -> uv-grid of the register-footprint scatters (k_grid_blk, k_grid_rec); every thread of the workgroup calls it
// after the barrier that ends the visibility loop.
// MINT: the smallest workgroup the caller launches (sizes the per-thread cell arrays of the plain flush)
template <int W, int KP, int LSO = 0, int MINT = 512>
__device__ __forceinline__ void blk_tile_to_grid(const GroupArgs &ga, const WorkItem &wi, const double *lds, int bu, int bv,
                                                 double2 *__restrict__ grid)
{
    const PlaneArgs &a = ga.a;
    constexpr int L = TILE + W - 1;
    constexpr int LS = LSO ? LSO : blk_stride(W, KP);  // (LSO: the caller's own tile stride)
    constexpr int LL = blk_tile_rows(W) * LS;
    const int BLK_THREADS = int(blockDim.x);
    // Tile -> uv-grid.  The (TILE + W - 1)^2 regions of tiles two apart in each direction are disjoint (W - 1 < TILE), so
    // within one launch of a single COLOUR (tile-row parity, tile-column parity) nobody else touches this region: a plain
    // coalesced read-add-write of whole complex cells.  Device-scope f64 atomics execute at the memory side, one 8-byte
    // operation per transaction: flushing every tile of C2 that way takes 2.7 ms per launch, the plain form 0.5 ms.
    // Work items flagged `shared` (several chunks of one tile in the same launch, or a plan without colours) keep the atomics.
    const bool shared = wi.pad != 0;
    if (shared) {
        for (int k = 0; k < KP; ++k) {
            double *gp = reinterpret_cast<double *>(grid + size_t(k) * ga.plane_stride);
            const double *lre = lds + (2 * k) * LL, *lim = lds + (2 * k + 1) * LL;
            for (int i = threadIdx.x; i < L * L; i += BLK_THREADS) {
                const int la = i / L, lb = i - la * L;
                const double re = lre[la * LS + lb], im = lim[la * LS + lb];
                if (re != 0.0 || im != 0.0) {
                    int gu = bu + la, gv = bv + lb;
                    gu = gu >= a.nu ? gu % a.nu : gu;
                    gv = gv >= a.nv ? gv % a.nv : gv;
                    const size_t o = (size_t(gu) * size_t(a.apitch) + size_t(gv)) * 2;
                    unsafeAtomicAdd(&gp[o], re);
                    unsafeAtomicAdd(&gp[o + 1], im);
                }
            }
        }
        return;
    }
    // all loads of the thread's cells first (a load -> add -> store chain per cell would expose the HBM latency once per
    // cell: ~9 cells per thread), then the stores
    constexpr int NJ = (L * L + MINT - 1) / MINT;  // cells per thread and plane at the smallest workgroup
    size_t off[NJ];
    int lo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = int(threadIdx.x) + j * BLK_THREADS;
        const int la = i / L, lb = i - la * L;
        int gu = bu + la, gv = bv + lb;
        gu = gu >= a.nu ? gu % a.nu : gu;
        gv = gv >= a.nv ? gv % a.nv : gv;
        off[j] = size_t(gu) * size_t(a.apitch) + size_t(gv);
        lo[j] = i < L * L ? la * LS + lb : -1;
    }
    double2 v[KP][NJ];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const double2 *gk = grid + size_t(k) * ga.plane_stride;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lo[j] >= 0) v[k][j] = gk[off[j]];
    }
    // Every sum is formed before the first store (round 3).  With the add and the store of a cell together under `if (nonzero)`
    // the compiler had to wait again for "possibly outstanding" loads behind the first stores, and since loads and stores share
    // the in-order vmcnt that wait covered the stores' own round trips -- twice per flush in the ISA.
    // (and the wait for the loads is spelled out, unconditionally: left to the compiler it sits inside the first `if`, and every
    // later block has to assume it was skipped)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const double *lre = lds + (2 * k) * LL, *lim = lds + (2 * k + 1) * LL;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lo[j] >= 0) {
                const double re = lre[lo[j]], im = lim[lo[j]];
                v[k][j].x += re;
                v[k][j].y += im;
                nz |= (re != 0.0 || im != 0.0) ? 1u << (k * NJ + j) : 0u;
            }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        double2 *gk = grid + size_t(k) * ga.plane_stride;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if ((nz >> (k * NJ + j)) & 1u) gk[off[j]] = v[k][j];
    }
}

// Block edge BC and lane layout of the register frame (round 4b, all three register-footprint scatters): the frame is FP = W + BC - 1
// cells a side.  FP <= 16: 16 columns x 4 row groups (rows 4 k + g), ceil(FP / 4) <= 4 cells per lane and every lane at work -- W <= 13
// with the 4 x 4-cell blocks of the tile sort, W = 14 / 15 with 2 x 2-cell blocks when the sort key carries them (key_sub = 256).
// FP > 16 (W = 16; W = 14 / 15 on a coarse key): 20 columns x 3 row groups (rows 3 k + g), up to 7 cells per lane.
__host__ __device__ constexpr bool blk_frame16(int W, int BC) { return W + BC - 1 <= 16; }
__host__ __device__ constexpr int blk_frame_rows(int W, int BC)
{
    return blk_frame16(W, BC) ? (W + BC - 1 + 3) / 4 : (W + BC - 1 + 2) / 3;
}
__host__ __device__ constexpr int blk_edge(int W, bool fine_key) { return (W == 14 || W == 15) && fine_key ? 2 : BLK_CELLS; }

template <int W, int KP, int BC>
__global__ void __launch_bounds__(blk_threads(KP)) k_grid_blk(GroupArgs ga, const double2 *__restrict__ sval,
                                                           double2 *__restrict__ grid)
{
    const PlaneArgs &a = ga.a;
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int LS = blk_stride(W, KP);
    constexpr int LL = blk_tile_rows(W) * LS;
    constexpr int FP = W + BC - 1;  // footprint edge of a block
    constexpr bool F16 = blk_frame16(W, BC);
    constexpr int NCOL = F16 ? 16 : 20, NGRP = F16 ? 4 : 3, SH = BC == 4 ? 2 : 1;
    constexpr int NR = blk_frame_rows(W, BC);
    constexpr int G = BC - 1;       // zero guard in front of the kernel values
    static_assert(BC == 2 || BC == 4, "block edge: 2 or 4 cells");
    static_assert(FP <= NCOL && NR * NGRP >= FP && NGRP * (NR - 1) + NGRP - 1 + G < 24, "frame does not fit the lane layout / scratch line");
    const int BLK_THREADS = int(blockDim.x);  // blk_threads(planes per pass of the plan) <= blk_threads(KP)
    extern __shared__ double lds[];
    double *wtab = lds + 2 * KP * LL;  // own layout: the tiles of this launch's KP planes, then the tables
    double *scr_all = wtab + W * (D + 1);

    uint32_t item = blockIdx.x;
    if (item >= a.nwork) return;
    const WorkItem wi = a.work[item];
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // (first loads requested before the tile is cleared: their latency hides behind it)
    // this wave's contiguous share of the chunk, taken 64 visibilities at a time: lane i loads the record of
    // visibility base + i and does the per-visibility scalar work (position, plane weights) for it -- 64 at once,
    // the next batch's loads in flight -- and the wave then walks the batch with v_readlane broadcasts.
    const uint32_t n = wi.end - wi.begin;
    const uint32_t NW = uint32_t(BLK_THREADS) / 64;
    const uint32_t j0 = wi.begin + uint32_t((uint64_t(n) * uint32_t(wave)) / NW);
    const uint32_t j1 = wi.begin + uint32_t((uint64_t(n) * uint32_t(wave + 1)) / NW);
    uint32_t jl = j0 + uint32_t(lane);
    bool lvalid = jl < j1;
    double npu, npv, npw;
    double2 nval;
    {
        const uint32_t jc = lvalid ? jl : wi.begin;
        const double qu = a.pu[jc], qv = a.pv[jc], qw = a.pw[jc]  /* (always allocated; dropped below without w-gridding) */;
        const double2 qs = sval[jc];
        npu = lvalid ? qu : 0.0;
        npv = lvalid ? qv : 0.0;
        npw = (lvalid && a.do_w) ? qw : 0.0;
        nval = lvalid ? qs : make_double2(0.0, 0.0);
    }
    for (int i = threadIdx.x; i < 2 * KP * LL; i += BLK_THREADS) lds[i] = 0.0;
    for (int i = threadIdx.x; i < W * (D + 1); i += BLK_THREADS) wtab[i] = a.ktab[i];
    for (int i = threadIdx.x; i < (BLK_THREADS / 64) * 2 * BLK_SCRATCH; i += BLK_THREADS) scr_all[i] = 0.0;

    double *scr = scr_all + wave * 2 * BLK_SCRATCH;
    const int b = lane & 15;
    double c[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) c[k] = (b < W && lane < 32) ? a.ktab[b * (D + 1) + k] : 0.0;
    // cell ownership: lane = NCOL g + cc, rows NGRP k + g
    const int g = lane / NCOL, cc = lane - NCOL * g;
    const bool act = g < NGRP && cc < FP;
    // SU[t + G] (lanes 0..15), SV[t + G] at scr + 24 (lanes 16..31); lanes 32..63 write (zeros) to dump slots 48..79
    const int wslot = lane < 16 ? lane + G : (lane < 32 ? lane + 8 + G : lane + 16);
    __syncthreads();

    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    const double shift = 1.0 - 0.5 * double(W);

    double are[NR][KP], aim[NR][KP];
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int p = 0; p < KP; ++p) are[k][p] = aim[k][p] = 0.0;

    auto flush = [&](int key) {
        const int r0 = (key >> 8) * BC, c0 = (key & 255) * BC;
        if (act) {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int r = NGRP * k + g;
                if (r < FP) {
                    const int off = (r0 + r) * LS + c0 + cc;
#pragma unroll
                    for (int p = 0; p < KP; ++p) {
                        unsafeAtomicAdd(&lds[(2 * p) * LL + off], are[k][p]);
                        unsafeAtomicAdd(&lds[(2 * p + 1) * LL + off], aim[k][p]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
            for (int p = 0; p < KP; ++p) are[k][p] = aim[k][p] = 0.0;
    };

    auto bcast = [](double v, int i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), i), hi = __builtin_amdgcn_readlane(__double2hiint(v), i);
        return __hiloint2double(hi, lo);
    };
    int cur = -1;
    for (uint32_t base = j0; base < j1; base += 64) {
        const double pu = npu, pv = npv, pw = npw;
        const double2 val = nval;
        const bool valid = lvalid;
        jl += 64;
        lvalid = jl < j1;
        {
            // the next batch's requests, branch-free from a clamped index (round 3: `lvalid ? a.pu[jl] : 0` compiled to a branch
            // around each load with s_waitcnt vmcnt(0) at the join -- the "prefetch" was waited for on the spot, one memory
            // round trip per 64 visibilities.  Measured neutral, C5 grid 195 -> 196 ms: the other waves cover it)
            const uint32_t jc = lvalid ? jl : wi.begin;  // (work items are not empty: a valid index)
            const double qu = a.pu[jc], qv = a.pv[jc], qw = a.pw[jc]  /* (always allocated; dropped below without w-gridding) */;
            const double2 qs = sval[jc];
            npu = lvalid ? qu : 0.0;
            npv = lvalid ? qv : 0.0;
            npw = (lvalid && a.do_w) ? qw : 0.0;
            nval = lvalid ? qs : make_double2(0.0, 0.0);
        }
        // per lane = per visibility of the batch
        const double fu = floor(pu + shift), fv = floor(pv + shift);
        const double zuv = 2.0 * ((pu + shift) - fu) - 1.0, zvv = 2.0 * ((pv + shift) - fv) - 1.0;
        const int lu = wrap_once((int)fu, a.nu) - bu, lv = wrap_once((int)fv, a.nv) - bv;
        double vrv[KP], viv[KP];
        bool touch = false;
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const double kw = valid ? plane_weight_of<W, D>(a, a.plane + p, ga.coefk[p], pw, wtab) : 0.0;
            touch = touch || (kw != 0.0);
            vrv[p] = val.x * kw;
            viv[p] = val.y * kw;
        }
        (void)touch;  // a visibility that touches no plane of the group carries zero values: it is walked like the others
        const int keyv = (lu << 8) | lv;  // lu, lv in [0, 32)
        const int nb = int(min(uint32_t(64), j1 - base));
        // Two-stage software pipeline over the batch: stage A(i) evaluates the kernel values of visibility i and
        // writes them to scratch line i & 1; stage B(i) reads them back and accumulates.  A(i + 1) is issued between
        // the reads and the FMAs of B(i) -- different scratch lines, so the Horner chain fills the LDS latency.
        auto stage_a = [&](int i) {
            const double zu = bcast(zuv, i), zv = bcast(zvv, i);
            const double kval = horner<D>(c, lane < 16 ? zu : zv);
            scr[(i & 1) * BLK_SCRATCH + wslot] = kval;  // lanes >= 32 write zeros into their own spare slots
        };
        stage_a(0);
        for (int i = 0; i < nb; ++i) {
            const int kk = __builtin_amdgcn_readlane(keyv, i);
            const int slu = kk >> 8, slv = kk & 255;
            const int key = ((slu >> SH) << 8) | (slv >> SH);
            if (key != cur) {
                if (cur >= 0) flush(cur);
                cur = key;
            }
            const double *sc = scr + (i & 1) * BLK_SCRATCH;
            const int du = slu & (BC - 1), dv = slv & (BC - 1);
            const double kvc = sc[24 + cc - dv + G];
            const double *su = sc + (g - du + G);
            double kuv[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) kuv[k] = su[NGRP * k];
            stage_a((i + 1) & 63);  // unconditional (one basic block); past the batch end it rewrites a line nobody reads
            double vr[KP], vi[KP];
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                vr[p] = bcast(vrv[p], i);
                vi[p] = bcast(viv[p], i);
            }
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const double t = kuv[k] * kvc;
#pragma unroll
                for (int p = 0; p < KP; ++p) {
                    are[k][p] = fma(vr[p], t, are[k][p]);
                    aim[k][p] = fma(vi[p], t, aim[k][p]);
                }
            }
        }
    }
    if (cur >= 0) flush(cur);
    __syncthreads();
    blk_tile_to_grid<W, KP>(ga, wi, lds, bu, bv, grid);
}

// ---------------------------------------------------------------------------------------------------------
// Scatter, register-footprint form driven by per-visibility RECORDS (k_grid_rec): single-pass plans with polynomial
// w-planes (every visibility touches every plane of the one pass).
//
// What the in-kernel stamps of k_grid_blk / the first k_grid_rec showed (tools/stamp_scatter.py, C2): a wave issues at
// most ONE instruction -- vector, scalar, LDS, branch or wait -- per 4 cycles, so a visibility costs a wave 4 x (all its
// instructions), not 4 x (its FMAs); the three waves of a SIMD overlap, but the SIMD arbitrates by age, so with equal
// shares the oldest wave of a SIMD finishes at 60 % of the youngest wave's time and the barrier waits for the youngest.
// Hence:
//  * everything that does not depend on the visibility VALUES is computed once per plan (k_vis_records) in the form
//    the loop consumes: kernel arguments (zu, zv), the 4 x 4-cell block id, and the two scratch-read byte offsets;
//  * the values arrive already multiplied by the plane weights (KP complex numbers per visibility: written by the
//    gather's epilogue in a Hessian apply, by k_plane_values otherwise);
//  * key / offsets / values are wave-uniform and come in through SCALAR loads (SGPR operands of the FMAs: no
//    v_readlane), the kernel argument through one vector load (lanes 0..15 read zu, lanes 16..31 zv); records are
//    addressed with running 32-bit byte offsets from the wave's own first record (arrays padded by REC_PAD entries: no clamping);
//  * three register sets and three scratch lines used round robin by a loop unrolled three times (static indices):
//    a visibility's data are requested two iterations before use;
//  * the scalar requests are issued BEHIND the wait for the scratch reads (scalar loads return out of order: a wait on
//    LDS data while one is pending waits for it too);
//  * age classes of waves get unequal shares (GroupArgs::wshare).
struct VisRec {
    double zu, zv;  // 2 frac - 1 of the first-tap offset on the plan's u and v axis
    int blk;        // ((lu >> 2) << 8) | (lv >> 2): 4 x 4-cell block of the first-tap cell inside the tile
    int offu;       // 8 (G - (lu & 3)): byte offset of the u-kernel line reads
    int offv;       // 8 (24 + G - (lv & 3)): byte offset of the v-kernel read
    int key;        // (lu << 8) | lv: the first-tap cell itself (gather; SKIP test of k_grid_rec)
};
static_assert(sizeof(VisRec) == 32, "VisRec layout");
// entries allocated past the last record / value: the loop requests up to 3 ahead, and the L2 warm-up (`touch`) one entry per
// lane up to 63 + 63 past the visibility it is at.  (With 8 the warm-up of the LAST work item read ~7 KB past the arrays: harmless
// inside an allocation's slack, a memory fault when a small plan's arrays ended at a page boundary.)
constexpr int REC_PAD = 136;

template <int W, int KP, int BC>
__global__ void __launch_bounds__(blk_threads(KP)) k_grid_rec(GroupArgs ga, const VisRec *__restrict__ rec,
                                                               const double2 *__restrict__ pval, double2 *__restrict__ grid)
{
    const PlaneArgs &a = ga.a;
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int LS = blk_stride(W, KP);
    constexpr int LL = blk_tile_rows(W) * LS;
    constexpr int FP = W + BC - 1;
    constexpr bool F16 = blk_frame16(W, BC);   // (frame and lane layout: see k_grid_blk)
    constexpr int NCOL = F16 ? 16 : 20, NGRP = F16 ? 4 : 3;
    constexpr int NR = blk_frame_rows(W, BC);
    constexpr int G = BC - 1;
    constexpr bool SKIP = !F16 && (FP % 3) == 1;
    static_assert(BC == 2 || BC == 4, "block edge: 2 or 4 cells");
    static_assert(FP <= NCOL && NR * NGRP >= FP && NGRP * (NR - 1) + NGRP - 1 + G < 24, "frame does not fit the lane layout / scratch line");
    constexpr int NLINE = 3;  // scratch lines per wave (blk_fixed_doubles reserves 3 * BLK_SCRATCH per wave for this kernel)
    const int BLK_THREADS = int(blockDim.x);
    extern __shared__ double lds[];
    double *wtab = lds + 2 * KP * LL;
    double *scr_all = wtab + W * (D + 1);

    const uint32_t item = blockIdx.x;
    if (item >= a.nwork) return;
    const bool stamp = ga.dbg != nullptr;
    const unsigned long long ts0 = stamp ? __builtin_readcyclecounter() : 0ull;
    const WorkItem wi = a.work[item];
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t n = wi.end - wi.begin;
    const uint32_t NW = uint32_t(BLK_THREADS) / 64;
    // share boundaries: waves of one age class get equal shares, the classes ga.wshare[] of the total
    auto share_begin = [&](uint32_t w) {
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
    const uint32_t pbytes = uint32_t(ga.kp_alloc) * 16u;  // bytes of values per visibility
    const char *rbase = reinterpret_cast<const char *>(rec);
    const char *pbase = reinterpret_cast<const char *>(pval);
    // Kernel values are evaluated for TWO visibilities at a time (round 4b): lanes 0..15 zu / 16..31 zv of the pair's first
    // visibility, 32..47 zu / 48..63 zv of its second -- one Horner round per pair with every lane at work (one visibility at a time
    // left lanes 32..63 idle).  Per-lane kernel-argument pointer: visibility j0 + vsel, + 2 per pair; coefficients requested before
    // the tile is cleared.
    const int axis = (lane >> 4) & 1, vsel = lane >> 5;
    const char *zptr = rbase + (size_t(j0) + size_t(vsel)) * 32 + (axis ? 8 : 0);
    const int b = lane & 15;
    double c[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) c[k] = b < W ? a.ktab[b * (D + 1) + k] : 0.0;
    // Warm the L2 with the records / values of 64 visibilities (one per lane) ahead of the walk, whose own requests --
    // two visibilities ahead -- then never wait for HBM.  The loaded words are consumed (an empty asm) at the next refresh.
    static_assert(REC_PAD >= 63 + 63 + 3, "the warm-up reads one entry per lane up to 63 + 63 past the current visibility");
    auto touch = [&](uint32_t first) {
        const uint32_t jt = first + uint32_t(lane);  // (padded arrays: no clamp)
        const int t0 = *reinterpret_cast<const int *>(rbase + size_t(jt) * 32 + 16);
        const double t1 = *reinterpret_cast<const double *>(pbase + size_t(jt) * pbytes);
        return double(t0) + t1;
    };
    double warm = touch(j0);
    double zq[3];  // z of the lane's axis for the next three PAIRS (slot = pair index mod 3)
#pragma unroll
    for (int u = 0; u < 3; ++u) zq[u] = *reinterpret_cast<const double *>(zptr + u * 64);
    zptr += 3 * 64;

    for (int i = threadIdx.x; i < 2 * KP * LL; i += BLK_THREADS) lds[i] = 0.0;
    for (int i = threadIdx.x; i < W * (D + 1); i += BLK_THREADS) wtab[i] = a.ktab[i];
    for (int i = threadIdx.x; i < (BLK_THREADS / 64) * NLINE * BLK_SCRATCH; i += BLK_THREADS) scr_all[i] = 0.0;

    char *scr = reinterpret_cast<char *>(scr_all + wave * NLINE * BLK_SCRATCH);
    const int g = lane / NCOL, cc = lane - NCOL * g;
    const bool act = g < NGRP && cc < FP;
    // where this lane writes its kernel value: slot b + G of SU (u lanes) / SV at + 24 (v lanes) of the line of ITS visibility.
    // Visibility s lives in line s mod 3; the pair (s + 1, s + 2) is evaluated while visibility s (odd within a block of six) is
    // processed, into the two lines that are not being read: wq[q] for s = 2 q + 1.
    char *const went = scr + ((axis ? 24 : 0) + b + G) * 8;
    char *wq[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) wq[q] = went + ((2 * q + 2 + vsel) % 3) * (BLK_SCRATCH * 8);
    const char *suptr = scr + g * 8;   // + offu + 8 NGRP k: u-kernel values of the lane's rows
    const char *svptr = scr + cc * 8;  // + offv: v-kernel value of the lane's column
    __syncthreads();

    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;

    double are[NR][KP], aim[NR][KP];
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int p = 0; p < KP; ++p) are[k][p] = aim[k][p] = 0.0;

    // footprint -> LDS tile: one address per flush (plane / row offsets are immediates where they fit 16 bits)
    char *const tile0 = reinterpret_cast<char *>(lds) + (g * LS + cc) * 8;
    auto flush = [&](int blk) {
        const int r0 = (blk >> 8) * BC, c0 = (blk & 255) * BC;
        char *base = tile0 + (r0 * LS + c0) * 8;
        if (act) {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                if (NGRP * k + g < FP) {
#pragma unroll
                    for (int p = 0; p < KP; ++p) {
                        unsafeAtomicAdd(reinterpret_cast<double *>(base + ((2 * p) * LL + NGRP * k * LS) * 8), are[k][p]);
                        unsafeAtomicAdd(reinterpret_cast<double *>(base + ((2 * p + 1) * LL + NGRP * k * LS) * 8), aim[k][p]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
            for (int p = 0; p < KP; ++p) are[k][p] = aim[k][p] = 0.0;
    };

    // kernel values of one visibility: Horner in two independent half chains (even / odd powers), which a single wave can
    // issue back to back (a 12-deep dependent f64 chain leaves it idle most of the time)
    auto kernel_value = [&](double z) {
        const double z2 = z * z;
        double e = c[D], o = c[D - 1];
#pragma unroll
        for (int k = D - 2; k >= 0; k -= 2) {
            e = fma(e, z2, c[k]);
            if (k >= 1) o = fma(o, z2, c[k - 1]);
        }
        return (D & 1) ? fma(o, z, e) : fma(o, z, e);
    };
    static_assert((D & 1) == 0, "kernel_value assumes an even polynomial degree");
    const unsigned long long ts1 = stamp ? __builtin_readcyclecounter() : 0ull;
    if (nmine > 0) {  // pair 0 -> lines 0, 1
        *reinterpret_cast<double *>(wq[2]) = kernel_value(zq[0]);
        zq[0] = *reinterpret_cast<const double *>(zptr);
        zptr += 64;
    }
    int cur = -1;
    int4 kq[3];
    double2 pq[3][KP];
    // Byte offsets of the next scalar requests (visibility s + 2 ...), RELATIVE to this wave's first record / value: a wave
    // walks at most CHUNK visibilities, so the 32-bit offsets stay below 2^19 whatever the plan's size.  (Absolute offsets
    // j0 * 32 and j0 * pbytes wrap at 1.3e8 / 6.7e7 active visibilities and silently address other visibilities' data.)
    const char *const rwave = rbase + size_t(j0) * 32;
    const char *const pwave = pbase + size_t(j0) * size_t(pbytes);
    uint32_t roff = 16u, poff = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        kq[u] = *reinterpret_cast<const int4 *>(rwave + roff);
#pragma unroll
        for (int p = 0; p < KP; ++p) pq[u][p] = *reinterpret_cast<const double2 *>(pwave + poff + p * 16);
        roff += 32u;
        poff += pbytes;
    }
    kq[2] = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < KP; ++p) pq[2][p] = make_double2(0.0, 0.0);
    // consume the first scalars here: with scalar loads still pending at the loop header the compiler waits for
    // lgkmcnt(0) at the top of EVERY iteration, right behind the loads it has just issued
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        asm volatile("" ::"s"(kq[u].x), "s"(kq[u].y), "s"(kq[u].z));
#pragma unroll
        for (int p = 0; p < KP; ++p) asm volatile("" ::"s"(pq[u][p].x), "s"(pq[u][p].y));
    }
    // block of a visibility: the record's 4 x 4 block, or the 2 x 2 block of its first-tap cell (key = (lu << 8) | lv); scratch byte
    // offsets: the record's (4 x 4 anchoring), or from the cell's position in its 2 x 2 block
    auto block_of = [](const int4 &r) { return BC == 4 ? r.x : ((r.w >> 1) & 0x0F0F); };
    if (nmine > 0) cur = block_of(kq[0]);
    for (uint32_t wb = 0; wb < nmine; wb += 60) {  // L2 window: 60 visibilities = 10 trips of the unrolled loop
        asm volatile("" ::"v"(warm));
        warm = touch(j0 + wb + 60);
        const uint32_t wend = min(wb + 60u, nmine);
    for (uint32_t sb = wb; sb < wend; sb += 6) {
#pragma unroll
        for (int i6 = 0; i6 < 6; ++i6) {
            if (sb + uint32_t(i6) >= wend) break;
            const int u = i6 % 3, ld = (u + 2) % 3;

            const int4 rk = kq[u];
            const int bnow = block_of(rk);
            if (bnow != cur) {
                flush(cur);
                cur = bnow;
            }
            const int offu = BC == 4 ? rk.y : 8 * (G - ((rk.w >> 8) & 1)), offv = BC == 4 ? rk.z : 8 * (24 + G - (rk.w & 1));
            const double kvc = *reinterpret_cast<const double *>(svptr + u * (BLK_SCRATCH * 8) + offv);
            const char *su = suptr + u * (BLK_SCRATCH * 8) + offu;
            double kuv[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) kuv[k] = *reinterpret_cast<const double *>(su + 8 * NGRP * k);
            if (i6 & 1) {  // kernel values of the pair (s + 1, s + 2) -> their lines
                const int q = i6 >> 1, slot = (q + 1) % 3;
                *reinterpret_cast<double *>(wq[q]) = kernel_value(zq[slot]);
                zq[slot] = *reinterpret_cast<const double *>(zptr);  // z of the pair three pairs on
                zptr += 64;
            }
            // The scalar requests for visibility s + 2 go out behind the wait for the scratch reads (the empty asm reads
            // their destinations and is a compiler barrier for memory operations).  Issued here they have the FMAs below
            // and the next iteration's kernel evaluation to arrive in.
#pragma unroll
            for (int k = 0; k < NR; ++k) asm volatile("" : "+v"(kuv[k])::"memory");
            double kvc_ = kvc;
            asm volatile("" : "+v"(kvc_)::"memory");
            kq[ld] = *reinterpret_cast<const int4 *>(rwave + roff);
#pragma unroll
            for (int p = 0; p < KP; ++p) pq[ld][p] = *reinterpret_cast<const double2 *>(pwave + poff + p * 16);
            roff += 32u;
            poff += pbytes;
            __builtin_amdgcn_sched_barrier(0);
            // SKIP shapes (the block footprint has 3 m + 1 rows; W = 16: 19): row group 0 holds rows 0..2, the last group row
            // 3 (NR - 1) only; a visibility whose origin sits in row du of its block touches rows du .. du + W - 1, i.e. group 0
            // unless du == 3 and the last group only if du == 3 (du = bits 8..9 of rk.w, wave-uniform)
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                bool on = true;
                if (SKIP && k == 0) on = (rk.w & 0x300) != 0x300;
                if (SKIP && k == NR - 1) on = (rk.w & 0x300) == 0x300;
                if (on) {
                    const double t = kuv[k] * kvc_;
#pragma unroll
                    for (int p = 0; p < KP; ++p) {
                        are[k][p] = fma(pq[u][p].x, t, are[k][p]);
                        aim[k][p] = fma(pq[u][p].y, t, aim[k][p]);
                    }
                }
            }
            // "use" the record's spare fourth word when its set is current: otherwise its register is handed out again while
            // the load that writes it is in flight, and that write waits for the load (an lgkmcnt(0) right behind the requests)
            asm volatile("" ::"s"(rk.x), "s"(rk.y), "s"(rk.z), "s"(rk.w));
        }
    }
    }
    asm volatile("" ::"v"(warm));
    if (cur >= 0) flush(cur);
    const unsigned long long ts2 = stamp ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();
    const unsigned long long ts3 = stamp ? __builtin_readcyclecounter() : 0ull;
    blk_tile_to_grid<W, KP>(ga, wi, lds, bu, bv, grid);
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts4 = __builtin_readcyclecounter();
        unsigned long long *d = ga.dbg + size_t(item) * 8;
        if (threadIdx.x == 0) {
            d[0] = ts1 - ts0;  // prologue (records requested, tile cleared, tables, barrier)
            d[1] = ts2 - ts1;  // wave 0: its share of the visibilities
            d[2] = ts3 - ts2;  // wave 0: wait for the slowest wave
            d[3] = ts4 - ts3;  // tile -> uv-grid
            d[4] = n;
            d[7] = wi.tile;
        }
        if (threadIdx.x == uint32_t(BLK_THREADS) - 64) {
            d[5] = ts2 - ts1;  // last wave's share
            d[6] = ts3 - ts2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Gather, row walk with DPP-broadcast FMAs (k_degrid_rw): single-pass plans with polynomial w-planes.
//
// k_degrid_mp walks a footprint along diagonals so that the 16 lanes of a visibility hold 16 different u-kernel values:
// per step 6 FMAs are accompanied by 3 DPP rotations and 2 address additions, and the per-visibility index arithmetic
// and plane weights are recomputed every apply -- 79 VALU instructions per visibility, 30 % of them FMAs, and the
// kernel is VALU-bound (a wave issues one instruction per 4 cycles whatever its kind).  gfx950 has a 64-bit DPP form of
// v_fmac_f64 whose first operand is BROADCAST from a fixed lane of each 16-lane row (row_newbcast): with lane b on
// footprint column cb and all 16 lanes on the same footprint row i, step i is
//     acc += ku[lane i of my row] * cell(row i, my column)        (one instruction, no rotation)
// and the cell address is base + i * (row pitch): an immediate.  The static per-visibility data come from the records
// of k_grid_rec (kernel arguments, first-tap cell) and a plan-time table of plane weights.
// The tile is always 48 x 48 cells here (rows / columns past TILE + W - 1 hold zeros: narrower kernels have zero
// coefficients on lanes >= W and read, but do not use, those cells).
constexpr int RW_LS = TILE + 16;  // cells per tile row and tile rows

template <int I>
__device__ __forceinline__ void fmac_row_bcast(double &acc, double ku, double cell)
{
    // acc += (ku of lane I of this lane's 16-lane row) * cell
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(ku), "v"(cell), "n"(I));
}

// planes 0, 1 are addressed from base, planes 2, 3 from base2 = base + 2 tiles: every offset is a 16-bit immediate
template <int KP, int I>
__device__ __forceinline__ void rw_load_row(const char *base, const char *base2, double2 (&cell)[KP])
{
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const char *bp = k < 2 ? base : base2;
        cell[k] = *reinterpret_cast<const double2 *>(bp + (size_t(k & 1) * RW_LS * RW_LS + size_t(I) * RW_LS) * 16);
    }
}

// Prefetch distance of the row walk, in footprint rows (experiment, round 3).  Left to itself the compiler keeps TWO
// ds_read_b128 in flight (issue 2, wait for the first, 2 FMAs ...).  With D > 0 the reads of row I + D are issued before the
// FMAs of row I, D * KP reads in flight (at most 15: lgkmcnt is a 4-bit counter), in D + 1 static register sets.
// MEASURED on C2 (W = 16, KP = 3): D = 3 is 26 % SLOWER (2.24 vs 1.77 ms) -- see DESIGN.md section 5.2; D = 0 (the compiler's
// schedule) stays the default, PFBHIP_RW_DEPTH selects 1..3 for the (16, 3) instantiation.
template <int KP, int D, int I>
__device__ __forceinline__ void rw_steps(const char *base, const char *base2, double ku, double (&sr)[KP], double (&si)[KP],
                                         double2 (&buf)[D + 1][KP])
{
    if constexpr (I < 16) {
        if constexpr (I + D < 16) rw_load_row<KP, I + D>(base, base2, buf[(I + D) % (D + 1)]);
        __builtin_amdgcn_sched_barrier(0);  // keeps the requests above in front of the FMAs below
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            fmac_row_bcast<I>(sr[k], ku, buf[I % (D + 1)][k].x);  // (v_mul_f64 has no DPP form: the accumulators start from zero)
            fmac_row_bcast<I>(si[k], ku, buf[I % (D + 1)][k].y);
        }
        __builtin_amdgcn_sched_barrier(0);
        rw_steps<KP, D, I + 1>(base, base2, ku, sr, si, buf);
    }
}
// (NS = W steps: the rows past the support carry zero kernel values)
template <int NS, int KP, int I>
__device__ __forceinline__ void rw_steps0(const char *base, const char *base2, double ku, double (&sr)[KP], double (&si)[KP])
{
    if constexpr (I < NS) {
        double2 cell[KP];
        rw_load_row<KP, I>(base, base2, cell);
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            fmac_row_bcast<I>(sr[k], ku, cell[k].x);
            fmac_row_bcast<I>(si[k], ku, cell[k].y);
        }
        rw_steps0<NS, KP, I + 1>(base, base2, ku, sr, si);
    }
}
template <int KP, int D, int I>
__device__ __forceinline__ void rw_prologue(const char *base, const char *base2, double2 (&buf)[D + 1][KP])
{
    if constexpr (I < D) {
        rw_load_row<KP, I>(base, base2, buf[I]);
        rw_prologue<KP, D, I + 1>(base, base2, buf);
    }
}

// sum over the 8 lanes of this lane's half row (lanes 0..7 / 8..15 of a 16-lane row), left in every lane of the half
__device__ __forceinline__ double half_row_sum(double v)
{
    auto dpp = [](double x, auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), C, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), C, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    };
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror: lane b <-> 7 - b
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    return v;
}

template <int W, int KP, int PD = 0>
__global__ void __launch_bounds__(MP_THREADS) k_degrid_rw(GroupArgs ga, const VisRec *__restrict__ rec,
                                                           const double *__restrict__ kwtab, const double2 *__restrict__ grid,
                                                           double2 *__restrict__ sacc, const double *__restrict__ swgt,
                                                           double2 *__restrict__ pval_out)
{
    const PlaneArgs &a = ga.a;
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int L = TILE + W - 1;
    constexpr int LL = RW_LS * RW_LS;
    extern __shared__ double lds[];
    double2 *tiles = reinterpret_cast<double2 *>(lds);  // KP tiles of LL complex cells

    const uint32_t item = blockIdx.x;
    if (item >= a.nwork) return;
    const bool stamp = ga.dbg != nullptr;
    const unsigned long long ts0 = stamp ? __builtin_readcyclecounter() : 0ull;
    const WorkItem wi = a.work[item];
    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane & 15, g = lane >> 4;
    const int kpa = ga.kp_alloc;
    const uint32_t stride = (MP_THREADS / 64) * 4;
    const uint32_t jlast = wi.end - 1;
    // first records requested before the tile: their latency hides behind the tile loads
    uint32_t j = wi.begin + wave * 4 + g;
    bool valid = j < wi.end;
    auto load_z = [&](uint32_t jj) { return *reinterpret_cast<const double2 *>(rec + min(jj, jlast)); };
    auto load_key = [&](uint32_t jj) { return rec[min(jj, jlast)].key; };
    double2 z = load_z(j);
    int key = load_key(j);
    double kw[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) kw[k] = kwtab[size_t(min(j, jlast)) * size_t(kpa) + k];
    const int bsel = (b & 7) < KP ? (b & 7) : 0;  // the plane whose value this lane writes (pval_out)
    {
        // every load of the thread's cells (all planes) in flight before the first LDS store
        constexpr int NJ = (LL + MP_THREADS - 1) / MP_THREADS;
        size_t off[NJ];
        bool in[NJ];
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const int i = int(threadIdx.x) + q * MP_THREADS;
            const int la = i / RW_LS, lb = i - la * RW_LS;
            int gu = bu + la, gv = bv + lb;
            gu = gu >= a.nu ? gu % a.nu : gu;
            gv = gv >= a.nv ? gv % a.nv : gv;
            in[q] = i < LL && la < L && lb < L;
            off[q] = size_t(gu) * size_t(a.apitch) + size_t(gv);
        }
        double2 v[KP][NJ];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const double2 *gk = grid + size_t(k) * ga.plane_stride;
#pragma unroll
            for (int q = 0; q < NJ; ++q) v[k][q] = in[q] ? gk[off[q]] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < KP; ++k)
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const int i = int(threadIdx.x) + q * MP_THREADS;
                if (i < LL) tiles[k * LL + i] = v[k][q];
            }
    }
    double c[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) c[k] = b < W ? a.ktab[b * (D + 1) + k] : 0.0;
    auto kernel_value = [&](double zz) {  // two independent half chains (even / odd powers)
        const double z2 = zz * zz;
        double e = c[D], o = c[D - 1];
#pragma unroll
        for (int k = D - 2; k >= 0; k -= 2) {
            e = fma(e, z2, c[k]);
            if (k >= 1) o = fma(o, z2, c[k - 1]);
        }
        return fma(o, zz, e);
    };
    static_assert((D & 1) == 0, "kernel_value assumes an even polynomial degree");
    __syncthreads();
    const unsigned long long ts1 = stamp ? __builtin_readcyclecounter() : 0ull;

    const char *tbase = reinterpret_cast<const char *>(tiles);
    for (uint32_t jb = wi.begin + wave * 4; jb < wi.end; jb += stride) {
        const uint32_t jn = j + stride;
        const bool nvalid = jn < wi.end;
        const double2 nz = load_z(jn);
        const int nkey = load_key(jn);
        double nkw[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) nkw[k] = kwtab[size_t(min(jn, jlast)) * size_t(kpa) + k];
        {
            double ku = kernel_value(z.x);
            const double kvb = kernel_value(z.y);
            const int lu = key >> 8, lv = key & 255;
            // lane b takes footprint column cb = (b - lv) mod 16: its LDS column lv + cb is congruent to b mod 16 whatever
            // the visibility, and with a row pitch of 0 mod 16 cells the 16-byte slot of every lane's read is b mod 16 -- the
            // lane groups of a ds_read_b128 mix two visibilities, and their slots stay distinct (no bank conflicts)
            const int cb = (b - lv) & 15;
            const double kv = __shfl(kvb, (lane & ~15) + cb);
            const char *base = tbase + (lu * RW_LS + lv + cb) * 16;
            const char *base2 = base + 2 * LL * 16;
            double sr[KP], si[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) sr[k] = si[k] = 0.0;
            asm volatile("s_nop 1" : "+v"(ku));  // VALU write -> DPP read of the same register needs 2 wait states
            if constexpr (PD == 0) {
                rw_steps0<W, KP, 0>(base, base2, ku, sr, si);
            } else {
                double2 cells[PD + 1][KP];
                rw_prologue<KP, PD, 0>(base, base2, cells);
                rw_steps<KP, PD, 0>(base, base2, ku, sr, si, cells);
            }
            double tr = 0.0, ti = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                tr = fma(sr[k], kw[k], tr);
                ti = fma(si[k], kw[k], ti);
            }
            tr *= kv;
            ti *= kv;
            // sum over the 16 columns: the halves of the row exchange one component, then lanes 0..7 reduce the real part
            // and lanes 8..15 the imaginary part
            const bool lo = b < 8;
            const double keep = lo ? tr : ti, give = lo ? ti : tr;
            const double tot = half_row_sum(keep + rotn_f64<8>(give));  // lanes 0..7: Re, lanes 8..15: Im
            if (valid) {
                if (pval_out != nullptr) {
                    if ((b & 7) < KP) {
                        const double wj = swgt[j];
                        double *o = reinterpret_cast<double *>(pval_out + size_t(j) * size_t(kpa) + size_t(bsel));
                        // ((v * swgt) * kw): the products k_scale_sorted and the scatter would form
                        double kwl = kw[0];
#pragma unroll
                        for (int k = 1; k < KP; ++k) kwl = bsel == k ? kw[k] : kwl;
                        o[lo ? 0 : 1] = (tot * wj) * kwl;
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
        for (int k = 0; k < KP; ++k) kw[k] = nkw[k];
    }
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts2 = __builtin_readcyclecounter();
        unsigned long long *d = ga.dbg + size_t(item) * 8;
        if (threadIdx.x == 0) {
            d[0] = ts1 - ts0;  // tile load
            d[1] = ts2 - ts1;  // wave 0: its rounds
            d[2] = 0;
            d[3] = 0;
            d[4] = wi.end - wi.begin;
            d[7] = wi.tile;
        }
        if (threadIdx.x == MP_THREADS - 64) {
            d[5] = ts2 - ts1;  // last wave's rounds
            d[6] = ts2 - ts0;
        }
    }
}

// plan time: plane weights of every visibility (wmode 1 / no w-gridding, single pass), kp_alloc per visibility
template <int W>
__global__ void k_plane_weights(GroupArgs ga, int64_t nactive, double *__restrict__ kwtab)
{
    constexpr int D = kernel_poly_degree_c(W);
    const int64_t j = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (j >= nactive + REC_PAD) return;
    const PlaneArgs &a = ga.a;
    const double pw = (a.do_w && j < nactive) ? a.pw[j] : 0.0;
    for (int p = 0; p < ga.kp_alloc; ++p)
        kwtab[size_t(j) * size_t(ga.kp_alloc) + size_t(p)] =
            (p < ga.kp && j < nactive) ? plane_weight_of<W, D>(a, a.plane + p, ga.coefk[p], pw, nullptr) : 0.0;
}

// plan time: the static half of the records (same expressions as the walk kernels evaluate per visibility)
template <int W>
__global__ void k_vis_records(int nu, int nv, int64_t nactive, const double *__restrict__ pu, const double *__restrict__ pv,
                              VisRec *__restrict__ rec)
{
    const int64_t j = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (j >= nactive + REC_PAD) return;
    VisRec r;
    if (j >= nactive) {  // padding: read, never used
        r.zu = r.zv = 0.0;
        r.blk = r.key = 0;
        r.offu = 8 * (BLK_CELLS - 1);
        r.offv = 8 * (24 + BLK_CELLS - 1);
        rec[j] = r;
        return;
    }
    const double shift = 1.0 - 0.5 * double(W);
    const double u = pu[j], v = pv[j];
    const double fu = floor(u + shift), fv = floor(v + shift);
    r.zu = 2.0 * ((u + shift) - fu) - 1.0;
    r.zv = 2.0 * ((v + shift) - fv) - 1.0;
    const int lu = wrap_once((int)fu, nu) % TILE, lv = wrap_once((int)fv, nv) % TILE;
    r.blk = ((lu >> 2) << 8) | (lv >> 2);
    r.offu = 8 * (BLK_CELLS - 1 - (lu & 3));
    r.offv = 8 * (24 + BLK_CELLS - 1 - (lv & 3));
    r.key = (lu << 8) | lv;
    rec[j] = r;
}

// per apply (outside Hessian applies): pval[j][p] = sval[j] * weight of plane p at the visibility's w
template <int W>
__global__ void k_plane_values(GroupArgs ga, int64_t nactive, const double2 *__restrict__ sval, double2 *__restrict__ pval)
{
    constexpr int D = kernel_poly_degree_c(W);
    const int64_t j = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (j >= nactive) return;
    const PlaneArgs &a = ga.a;
    const double2 val = sval[j];
    const double pw = a.do_w ? a.pw[j] : 0.0;
    for (int p = 0; p < ga.kp; ++p) {
        const double kw = plane_weight_of<W, D>(a, a.plane + p, ga.coefk[p], pw, nullptr);
        pval[size_t(j) * size_t(ga.kp_alloc) + size_t(p)] = make_double2(val.x * kw, val.y * kw);
    }
}

// The same for one pass of an ES-kernel plane stack (round 3: the record scatter serves multi-pass plans too), driven by the
// pass's WORK LIST (one workgroup per work item, all colours): with the plane-sorted order a pass's items hold exactly the
// visibilities whose W planes meet its planes [a.plane, a.plane + kp), so a pass costs what it scatters, not nactive; plans with
// too few planes for that order walk every visibility in every pass, zeros included.
template <int W>
__global__ void __launch_bounds__(256) k_plane_values_es(GroupArgs ga, const double2 *__restrict__ sval, double2 *__restrict__ pval)
{
    constexpr int D = kernel_poly_degree_c(W);
    const PlaneArgs &a = ga.a;
    if (blockIdx.x >= a.nwork) return;
    const WorkItem wi = a.work[blockIdx.x];
    const double shift = 1.0 - 0.5 * double(W);
    for (uint32_t j = wi.begin + threadIdx.x; j < wi.end; j += blockDim.x) {
        const double pws = a.pw[j] + shift;
        const double fl = floor(pws);
        const int dp0 = a.plane - int(fl);  // the pass's first plane relative to the visibility's first plane
        const double z = 2.0 * (pws - fl) - 1.0;
        const double2 val = sval[j];
        for (int p = 0; p < ga.kp; ++p) {
            const int dp = dp0 + p;
            double kw = 0.0;
            if (dp >= 0 && dp < W) {
                const double *c = a.ktab + dp * (D + 1);
                kw = c[D];
#pragma unroll
                for (int k = D - 1; k >= 0; --k) kw = fma(kw, z, c[k]);
            }
            pval[size_t(j) * size_t(ga.kp_alloc) + size_t(p)] = make_double2(val.x * kw, val.y * kw);
        }
    }
}

// pval_out != NULL (single-pass plans inside a Hessian apply): instead of accumulating the model visibility into sacc,
// lane b < KP of the visibility's row writes it multiplied by the imaging weight and by the weight of plane b --
// the input of k_grid_rec ((v * swgt) * kw, the products k_scale_sorted and the scatter would form).
template <int W, int KP>
__global__ void __launch_bounds__(MP_THREADS) k_degrid_mp(GroupArgs ga, const double2 *__restrict__ grid,
                                                           double2 *__restrict__ sacc, const double *__restrict__ swgt,
                                                           double2 *__restrict__ pval_out)
{
    const PlaneArgs &a = ga.a;
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int L = TILE + W - 1;
    constexpr int LS = tile_stride(W);
    constexpr int LL = tile_rows(W) * LS;
    extern __shared__ double lds[];
    double2 *tiles = reinterpret_cast<double2 *>(lds);  // kp_alloc tiles of LL complex
    double *wtab = lds + 2 * ga.kp_alloc * LL;
    constexpr int kp = KP;

    uint32_t item = blockIdx.x;
    if (item >= a.nwork) return;
    const WorkItem wi = a.work[item];
    const int bu = int(wi.tile / uint32_t(a.ntv)) * TILE;
    const int bv = int(wi.tile % uint32_t(a.ntv)) * TILE;
    // the first records are requested before the tile: their latency hides behind the tile loads
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane & 15, g = lane >> 4;
    const uint32_t stride = (MP_THREADS / 64) * 4;
    uint32_t j = wi.begin + wave * 4 + g;
    bool valid = j < wi.end;
    // (requests branch-free from a clamped index, round 3: `valid ? a.pu[j] : 0` is a branch around each load with a
    // s_waitcnt vmcnt(0) at the join -- the next round's "prefetch" below was waited for on the spot; the running sum of the
    // multi-pass accumulation travels with it instead of being read, waited for and written at the end of every round.
    // Measured neutral: C5 degrid 107 -> 108 ms)
    const bool rmw = pval_out == nullptr;
    const double2 *accp = rmw ? sacc : grid;  // (always a valid address: the plane-value form does not read the running sum)
    double pu, pv, pw;
    double2 acc0;
    {
        const uint32_t jc = valid ? j : wi.begin;
        const double qu = a.pu[jc], qv = a.pv[jc], qw = a.pw[jc]  /* (always allocated; dropped below without w-gridding) */;
        acc0 = accp[rmw ? jc : 0u];
        pu = valid ? qu : 0.0;
        pv = valid ? qv : 0.0;
        pw = (valid && a.do_w) ? qw : 0.0;
    }
    {
        // every load of the thread's cells (all planes) in flight before the first LDS store: a load -> store chain per
        // cell exposes the HBM latency once per cell and plane
        constexpr int NJ = (LL + MP_THREADS - 1) / MP_THREADS;
        size_t off[NJ];
        bool in[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = int(threadIdx.x) + j * MP_THREADS;
            const int la = i / LS, lb = i - la * LS;
            int gu = bu + la, gv = bv + lb;
            gu = gu >= a.nu ? gu % a.nu : gu;
            gv = gv >= a.nv ? gv % a.nv : gv;
            in[j] = i < LL && la < L && lb < L;
            off[j] = size_t(gu) * size_t(a.apitch) + size_t(gv);
        }
        double2 v[KP][NJ];
#pragma unroll
        for (int k = 0; k < kp; ++k) {
            const double2 *gk = grid + size_t(k) * ga.plane_stride;
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[k][j] = in[j] ? gk[off[j]] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < kp; ++k)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int i = int(threadIdx.x) + j * MP_THREADS;
                if (i < LL) tiles[k * LL + i] = v[k][j];
            }
    }
    for (int i = threadIdx.x; i < W * (D + 1); i += MP_THREADS) wtab[i] = a.ktab[i];

    double c[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) c[k] = b < W ? a.ktab[b * (D + 1) + k] : 0.0;
    const double mycoef = b < KP_MAX ? ga.coefk[b] : 0.0;
    __syncthreads();

    const double shift = 1.0 - 0.5 * double(W);
    for (uint32_t jb = wi.begin + wave * 4; jb < wi.end; jb += stride) {
        const uint32_t jn = j + stride;
        const bool nvalid = jn < wi.end;
        const uint32_t jc = nvalid ? jn : wi.begin;
        const double qu = a.pu[jc], qv = a.pv[jc], qw = a.pw[jc]  /* (always allocated; dropped below without w-gridding) */;
        const double2 nacc = accp[rmw ? jc : 0u];
        const double npu = nvalid ? qu : 0.0, npv = nvalid ? qv : 0.0, npw = (nvalid && a.do_w) ? qw : 0.0;
        {
            double kwl = (b < kp && valid) ? plane_weight_of<W, D>(a, a.plane + b, mycoef, pw, wtab) : 0.0;
            const double fu = floor(pu + shift), fv = floor(pv + shift);
            const double zu = 2.0 * ((pu + shift) - fu) - 1.0, zv = 2.0 * ((pv + shift) - fv) - 1.0;
            double ku = horner<D>(c, zu);
            const double kvb = horner<D>(c, zv);
            const int lu = wrap_once((int)fu, a.nu) - bu, lv = wrap_once((int)fv, a.nv) - bv;
            // Lane b takes footprint column cb = (b - lv) mod 16, so its LDS column lv + cb is congruent to b mod 16
            // whatever the visibility: with a row stride of 0 mod 16 cells the 16-byte slot of every lane's read is b mod 16.
            // A ds_read_b128 is serviced in groups of 16 lanes that mix two DPP rows (two visibilities, lanes {0-3, 12-15} of
            // one and {4-11} of the other): the 16 slots of a group are then always distinct -- no bank conflicts between
            // the two footprints (they cost ~45 % of the LDS cycles with lane b on column b).
            const int cb = (b - lv) & 15;
            const double kv = __shfl(kvb, (lane & ~15) + cb);  // the v-kernel value of tap cb, from the lane that evaluated it
            double kw[KP];
            bool touch = false;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                kw[k] = row_bcast_f64(kwl, k);
                touch = touch || (kw[k] != 0.0);
            }
            const int colbase = lu * LS + lv + cb;
            int arow = b;
            double sr[KP], si[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) sr[k] = si[k] = 0.0;
            if (touch) {
                if (W >= 14) {
                    // software pipeline: the reads of step i + 1 are issued before the FMAs of step i; the
                    // row offset arow * LS is rotated itself (one DPP move + one add per step)
                    int rowoff = arow * LS;
                    double2 cur[KP];
#pragma unroll
                    for (int k = 0; k < KP; ++k) cur[k] = tiles[k * LL + colbase + rowoff];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        double2 nxt[KP];
                        rowoff = rot1_i32(rowoff);
                        if (i < 15) {
#pragma unroll
                            for (int k = 0; k < KP; ++k) nxt[k] = tiles[k * LL + colbase + rowoff];
                        }
#pragma unroll
                        for (int k = 0; k < KP; ++k) {
                            sr[k] = fma(cur[k].x, ku, sr[k]);
                            si[k] = fma(cur[k].y, ku, si[k]);
                        }
                        ku = rot1_f64(ku);
                        if (i < 15) {
#pragma unroll
                            for (int k = 0; k < KP; ++k) cur[k] = nxt[k];
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int off = colbase + arow * LS;
                        if (arow < W && cb < W) {
#pragma unroll
                            for (int k = 0; k < KP; ++k) {
                                const double2 gval = tiles[k * LL + off];
                                sr[k] = fma(gval.x, ku, sr[k]);
                                si[k] = fma(gval.y, ku, si[k]);
                            }
                        }
                        ku = rot1_f64(ku);
                        arow = rot1_i32(arow);
                    }
                }
            }
            double tr = 0.0, ti = 0.0;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                tr = fma(sr[k], kw[k], tr);
                ti = fma(si[k], kw[k], ti);
            }
            tr *= kv;
            ti *= kv;
            tr += rotn_f64<8>(tr);
            ti += rotn_f64<8>(ti);
            tr += rotn_f64<4>(tr);
            ti += rotn_f64<4>(ti);
            tr += rotn_f64<2>(tr);
            ti += rotn_f64<2>(ti);
            tr += rotn_f64<1>(tr);
            ti += rotn_f64<1>(ti);
            if (pval_out != nullptr) {
                if (b < KP && valid) {
                    const double wj = swgt[j];
                    pval_out[size_t(j) * size_t(ga.kp_alloc) + size_t(b)] = make_double2((tr * wj) * kwl, (ti * wj) * kwl);
                }
            } else if (b == 0 && touch) {
                sacc[j] = make_double2(acc0.x + tr, acc0.y + ti);
            }
        }
        j = jn;
        acc0 = nacc;
        valid = nvalid;
        pu = npu;
        pv = npv;
        pw = npw;
    }
}

}  // namespace pfbhip
