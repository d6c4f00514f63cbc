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
                const size_t o = (size_t(gu) * size_t(a.nv) + size_t(gv)) * 2;
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
__host__ __device__ constexpr size_t blk_fixed_doubles(int W, int waves) { return size_t(W) * (kernel_poly_degree_c(W) + 1) + size_t(waves) * 2 * BLK_SCRATCH; }
__host__ __device__ constexpr int blk_stride(int W, int KP)
{
    const int L = TILE + W - 1;
    const int wide = L <= 43 ? 43 : 53;
    const size_t bytes = (size_t(2) * KP * L * wide + blk_fixed_doubles(W, 12)) * sizeof(double);
    return bytes <= size_t(160) * 1024 ? wide : ((L & 1) ? L + 1 : L);
}

template <int W, int KP>
__global__ void __launch_bounds__(blk_threads(KP)) k_grid_blk(GroupArgs ga, const double2 *__restrict__ sval,
                                                           double2 *__restrict__ grid)
{
    const PlaneArgs &a = ga.a;
    constexpr int D = kernel_poly_degree_c(W);
    constexpr int L = TILE + W - 1;
    constexpr int LS = blk_stride(W, KP);
    constexpr int LL = blk_tile_rows(W) * LS;
    constexpr int FP = W + BLK_CELLS - 1;  // footprint edge of a block
    constexpr int NR = blk_rows_per_lane(W);
    constexpr int G = BLK_CELLS - 1;       // zero guard in front of the kernel values
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
    double npu = lvalid ? a.pu[jl] : 0.0, npv = lvalid ? a.pv[jl] : 0.0, npw = (lvalid && a.do_w) ? a.pw[jl] : 0.0;
    double2 nval = lvalid ? sval[jl] : make_double2(0.0, 0.0);
    for (int i = threadIdx.x; i < 2 * KP * LL; i += BLK_THREADS) lds[i] = 0.0;
    for (int i = threadIdx.x; i < W * (D + 1); i += BLK_THREADS) wtab[i] = a.ktab[i];
    for (int i = threadIdx.x; i < (BLK_THREADS / 64) * 2 * BLK_SCRATCH; i += BLK_THREADS) scr_all[i] = 0.0;

    double *scr = scr_all + wave * 2 * BLK_SCRATCH;
    const int b = lane & 15;
    double c[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) c[k] = (b < W && lane < 32) ? a.ktab[b * (D + 1) + k] : 0.0;
    // cell ownership: lane = 20 g + cc, rows 3 k + g
    const int g = lane / 20, cc = lane - 20 * g;
    const bool act = g < 3 && cc < FP;
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
        const int r0 = (key >> 8) * BLK_CELLS, c0 = (key & 255) * BLK_CELLS;
        if (act) {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int r = 3 * k + g;
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
        npu = lvalid ? a.pu[jl] : 0.0;
        npv = lvalid ? a.pv[jl] : 0.0;
        npw = (lvalid && a.do_w) ? a.pw[jl] : 0.0;
        nval = lvalid ? sval[jl] : make_double2(0.0, 0.0);
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
            const int key = ((slu >> 2) << 8) | (slv >> 2);
            if (key != cur) {
                if (cur >= 0) flush(cur);
                cur = key;
            }
            const double *sc = scr + (i & 1) * BLK_SCRATCH;
            const int du = slu & 3, dv = slv & 3;
            const double kvc = sc[24 + cc - dv + G];
            const double *su = sc + (g - du + G);
            double kuv[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) kuv[k] = su[3 * k];
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
                    const size_t o = (size_t(gu) * size_t(a.nv) + size_t(gv)) * 2;
                    unsafeAtomicAdd(&gp[o], re);
                    unsafeAtomicAdd(&gp[o + 1], im);
                }
            }
        }
        return;
    }
    // all loads of the thread's cells first (a load -> add -> store chain per cell would expose the HBM latency once per
    // cell: ~9 cells per thread), then the stores
    constexpr int NJ = (L * L + 511) / 512;  // cells per thread and plane at the smallest workgroup
    size_t off[NJ];
    int lo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = int(threadIdx.x) + j * BLK_THREADS;
        const int la = i / L, lb = i - la * L;
        int gu = bu + la, gv = bv + lb;
        gu = gu >= a.nu ? gu % a.nu : gu;
        gv = gv >= a.nv ? gv % a.nv : gv;
        off[j] = size_t(gu) * size_t(a.nv) + size_t(gv);
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
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        double2 *gk = grid + size_t(k) * ga.plane_stride;
        const double *lre = lds + (2 * k) * LL, *lim = lds + (2 * k + 1) * LL;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lo[j] >= 0) {
                const double re = lre[lo[j]], im = lim[lo[j]];
                if (re != 0.0 || im != 0.0) gk[off[j]] = make_double2(v[k][j].x + re, v[k][j].y + im);
            }
    }
}

template <int W, int KP>
__global__ void __launch_bounds__(MP_THREADS) k_degrid_mp(GroupArgs ga, const double2 *__restrict__ grid,
                                                           double2 *__restrict__ sacc)
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
    double pu = valid ? a.pu[j] : 0.0, pv = valid ? a.pv[j] : 0.0, pw = (valid && a.do_w) ? a.pw[j] : 0.0;
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
            off[j] = size_t(gu) * size_t(a.nv) + size_t(gv);
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
        const double npu = nvalid ? a.pu[jn] : 0.0, npv = nvalid ? a.pv[jn] : 0.0;
        const double npw = (nvalid && a.do_w) ? a.pw[jn] : 0.0;
        {
            double kwl = (b < kp && valid) ? plane_weight_of<W, D>(a, a.plane + b, mycoef, pw, wtab) : 0.0;
            const double fu = floor(pu + shift), fv = floor(pv + shift);
            const double zu = 2.0 * ((pu + shift) - fu) - 1.0, zv = 2.0 * ((pv + shift) - fv) - 1.0;
            double ku = horner<D>(c, zu);
            const double kv = horner<D>(c, zv);
            const int lu = wrap_once((int)fu, a.nu) - bu, lv = wrap_once((int)fv, a.nv) - bv;
            double kw[KP];
            bool touch = false;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                kw[k] = row_bcast_f64(kwl, k);
                touch = touch || (kw[k] != 0.0);
            }
            const int colbase = lu * LS + lv + b;
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
                        if (arow < W && b < W) {
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
            if (b == 0 && touch) {
                double2 acc = sacc[j];
                acc.x += tr;
                acc.y += ti;
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
