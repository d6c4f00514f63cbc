// rowfft.hip -- kernels built on rowfft.hpp: plain batched row transform (debug / benchmark entry)
// and the fused second-axis passes of the gridder's plane transform.
// Compiled with FMA contraction ON (no bit-exact index arithmetic lives in this file).
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "rowfft.hpp"
#include "rowfft_api.hpp"

namespace pfbhip {

// (load functors: two-step form, see rf_has_fetch in rowfft.hpp -- fetch() is the bare, branch-free request)
struct PlainLoad {
    const double2 *row;
    __device__ __forceinline__ double2 fetch(int i, int) const { return row[i]; }
    __device__ __forceinline__ double2 finish(double2 v, int, int) const { return v; }
    __device__ __forceinline__ double2 operator()(int i, int) const { return row[i]; }
};
struct PlainStore {
    double2 *row;
    __device__ __forceinline__ void operator()(int i, double2 v) const { row[i] = v; }
};
// Row of a uv-plane of which only the columns [r.x, r.y) and [r.z, r.w) are in use (the tile columns the row's tile row has
// visibilities in, plus their halo): the rest is zero by construction on the gridding side and never read on the degridding
// side -- with a disc-shaped uv coverage two thirds of an occupied row.
struct RunLoad {
    const double2 *row;
    int4 r;
    __device__ __forceinline__ bool in(int i) const { return (i >= r.x && i < r.y) || (i >= r.z && i < r.w); }
    // outside the runs: element 0 of the row (one hot line per wave), dropped by finish()
    __device__ __forceinline__ double2 fetch(int i, int) const { return row[in(i) ? i : 0]; }
    __device__ __forceinline__ double2 finish(double2 v, int i, int) const { return in(i) ? v : make_double2(0.0, 0.0); }
    __device__ __forceinline__ double2 operator()(int i, int s) const { return finish(fetch(i, s), i, s); }
};
struct RunStore {
    double2 *row;
    int4 r;
    __device__ __forceinline__ void operator()(int i, double2 v) const
    {
        if ((i >= r.x && i < r.y) || (i >= r.z && i < r.w)) row[i] = v;
    }
};

// One row per workgroup, S::T threads; the register budget follows from the workgroup size
// (512 threads -> 256 VGPRs, 640/768 -> 168, 1024 -> 128); the straight-line passes need ~165.
// INV is a template parameter: a run-time direction costs 128 v_cndmask per row and 20 % of the rate.
template <class S, bool INV>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_rowfft_plain(const double2 *tw, double2 *data, int nrows, size_t pitch)
{
    extern __shared__ double rf_lds[];
    const int row = blockIdx.x;
    if (row >= nrows) return;
    PlainLoad ld{data + size_t(row) * pitch};
    PlainStore st{data + size_t(row) * pitch};
    rf_row<S>(tw, ld, st, INV, rf_lds);
}

// LDS beyond 64 KiB needs the opt-in attribute, per (device, kernel): see allow_dynamic_lds()
template <class Kern>
static void rf_allow_lds(Kern kern, bool *)
{
    allow_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
}

bool RowFFT::init(int64_t N)
{
    release();
    if (!rowfft_make_plan(N, &pl)) return false;
    const long double pi = 3.141592653589793238462643383279502884L;
    // plain shapes: exp(-2 pi i k / N), k < N.  Doubled shapes: the N1-point table, then exp(-2 pi i k / N), k < N1.
    const int64_t n1 = pl.doubled ? N / 2 : N;
    // (host tables are kept per length: a plan has two axes, usually of one length, and a process builds many plans)
    static std::mutex tw_mu;
    static std::map<int64_t, std::vector<double2>> tw_cache;
    std::vector<double2> tw;
    {
        std::lock_guard<std::mutex> lk(tw_mu);
        auto it = tw_cache.find(N);
        if (it != tw_cache.end()) tw = it->second;
    }
    if (tw.empty()) {
        tw.assign(static_cast<size_t>(pl.doubled ? 2 * n1 : n1), make_double2(0.0, 0.0));
        for (int64_t k = 0; k < n1; ++k) {
            long double a = -2.0L * pi * (long double)k / (long double)n1;
            tw[size_t(k)] = make_double2(double(cosl(a)), double(sinl(a)));
        }
        if (pl.doubled)
            for (int64_t k = 0; k < n1; ++k) {
                long double a = -2.0L * pi * (long double)k / (long double)N;
                tw[size_t(n1 + k)] = make_double2(double(cosl(a)), double(sinl(a)));
            }
        std::lock_guard<std::mutex> lk(tw_mu);
        if (tw_cache.size() >= 16) tw_cache.clear();
        tw_cache[N] = tw;
    }
    d_tw = static_cast<double2 *>(dev_alloc(tw.size() * sizeof(double2)));
    d_tw_bytes = tw.size() * sizeof(double2);
    PFB_HIP(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice));
    pl.twiddle = d_tw;
    ok = true;
    return true;
}

void RowFFT::release()
{
    if (d_tw) dev_free(d_tw, d_tw_bytes);
    d_tw = nullptr;
    d_tw_bytes = 0;
    ok = false;
}

template <class S, bool INV>
static void launch_plain(const RowFFTPlan &pl, double2 *data_dev, int nrows, size_t pitch, hipStream_t stream)
{
    static bool attr = false;
    rf_allow_lds(&k_rowfft_plain<S, INV>, &attr);
    hipLaunchKernelGGL((k_rowfft_plain<S, INV>), dim3(uint32_t(nrows)), dim3(S::T), size_t(S::LDS_BYTES), stream,
                       pl.twiddle, data_dev, nrows, pitch);
}

void rowfft_plain(const RowFFTPlan &pl, double2 *data_dev, int nrows, bool inverse, hipStream_t stream, size_t pitch)
{
    if (pitch == 0) pitch = size_t(pl.N);
    switch (pl.N) {
#define RF_X(L, K)                                                                  \
    case (L << K):                                                                  \
        if (inverse) launch_plain<RfShape<L, K>, true>(pl, data_dev, nrows, pitch, stream); \
        else launch_plain<RfShape<L, K>, false>(pl, data_dev, nrows, pitch, stream);        \
        break;
        RF_FOR_SHAPES(RF_X)
#undef RF_X
#define RF_X(L, K)                                                                                 \
    case 2 * (L << K):                                                                             \
        if (inverse) launch_plain<RfShape2<RfShape<L, K>>, true>(pl, data_dev, nrows, pitch, stream);     \
        else launch_plain<RfShape2<RfShape<L, K>>, false>(pl, data_dev, nrows, pitch, stream);            \
        break;
        RF_FOR_SHAPES2(RF_X)
#undef RF_X
        default: PFB_REQUIRE(false, "row length %d is not supported by the hand-written FFT", pl.N);
    }
    PFB_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
// first-axis passes with the transpose folded in (no separate k_a2b / k_b2a pass over the plane)
// ---------------------------------------------------------------------------------------
// grid side:   B[y][u] = IFFT_v(A[u][:])[wrap(y - ny/2, nv)]      (crop + transpose in the store)
// degrid side: A[u][:] = FFT_v( v -> B[y(v)][u], 0 outside the image )  (pad + transpose in the load)
// One workgroup owns one row u, so its B accesses are 16-byte pieces nu * 16 bytes apart.  What makes that
// affordable is WHICH workgroups run together: rowmap[] hands the 8 rows of every 128-byte line of B to 8
// workgroups that share an XCD (blockIdx equal mod 8) and are dispatched within the same 64 block ids, so the 8
// pieces of a line meet in that XCD's L2 -- stores leave it as whole lines, loads miss once per line.
struct CropTStore {
    double2 *B;
    int u, nu, ny, nv, hy;
    __device__ __forceinline__ void operator()(int v, double2 val) const
    {
        int y = -1;
        if (v < ny - hy) y = v + hy;
        else if (v >= nv - hy) y = v - (nv - hy);
        if (y >= 0) B[size_t(y) * size_t(nu) + size_t(u)] = val;
    }
};
struct PadTLoad {
    const double2 *B;
    int u, nu, ny, nv, hy;
    __device__ __forceinline__ int yof(int v) const { return v < ny - hy ? v + hy : (v >= nv - hy ? v - (nv - hy) : -1); }
    __device__ __forceinline__ double2 fetch(int v, int) const { return B[size_t(max(yof(v), 0)) * size_t(nu) + size_t(u)]; }
    __device__ __forceinline__ double2 finish(double2 x, int v, int) const { return yof(v) >= 0 ? x : make_double2(0.0, 0.0); }
    __device__ __forceinline__ double2 operator()(int v, int s) const { return finish(fetch(v, s), v, s); }
};

// (Round 3 tried an L2 warm-up here -- every workgroup, once its own requests were out, requested one dword per 128-byte line
// of the row its CU would take next, with all twiddles requested up front so that no later wait drained those requests.
// Measured on C2: first axis 2.17 -> 2.32 ms, fused second axis unchanged; removed.  DESIGN.md section 5.1.)
// blockIdx.y = plane of the launch (astride / bstride elements apart): one launch for all planes of a pass leaves one
// partially filled round of workgroups instead of one per plane (4896 rows on 256 CUs: 19.1 rounds each)
template <class S>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_rowfft_a2b(const double2 *tw, const double2 *A, double2 *B,
                                                                          const int *rowmap, int nrows, int nu, int ny, size_t apitch,
                                                                          size_t astride, size_t bstride, const int4 *colruns)
{
    extern __shared__ double rf_lds[];
    if (int(blockIdx.x) >= nrows) return;
    const int u = rowmap[blockIdx.x];
    A += size_t(blockIdx.y) * astride;
    B += size_t(blockIdx.y) * bstride;
    RunLoad ld{A + size_t(u) * apitch, colruns[u >> 5]};
    CropTStore st{B, u, nu, ny, S::N, ny / 2};
    rf_row<S>(tw, ld, st, true, rf_lds);
}

// Row u of Bt (tpitch elements per row, written by the fused pad kernel with the transpose in ITS stores): contiguous loads.
struct PadRowLoad {
    const double2 *row;
    int ny, nv, hy;
    __device__ __forceinline__ int yof(int v) const { return v < ny - hy ? v + hy : (v >= nv - hy ? v - (nv - hy) : -1); }
    __device__ __forceinline__ double2 fetch(int v, int) const { return row[max(yof(v), 0)]; }
    __device__ __forceinline__ double2 finish(double2 x, int v, int) const { return yof(v) >= 0 ? x : make_double2(0.0, 0.0); }
    __device__ __forceinline__ double2 operator()(int v, int s) const { return finish(fetch(v, s), v, s); }
};

template <class S, bool TR>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_rowfft_b2a(const double2 *tw, const double2 *B, double2 *A,
                                                                          const int *rowmap, int nrows, int nu, int ny, size_t apitch,
                                                                          int tpitch, size_t astride, size_t bstride, const int4 *colruns)
{
    extern __shared__ double rf_lds[];
    if (int(blockIdx.x) >= nrows) return;
    const int u = rowmap[blockIdx.x];
    A += size_t(blockIdx.y) * astride;
    B += size_t(blockIdx.y) * bstride;
    RunStore st{A + size_t(u) * apitch, colruns[u >> 5]};
    if constexpr (TR) {
        PadRowLoad ld{B + size_t(u) * size_t(tpitch), ny, S::N, ny / 2};
        rf_row<S>(tw, ld, st, false, rf_lds);
    } else {
        PadTLoad ld{B, u, nu, ny, S::N, ny / 2};
        rf_row<S>(tw, ld, st, false, rf_lds);
    }
}

// slots of the waiting half that fit behind the exchange buffer (the rest stay in registers); 0: no stash
template <class S1>
constexpr int rf_stash_slots()
{
    const int room = (160 * 1024 - S1::LDS_BYTES) / int(sizeof(double)) / S1::T;
    return room >= S1::E - 2 ? (room < S1::E ? room : S1::E) : 0;
}

// Doubled shapes: the generic rf_row<RfShape2> keeps all 2 x 16 outputs (128 VGPRs) next to the waiting half and spills 200+
// registers; here, as in the fused second-axis kernels, the even half's real parts are parked in LDS while the odd half is
// transformed (STASH) and every output pair is stored as soon as it is combined.
template <class S1, bool STASH, class Load, class Store>
__device__ __forceinline__ void rf_row2(const double2 *__restrict__ tw, Load &ld, Store &st, bool inverse, double *rf_lds)
{
    double *stash = rf_lds + S1::LDS_BYTES / sizeof(double);
    RfHalfLoad<Load, 0, 0> ld_e{ld};
    RfHalfLoad<Load, 1, S1::NSLOT> ld_o{ld};
    double er[S1::E], ei[S1::E], orr[S1::E], oi[S1::E];
    int t;
    rf_row_compute<S1>(tw, ld_e, inverse, rf_lds, t, er, ei);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NST = STASH ? rf_stash_slots<S1>() : 0;
#pragma unroll
    for (int e = 0; e < NST; ++e) stash[e * S1::T + t] = er[e];
    rf_row_compute<S1, decltype(ld_o), 2>(tw, ld_o, inverse, rf_lds, t, orr, oi);
    __builtin_amdgcn_sched_barrier(0);
    rf_opaque(t);
    const double2 *__restrict__ tw2 = tw + S1::N;
#pragma unroll
    for (int e = 0; e < S1::E; ++e) {
        const int k1 = S1::out_pos(t, e);
        const double2 w = tw2[k1];
        const double ere = e < NST ? stash[e * S1::T + t] : er[e];
        // (the inverse is the forward transform of the (im, re)-swapped row, swapped back on the way out: the combination
        // below works in that swapped domain with the forward twiddle, like rf_row_compute's doubled branch)
        const double tr = orr[e] * w.x - oi[e] * w.y, ti = orr[e] * w.y + oi[e] * w.x;
        const double2 lo = make_double2(ere + tr, ei[e] + ti), hi = make_double2(ere - tr, ei[e] - ti);
        st(k1, inverse ? make_double2(lo.y, lo.x) : lo);
        st(k1 + S1::N, inverse ? make_double2(hi.y, hi.x) : hi);
        // (four combination twiddles requested at a time: all 16 would cost 64 VGPRs.  Computing them instead -- w^t times a
        // scalar-loaded w^(c T), r03g, or times the literal 32nd root of unity exp(-i pi c / 16), r03q -- was measured 4-10 %
        // slower on C5 both times; so was a two-phase form with all stores behind all combinations: 114 spilled VGPRs)
        if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

template <class S1, bool STASH>
__global__ void __launch_bounds__(S1::T, ((S1::T + 63) / 64 + 3) / 4) k_rowfft_a2b2(const double2 *tw, const double2 *A, double2 *B,
                                                                                    const int *rowmap, int nrows, int nu, int ny,
                                                                                    size_t apitch, size_t astride, size_t bstride,
                                                                                    const int4 *colruns)
{
    extern __shared__ double rf_lds[];
    if (int(blockIdx.x) >= nrows) return;
    const int u = rowmap[blockIdx.x];
    A += size_t(blockIdx.y) * astride;
    B += size_t(blockIdx.y) * bstride;
    RunLoad ld{A + size_t(u) * apitch, colruns[u >> 5]};
    CropTStore st{B, u, nu, ny, 2 * S1::N, ny / 2};
    rf_row2<S1, STASH>(tw, ld, st, true, rf_lds);
}
template <class S1, bool TR, bool STASH>
__global__ void __launch_bounds__(S1::T, ((S1::T + 63) / 64 + 3) / 4) k_rowfft_b2a2(const double2 *tw, const double2 *B, double2 *A,
                                                                                    const int *rowmap, int nrows, int nu, int ny,
                                                                                    size_t apitch, int tpitch, size_t astride,
                                                                                    size_t bstride, const int4 *colruns)
{
    extern __shared__ double rf_lds[];
    if (int(blockIdx.x) >= nrows) return;
    const int u = rowmap[blockIdx.x];
    A += size_t(blockIdx.y) * astride;
    B += size_t(blockIdx.y) * bstride;
    RunStore st{A + size_t(u) * apitch, colruns[u >> 5]};
    if constexpr (TR) {
        PadRowLoad ld{B + size_t(u) * size_t(tpitch), ny, 2 * S1::N, ny / 2};
        rf_row2<S1, STASH>(tw, ld, st, false, rf_lds);
    } else {
        PadTLoad ld{B, u, nu, ny, 2 * S1::N, ny / 2};
        rf_row2<S1, STASH>(tw, ld, st, false, rf_lds);
    }
}
template <class S1>
static void launch_a2b2(const RowFFTPlan &pl, const double2 *A, double2 *B, const int *rowmap, int nrows, int nu, int ny,
                        size_t apitch, int nplanes, size_t astride, size_t bstride, const int4 *colruns, hipStream_t stream)
{
    static bool attr = false;
    constexpr bool STASH = rf_stash_slots<S1>() > 0;
    rf_allow_lds(&k_rowfft_a2b2<S1, STASH>, &attr);
    hipLaunchKernelGGL((k_rowfft_a2b2<S1, STASH>), dim3(uint32_t((nrows + 7) / 8 * 8), uint32_t(nplanes)), dim3(S1::T),
                       size_t(S1::LDS_BYTES) + size_t(rf_stash_slots<S1>()) * S1::T * sizeof(double), stream, pl.twiddle, A, B, rowmap,
                       nrows, nu, ny, apitch, astride, bstride, colruns);
}
template <class S1>
static void launch_b2a2(const RowFFTPlan &pl, const double2 *B, double2 *A, const int *rowmap, int nrows, int nu, int ny,
                        size_t apitch, int tpitch, int nplanes, size_t astride, size_t bstride, const int4 *colruns, hipStream_t stream)
{
    static bool attr = false, attr_t = false;
    constexpr bool STASH = rf_stash_slots<S1>() > 0;
    const dim3 grid(uint32_t((nrows + 7) / 8 * 8), uint32_t(nplanes));
    const size_t lds = size_t(S1::LDS_BYTES) + size_t(rf_stash_slots<S1>()) * S1::T * sizeof(double);
    if (tpitch > 0) {
        rf_allow_lds(&k_rowfft_b2a2<S1, true, STASH>, &attr_t);
        hipLaunchKernelGGL((k_rowfft_b2a2<S1, true, STASH>), grid, dim3(S1::T), lds, stream, pl.twiddle, B, A, rowmap, nrows, nu, ny,
                           apitch, tpitch, astride, bstride, colruns);
        return;
    }
    rf_allow_lds(&k_rowfft_b2a2<S1, false, STASH>, &attr);
    hipLaunchKernelGGL((k_rowfft_b2a2<S1, false, STASH>), grid, dim3(S1::T), lds, stream, pl.twiddle, B, A, rowmap, nrows, nu, ny, apitch,
                       0, astride, bstride, colruns);
}

template <class S>
static void launch_a2b(const RowFFTPlan &pl, const double2 *A, double2 *B, const int *rowmap, int nrows, int nu, int ny,
                       size_t apitch, int nplanes, size_t astride, size_t bstride, const int4 *colruns, hipStream_t stream)
{
    static bool attr = false;
    rf_allow_lds(&k_rowfft_a2b<S>, &attr);
    // (grid.x a multiple of 8: a plane's workgroups then start on XCD 0 like the first plane's -- rowmap relies on it)
    hipLaunchKernelGGL((k_rowfft_a2b<S>), dim3(uint32_t((nrows + 7) / 8 * 8), uint32_t(nplanes)), dim3(S::T), size_t(S::LDS_BYTES),
                       stream, pl.twiddle, A, B, rowmap, nrows, nu, ny, apitch, astride, bstride, colruns);
}
template <class S>
static void launch_b2a(const RowFFTPlan &pl, const double2 *B, double2 *A, const int *rowmap, int nrows, int nu, int ny,
                       size_t apitch, int tpitch, int nplanes, size_t astride, size_t bstride, const int4 *colruns, hipStream_t stream)
{
    static bool attr = false, attr_t = false;
    const dim3 grid(uint32_t((nrows + 7) / 8 * 8), uint32_t(nplanes));
    if (tpitch > 0) {
        rf_allow_lds(&k_rowfft_b2a<S, true>, &attr_t);
        hipLaunchKernelGGL((k_rowfft_b2a<S, true>), grid, dim3(S::T), size_t(S::LDS_BYTES), stream, pl.twiddle, B, A, rowmap, nrows,
                           nu, ny, apitch, tpitch, astride, bstride, colruns);
        return;
    }
    rf_allow_lds(&k_rowfft_b2a<S, false>, &attr);
    hipLaunchKernelGGL((k_rowfft_b2a<S, false>), grid, dim3(S::T), size_t(S::LDS_BYTES), stream, pl.twiddle, B, A, rowmap, nrows, nu,
                       ny, apitch, 0, astride, bstride, colruns);
}

void rowfft_a2b(const RowFFTPlan &pl, const double2 *A_dev, double2 *B_dev, const int *rowmap_dev, int nrows, int nu, int ny,
                size_t apitch, int nplanes, size_t astride, size_t bstride, const int4 *colruns, hipStream_t stream)
{
    switch (pl.N) {
#define RF_X(L, K)                                                                       \
    case (L << K): launch_a2b<RfShape<L, K>>(pl, A_dev, B_dev, rowmap_dev, nrows, nu, ny, apitch, nplanes, astride, bstride, colruns, stream); break;
        RF_FOR_SHAPES(RF_X)
#undef RF_X
#define RF_X(L, K)                                                                       \
    case 2 * (L << K): launch_a2b2<RfShape<L, K, false>>(pl, A_dev, B_dev, rowmap_dev, nrows, nu, ny, apitch, nplanes, astride, bstride, colruns, stream); break;
        RF_FOR_SHAPES2(RF_X)
#undef RF_X
        default: PFB_REQUIRE(false, "row length %d is not supported by the hand-written FFT", pl.N);
    }
    PFB_HIP(hipGetLastError());
}

void rowfft_b2a(const RowFFTPlan &pl, const double2 *B_dev, double2 *A_dev, const int *rowmap_dev, int nrows, int nu, int ny,
                size_t apitch, int tpitch, int nplanes, size_t astride, size_t bstride, const int4 *colruns, hipStream_t stream)
{
    switch (pl.N) {
#define RF_X(L, K)                                                                       \
    case (L << K): launch_b2a<RfShape<L, K>>(pl, B_dev, A_dev, rowmap_dev, nrows, nu, ny, apitch, tpitch, nplanes, astride, bstride, colruns, stream); break;
        RF_FOR_SHAPES(RF_X)
#undef RF_X
#define RF_X(L, K)                                                                       \
    case 2 * (L << K): launch_b2a2<RfShape<L, K, false>>(pl, B_dev, A_dev, rowmap_dev, nrows, nu, ny, apitch, tpitch, nplanes, astride, bstride, colruns, stream); break;
        RF_FOR_SHAPES2(RF_X)
#undef RF_X
        default: PFB_REQUIRE(false, "row length %d is not supported by the hand-written FFT", pl.N);
    }
    PFB_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
// fused second-axis passes of the plane transform
// ---------------------------------------------------------------------------------------

// Row of B with the unoccupied 32-column blocks read as zero.  The occupancy of the thread's
// elements is looked up once per workgroup (bit `slot` of mask), not once per element and plane: a
// byte load in front of every row load doubles the latency of the load phase.
struct OccLoad {
    const double2 *row;
    uint64_t mask;
    __device__ __forceinline__ bool on(int slot) const { return ((mask >> slot) & 1ull) != 0; }
    // unoccupied blocks were never written: request element 0 of the row instead (one hot line), dropped by finish()
    __device__ __forceinline__ double2 fetch(int u, int slot) const { return row[on(slot) ? u : 0]; }
    __device__ __forceinline__ double2 finish(double2 v, int, int slot) const { return on(slot) ? v : make_double2(0.0, 0.0); }
    __device__ __forceinline__ double2 operator()(int u, int slot) const { return finish(fetch(u, slot), u, slot); }
};

// lds_row: the kernel keeps the running sum over the launch's planes of its image row in LDS (nx doubles
// behind the transpose buffer) and touches accT once; otherwise (no room) every plane read-modify-writes accT.
//
// Memory requests are kept out of divergent control flow (round 3; see rf_has_fetch in rowfft.hpp): the row of B comes in
// through OccLoad's branch-free fetch, and the epilogue of a plane that goes to the image requests the running sum, the
// correction, the beam and x for two outputs at a time from clamped addresses, one group ahead of the group whose
// screens are being evaluated; results wait in the registers of the transform and are stored after the last group (a
// store between two request groups would make the compiler wait for it: loads and stores share the in-order vmcnt).
template <class S, bool SC>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_fused_fft_crop(const double2 *tw, FusedGeom g, const uint8_t *occ,
                                                          const double2 *B, size_t bstride, FusedPlanes planes,
                                                          int do_w, int first, int lds_row, double *accT, FusedFinal fin)
{
    extern __shared__ double rf_lds[];
    double *acc = rf_lds + S::LDS_BYTES / sizeof(double);
    const int y = blockIdx.x;
    double *arow = accT + size_t(y) * g.nx;
    uint64_t mask = 0;
    rf_for_each_load<S>(int(threadIdx.x), [&](int u, int slot) { mask |= (occ[u >> 5] ? 1ull : 0ull) << slot; });
    for (int k = 0; k < planes.kp; ++k) {
        OccLoad ld{B + size_t(k) * bstride + size_t(y) * size_t(g.bpitch), mask};
        double re[S::E], im[S::E];
        int t;
        rf_row_compute<S>(tw, ld, true, rf_lds, t, re, im);
        rf_opaque(t);
        const double wk = planes.w[k];
        double cc[FUSED_SCMAX], ss[FUSED_SCMAX];  // SC: this plane's composite screen polynomials
        if constexpr (SC) {
#pragma unroll
            for (int q = 0; q < FUSED_SCMAX; ++q) {
                cc[q] = planes.cs[k][q];
                ss[q] = planes.sn[k][q];
            }
        }
        const bool last = k == planes.kp - 1;
        // what must be added to this plane's value: the LDS running sum (thread-private cells, no barrier)
        // and / or the image so far
        const bool add_lds = lds_row && k > 0;
        const bool add_img = lds_row ? (last && !first) : !(first && k == 0);
        const bool to_img = !lds_row || last;
        const bool finalize = last && to_img && fin.corr != nullptr;
        // value of output e of this plane: Re( (im + i re) * (c - i s) ) (inverse transform: value = (im, re))
        auto plane_value = [&](int e, int ix) {
            double r = im[e];
            if (do_w) {
                double s, c;
                if constexpr (SC) {
                    fg_screen_poly(g, cc, ss, ix, y, s, c, planes.nsc);
                } else {
                    double ph = wk * fg_t_poly(g, ix, y);
                    ph -= rint(ph);
                    fg_sincos2pi(ph, s, c);
                }
                r = im[e] * c + re[e] * s;
            }
            return r;
        };
        if (!to_img) {  // running sum in LDS: no global memory in this epilogue
#pragma unroll
            for (int e = 0; e < S::E; ++e) {
                const int ix = fg_ix(g, S::out_pos(t, e));
                if (ix >= 0) {
                    double r = plane_value(e, ix);
                    if (add_lds) r += acc[ix];
                    acc[ix] = r;
                }
                if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the number of sincos chains in flight
            }
            continue;
        }
        // rows the groups request from (always valid memory: absent arrays alias the accumulator row and are dropped)
        // (the FIN split of k_fused_fft_crop2 -- a lean instance for the planes that only update the running sum -- was tried
        // here too: the 1024-thread shapes, 128 VGPRs, then spill 14-21 registers; one body stays)
        const size_t ro = size_t(y) * size_t(g.nx);
        const bool has_beam = finalize && fin.beam != nullptr, has_x = finalize && fin.x != nullptr;
        const double *crow = finalize ? fin.corr + ro : arow;
        const double *brow = has_beam ? fin.beam + ro : crow;
        const double *xrow = has_x ? fin.x + ro : crow;
        // (a running sum that is not added is not read either: r03t's counters showed 0.54 GB per launch of it at C2)
        const double *irow = add_img ? arow : crow;
        constexpr int GQ = 2;  // outputs per group (four: 64 VGPRs of requests in flight next to the 64 of the transform -- spills)
        constexpr int NG = S::E / GQ;
        double qi[2][GQ], qc[2][GQ], qb[2][GQ], qx[2][GQ];
        auto request = [&](int e0, int buf) {
#pragma unroll
            for (int j = 0; j < GQ; ++j) {
                const int ixc = max(fg_ix(g, S::out_pos(t, e0 + j)), 0);
                qi[buf][j] = irow[ixc];
                qc[buf][j] = crow[ixc];
                qb[buf][j] = brow[ixc];
                qx[buf][j] = xrow[ixc];
            }
        };
        request(0, 0);
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) request(GQ * (gq + 1), (gq + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < GQ; ++j) {
                const int e = GQ * gq + j;
                const int ix = fg_ix(g, S::out_pos(t, e));
                double v = 0.0;
                if (ix >= 0) {
                    double r = plane_value(e, ix);
                    if (add_lds) r += acc[ix];
                    if (add_img) r += qi[gq & 1][j];
                    if (finalize) {  // finalize in place of a separate pass over the image
                        const double c = has_beam ? qc[gq & 1][j] * qb[gq & 1][j] : qc[gq & 1][j];
                        v = r * c * fin.scale;
                        if (has_x) v += fin.eta * qx[gq & 1][j];
                    } else {
                        v = r;
                    }
                }
                re[e] = v;  // (re[e] / im[e] are dead from here on)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        double *orow = finalize ? fin.out + ro : arow;
#pragma unroll
        for (int e = 0; e < S::E; ++e) {
            const int ix = fg_ix(g, S::out_pos(t, e));
            if (ix >= 0) orow[ix] = re[e];
        }
    }
}

// Degridding side.  The prepared image row x * corr [* beam] lives in LDS across the planes (lds_row; filled by a pre-pass
// whose requests -- three arrays, half the thread's pixels at a time -- are all in flight together), or, where it does not
// fit, is read per plane from a PREPARED image (the caller ran k_prepare_img).  Either way the load functor's fetch is one
// branch-free read and its finish the screen.
template <bool SC, bool LROW>
struct PadLoadT {
    const double *src;   // LROW: the LDS row; else row y of the prepared image
    const FusedGeom &g;  // the kernel argument itself (a copy would put the coefficient array in scratch)
    const double (&cc)[FUSED_SCMAX];  // SC: the plane's composite screen polynomials (registers of the kernel)
    const double (&ss)[FUSED_SCMAX];
    int y, do_w;
    double wk;
    int nsc;  // SC: coefficients the chains evaluate (FusedPlanes::nsc)
    __device__ __forceinline__ double fetch(int u, int) const { return src[max(fg_ix(g, u), 0)]; }
    __device__ __forceinline__ double2 finish(double val, int u, int) const
    {
        const int ix = fg_ix(g, u);
        if (ix < 0) return make_double2(0.0, 0.0);
        if (!do_w) return make_double2(val, 0.0);
        double s, c;
        if constexpr (SC) {
            fg_screen_poly(g, cc, ss, ix, y, s, c, nsc);
        } else {
            double ph = wk * fg_t_poly(g, ix, y);
            ph -= rint(ph);
            fg_sincos2pi(ph, s, c);
        }
        return make_double2(val * c, val * s);
    }
    __device__ __forceinline__ double2 operator()(int u, int s) const { return finish(fetch(u, s), u, s); }
};

using PadLoad = PadLoadT<false, false>;

template <class S, bool SC>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_fused_pad_fft(const double2 *tw, FusedGeom g, const uint8_t *occ,
                                                         const double *dcT, FusedPrep prep, FusedPlanes planes, int do_w,
                                                         int lds_row, double2 *B, size_t bstride)
{
    extern __shared__ double rf_lds[];
    double *lrow = rf_lds + S::LDS_BYTES / sizeof(double);
    int y = blockIdx.x;
    if (g.tpitch > 0 && y < (g.ny & ~63)) {
        // transposed stores (below): the 8 rows y whose 16-byte pieces make up one 128-byte line of Bt go to 8 workgroups of
        // ONE XCD (block ids equal mod 8) inside the same 64 block ids, so that the line leaves that XCD's L2 whole
        const int r = y & 63;
        y = (y & ~63) + (r & 7) * 8 + (r >> 3);
    }
    uint32_t omask = 0;  // occupancy of the thread's output columns
#pragma unroll
    for (int e = 0; e < S::E; ++e) omask |= (occ[S::out_pos(int(threadIdx.x), e) >> 5] ? 1u : 0u) << e;
    const size_t ro = size_t(y) * size_t(g.nx);
    if (lds_row) {
        // pre-pass: lrow[ix] = x * corr [* beam] (or the prepared value) for the pixels this thread will be asked for
        // (thread-private cells: the same thread reads them back on every plane, no barrier)
        const bool raw = prep.x != nullptr, has_beam = raw && prep.beam != nullptr;
        const double *xr = (raw ? prep.x : dcT) + ro;
        const double *cr = raw ? prep.corr + ro : xr;
        const double *br = has_beam ? prep.beam + ro : cr;
        constexpr int NH = (S::NSLOT + 1) / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double vx[NH], vc[NH], vb[NH];
            rf_for_each_load<S>(int(threadIdx.x), [&](int u, int slot) {
                if (slot / NH == h) {
                    const int ixc = max(fg_ix(g, u), 0);
                    vx[slot % NH] = xr[ixc];
                    vc[slot % NH] = cr[ixc];
                    vb[slot % NH] = br[ixc];
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            rf_for_each_load<S>(int(threadIdx.x), [&](int u, int slot) {
                if (slot / NH == h) {
                    const int ix = fg_ix(g, u);
                    double v = vx[slot % NH];
                    if (raw) {  // (x * corr) * beam: the rounding order of k_prepare_img, which the read-modify-write form uses
                        v *= vc[slot % NH];
                        if (has_beam) v *= vb[slot % NH];
                    }
                    if (ix >= 0) lrow[ix] = v;
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    auto planes_loop = [&](auto lrow_tag) {
        constexpr bool LROW = decltype(lrow_tag)::value;
        for (int k = 0; k < planes.kp; ++k) {
            double cc[FUSED_SCMAX], ss[FUSED_SCMAX];
#pragma unroll
            for (int q = 0; q < FUSED_SCMAX; ++q) {
                cc[q] = SC ? planes.cs[k][q] : 0.0;
                ss[q] = SC ? planes.sn[k][q] : 0.0;
            }
            PadLoadT<SC, LROW> ld{LROW ? lrow : dcT + ro, g, cc, ss, y, do_w, planes.w[k], planes.nsc};
            double re[S::E], im[S::E];
            int t;
            rf_row_compute<S>(tw, ld, false, rf_lds, t, re, im);
            rf_opaque(t);
            __builtin_amdgcn_sched_barrier(0);
            if (g.tpitch > 0) {  // Bt[u][y]: the first-axis transform of row u then reads contiguously (k_rowfft_b2a<S, true>)
                double2 *bcol = B + size_t(k) * bstride + size_t(y);
#pragma unroll
                for (int e = 0; e < S::E; ++e)
                    if ((omask >> e) & 1u) bcol[size_t(S::out_pos(t, e)) * size_t(g.tpitch)] = make_double2(re[e], im[e]);
            } else {
                double2 *brow = B + size_t(k) * bstride + size_t(y) * size_t(g.bpitch);
#pragma unroll
                for (int e = 0; e < S::E; ++e)
                    if ((omask >> e) & 1u) brow[S::out_pos(t, e)] = make_double2(re[e], im[e]);
            }
        }
    };
    if (lds_row) planes_loop(std::true_type{});
    else planes_loop(std::false_type{});
}

// Single-plane launches, PERSISTENT form (round 4: the one-plane w-scheme).  Knock-outs on k_fused_pad_fft
// at C2 with one plane (0.93 ms): without the pre-pass loads 0.69, without the stores 0.70, the transform alone 0.49 -- one
// workgroup per CU serialises its load phase (128 KB per row at the ~10 B per clock one CU pulls), its passes and its store
// phase.  Here a workgroup walks its rows; the NEXT row's x, corr [, beam] are requested in S::NP batches at the hook points of
// the transform (rowfft.hpp: RfNoHook) and banked into the LDS image row -- thread-private cells the first pass of THIS row has
// already read -- one pass later, so the requests fly under the passes; the row's scattered stores are issued last and drain
// under the next row's passes.
template <class S, bool BEAM>
struct PadPrefetch {
    static constexpr int NB = S::NP;                              // batches = hook points - 1
    static constexpr int BS = (S::NSLOT + NB - 1) / NB;           // slots per batch
    const FusedGeom &g;
    const double *xr, *cr, *br;  // the image row after the one being transformed
    double *lrow;
    int t;
    double vx[BS], vc[BS], vb[BEAM ? BS : 1];
    template <int B>
    __device__ __forceinline__ void issue(const double *x, const double *c, const double *bm)
    {
        rf_for_each_load<S>(t, [&](int u, int slot) {
            if (slot / BS == B) {
                const int ixc = max(fg_ix(g, u), 0);
                vx[slot % BS] = x[ixc];
                vc[slot % BS] = c[ixc];
                if constexpr (BEAM) vb[slot % BS] = bm[ixc];
            }
        });
    }
    template <int B>
    __device__ __forceinline__ void bank()
    {
        rf_for_each_load<S>(t, [&](int u, int slot) {
            if (slot / BS == B) {
                const int ix = fg_ix(g, u);
                double v = vx[slot % BS] * vc[slot % BS];  // (x * corr) * beam: the rounding order of k_prepare_img
                if constexpr (BEAM) v *= vb[slot % BS];
                if (ix >= 0) lrow[ix] = v;
            }
        });
    }
    // point P: bank batch P - 1 (requested one pass earlier), request batch P.  (Requesting the first batch in front of the
    // previous row's stores instead, so that it does not queue behind them, was measured and is slower: 0.97 vs 0.78 ms.)
    template <int P>
    __device__ __forceinline__ void at()
    {
        if constexpr (P >= 1 && P <= NB) bank<P - 1>();
        if constexpr (P < NB) issue<P>(xr, cr, br);
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <class S, bool SC, bool BEAM>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_fused_pad_fft_p(const double2 *tw, FusedGeom g, const uint8_t *occ,
                                                           FusedPrep prep, FusedPlanes planes, int do_w, double2 *B)
{
    extern __shared__ double rf_lds[];
    double *lrow = rf_lds + S::LDS_BYTES / sizeof(double);
    auto row_of = [&](int blk) {  // (the XCD grouping of the transposed stores: see k_fused_pad_fft; gridDim.x is a multiple of 64)
        if (g.tpitch > 0 && blk < (g.ny & ~63)) {
            const int r = blk & 63;
            return (blk & ~63) + (r & 7) * 8 + (r >> 3);
        }
        return blk;
    };
    uint32_t omask = 0;
#pragma unroll
    for (int e = 0; e < S::E; ++e) omask |= (occ[S::out_pos(int(threadIdx.x), e) >> 5] ? 1u : 0u) << e;
    double cc[FUSED_SCMAX], ss[FUSED_SCMAX];
#pragma unroll
    for (int q = 0; q < FUSED_SCMAX; ++q) {
        cc[q] = SC ? planes.cs[0][q] : 0.0;
        ss[q] = SC ? planes.sn[0][q] : 0.0;
    }
    int blk = blockIdx.x;
    if (blk >= g.ny) return;
    const int G = int(gridDim.x);
    auto rowptr = [&](const double *base, int bk) {  // row of block bk (past the end: this workgroup's first row again, read and dropped)
        const int yy = row_of(bk < g.ny ? bk : int(blockIdx.x));
        return base + size_t(yy) * size_t(g.nx);
    };
    const double *bsrc = BEAM ? prep.beam : prep.corr;
    PadPrefetch<S, BEAM> hook{g, nullptr, nullptr, nullptr, lrow, int(threadIdx.x), {}, {}, {}};
    {   // the first row's image values: every batch in turn (pre-pass)
        const double *x0 = rowptr(prep.x, blk), *c0 = rowptr(prep.corr, blk), *b0 = rowptr(bsrc, blk);
        hook.template issue<0>(x0, c0, b0);
        hook.template bank<0>();
        if constexpr (S::NP > 1) {
            hook.template issue<1>(x0, c0, b0);
            hook.template bank<1>();
        }
        if constexpr (S::NP > 2) {
            hook.template issue<2>(x0, c0, b0);
            hook.template bank<2>();
        }
        if constexpr (S::NP > 3) {
            hook.template issue<3>(x0, c0, b0);
            hook.template bank<3>();
        }
        static_assert(S::NP <= 4, "PadPrefetch: at most four batches");
    }
    for (; blk < g.ny; blk += G) {
        const int y = row_of(blk);
        hook.xr = rowptr(prep.x, blk + G);
        hook.cr = rowptr(prep.corr, blk + G);
        hook.br = rowptr(bsrc, blk + G);
        PadLoadT<SC, true> ld{lrow, g, cc, ss, y, do_w, planes.w[0], planes.nsc};
        double re[S::E], im[S::E];
        int t;
        rf_row_compute<S>(tw, ld, false, rf_lds, t, re, im, hook);
        rf_opaque(t);
        __builtin_amdgcn_sched_barrier(0);
        if (g.tpitch > 0) {
            double2 *bcol = B + size_t(y);
#pragma unroll
            for (int e = 0; e < S::E; ++e)
                if ((omask >> e) & 1u) bcol[size_t(S::out_pos(t, e)) * size_t(g.tpitch)] = make_double2(re[e], im[e]);
        } else {
            double2 *brow = B + size_t(y) * size_t(g.bpitch);
#pragma unroll
            for (int e = 0; e < S::E; ++e)
                if ((omask >> e) & 1u) brow[S::out_pos(t, e)] = make_double2(re[e], im[e]);
        }
    }
}

void fused_geom_fit(FusedGeom &g)
{
    g.npoly = 0;
    // largest r2 over the image (a corner)
    long double zmax = 0.0L;
    for (int cx = 0; cx < 2; ++cx)
        for (int cy = 0; cy < 2; ++cy) {
            long double l = (long double)g.lshift + (long double)((cx ? g.nx - 1 : 0) - g.nx / 2) * (long double)g.px;
            long double m = (long double)g.mshift + (long double)((cy ? g.ny - 1 : 0) - g.ny / 2) * (long double)g.py;
            zmax = std::max(zmax, l * l + m * m);
        }
    if (!(zmax > 0.0L) || zmax > 0.5L) return;
    auto f = [](long double z) { return -z / (1.0L + sqrtl(1.0L - z)); };  // sqrt(1 - z) - 1
    // Chebyshev coefficients of f on [0, zmax] (s = 2 z / zmax - 1)
    const int M = 64;
    const long double pi = 3.141592653589793238462643383279502884L;
    std::vector<long double> fv(M), c(M, 0.0L);
    for (int j = 0; j < M; ++j) fv[size_t(j)] = f(0.5L * zmax * (cosl(pi * (j + 0.5L) / M) + 1.0L));
    for (int k = 0; k < M; ++k) {
        long double acc = 0.0L;
        for (int j = 0; j < M; ++j) acc += fv[size_t(j)] * cosl(pi * k * (j + 0.5L) / M);
        c[size_t(k)] = acc * (k == 0 ? 1.0L : 2.0L) / M;
    }
    const long double fmax = fabsl(f(zmax));
    int deg = -1;
    for (int d = 1; d <= FUSED_MAXPOLY; ++d) {
        long double tail = 0.0L;
        for (int k = d + 1; k < M / 2; ++k) tail += fabsl(c[size_t(k)]);
        if (tail <= 1e-17L * fmax) { deg = d; break; }
    }
    if (deg < 0) return;
    // Chebyshev -> monomial in s:  T_0 = 1, T_1 = s, T_{k+1} = 2 s T_k - T_{k-1}
    std::vector<long double> mono(size_t(deg + 1), 0.0L), t0(size_t(deg + 1), 0.0L), t1(size_t(deg + 1), 0.0L);
    t0[0] = 1.0L;
    t1[1] = 1.0L;
    for (int k = 0; k <= deg; ++k) {
        const std::vector<long double> &tk = k == 0 ? t0 : t1;
        for (int i = 0; i <= deg; ++i) mono[size_t(i)] += c[size_t(k)] * tk[size_t(i)];
        if (k >= 1) {
            std::vector<long double> t2(size_t(deg + 1), 0.0L);
            for (int i = 0; i < deg; ++i) t2[size_t(i + 1)] = 2.0L * t1[size_t(i)];
            for (int i = 0; i <= deg; ++i) t2[size_t(i)] -= t0[size_t(i)];
            t0 = t1;
            t1 = t2;
        }
    }
    FusedGeom trial = g;
    trial.npoly = deg + 1;
    trial.za = double(2.0L / zmax);
    trial.zb = -1.0;
    for (int i = 0; i <= FUSED_MAXPOLY; ++i) trial.pc[i] = 0.0;
    for (int i = 0; i <= deg; ++i) trial.pc[FUSED_MAXPOLY - deg + i] = double(mono[size_t(deg - i)]);  // right-aligned
    // verify in double arithmetic, as the kernel evaluates it
    long double worst = 0.0L;
    const int NS = 4097;
    for (int j = 0; j < NS; ++j) {
        const double z = double(zmax * j / (NS - 1));
        const double sv = z * trial.za + trial.zb;
        double acc = trial.pc[FUSED_MAXPOLY + 1 - trial.npoly];
        for (int k = FUSED_MAXPOLY + 2 - trial.npoly; k <= FUSED_MAXPOLY; ++k) acc = acc * sv + trial.pc[k];
        worst = std::max(worst, fabsl((long double)acc - f((long double)z)));
    }
    if (worst <= 4e-16L * fmax) g = trial;
}

void fused_planes_fit(const FusedGeom &g, FusedPlanes &pl, bool residual)
{
    pl.nsc = 0;
    pl.sep = 0;
    if (g.npoly <= 0 || pl.kp <= 0) return;
    const long double pi = 3.141592653589793238462643383279502884L;
    const long double zmax = 2.0L / (long double)g.za;  // s = r2 * za - 1, r2 in [0, zmax]
    // the whole n - 1 + nshift, or what is left of n - 1 behind its linear term: R(z) = sqrt(1 - z) - 1 + z / 2
    auto tfun = [&](long double z) {
        const long double sq = 1.0L + sqrtl(1.0L - z);
        return residual ? -z * z / (2.0L * sq * sq) : -z / sq + (long double)g.nshift;
    };
    const int M = 64;
    int need = 0;
    std::vector<std::vector<long double>> mono(size_t(2 * pl.kp));
    for (int k = 0; k < pl.kp; ++k)
        for (int part = 0; part < 2; ++part) {
            std::vector<long double> fv(M), c(M, 0.0L);
            for (int j = 0; j < M; ++j) {
                const long double z = 0.5L * zmax * (cosl(pi * (j + 0.5L) / M) + 1.0L);
                long double ph = (long double)pl.w[k] * tfun(z);
                ph -= rintl(ph);
                fv[size_t(j)] = part == 0 ? cosl(2.0L * pi * ph) : sinl(2.0L * pi * ph);
            }
            for (int q = 0; q < M; ++q) {
                long double acc = 0.0L;
                for (int j = 0; j < M; ++j) acc += fv[size_t(j)] * cosl(pi * q * (j + 0.5L) / M);
                c[size_t(q)] = acc * (q == 0 ? 1.0L : 2.0L) / M;
            }
            int deg = -1;
            for (int d = 0; d < FUSED_SCMAX; ++d) {
                long double tail = 0.0L;
                for (int q = d + 1; q < M / 2; ++q) tail += fabsl(c[size_t(q)]);
                // (1e-14 of a unit-modulus screen: three orders below the tightest epsilon a plan accepts, ~1e-11; 2e-17 -- the
                // rounding level -- cost C2 two to three more coefficients per chain)
                if (tail <= 1e-14L) { deg = d; break; }
            }
            if (deg < 0) return;  // phase too large for a short polynomial: general path
            need = std::max(need, deg + 1);
            // Chebyshev -> monomial in s
            std::vector<long double> mo(size_t(deg + 1), 0.0L), t0(size_t(deg + 1), 0.0L), t1(size_t(deg + 1), 0.0L);
            t0[0] = 1.0L;
            if (deg >= 1) t1[1] = 1.0L;
            for (int q = 0; q <= deg; ++q) {
                const std::vector<long double> &tq = q == 0 ? t0 : t1;
                for (int i = 0; i <= deg; ++i) mo[size_t(i)] += c[size_t(q)] * tq[size_t(i)];
                if (q >= 1) {
                    std::vector<long double> t2(size_t(deg + 1), 0.0L);
                    for (int i = 0; i < deg; ++i) t2[size_t(i + 1)] = 2.0L * t1[size_t(i)];
                    for (int i = 0; i <= deg; ++i) t2[size_t(i)] -= t0[size_t(i)];
                    t0 = t1;
                    t1 = t2;
                }
            }
            mono[size_t(2 * k + part)] = mo;
        }
    if (need > FUSED_SCMAX) return;
    const int neval = need <= 4 ? 4 : (need <= 6 ? 6 : FUSED_SCMAX);  // what the kernels evaluate (fg_screen_poly)
    need = FUSED_SCMAX;  // the arrays hold FUSED_SCMAX coefficients, highest power first, padded with leading zeros
    FusedPlanes trial = pl;
    trial.nsc = neval;
    trial.sep = residual ? 1 : 0;
    for (int k = 0; k < pl.kp; ++k)
        for (int part = 0; part < 2; ++part) {
            const auto &mo = mono[size_t(2 * k + part)];
            double *dst = part == 0 ? trial.cs[k] : trial.sn[k];
            for (int i = 0; i < need; ++i) {  // highest power first, padded with leading zeros
                const int power = need - 1 - i;
                dst[i] = power < int(mo.size()) ? double(mo[size_t(power)]) : 0.0;
            }
        }
    // verify in double arithmetic, as the kernel evaluates it
    long double worst = 0.0L;
    const int NS = 2049;
    for (int k = 0; k < pl.kp; ++k)
        for (int j = 0; j < NS; ++j) {
            const double z = double(zmax * j / (NS - 1));
            const double sv = z * g.za + g.zb;
            double cc = trial.cs[k][0], ss = trial.sn[k][0];
            for (int i = 1; i < need; ++i) {
                cc = cc * sv + trial.cs[k][i];
                ss = ss * sv + trial.sn[k][i];
            }
            long double ph = (long double)pl.w[k] * tfun((long double)z);
            worst = std::max(worst, fabsl((long double)cc - cosl(2.0L * pi * ph)));
            worst = std::max(worst, fabsl((long double)ss - sinl(2.0L * pi * ph)));
        }
    if (worst <= 1e-13L) pl = trial;
}

__global__ void k_screen_table(FusedGeom g, const double *w, int nplanes, double2 *tau)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
    if (ix >= g.nx || p >= nplanes) return;
    const double l = g.lshift + double(ix - g.nx / 2) * g.px;
    double ph = w[p] * (-0.5 * (l * l));
    ph -= rint(ph);
    double sn, cs;
    fg_sincos2pi(ph, sn, cs);
    tau[size_t(p) * size_t(g.nx) + size_t(ix)] = make_double2(cs, sn);
}

void fused_screen_table(const FusedGeom &g, const double *w_dev, int nplanes, double2 *tau_dev, hipStream_t stream)
{
    if (nplanes <= 0) return;
    hipLaunchKernelGGL(k_screen_table, dim3(uint32_t((g.nx + 255) / 256), uint32_t(nplanes)), dim3(256), 0, stream, g, w_dev, nplanes,
                       tau_dev);
    PFB_HIP(hipGetLastError());
}

// ---- doubled shapes (N = 2 N1): dedicated fused kernels ---------------------------------------------------
// The generic kernels above on RfShape2 materialise all 2 x 16 outputs per thread (128 VGPRs) next to the
// waiting half transform and spill ~250 registers.  Here the even / odd half transforms are combined pair by
// pair, X[k] = E[k] + w^k O[k], X[k + N1] = E[k] - w^k O[k], and every pair is consumed at once.  The image row
// does not fit LDS at these sizes: read-modify-write form.
struct OccLoad2 {
    const double2 *row;
    uint32_t mask;
    int par;
    __device__ __forceinline__ bool on(int slot) const { return ((mask >> slot) & 1u) != 0; }
    __device__ __forceinline__ double2 fetch(int pos, int slot) const { return row[on(slot) ? 2 * pos + par : 0]; }
    __device__ __forceinline__ double2 finish(double2 v, int, int slot) const { return on(slot) ? v : make_double2(0.0, 0.0); }
    __device__ __forceinline__ double2 operator()(int pos, int slot) const { return finish(fetch(pos, slot), pos, slot); }
};

// STASH (round 3; shapes with 2 N1 doubles of LDS next to each other: 20480 points): the half transform that waits is parked in
// LDS, not in registers -- er behind the exchange buffer while the odd half is transformed, ei in the (then idle) exchange
// buffer during the epilogue; thread-private cells, one barrier per plane.  With both halves in registers next to a running
// transform the kernel spilled ~100 VGPRs and lost to the unfused path (36.8 vs 35.6 ms per 4 planes at C5's size).  The
// epilogue requests the running sum / correction / beam / x of FOUR outputs at a time from clamped addresses, one group ahead
// (see k_fused_fft_crop), and stores after the last group.
// MODE 0: n - 1 polynomial + sincos per pixel and plane; 2: separable screen (FusedPlanes::sep: table + row factor + residual)
template <class S1, bool STASH, int MODE>
__global__ void __launch_bounds__(S1::T, ((S1::T + 63) / 64 + 3) / 4)
    k_fused_fft_crop2(const double2 *tw, FusedGeom g, const uint8_t *occ, const double2 *B, size_t bstride, FusedPlanes planes,
                      int do_w, int first, double *accT, FusedFinal fin)
{
    extern __shared__ double rf_lds[];
    double *stash = rf_lds + S1::LDS_BYTES / sizeof(double);  // STASH: S1::N doubles behind the exchange buffer
    const int y = blockIdx.x;
    double *arow = accT + size_t(y) * g.nx;
    // samples 2 pos and 2 pos + 1 lie in the same 32-column block: one occupancy mask serves both halves
    uint32_t mask = 0;
    rf_for_each_load<S1>(int(threadIdx.x), [&](int pos, int slot) { mask |= (occ[(2 * pos) >> 5] ? 1u : 0u) << slot; });
    for (int k = 0; k < planes.kp; ++k) {
        const double2 *row = B + size_t(k) * bstride + size_t(y) * size_t(g.bpitch);
        OccLoad2 ld_e{row, mask, 0}, ld_o{row, mask, 1};
        double er[S1::E], ei[S1::E], orr[S1::E], oi[S1::E];
        int t;
        rf_row_compute<S1>(tw, ld_e, true, rf_lds, t, er, ei);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NST = STASH ? rf_stash_slots<S1>() : 0;
#pragma unroll
        for (int e = 0; e < NST; ++e) stash[e * S1::T + t] = er[e];
        rf_row_compute<S1, decltype(ld_o), 2>(tw, ld_o, true, rf_lds, t, orr, oi);
        __builtin_amdgcn_sched_barrier(0);
        rf_opaque(t);
        if constexpr (STASH) {  // (behind the transform's last barrier nobody reads the exchange buffer any more)
#pragma unroll
            for (int e = 0; e < S1::E; ++e) rf_lds[e * S1::T + t] = ei[e];
        }
        const double wk = planes.w[k];
        double cc[FUSED_SCMAX], ss[FUSED_SCMAX];  // MODE 2: the plane's residual polynomials, its row factor and its table
        double rs = 0.0, rc = 1.0;
        const double2 *trow = nullptr;
        if constexpr (MODE == 2) {
#pragma unroll
            for (int q = 0; q < FUSED_SCMAX; ++q) {
                cc[q] = planes.cs[k][q];
                ss[q] = planes.sn[k][q];
            }
            fg_row_factor(g, wk, y, rs, rc);
            trow = planes.tau + size_t(k) * size_t(g.nx);
        }
        const bool last = k == planes.kp - 1;
        const bool add_img = !(first && k == 0);
        const bool finalize = last && fin.corr != nullptr;
        const size_t ro = size_t(y) * size_t(g.nx);
        // The epilogue exists twice: FIN = false (all planes but the band's last) requests only the running sum, FIN = true also
        // the correction, beam and x.  One body for both kept 12 more request registers live on every plane, and the spill
        // reloads they caused (scratch loads wait on vmcnt(0)) drained the next group's requests each time.
        const double2 *__restrict__ tw2 = tw + S1::N;
        auto epilogue = [&](auto fin_tag) {
            constexpr bool FIN = decltype(fin_tag)::value;
            const bool has_beam = FIN && fin.beam != nullptr, has_x = FIN && fin.x != nullptr;
            const double *crow = FIN ? fin.corr + ro : arow;
            const double *brow = has_beam ? fin.beam + ro : crow;
            const double *xrow = has_x ? fin.x + ro : crow;
            // group = one slot e = two outputs (k1 and k1 + N1)
            constexpr int NG = S1::E;
            double qi[2][2], qc[2][2], qb[2][2], qx[2][2];
            double2 qt[2][2], qw[2];  // (the combination twiddle travels with the requests: loaded inside its group it was the
                                      // youngest load in flight, and waiting for it -- vmcnt(0) -- drained the next group's requests)
            auto request = [&](int e0, int buf) {
                qw[buf] = tw2[S1::out_pos(t, e0)];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ixc = max(fg_ix(g, S1::out_pos(t, e0) + j * S1::N), 0);
                    qi[buf][j] = arow[ixc];
                    if constexpr (FIN) {
                        qc[buf][j] = crow[ixc];
                        qb[buf][j] = brow[ixc];
                        qx[buf][j] = xrow[ixc];
                    }
                    if constexpr (MODE == 2) qt[buf][j] = trow[ixc];
                }
            };
            request(0, 0);
            // results wait in the LDS cells their slot's parked values leave behind (STASH), else in registers, and are stored
            // after the last group
            double res[STASH ? 2 * (S1::E - NST) + 1 : 2 * S1::E];
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                if (gq + 1 < NG) request(gq + 1, (gq + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int e = gq;
                    const int k1 = S1::out_pos(t, e);
                    const double2 w = qw[gq & 1];
                    const double ere = e < NST ? stash[e * S1::T + t] : er[e], eim = STASH ? rf_lds[e * S1::T + t] : ei[e];
                    const double tr = orr[e] * w.x - oi[e] * w.y, ti = orr[e] * w.y + oi[e] * w.x;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const double vre = h ? ere - tr : ere + tr, vim = h ? eim - ti : eim + ti;
                        const int ix = fg_ix(g, k1 + h * S1::N);
                        const int j = h;
                        double v = 0.0;
                        if (ix >= 0) {
                            double r = vim;  // inverse transform: value = (im, re)
                            if (do_w) {
                                double sn, cs;
                                if constexpr (MODE == 2) {
                                    fg_screen_sep(g, cc, ss, ix, y, qt[gq & 1][j], rs, rc, sn, cs, planes.nsc);
                                } else {
                                    double ph = wk * fg_t_poly(g, ix, y);
                                    ph -= rint(ph);
                                    fg_sincos2pi(ph, sn, cs);
                                }
                                r = vim * cs + vre * sn;
                            }
                            if (add_img) r += qi[gq & 1][j];
                            if constexpr (FIN) {
                                const double c = has_beam ? qc[gq & 1][j] * qb[gq & 1][j] : qc[gq & 1][j];
                                v = r * c * fin.scale;
                                if (has_x) v += fin.eta * qx[gq & 1][j];
                            } else {
                                v = r;
                            }
                        }
                        if constexpr (STASH) {
                            if (h == 1) rf_lds[e * S1::T + t] = v;
                            else if (e < NST) stash[e * S1::T + t] = v;
                            else res[2 * (e - NST)] = v;
                        } else {
                            res[2 * e + h] = v;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            double *orow = FIN ? fin.out + ro : arow;
#pragma unroll
            for (int e = 0; e < S1::E; ++e) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ix = fg_ix(g, S1::out_pos(t, e) + h * S1::N);
                    double v;
                    if constexpr (STASH) v = h == 1 ? rf_lds[e * S1::T + t] : (e < NST ? stash[e * S1::T + t] : res[2 * (e - NST)]);
                    else v = res[2 * e + h];
                    if (ix >= 0) orow[ix] = v;
                }
            }
        };
        if (finalize) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if constexpr (STASH) rf_barrier();  // the next plane's transform writes the exchange buffer (the parked ei of slow waves)
    }
}

struct PadLoad2 {
    PadLoad base;
    int par;
    __device__ __forceinline__ double fetch(int pos, int slot) const { return base.fetch(2 * pos + par, slot); }
    __device__ __forceinline__ double2 finish(double v, int pos, int slot) const { return base.finish(v, 2 * pos + par, slot); }
    __device__ __forceinline__ double2 operator()(int pos, int slot) const { return base(2 * pos + par, slot); }
};

template <class S1, bool STASH>
__global__ void __launch_bounds__(S1::T, ((S1::T + 63) / 64 + 3) / 4)
    k_fused_pad_fft2(const double2 *tw, FusedGeom g, const uint8_t *occ, const double *dcT, FusedPrep prep, FusedPlanes planes,
                     int do_w, double2 *B, size_t bstride)
{
    extern __shared__ double rf_lds[];
    double *stash = rf_lds + S1::LDS_BYTES / sizeof(double);  // STASH: the even half's real parts wait here (see k_fused_fft_crop2)
    int y = blockIdx.x;
    if (g.tpitch > 0 && y < (g.ny & ~63)) {  // transposed stores: the 8 rows of a 128-byte line of Bt on one XCD (see k_fused_pad_fft)
        const int r = y & 63;
        y = (y & ~63) + (r & 7) * 8 + (r >> 3);
    }
    uint32_t omask = 0;  // bit e: column out_pos(e) occupied, bit 16 + e: column N1 + out_pos(e)
#pragma unroll
    for (int e = 0; e < S1::E; ++e) {
        const int k1 = S1::out_pos(int(threadIdx.x), e);
        omask |= (occ[k1 >> 5] ? 1u : 0u) << e;
        omask |= (occ[(k1 + S1::N) >> 5] ? 1u : 0u) << (16 + e);
    }
    const size_t ro = size_t(y) * size_t(g.nx);
    const double zc[FUSED_SCMAX] = {};  // (the doubled shapes evaluate the screen the general way)
    for (int k = 0; k < planes.kp; ++k) {
        PadLoad base{dcT + ro, g, zc, zc, y, do_w, planes.w[k], 0};  // (the doubled shapes read a PREPARED image: fused_pad_takes_prep)
        PadLoad2 ld_e{base, 0}, ld_o{base, 1};
        double er[S1::E], ei[S1::E], orr[S1::E], oi[S1::E];
        int t;
        rf_row_compute<S1>(tw, ld_e, false, rf_lds, t, er, ei);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NST = STASH ? rf_stash_slots<S1>() : 0;
#pragma unroll
        for (int e = 0; e < NST; ++e) stash[e * S1::T + t] = er[e];
        rf_row_compute<S1, decltype(ld_o), 2>(tw, ld_o, false, rf_lds, t, orr, oi);
        __builtin_amdgcn_sched_barrier(0);
        rf_opaque(t);
        // B[y][u], or Bt[u][y] for the transposing first axis (g.tpitch > 0)
        double2 *bbase = B + size_t(k) * bstride + (g.tpitch > 0 ? size_t(y) : size_t(y) * size_t(g.bpitch));
        const size_t ustep = g.tpitch > 0 ? size_t(g.tpitch) : size_t(1);
        const double2 *__restrict__ tw2 = tw + S1::N;
#pragma unroll
        for (int e = 0; e < S1::E; ++e) {
            const int k1 = S1::out_pos(t, e);
            const double2 w = tw2[k1];
            const double ere = e < NST ? stash[e * S1::T + t] : er[e];
            const double tr = orr[e] * w.x - oi[e] * w.y, ti = orr[e] * w.y + oi[e] * w.x;
            if ((omask >> e) & 1u) bbase[size_t(k1) * ustep] = make_double2(ere + tr, ei[e] + ti);
            if ((omask >> (16 + e)) & 1u) bbase[size_t(k1 + S1::N) * ustep] = make_double2(ere - tr, ei[e] - ti);
            if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (at most four combination twiddles in flight)
        }
    }
}

template <class S1>
static void launch_crop2(const RowFFTPlan &pl, const FusedGeom &g, const uint8_t *occ_dev, const double2 *B_dev, size_t bstride,
                         const FusedPlanes &planes, int do_w, bool first, double *accT_dev, const FusedFinal &fin,
                         hipStream_t stream)
{
    static bool attr = false, attr_sep = false;
    constexpr bool STASH = rf_stash_slots<S1>() > 0;
    const size_t lds = size_t(S1::LDS_BYTES) + size_t(rf_stash_slots<S1>()) * S1::T * sizeof(double);
    if (planes.sep && planes.nsc > 0 && do_w) {
        rf_allow_lds(&k_fused_fft_crop2<S1, STASH, 2>, &attr_sep);
        hipLaunchKernelGGL((k_fused_fft_crop2<S1, STASH, 2>), dim3(uint32_t(g.ny)), dim3(S1::T), lds, stream, pl.twiddle, g, occ_dev,
                           B_dev, bstride, planes, do_w, first ? 1 : 0, accT_dev, fin);
        return;
    }
    rf_allow_lds(&k_fused_fft_crop2<S1, STASH, 0>, &attr);
    hipLaunchKernelGGL((k_fused_fft_crop2<S1, STASH, 0>), dim3(uint32_t(g.ny)), dim3(S1::T), lds, stream, pl.twiddle, g, occ_dev,
                       B_dev, bstride, planes, do_w, first ? 1 : 0, accT_dev, fin);
}
template <class S1>
static void launch_pad2(const RowFFTPlan &pl, const FusedGeom &g, const uint8_t *occ_dev, const double *dcT_dev,
                        const FusedPrep &prep, const FusedPlanes &planes, int do_w, double2 *B_dev, size_t bstride,
                        hipStream_t stream)
{
    static bool attr = false;
    PFB_REQUIRE(prep.x == nullptr, "fused pad kernel of the doubled shapes needs a prepared image (fused_pad_takes_prep)");
    constexpr bool STASH = rf_stash_slots<S1>() > 0;
    rf_allow_lds(&k_fused_pad_fft2<S1, STASH>, &attr);
    hipLaunchKernelGGL((k_fused_pad_fft2<S1, STASH>), dim3(uint32_t(g.ny)), dim3(S1::T),
                       size_t(S1::LDS_BYTES) + size_t(rf_stash_slots<S1>()) * S1::T * sizeof(double), stream,
                       pl.twiddle, g, occ_dev, dcT_dev, prep, planes, do_w, B_dev, bstride);
}

// The fused kernels transpose one component at a time (N doubles of LDS), which leaves room for the
// workgroup's image row (nx doubles) when (N + nx) * 8 <= 160 KiB.
static bool fused_row_fits(int lds_bytes, int nx)
{
    // PFBHIP_FUSED_LDSROW=0 forces the read-modify-write form (what large grids use) so that tests reach it
    const char *env = std::getenv("PFBHIP_FUSED_LDSROW");
    if (env != nullptr && env[0] == '0') return false;
    return size_t(lds_bytes) + size_t(nx) * sizeof(double) <= 160 * 1024;
}

template <class S>
static void launch_crop(const RowFFTPlan &pl, const FusedGeom &g, const uint8_t *occ_dev, const double2 *B_dev,
                        size_t bstride, const FusedPlanes &planes, int do_w, bool first, double *accT_dev,
                        const FusedFinal &fin, hipStream_t stream)
{
    static bool attr = false, attr_sc = false;
    const bool row = fused_row_fits(S::LDS_BYTES, g.nx) && planes.kp > 1;
    const size_t lds = size_t(S::LDS_BYTES) + (row ? size_t(g.nx) * sizeof(double) : 0);
    if (planes.nsc > 0 && !planes.sep && do_w) {
        rf_allow_lds(&k_fused_fft_crop<S, true>, &attr_sc);
        hipLaunchKernelGGL((k_fused_fft_crop<S, true>), dim3(uint32_t(g.ny)), dim3(S::T), lds, stream, pl.twiddle, g, occ_dev, B_dev,
                           bstride, planes, do_w, first ? 1 : 0, row ? 1 : 0, accT_dev, fin);
        return;
    }
    rf_allow_lds(&k_fused_fft_crop<S, false>, &attr);
    hipLaunchKernelGGL((k_fused_fft_crop<S, false>), dim3(uint32_t(g.ny)), dim3(S::T), lds, stream, pl.twiddle, g, occ_dev, B_dev,
                       bstride, planes, do_w, first ? 1 : 0, row ? 1 : 0, accT_dev, fin);
}

template <class S>
static void launch_pad(const RowFFTPlan &pl, const FusedGeom &g, const uint8_t *occ_dev, const double *dcT_dev,
                       const FusedPrep &prep, const FusedPlanes &planes, int do_w, double2 *B_dev, size_t bstride,
                       hipStream_t stream)
{
    static bool attr = false, attr_sc = false;
    const bool row = fused_row_fits(S::LDS_BYTES, g.nx);
    PFB_REQUIRE(row || prep.x == nullptr, "fused pad kernel without an LDS row needs a prepared image (fused_pad_takes_prep)");
    const size_t lds = size_t(S::LDS_BYTES) + (row ? size_t(g.nx) * sizeof(double) : 0);
    {   // single-plane launches on the raw image: the persistent form (PFBHIP_PAD_PERSIST=0 keeps one workgroup per row)
        static const bool persist = [] { const char *e = std::getenv("PFBHIP_PAD_PERSIST"); return !(e != nullptr && e[0] == '0'); }();
        if (persist && row && planes.kp == 1 && prep.x != nullptr) {
            int dev = 0, ncu = 256;
            PFB_HIP(hipGetDevice(&dev));
            PFB_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
            // one workgroup per CU, a multiple of 64 of them (the row -> block map groups blocks in 64s)
            const uint32_t nwg = uint32_t(std::min<int>(g.ny, std::max(64, (ncu / 64) * 64)));
            const bool sc = planes.nsc > 0 && !planes.sep && do_w, beam = prep.beam != nullptr;
            static bool ap[4] = {false, false, false, false};
#define PFB_PADP(SCV, BV)                                                                                                        \
    do {                                                                                                                         \
        rf_allow_lds(&k_fused_pad_fft_p<S, SCV, BV>, &ap[(SCV ? 2 : 0) + (BV ? 1 : 0)]);                                          \
        hipLaunchKernelGGL((k_fused_pad_fft_p<S, SCV, BV>), dim3(nwg), dim3(S::T), lds, stream, pl.twiddle, g, occ_dev, prep, planes, \
                           do_w, B_dev);                                                                                         \
    } while (0)
            if (sc && beam) PFB_PADP(true, true);
            else if (sc) PFB_PADP(true, false);
            else if (beam) PFB_PADP(false, true);
            else PFB_PADP(false, false);
#undef PFB_PADP
            return;
        }
    }
    if (planes.nsc > 0 && !planes.sep && do_w) {
        rf_allow_lds(&k_fused_pad_fft<S, true>, &attr_sc);
        hipLaunchKernelGGL((k_fused_pad_fft<S, true>), dim3(uint32_t(g.ny)), dim3(S::T), lds, stream, pl.twiddle, g, occ_dev, dcT_dev,
                           prep, planes, do_w, row ? 1 : 0, B_dev, bstride);
        return;
    }
    rf_allow_lds(&k_fused_pad_fft<S, false>, &attr);
    hipLaunchKernelGGL((k_fused_pad_fft<S, false>), dim3(uint32_t(g.ny)), dim3(S::T), lds, stream, pl.twiddle, g, occ_dev, dcT_dev,
                       prep, planes, do_w, row ? 1 : 0, B_dev, bstride);
}

void fused_fft_crop(const RowFFT &f, const FusedGeom &g, const uint8_t *occ_dev, const double2 *B_dev, size_t bstride,
                    const FusedPlanes &planes, int do_w, bool first, double *accT_dev, const FusedFinal &fin, hipStream_t stream)
{
    switch (f.pl.N) {
#define RF_X(L, K)                                                                                                   \
    case (L << K):                                                                                                   \
        launch_crop<RfShape<L, K, false>>(f.pl, g, occ_dev, B_dev, bstride, planes, do_w, first, accT_dev, fin, stream); \
        break;
        RF_FOR_SHAPES(RF_X)
#undef RF_X
#define RF_X(L, K)                                                                                                        \
    case 2 * (L << K):                                                                                                    \
        launch_crop2<RfShape<L, K, false>>(f.pl, g, occ_dev, B_dev, bstride, planes, do_w, first, accT_dev, fin, stream);    \
        break;
        RF_FOR_SHAPES2(RF_X)
#undef RF_X
        default: PFB_REQUIRE(false, "row length %d is not supported by the hand-written FFT", f.pl.N);
    }
    PFB_HIP(hipGetLastError());
}

bool fused_doubled_stashes(const RowFFT &f)
{
    if (!f.ok || !f.pl.doubled) return false;
    switch (f.pl.N) {
#define RF_X(L, K) \
    case 2 * (L << K): return rf_stash_slots<RfShape<L, K, false>>() > 0;
        RF_FOR_SHAPES2(RF_X)
#undef RF_X
        default: return false;
    }
}

bool fused_pad_takes_prep(const RowFFT &f, const FusedGeom &g)
{
    if (!f.ok || f.pl.doubled) return false;
    switch (f.pl.N) {
#define RF_X(L, K) \
    case (L << K): return fused_row_fits(RfShape<L, K, false>::LDS_BYTES, g.nx);
        RF_FOR_SHAPES(RF_X)
#undef RF_X
        default: return false;
    }
}

void fused_pad_fft(const RowFFT &f, const FusedGeom &g, const uint8_t *occ_dev, const double *dcT_dev, const FusedPrep &prep,
                   const FusedPlanes &planes, int do_w, double2 *B_dev, size_t bstride, hipStream_t stream)
{
    switch (f.pl.N) {
#define RF_X(L, K)                                                                                           \
    case (L << K):                                                                                           \
        launch_pad<RfShape<L, K, false>>(f.pl, g, occ_dev, dcT_dev, prep, planes, do_w, B_dev, bstride, stream); \
        break;
        RF_FOR_SHAPES(RF_X)
#undef RF_X
#define RF_X(L, K)                                                                                                      \
    case 2 * (L << K):                                                                                                  \
        launch_pad2<RfShape<L, K, false>>(f.pl, g, occ_dev, dcT_dev, prep, planes, do_w, B_dev, bstride, stream);          \
        break;
        RF_FOR_SHAPES2(RF_X)
#undef RF_X
        default: PFB_REQUIRE(false, "row length %d is not supported by the hand-written FFT", f.pl.N);
    }
    PFB_HIP(hipGetLastError());
}

}  // namespace pfbhip

using namespace pfbhip;

// Diagnostic: the plain row transform with its global loads (MODE & 1) and / or stores (MODE & 2) taken out -- what is left
// of the row time tells how much of it is the passes themselves.  PFBHIP_RF_MODE selects it in pfbhip_debug_rowfft's timing.
template <int MODE>
struct DbgLoad {
    const double2 *row;
    __device__ __forceinline__ double2 operator()(int i, int) const
    {
        return (MODE & 1) ? make_double2(double(i), 1.0) : row[i];
    }
};
template <int MODE>
struct DbgStore {
    double2 *row;
    __device__ __forceinline__ void operator()(int i, double2 v) const
    {
        if (!(MODE & 2) || v.x == 1.2345e300) row[i] = v;
    }
};
template <class S, int MODE>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_rowfft_dbg(const double2 *tw, double2 *data, int nrows, size_t pitch)
{
    extern __shared__ double rf_lds[];
    const int row = blockIdx.x;
    if (row >= nrows) return;
    DbgLoad<MODE> ld{data + size_t(row) * pitch};
    DbgStore<MODE> st{data + size_t(row) * pitch};
    rf_row<S>(tw, ld, st, true, rf_lds);
}
template <class S, int MODE>
static void launch_dbg(const RowFFTPlan &pl, double2 *d, int nrows, size_t pitch)
{
    static bool attr = false;
    rf_allow_lds(&k_rowfft_dbg<S, MODE>, &attr);
    hipLaunchKernelGGL((k_rowfft_dbg<S, MODE>), dim3(uint32_t(nrows)), dim3(S::T), size_t(S::LDS_BYTES), nullptr, pl.twiddle, d,
                       nrows, pitch);
}

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
            const char *dm = std::getenv("PFBHIP_RF_MODE");
            if (dm != nullptr && n == 10240) {
                using S = RfShape<5, 11>;
                for (int r = 0; r < reps; ++r) switch (std::atoi(dm)) {
                        case 0: launch_dbg<S, 0>(pl, s.p, int(nrows), size_t(n)); break;
                        case 1: launch_dbg<S, 1>(pl, s.p, int(nrows), size_t(n)); break;
                        case 2: launch_dbg<S, 2>(pl, s.p, int(nrows), size_t(n)); break;
                        default: launch_dbg<S, 3>(pl, s.p, int(nrows), size_t(n)); break;
                    }
            } else {
                for (int r = 0; r < reps; ++r) rowfft_plain(pl, s.p, int(nrows), inverse != 0, nullptr);
            }
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
