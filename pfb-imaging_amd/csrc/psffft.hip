// psffft.hip -- PSF convolution (psf_convolve_* / hessian_psf_*, src/pfb_imaging/operators/psf.py:14-104,
// operators/hessian.py:92-140, 326-348) on the hand-written row FFT: see psffft_api.hpp for the pipeline.
// Compiled with FMA contraction ON (no bit-exact index arithmetic lives in this file).
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.hpp"
#include "psffft_api.hpp"
#include "rowfft.hpp"

namespace pfbhip {

// y axis (passes 1 and 3): every plain shape of the row FFT.  x axis (pass 2: forward, multiply, inverse on the same
// row): powers of two chain the two transforms in registers; lengths with a leading radix-3/5 pass hand the row
// over through LDS (both components: N * 16 bytes <= 160 KiB, i.e. N <= 10240).
#define PSF_FOR_SHAPES_Y(X) RF_FOR_SHAPES(X)
#define PSF_FOR_SHAPES_X(X)                                                                                     \
    X(1, 10) X(1, 11) X(1, 12) X(1, 13) X(1, 14) X(3, 9) X(3, 10) X(3, 11) X(5, 8) X(5, 9) X(5, 10) X(5, 11) X(7, 8) \
    X(7, 9) X(7, 10) X(9, 7) X(9, 8) X(9, 9) X(9, 10) X(15, 7) X(15, 8) X(15, 9)
// (32 complex per thread at 16384 points -- 512 threads, 256 VGPRs -- was tried against the 1024-thread /
// 128-VGPR layout and its 35-120 spilled registers in these fused kernels: it spills more.)
template <int L, int K>
using PsfShape = RfShape<L, K, true, 0, false>;  // (both components in LDS: see the hand-over below)

// out (cols, rows) = in (rows, cols)^T, 32 x 32 tiles through LDS
template <class T>
__global__ void __launch_bounds__(256) k_transpose_any(const T *__restrict__ in, int rows, int cols, size_t ld_in,
                                                        T *__restrict__ out, size_t ld_out)
{
    __shared__ T tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + int(threadIdx.x);
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[size_t(r) * ld_in + size_t(c)];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + int(threadIdx.x);
        if (r < rows && c < cols) out[size_t(c) * ld_out + size_t(r)] = tile[threadIdx.x][j];
    }
}

template <class T>
static void transpose_any(const T *in, int64_t rows, int64_t cols, size_t ld_in, T *out, size_t ld_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_transpose_any<T>, dim3(uint32_t(ceil_div(cols, 32)), uint32_t(ceil_div(rows, 32))), dim3(32, 8), 0, st,
                       in, int(rows), int(cols), ld_in, out, ld_out);
    PFB_HIP(hipGetLastError());
}

template <class S>
__device__ __forceinline__ int rf_neg_index(int k) { return k == 0 ? 0 : S::N - k; }  // (N - k) mod N

// ---- pass 1: rows of beam * x, zero-padded to nyp, forward along y, half spectrum kept --------------
// Two real rows share one complex transform: z = a + i b, A[k] = (Z[k] + conj Z[N-k]) / 2,
// B[k] = (Z[k] - conj Z[N-k]) / 2i.  The partner Z[N-k] lives in another thread: one more LDS exchange
// (component by component) after the last pass -- 10 % of a transform to save a whole one.
struct PsfPadLoad2 {
    const double *xa, *xb, *ba, *bb;  // rows 2r and 2r+1 (xb == NULL: no second row)
    int ny;
    __device__ __forceinline__ double2 operator()(int u, int) const
    {
        if (u >= ny) return make_double2(0.0, 0.0);
        double va = xa[u], vb = xb != nullptr ? xb[u] : 0.0;
        if (ba != nullptr) {
            va *= ba[u];
            if (xb != nullptr) vb *= bb[u];
        }
        return make_double2(va, vb);
    }
};

// Transposed stores (passes 1 and 2 write the layout the next pass reads row-wise; no transpose kernels): the 16-byte
// pieces of a 128-byte line of the destination come from 8 consecutive source rows, which this map hands to 8 workgroups
// of ONE XCD (block ids equal mod 8) inside the same 64 block ids -- the line leaves that XCD's L2 whole.
__device__ __forceinline__ int psf_xcd_row(int b, int n)
{
    if (b >= (n & ~63)) return b;
    const int r = b & 63;
    return (b & ~63) + (r & 7) * 8 + (r >> 3);
}

template <class S>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_psf_rows_fwd(const double2 *tw, const double *x, const double *beam,
                                                                          int nx, int ny, double2 *t2, size_t ld2)
{
    extern __shared__ double rf_lds[];
    const size_t r0 = size_t(psf_xcd_row(int(blockIdx.x), (nx + 1) / 2)) * 2;  // (4 row pairs = 8 rows = one line of T2)
    const bool two = r0 + 1 < size_t(nx);
    PsfPadLoad2 ld{x + r0 * size_t(ny), two ? x + (r0 + 1) * size_t(ny) : nullptr,
                   beam != nullptr ? beam + r0 * size_t(ny) : nullptr,
                   (beam != nullptr && two) ? beam + (r0 + 1) * size_t(ny) : nullptr, ny};
    double re[S::E], im[S::E];
    int t;
    rf_row_compute<S>(tw, ld, false, rf_lds, t, re, im);
    rf_opaque(t);
    // partner values Z[(N - k) mod N], one component at a time through the (now free) transpose buffer
    double pre[S::E], pim[S::E];
    rf_barrier();
#pragma unroll
    for (int e = 0; e < S::E; ++e) rf_lds[rf_swz(S::out_pos(t, e))] = re[e];
    rf_barrier();
#pragma unroll
    for (int e = 0; e < S::E; ++e) pre[e] = rf_lds[rf_swz(rf_neg_index<S>(S::out_pos(t, e)))];
    rf_barrier();
#pragma unroll
    for (int e = 0; e < S::E; ++e) rf_lds[rf_swz(S::out_pos(t, e))] = im[e];
    rf_barrier();
#pragma unroll
    for (int e = 0; e < S::E; ++e) pim[e] = rf_lds[rf_swz(rf_neg_index<S>(S::out_pos(t, e)))];
    // T2[k][r0], T2[k][r0 + 1]: 32 contiguous bytes per frequency
    double2 *col = t2 + r0;
#pragma unroll
    for (int e = 0; e < S::E; ++e) {
        const int k = S::out_pos(t, e);
        if (k <= S::N / 2) {
            double2 *o = col + size_t(k) * ld2;
            o[0] = make_double2(0.5 * (re[e] + pre[e]), 0.5 * (im[e] - pim[e]));
            if (two) o[1] = make_double2(0.5 * (im[e] + pim[e]), 0.5 * (pre[e] - re[e]));
        }
    }
}

// ---- pass 2: per y-frequency: forward along x, times f(psfhat), inverse along x ----------------------
struct PsfColLoad {
    const double2 *row;
    int nx;
    __device__ __forceinline__ double2 operator()(int p, int) const
    {
        return p < nx ? row[p] : make_double2(0.0, 0.0);
    }
};

struct PsfLdsLoad {
    static constexpr bool FROM_LDS = true;
    const double *l;
    int N;
    __device__ __forceinline__ double2 operator()(int p, int) const { return make_double2(l[p], l[N + p]); }
};

// mode 0: psf, 1: psf + shift, 2: 1 / (psf + shift); norm = 1 / (nxp nyp) folded in
template <class S>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_psf_cols(const double2 *tw, const double2 *t2, size_t ld2, int nx, int nyo2,
                                                                      const double *psfT, int is_complex, int mode, double shift,
                                                                      double norm, double2 *t1, size_t ld1)
{
    extern __shared__ double rf_lds[];
    const size_t k = size_t(psf_xcd_row(int(blockIdx.x), nyo2));
    const double2 *row = t2 + k * ld2;
    PsfColLoad ld{row, nx};
    double re[S::E], im[S::E];
    int t;
    rf_row_compute<S>(tw, ld, false, rf_lds, t, re, im);
    // The forward transform leaves position t + rf_last_slot(e) T in slot e: a permutation of the
    // thread's own natural set {t + e T}.  The inverse transform therefore starts from registers
    // (power-of-two lengths: no leading radix-3/5 pass, whose inputs would belong to other threads).
    double re2[S::E], im2[S::E];
    const double *prow = psfT + k * size_t(S::N) * (is_complex ? 2 : 1);
    __builtin_amdgcn_sched_barrier(0);  // the psfhat loads belong between the transforms, not inside the first
#pragma unroll
    for (int e0 = 0; e0 < S::E; e0 += 4) {
        double pr[4], pi[4];
#pragma unroll
        for (int e = e0; e < e0 + 4; ++e) {
            const int kx = S::out_pos(t, e);
            if (is_complex) {
                const double2 pv = reinterpret_cast<const double2 *>(prow)[kx];
                pr[e - e0] = pv.x;
                pi[e - e0] = pv.y;
            } else {
                pr[e - e0] = prow[kx];
                pi[e - e0] = 0.0;
            }
        }
#pragma unroll
        for (int e = e0; e < e0 + 4; ++e) {
            double p_r = pr[e - e0];
            const double p_i = pi[e - e0];
            if (mode != 0) p_r += shift;
            double yr, yi;
            if (mode == 2) {  // v / (pr + i pi)
                const double d = norm / (p_r * p_r + p_i * p_i);
                yr = (re[e] * p_r + im[e] * p_i) * d;
                yi = (im[e] * p_r - re[e] * p_i) * d;
            } else {
                yr = (re[e] * p_r - im[e] * p_i) * norm;
                yi = (re[e] * p_i + im[e] * p_r) * norm;
            }
            const int s = rf_last_slot(S::RLAST, S::E, e);
            re2[s] = yi;  // inverse transform = forward transform of the swapped components
            im2[s] = yr;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // a second opaque copy of the thread index: otherwise the LDS / twiddle addresses of the second
    // transform are common subexpressions of the first and stay live through it (130 spilled VGPRs)
    int tb = t;
    rf_opaque(tb);
    if constexpr (S::LEAD == 1) {
        const double2 w0[S::E / rf_radix(S::K, 0)] = {};
        rf_passes<S, 0, 1>(re2, im2, tb, tw, rf_lds, w0);
    } else {
        // the leading radix-3/5 pass of the second transform needs other threads' values: hand the row over through LDS
        static_assert(S::DUAL, "the LDS hand-over needs both components in LDS");
#pragma unroll
        for (int e = 0; e < S::E; ++e) {  // re2 / im2 are in natural slot order: slot e <-> position tb + e T
            rf_lds[tb + e * S::T] = re2[e];
            rf_lds[S::N + tb + e * S::T] = im2[e];
        }
        rf_barrier();
        PsfLdsLoad ll{rf_lds, S::N};
        int tc;
        rf_row_compute<S>(tw, ll, false, rf_lds, tc, re2, im2);
        tb = tc;
    }
#pragma unroll
    for (int e = 0; e < S::E; ++e) {
        const int p = S::out_pos(tb, e);
        if (p < nx) t1[size_t(p) * ld1 + k] = make_double2(im2[e], re2[e]);  // T1[p][k]: pass 3 reads rows of T1
    }
}

// ---- pass 3: rows: Hermitian-extended inverse along y, real part, crop, beam, scale, eta --------------
// Two output rows per transform: Z = A + i B (both Hermitian-extended), z = a + i b.
struct PsfHermLoad2 {
    const double2 *rowa, *rowb;  // rowb == NULL: single row
    int N;
    __device__ __forceinline__ double2 operator()(int k, int) const
    {
        const bool lo = k <= N / 2;
        const int kk = lo ? k : N - k;
        const double2 a = rowa[kk];
        const double2 b = rowb != nullptr ? rowb[kk] : make_double2(0.0, 0.0);
        // lo: a + i b ; hi: conj(a) + i conj(b)
        return lo ? make_double2(a.x - b.y, a.y + b.x) : make_double2(a.x + b.y, b.x - a.y);
    }
};

template <class S>
__global__ void __launch_bounds__(S::T, S::WAVES_PER_SIMD) k_psf_rows_inv(const double2 *tw, const double2 *t1, size_t ld1,
                                                                          const double *beam, const double *x, int nx, int ny,
                                                                          double scale, double eta, int accumulate, double *out)
{
    extern __shared__ double rf_lds[];
    const size_t r0 = size_t(blockIdx.x) * 2;
    const bool two = r0 + 1 < size_t(nx);
    PsfHermLoad2 ld{t1 + r0 * ld1, two ? t1 + (r0 + 1) * ld1 : nullptr, S::N};
    double re[S::E], im[S::E];
    int t;
    rf_row_compute<S>(tw, ld, true, rf_lds, t, re, im);
    rf_opaque(t);
#pragma unroll
    for (int e = 0; e < S::E; ++e) {
        const int u = S::out_pos(t, e);
        if (u < ny) {
            // inverse transform: value = (im, re) = (row 2r, row 2r+1)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && !two) break;
                const size_t o = (r0 + size_t(h)) * size_t(ny) + size_t(u);
                double v = h == 0 ? im[e] : re[e];
                if (beam != nullptr) v *= beam[o];
                v *= scale;
                if (eta != 0.0) v += eta * x[o];
                out[o] = accumulate ? out[o] + v : v;
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------

template <class Kern>
static void psf_allow_lds(Kern kern, bool *)
{
    allow_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024);  // per (device, kernel)
}

bool PsfFFT::init(int64_t nx_, int64_t ny_, int64_t nxp_, int64_t nyp_)
{
    ok = false;
    const char *env = std::getenv("PFBHIP_PSF_ROWFFT");
    if (env != nullptr && env[0] == '0') return false;
    RowFFTPlan a, b;
    if (!rowfft_make_plan(nxp_, &a) || !rowfft_make_plan(nyp_, &b) || a.doubled || b.doubled) return false;
    if (a.lead != 1 && nxp_ * 16 > 160 * 1024) return false;  // pass 2 hands rows with a radix-3/5 lead over through LDS
    if (nx_ > nxp_ || ny_ > nyp_) return false;
    nx = nx_;
    ny = ny_;
    nxp = nxp_;
    nyp = nyp_;
    nyo2 = nyp / 2 + 1;
    ld1 = size_t(ceil_div(nyo2, 32) * 32);
    if (!fy.init(nyp) || !fx.init(nxp)) return false;
    t1.alloc(size_t(nx) * ld1);
    ld2 = size_t(nx) + 8;  // (off the power of two: the transposed stores of pass 1 walk the rows of T2)
    t2.alloc(size_t(nyo2) * ld2);
    ok = true;
    return true;
}

void PsfFFT::transpose_psf(const double *src_dev, bool is_complex, double *dst_dev, hipStream_t st) const
{
    if (is_complex)
        transpose_any(reinterpret_cast<const double2 *>(src_dev), nxp, nyo2, size_t(nyo2), reinterpret_cast<double2 *>(dst_dev),
                      size_t(nxp), st);
    else
        transpose_any(src_dev, nxp, nyo2, size_t(nyo2), dst_dev, size_t(nxp), st);
}

template <class S>
static void launch_rows_fwd(const PsfFFT &p, const double *x, const double *beam, hipStream_t st)
{
    static bool attr = false;
    psf_allow_lds(&k_psf_rows_fwd<S>, &attr);
    hipLaunchKernelGGL(k_psf_rows_fwd<S>, dim3(uint32_t((p.nx + 1) / 2)), dim3(S::T), size_t(S::LDS_BYTES), st, p.fy.pl.twiddle,
                       x, beam, int(p.nx), int(p.ny), p.t2.p, p.ld2);
}
template <class S>
static void launch_cols(const PsfFFT &p, const double *psfT, bool is_complex, int mode, double shift, hipStream_t st)
{
    static bool attr = false;
    psf_allow_lds(&k_psf_cols<S>, &attr);
    hipLaunchKernelGGL(k_psf_cols<S>, dim3(uint32_t(p.nyo2)), dim3(S::T), size_t(S::LDS_BYTES), st, p.fx.pl.twiddle, p.t2.p, p.ld2,
                       int(p.nx), int(p.nyo2), psfT, is_complex ? 1 : 0, mode, shift, 1.0 / (double(p.nxp) * double(p.nyp)), p.t1.p,
                       p.ld1);
}
template <class S>
static void launch_rows_inv(const PsfFFT &p, const double *beam, const double *x, double scale, double eta, int accumulate,
                            double *out, hipStream_t st)
{
    static bool attr = false;
    psf_allow_lds(&k_psf_rows_inv<S>, &attr);
    hipLaunchKernelGGL(k_psf_rows_inv<S>, dim3(uint32_t((p.nx + 1) / 2)), dim3(S::T), size_t(S::LDS_BYTES), st, p.fy.pl.twiddle,
                       p.t1.p, p.ld1, beam, x, int(p.nx), int(p.ny), scale, eta, accumulate, out);
}

void PsfFFT::apply(const double *x_dev, const double *beam_dev, const double *psfT_dev, bool is_complex, int mode, double shift,
                   double scale, double eta, int accumulate, double *out_dev, hipStream_t st)
{
    PFB_REQUIRE(ok, "PSF row-FFT plan is not initialised");
    switch (nyp) {
#define RF_X(L, K) \
    case (L << K): launch_rows_fwd<PsfShape<L, K>>(*this, x_dev, beam_dev, st); break;
        PSF_FOR_SHAPES_Y(RF_X)
#undef RF_X
        default: PFB_REQUIRE(false, "unsupported padded size %lld", (long long)nyp);
    }
    PFB_HIP(hipGetLastError());
    switch (nxp) {
#define RF_X(L, K) \
    case (L << K): launch_cols<PsfShape<L, K>>(*this, psfT_dev, is_complex, mode, shift, st); break;
        PSF_FOR_SHAPES_X(RF_X)
#undef RF_X
        default: PFB_REQUIRE(false, "unsupported padded size %lld", (long long)nxp);
    }
    PFB_HIP(hipGetLastError());
    switch (nyp) {
#define RF_X(L, K) \
    case (L << K): launch_rows_inv<PsfShape<L, K>>(*this, beam_dev, x_dev, scale, eta, accumulate, out_dev, st); break;
        PSF_FOR_SHAPES_Y(RF_X)
#undef RF_X
        default: PFB_REQUIRE(false, "unsupported padded size %lld", (long long)nyp);
    }
    PFB_HIP(hipGetLastError());
}

}  // namespace pfbhip
