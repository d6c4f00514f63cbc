// rowfft.hpp -- hand-written batched row FFT (double-complex) for gfx950 with fusable load/store.
//
// Why: the second axis of the plane transform runs on the cropped, transposed plane B (ny, nu).
// With rocFFT that pass needs B materialised twice (pad/crop kernel + in-place transform).  This
// kernel takes a LOAD functor (element index -> value) and a STORE functor (element index, value),
// so "pad + w-screen" feeds the transform directly from the image and "crop + w-screen + accumulate"
// consumes it directly into the image: B is read or written once, not three times.
//
// Algorithm: Stockham autosort, one row per workgroup, the row held in REGISTERS (16 complex per
// thread, T = N/16 threads), passes of radix 16/8/4/2 (and one leading radix 3 or 5 pass for
// N = 3*2^a, 5*2^a).  Between passes the row is transposed through LDS one component at a time
// (N doubles = 80 KiB at N = 10240): writes are scattered (XOR-swizzled against bank conflicts),
// reads are unit-stride.  With T = N/16 every power-of-two pass finds thread t holding exactly the
// positions {t + e T, e < 16}, which is also the coalesced global layout, so the first pass reads
// global memory directly and the last one writes it directly (no LDS round trip for I/O).
// Twiddles: one load per butterfly from an L2-resident table exp(-2 pi i k / N), powers by complex
// multiplication.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pfbhip {

constexpr int RF_E = 16;        // complex elements per thread
constexpr int RF_MAXPASS = 8;

struct RowFFTPlan {
    int N = 0, T = 0, npass = 0;
    int radix[RF_MAXPASS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double2 *twiddle = nullptr;  // device table exp(-2 pi i k / N), k < N (filled by the owner of the plan)
};

// N = m * 2^a with m in {1, 3, 5}, 1024 <= N <= 16384, N % 16 == 0
inline bool rowfft_make_plan(int64_t N, RowFFTPlan *p)
{
    if (N < 1024 || N > 16384 || (N % 16) != 0) return false;
    int64_t pow2 = N, m = 1;
    if (pow2 % 5 == 0) { m = 5; pow2 /= 5; }
    else if (pow2 % 3 == 0) { m = 3; pow2 /= 3; }
    if (pow2 & (pow2 - 1)) return false;
    if (pow2 < 16) return false;
    RowFFTPlan pl;
    pl.N = int(N);
    pl.T = int(N / RF_E);
    if (pl.T > 1024) return false;
    int np = 0;
    if (m > 1) pl.radix[np++] = int(m);
    while (pow2 > 1) {
#ifdef RF_SMALLRADIX
        int r = pow2 >= 4 ? 4 : int(pow2);
#else
        int r = pow2 >= 16 ? 16 : int(pow2);
#endif
        if (np >= RF_MAXPASS) return false;
        pl.radix[np++] = r;
        pow2 /= r;
    }
    pl.npass = np;
    *p = pl;
    return true;
}

#if defined(__HIPCC__)

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }  // -i * a

// ---- forward DFTs of small size, natural order in and out ----
template <int R>
__device__ __forceinline__ void dft(double2 (&v)[R]);

template <>
__device__ __forceinline__ void dft<2>(double2 (&v)[2])
{
    double2 a = v[0];
    v[0] = cadd(a, v[1]);
    v[1] = csub(a, v[1]);
}

template <>
__device__ __forceinline__ void dft<4>(double2 (&v)[4])
{
    double2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(a, c);
    v[2] = csub(a, c);
    v[1] = cadd(b, d);
    v[3] = csub(b, d);
}

template <>
__device__ __forceinline__ void dft<3>(double2 (&v)[3])
{
    const double h = 0.86602540378443864676;  // sqrt(3)/2
    double2 t1 = cadd(v[1], v[2]);
    double2 t2 = make_double2(v[0].x - 0.5 * t1.x, v[0].y - 0.5 * t1.y);
    double2 t3 = mul_mi(make_double2(h * (v[1].x - v[2].x), h * (v[1].y - v[2].y)));  // -i * (sqrt3/2)(x1 - x2)
    v[0] = cadd(v[0], t1);
    v[1] = cadd(t2, t3);
    v[2] = csub(t2, t3);
}

template <>
__device__ __forceinline__ void dft<5>(double2 (&v)[5])
{
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;  // cos(2pi/5), cos(4pi/5)
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;   // sin(2pi/5), sin(4pi/5)
    double2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    double2 p1 = make_double2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    double2 p2 = make_double2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    double2 q1 = mul_mi(make_double2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
    double2 q2 = mul_mi(make_double2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = cadd(p1, q1);
    v[4] = csub(p1, q1);
    v[2] = cadd(p2, q2);
    v[3] = csub(p2, q2);
}

// exp(-2 pi i m / 16), m = 0..9 (m is a compile-time constant at every call site)
__device__ __forceinline__ double2 w16(int m)
{
    const double c1 = 0.92387953251128675613, c2 = 0.70710678118654752440, c3 = 0.38268343236508977173;
    switch (m) {
        case 0: return make_double2(1.0, 0.0);
        case 1: return make_double2(c1, -c3);
        case 2: return make_double2(c2, -c2);
        case 3: return make_double2(c3, -c1);
        case 4: return make_double2(0.0, -1.0);
        case 5: return make_double2(-c3, -c1);
        case 6: return make_double2(-c2, -c2);
        case 7: return make_double2(-c1, -c3);
        case 8: return make_double2(-1.0, 0.0);
        default: return make_double2(-c1, c3);
    }
}

// dft<8> and dft<16> work in place and leave the result in TRANSPOSED digit order (no second
// register copy of the row): slot s holds X[dft_index<R>(s)].
template <int R>
__host__ __device__ constexpr int dft_index(int s)
{
    return R == 16 ? (s >> 2) + 4 * (s & 3) : (R == 8 ? (s >> 1) + 4 * (s & 1) : s);
}

template <>
__device__ __forceinline__ void dft<8>(double2 (&v)[8])
{
    // n = 2 n1 + n2 (n1 < 4, n2 < 2), k = k1 + 4 k2 ; result X[k1 + 4 k2] in slot 2 k1 + k2
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
        double2 t[4] = {v[n2], v[2 + n2], v[4 + n2], v[6 + n2]};
        dft<4>(t);
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) v[2 * k1 + n2] = (n2 == 0 || k1 == 0) ? t[k1] : cmul(t[k1], w16(2 * k1));
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        double2 a = v[2 * k1];
        v[2 * k1] = cadd(a, v[2 * k1 + 1]);
        v[2 * k1 + 1] = csub(a, v[2 * k1 + 1]);
    }
}

template <>
__device__ __forceinline__ void dft<16>(double2 (&v)[16])
{
    // n = 4 n1 + n2, k = k1 + 4 k2 ; result X[k1 + 4 k2] in slot 4 k1 + k2
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        double2 t[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
        dft<4>(t);
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) v[4 * k1 + n2] = (n2 == 0 || k1 == 0) ? t[k1] : cmul(t[k1], w16(n2 * k1));
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        double2 t[4] = {v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]};
        dft<4>(t);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) v[4 * k1 + k2] = t[k2];
    }
}

__device__ __forceinline__ int rf_swz(int p) { return p ^ ((p >> 4) & 15); }

// Transpose through LDS: every thread scatters its NE values to positions pos[e] (or skips pos < 0),
// then gathers the standard layout {t + e T}.  src and dst may be the same array.
template <int NE>
__device__ __forceinline__ void rf_exchange(const double (&src)[NE], const int (&pos)[NE], double (&dst)[RF_E], int t,
                                            int T, double *lds)
{
#pragma unroll
    for (int e = 0; e < NE; ++e)
        if (pos[e] >= 0) lds[rf_swz(pos[e])] = src[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < RF_E; ++e) dst[e] = lds[rf_swz(t + e * T)];
    __syncthreads();
}

// LDS transpose after a radix-R pass: slot i + s IT holds output dft_index<R>(s) of butterfly
// j = t + i T, i.e. position expand(j) + dft_index<R>(s) Ns; afterwards slot e holds position t + e T.
template <int R>
__device__ __forceinline__ void rf_transpose(double (&a)[RF_E], int t, int T, int Ns, double *lds)
{
    constexpr int IT = RF_E / R;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int j = t + i * T;
        const int k = j % Ns;
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int s = 0; s < R; ++s) lds[rf_swz(j0 + dft_index<R>(s) * Ns)] = a[i + s * IT];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < RF_E; ++e) a[e] = lds[rf_swz(t + e * T)];
    __syncthreads();
}

// One power-of-two pass.  On entry slot e holds position t + e T of the current array; on exit the
// same holds for the next array (after the LDS transpose).  The LAST pass skips the transpose and
// leaves the outputs in the butterfly's own slot order: slot i + s IT holds position
// t + (i + dft_index<R>(s) IT) T (the caller's store uses rf_last_slot).
template <int R>
__device__ __forceinline__ void rf_pass(double (&re)[RF_E], double (&im)[RF_E], int t, int T, int N, int Ns, bool last,
                                        const double2 *__restrict__ tw, double *lds)
{
    constexpr int IT = RF_E / R;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int j = t + i * T;
        const int k = j % Ns;
        double2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = make_double2(re[i + q * IT], im[i + q * IT]);
        if (Ns > 1) {
            const double2 w1 = tw[k * (N / (Ns * R))];  // exp(-2 pi i k / (Ns R))
            double2 w = w1;
#pragma unroll
            for (int q = 1; q < R; ++q) {
                v[q] = cmul(v[q], w);
                if (q + 1 < R) w = cmul(w, w1);
            }
        }
        dft<R>(v);
#pragma unroll
        for (int s = 0; s < R; ++s) {
            re[i + s * IT] = v[s].x;
            im[i + s * IT] = v[s].y;
        }
    }
    if (!last) {
        rf_transpose<R>(re, t, T, Ns, lds);
        rf_transpose<R>(im, t, T, Ns, lds);
    }
}

// position (divided by T, minus t) held by slot e after the LAST pass of radix R
__host__ __device__ constexpr int rf_last_slot(int R, int e)
{
    // slot e = i + s IT, IT = 16 / R  ->  i + dft_index<R>(s) IT
    return R == 16 ? ((e >> 2) + 4 * (e & 3))
                   : (R == 8 ? ((e & 1) + 2 * (((e >> 1) >> 1) + 4 * ((e >> 1) & 1))) : e);
}

// Leading odd pass (radix M = 3 or 5, Ns = 1): N/M butterflies, ceil(16/M) per thread, inputs read
// straight from the load functor; always followed by the LDS transpose.
template <int M, class Load>
__device__ __forceinline__ void rf_first_odd(double (&re)[RF_E], double (&im)[RF_E], int t, int T, int N, Load &ld,
                                             bool inverse, double *lds)
{
    constexpr int IT = (RF_E + M - 1) / M;
    const int nbf = N / M;
    double ore[IT * M], oim[IT * M];
    int pos[IT * M];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int j = t + i * T;
        double2 v[M] = {};
        if (j < nbf) {
#pragma unroll
            for (int q = 0; q < M; ++q) {
                double2 x = ld(j + q * nbf);
                v[q] = inverse ? make_double2(x.y, x.x) : x;
            }
            dft<M>(v);
        }
#pragma unroll
        for (int q = 0; q < M; ++q) {
            ore[i * M + q] = v[q].x;
            oim[i * M + q] = v[q].y;
            pos[i * M + q] = j < nbf ? j * M + q : -1;
        }
        // keep the scheduler from interleaving all butterflies' load functors (each may carry a
        // sincospi chain): that blows the register budget of the 640-thread workgroup
        __builtin_amdgcn_sched_barrier(0);
    }
    rf_exchange<IT * M>(ore, pos, re, t, T, lds);
    rf_exchange<IT * M>(oim, pos, im, t, T, lds);
}

// position of the value left in slot e after the last pass
__device__ __forceinline__ int rf_out_pos(const RowFFTPlan &pl, int t, int e)
{
    const int Rl = pl.radix[pl.npass - 1];
    const int slot = Rl == 16 ? rf_last_slot(16, e) : (Rl == 8 ? rf_last_slot(8, e) : e);
    return t + slot * pl.T;
}

// load -> passes; on return slot e holds the transform at position rf_out_pos(pl, t, e) as
// (re[e], im[e]) for the forward transform and as (im[e], re[e]) for the (unnormalised) inverse.
template <class Load>
__device__ __forceinline__ void rf_row_compute(const RowFFTPlan &pl, Load &ld, bool inverse, double *lds, int &t_out,
                                               double (&re)[RF_E], double (&im)[RF_E])
{
    int t = threadIdx.x;
    // Opaque to the optimiser: otherwise every pass's (row-invariant) LDS and global addresses are
    // hoisted out of the caller's loops and kept live -- hundreds of VGPRs of loop invariants.
    asm volatile("" : "+v"(t));
    t_out = t;
    const int T = pl.T, N = pl.N;
    int Ns = 1, p = 0;
    if (pl.radix[0] == 5) {
        rf_first_odd<5>(re, im, t, T, N, ld, inverse, lds);
        Ns = 5;
        p = 1;
    } else if (pl.radix[0] == 3) {
        rf_first_odd<3>(re, im, t, T, N, ld, inverse, lds);
        Ns = 3;
        p = 1;
    } else {
#pragma unroll
        for (int e = 0; e < RF_E; ++e) {
            double2 x = ld(t + e * T);
            re[e] = inverse ? x.y : x.x;
            im[e] = inverse ? x.x : x.y;
        }
    }
    for (; p < pl.npass; ++p) {
        const bool last = p == pl.npass - 1;
        const int R = pl.radix[p];
#ifndef RF_SMALLRADIX
        if (R == 16) rf_pass<16>(re, im, t, T, N, Ns, last, pl.twiddle, lds);
        else if (R == 8) rf_pass<8>(re, im, t, T, N, Ns, last, pl.twiddle, lds);
        else
#endif
        if (R == 4) rf_pass<4>(re, im, t, T, N, Ns, last, pl.twiddle, lds);
        else rf_pass<2>(re, im, t, T, N, Ns, last, pl.twiddle, lds);
        Ns *= R;
    }
}

// The whole row: load -> passes -> store.
template <class Load, class Store>
__device__ __forceinline__ void rf_row(const RowFFTPlan &pl, Load &ld, Store &st, bool inverse, double *lds)
{
    double re[RF_E], im[RF_E];
    int t;
    rf_row_compute(pl, ld, inverse, lds, t, re, im);
#pragma unroll
    for (int e = 0; e < RF_E; ++e)
        st(rf_out_pos(pl, t, e), inverse ? make_double2(im[e], re[e]) : make_double2(re[e], im[e]));
}

#endif  // __HIPCC__

}  // namespace pfbhip
