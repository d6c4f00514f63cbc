// rowfft.hpp -- hand-written batched row FFT (double-complex) for gfx950 with fusable load/store.
//
// Why: the second axis of the plane transform runs on the cropped, transposed plane B (ny, nu).
// With rocFFT that pass needs B materialised twice (pad/crop kernel + in-place transform).  This
// kernel takes a LOAD functor (element index, slot -> value) and a STORE functor (element index, value),
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
#include <type_traits>

namespace pfbhip {

// Supported row lengths: N = LEAD * 2^K with (LEAD, K) in RF_FOR_SHAPES, i.e. 1024 <= N <= 16384 of
// the forms {1, 3, 5, 7, 9, 15} * 2^a.  Every shape is its own kernel instantiation: the pass sequence, N and
// T are compile-time constants (straight-line code; a run-time radix switch costs ~60 more VGPRs and
// 35 % of the throughput).
#define RF_FOR_SHAPES(X)                                                                                              \
    X(1, 10) X(1, 11) X(1, 12) X(1, 13) X(1, 14) X(3, 9) X(3, 10) X(3, 11) X(3, 12) X(5, 8) X(5, 9) X(5, 10) X(5, 11) \
    X(7, 8) X(7, 9) X(7, 10) X(7, 11) X(9, 7) X(9, 8) X(9, 9) X(9, 10) X(15, 7) X(15, 8) X(15, 9) X(15, 10)

// complex elements per thread: 32 (T = N/32 threads per row) where that leaves room for TWO rows per
// CU (registers: 10 waves of <= 168 VGPRs; LDS: 2 x N doubles), so that one row's butterflies overlap
// the other row's LDS transposes; 16 otherwise.
#ifndef RF_E32_MAXN
#define RF_E32_MAXN 0
#endif
#ifndef RF_E32_MINN
#define RF_E32_MINN 4096
#endif
#ifndef RF_XPAD
#define RF_XPAD 2
#endif
#ifndef RF_TWO_WG_MAXT
#define RF_TWO_WG_MAXT 1024
#endif
constexpr int rf_elems(int N) { return (N >= RF_E32_MINN && N <= RF_E32_MAXN) ? 32 : 16; }

// Doubled shapes N = 2 N1 (N1 = LEAD * 2^K from this list): one workgroup runs the N1-point transforms of the
// even and the odd samples one after the other and combines them (X[k] = E[k] + w^k O[k], X[k + N1] = E[k] -
// w^k O[k]); E waits in registers / compiler-managed scratch meanwhile.  Extends the range to 20480, 24576 and 32768
// points (uv-grids of 16k^2 images).
#define RF_FOR_SHAPES2(X) X(5, 11) X(3, 12) X(1, 14)

struct RowFFTPlan {
    int N = 0, T = 0, lead = 0, K = 0;
    int doubled = 0;                   // 1: N = 2 N1, run as two N1-point transforms + combine
    // device table (filled by the owner of the plan): exp(-2 pi i k / N), k < N; doubled shapes: the N1-point table
    // followed by exp(-2 pi i k / N), k < N1
    const double2 *twiddle = nullptr;
};

// radix of pass p of the power-of-two part 2^K (after the optional leading radix-3/5 pass)
constexpr int rf_npass(int K) { return (K + 3) / 4; }
constexpr int rf_radix(int K, int p)
{
    // 7: 16 8 | 8: 16 16 | 9: 16 8 4 | 10: 16 16 4 | 11: 16 16 8 | 12: 16 16 16 | 13: 16 16 8 4 | 14: 16 16 16 4
    return p == 0 ? 16
         : p == 1 ? (K == 9 || K == 7 ? 8 : 16)
         : p == 2 ? (K == 9 || K == 10 ? 4 : (K == 11 || K == 13 ? 8 : 16))
                  : 4;
}

inline bool rowfft_make_plan(int64_t N, RowFFTPlan *p)
{
#define RF_X(L, KK)                      \
    if (N == (int64_t(L) << KK)) {       \
        p->N = int(N);                   \
        p->T = int(N) / rf_elems(int(N)); \
        p->lead = L;                     \
        p->K = KK;                       \
        return true;                     \
    }
    RF_FOR_SHAPES(RF_X)
#undef RF_X
#define RF_X(L, KK)                          \
    if (N == 2 * (int64_t(L) << KK)) {       \
        p->N = int(N);                       \
        p->T = int(N / 2) / rf_elems(int(N / 2)); \
        p->lead = L;                         \
        p->K = KK + 1;                       \
        p->doubled = 1;                      \
        return true;                         \
    }
    RF_FOR_SHAPES2(RF_X)
#undef RF_X
    return false;
}

// smallest supported row length >= x (0 if none)
inline int64_t rowfft_size_at_least(double x)
{
    int64_t best = 0;
#define RF_X(L, KK)                                                       \
    if (double(int64_t(L) << KK) >= x - 1e-9 && (best == 0 || (int64_t(L) << KK) < best)) best = int64_t(L) << KK;
    RF_FOR_SHAPES(RF_X)
#undef RF_X
#define RF_X(L, KK)                                                               \
    if (double(2 * (int64_t(L) << KK)) >= x - 1e-9 && (best == 0 || 2 * (int64_t(L) << KK) < best)) \
        best = 2 * (int64_t(L) << KK);
    RF_FOR_SHAPES2(RF_X)
#undef RF_X
    return best;
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

template <>
__device__ __forceinline__ void dft<7>(double2 (&v)[7])
{
    const double c1 = 0.62348980185873353053, c2 = -0.22252093395631440429, c3 = -0.90096886790241912624;
    const double s1 = 0.78183148246802980871, s2 = 0.97492791218182360702, s3 = 0.43388373911755812048;
    const double2 a1 = cadd(v[1], v[6]), a2 = cadd(v[2], v[5]), a3 = cadd(v[3], v[4]);
    const double2 b1 = csub(v[1], v[6]), b2 = csub(v[2], v[5]), b3 = csub(v[3], v[4]);
    const double2 x0 = v[0];
    // X_k = x0 + sum_j cos(2 pi j k / 7) a_j - i sum_j sin(2 pi j k / 7) b_j ; X_{7-k} with + i
    const double2 p1 = make_double2(x0.x + c1 * a1.x + c2 * a2.x + c3 * a3.x, x0.y + c1 * a1.y + c2 * a2.y + c3 * a3.y);
    const double2 p2 = make_double2(x0.x + c2 * a1.x + c3 * a2.x + c1 * a3.x, x0.y + c2 * a1.y + c3 * a2.y + c1 * a3.y);
    const double2 p3 = make_double2(x0.x + c3 * a1.x + c1 * a2.x + c2 * a3.x, x0.y + c3 * a1.y + c1 * a2.y + c2 * a3.y);
    const double2 q1 = mul_mi(make_double2(s1 * b1.x + s2 * b2.x + s3 * b3.x, s1 * b1.y + s2 * b2.y + s3 * b3.y));
    const double2 q2 = mul_mi(make_double2(s2 * b1.x - s3 * b2.x - s1 * b3.x, s2 * b1.y - s3 * b2.y - s1 * b3.y));
    const double2 q3 = mul_mi(make_double2(s3 * b1.x - s1 * b2.x + s2 * b3.x, s3 * b1.y - s1 * b2.y + s2 * b3.y));
    v[0] = cadd(x0, cadd(a1, cadd(a2, a3)));
    v[1] = cadd(p1, q1);
    v[6] = csub(p1, q1);
    v[2] = cadd(p2, q2);
    v[5] = csub(p2, q2);
    v[3] = cadd(p3, q3);
    v[4] = csub(p3, q3);
}

template <>
__device__ __forceinline__ void dft<9>(double2 (&v)[9])
{
    // n = 3 n1 + n2, k = k1 + 3 k2 : t[n2][k1] = DFT3_{n1} x[3 n1 + n2] * w9^(n2 k1) ; X[k1 + 3 k2] = DFT3_{n2} t[n2][k1]
    const double2 w1 = make_double2(0.76604444311897803520, -0.64278760968653932632);   // exp(-2 pi i / 9)
    const double2 w2 = make_double2(0.17364817766693034885, -0.98480775301220805937);   // ^2
    const double2 w4 = make_double2(-0.93969262078590838405, -0.34202014332566873304);  // ^4
    double2 t[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
        double2 u[3] = {v[n2], v[3 + n2], v[6 + n2]};
        dft<3>(u);
        t[n2][0] = u[0];
        t[n2][1] = n2 == 0 ? u[1] : cmul(u[1], n2 == 1 ? w1 : w2);
        t[n2][2] = n2 == 0 ? u[2] : cmul(u[2], n2 == 1 ? w2 : w4);
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        double2 u[3] = {t[0][k1], t[1][k1], t[2][k1]};
        dft<3>(u);
        v[k1] = u[0];
        v[k1 + 3] = u[1];
        v[k1 + 6] = u[2];
    }
}

template <>
__device__ __forceinline__ void dft<15>(double2 (&v)[15])
{
    // prime-factor 3 x 5 (no twiddles): input (5 n1 + 3 n2) mod 15, output (10 k1 + 6 k2) mod 15
    double2 t[5][3];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        double2 u[3] = {v[(3 * n2) % 15], v[(5 + 3 * n2) % 15], v[(10 + 3 * n2) % 15]};
        dft<3>(u);
        t[n2][0] = u[0];
        t[n2][1] = u[1];
        t[n2][2] = u[2];
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        double2 u[5] = {t[0][k1], t[1][k1], t[2][k1], t[3][k1], t[4][k1]};
        dft<5>(u);
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) v[(10 * k1 + 6 * k2) % 15] = u[k2];
    }
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

// LDS address swizzle of the transposes (a bijection on [0, N), N % 16 == 0): the low four bits are
// xored with bits 4..7, which spreads the stride-16 writes of the early passes over the banks.  The
// counters still show ~45 % of the LDS cycles as bank conflicts at N = 10240 (the period-5 write
// patterns); per-pass transforms that remove two thirds of them (rotation of the low five bits by p >> 5
// for sub-lengths <= 5, none for the radix-5 exchange) were measured 2 % SLOWER: the transposes are not
// on the critical path, the extra address arithmetic is.
__device__ __forceinline__ int rf_swz(int p) { return p ^ ((p >> 4) & 15); }
// Shapes with an odd leading factor LEAD keep the natural layout instead (SWZ = false): their exchanges scatter with
// strides LEAD (conflict-free) and 16 LEAD + k (two-way conflicts in ONE exchange of the row), and without the xor every
// LDS address of an exchange is one per-thread base plus a compile-time offset -- the swizzled form spends 4 integer
// VALU operations per access, a third of the kernel's VALU time, and these kernels are VALU-bound, not LDS-bound.
template <bool SWZ>
__device__ __forceinline__ int rf_pos(int p) { return SWZ ? rf_swz(p) : p; }

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the
// vector-memory counter, which would serialise the prefetched twiddle loads (and the previous row's
// stores) with every LDS transpose.
__device__ __forceinline__ void rf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS transpose after a radix-R pass: slot i + s IT holds output dft_index<R>(s) of butterfly
// j = t + i T, i.e. position expand(j) + dft_index<R>(s) Ns; afterwards slot e holds position t + e T.
// The row moves one component at a time (N doubles of LDS) or, DUAL, both at once (2 N doubles, half
// the barriers).
template <int R, int E, bool DUAL, bool SWZ, int T, int N, int Ns, int PAD, int NL>
__device__ __forceinline__ void rf_transpose(double (&re)[E], double (&im)[E], int t, double *lds)
{
    constexpr int IT = E / R;
    static_assert(T % Ns == 0, "the butterfly's offset within its sub-transform must not depend on i");
    static_assert(PAD == 0 || (!SWZ && T % (R * Ns) == 0), "padded exchange: natural layout, whole blocks per e");
    static_assert(NL >= N + PAD * (N / (R * Ns)), "component length of the LDS buffer");
    // write position of slot i + s IT (recomputed per component: E address registers are worth more
    // than E integer operations): one base per thread plus compile-time offsets.
    // PAD > 0: every block of R Ns positions (one butterfly group a = t / Ns) is followed by PAD unused doubles.  In the
    // natural layout the lanes of a write instruction sit at R Ns a + k (k < Ns), i.e. on (R Ns mod 32) a + k double-banks:
    // with R Ns = 16 LEAD that is 16 a + k, LEAD + LEAD of 32 bank pairs, a 3- to 4-way conflict; two more doubles per
    // block walk the groups over the banks (18 a + k).
    const int k = t % Ns;
    const int p0 = (t - k) * R + k + PAD * ((t - k) / Ns);
    auto wpos = [&](int i, int s) { return rf_pos<SWZ>(p0 + i * (T * R + PAD * (T / Ns)) + dft_index<R>(s) * Ns); };
    const int r0 = PAD > 0 ? t + PAD * (t / (R * Ns)) : t;
    constexpr int RS = PAD > 0 ? T + PAD * (T / (R * Ns)) : T;  // reader's step per slot
    if (DUAL) {
        double *l2 = lds + NL;
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
            for (int s = 0; s < R; ++s) {
                const int p = wpos(i, s);
                lds[p] = re[i + s * IT];
                l2[p] = im[i + s * IT];
            }
        rf_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            re[e] = lds[rf_pos<SWZ>(r0 + e * RS)];
            im[e] = l2[rf_pos<SWZ>(r0 + e * RS)];
        }
        rf_barrier();
    } else {
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
            for (int s = 0; s < R; ++s) lds[wpos(i, s)] = re[i + s * IT];
        rf_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e) re[e] = lds[rf_pos<SWZ>(r0 + e * RS)];
        rf_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
            for (int s = 0; s < R; ++s) lds[wpos(i, s)] = im[i + s * IT];
        rf_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e) im[e] = lds[rf_pos<SWZ>(r0 + e * RS)];
        rf_barrier();
    }
}

// first twiddle exp(-2 pi i k / (Ns R)) of each of the thread's butterflies in a radix-R pass
template <int R, int E>
__device__ __forceinline__ void rf_load_twiddles(double2 (&w1)[E / R], int t, int T, int N, int Ns,
                                                 const double2 *__restrict__ tw)
{
#pragma unroll
    for (int i = 0; i < E / R; ++i) w1[i] = tw[((t + i * T) % Ns) * (N / (Ns * R))];
}

// The butterflies of one power-of-two pass, in place: on entry slot e holds position t + e T of the
// current array; on exit slot i + s IT holds output dft_index<R>(s) of butterfly t + i T.  After the
// LAST pass that is position t + (i + dft_index<R>(s) IT) T (the caller's store uses rf_last_slot).
template <int R, int E>
__device__ __forceinline__ void rf_butterflies(double (&re)[E], double (&im)[E], bool twiddle,
                                               const double2 (&w1s)[E / R])
{
    constexpr int IT = E / R;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        double2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = make_double2(re[i + q * IT], im[i + q * IT]);
        if (twiddle) {
            const double2 w1 = w1s[i];
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
        if (E > 16 && R >= 8) __builtin_amdgcn_sched_barrier(0);  // one wide butterfly in flight at a time
    }
}

// position (divided by T, minus t) held by slot e = i + s IT after the LAST pass of radix R
__host__ __device__ constexpr int rf_last_slot(int R, int E, int e)
{
    const int IT = E / R, i = e % IT, s = e / IT;
    const int d = R == 16 ? (s >> 2) + 4 * (s & 3) : (R == 8 ? (s >> 1) + 4 * (s & 1) : s);
    return i + d * IT;
}

// A load functor that reads the transpose buffer itself (static constexpr bool FROM_LDS = true) needs all of the
// workgroup's loads finished before the first transpose write: the leading odd pass then gathers its inputs first.
template <class Load, class = void>
struct rf_load_from_lds : std::false_type {};
template <class Load>
struct rf_load_from_lds<Load, std::void_t<decltype(Load::FROM_LDS)>> : std::bool_constant<Load::FROM_LDS> {};

// Two-step load functors.  A functor may split its work into
//     Raw     fetch(int pos, int slot) const    the memory request alone: BRANCH-FREE (out-of-range positions read a
//                                               clamped, valid address), no arithmetic on the result
//     double2 finish(Raw, int pos, int slot)    zero padding, screens ... on the fetched word
// and the transforms then issue EVERY request of the row before the first finish.  With the one-step form
// (`cond ? row[i] : 0` inside operator()) hipcc puts each load behind its own branch and waits vmcnt(0) at the join:
// the leading radix-5 pass of a 10240-point row made four dependent round trips to memory (one per butterfly), a
// load -> multiply -> load chain one per element (ISA of round 2's kernels).
template <class Load, class = void>
struct rf_has_fetch : std::false_type {};
template <class Load>
struct rf_has_fetch<Load, std::void_t<decltype(std::declval<const Load &>().fetch(0, 0))>> : std::true_type {};
template <class Load, bool = rf_has_fetch<Load>::value>
struct rf_raw_type { using type = double2; };
template <class Load>
struct rf_raw_type<Load, true> { using type = decltype(std::declval<const Load &>().fetch(0, 0)); };
template <class Load>
__device__ __forceinline__ typename rf_raw_type<Load>::type rf_fetch(Load &ld, int pos, int slot)
{
    if constexpr (rf_has_fetch<Load>::value) return ld.fetch(pos, slot);
    else return ld(pos, slot);
}
template <class Load>
__device__ __forceinline__ double2 rf_finish(Load &ld, const typename rf_raw_type<Load>::type &raw, int pos, int slot)
{
    if constexpr (rf_has_fetch<Load>::value) return ld.finish(raw, pos, slot);
    else return raw;
}

// Leading odd pass (radix M = 3 or 5, Ns = 1): N/M butterflies, ceil(E/M) per thread, inputs read
// straight from the load functor, outputs written straight into the LDS transpose.
// NB: butterflies whose requests are in flight together (two-step functors; RfShape::LOAD_BATCH: all of them where the
// register budget allows, fewer in the 1024-thread shapes)
template <int M, int E, bool DUAL, bool SWZ, int T, int N, int NL, int NB, class Load>
__device__ __forceinline__ void rf_first_odd(double (&re)[E], double (&im)[E], int t, Load &ld, bool inverse, double *lds)
{
    constexpr int IT = (E + M - 1) / M;
    constexpr bool PRE = rf_load_from_lds<Load>::value;
    constexpr bool BULK = PRE || rf_has_fetch<Load>::value;  // requests first (see rf_has_fetch)
    constexpr int BATCH = PRE ? IT : (NB < 1 ? 1 : (NB > IT ? IT : NB));
    const int nbf = N / M;
    double *l2 = lds + NL;
    double oim[DUAL ? 1 : IT * M];
    typename rf_raw_type<Load>::type vin[BULK ? IT * M : 1];
    auto request = [&](int i) {
        const int j = t + i * T;
        // only the last butterfly of a thread can lie past the end (t + i T < N / M for i < IT - 1): it requests the
        // thread's first butterfly again and drops it below -- no branch around the requests
        const int jc = (i == IT - 1 && IT * T > N / M) ? (j < nbf ? j : t) : j;
#pragma unroll
        for (int q = 0; q < M; ++q) vin[i * M + q] = rf_fetch(ld, jc + q * nbf, i * M + q);
    };
    if constexpr (BULK) {
#pragma unroll
        for (int i = 0; i < BATCH; ++i) request(i);
        if constexpr (PRE) rf_barrier();
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        if constexpr (BULK && BATCH < IT) {
            if (i > 0 && i % BATCH == 0) {
#pragma unroll
                for (int i2 = i; i2 < i + BATCH && i2 < IT; ++i2) request(i2);
            }
        }
        const int j = t + i * T;
        double2 v[M] = {};
        if (j < nbf) {
#pragma unroll
            for (int q = 0; q < M; ++q) {
                double2 x;
                if constexpr (BULK) x = rf_finish(ld, vin[i * M + q], j + q * nbf, i * M + q);
                else x = ld(j + q * nbf, i * M + q);
                v[q] = inverse ? make_double2(x.y, x.x) : x;
            }
            dft<M>(v);
#pragma unroll
            for (int q = 0; q < M; ++q) {
                lds[rf_pos<SWZ>(j * M + q)] = v[q].x;
                if (DUAL) l2[rf_pos<SWZ>(j * M + q)] = v[q].y;
            }
        }
        if constexpr (!DUAL) {
#pragma unroll
            for (int q = 0; q < M; ++q) oim[i * M + q] = v[q].y;
        }
        // keep the scheduler from interleaving all butterflies' load functors (each may carry a
        // sincospi chain): that blows the register budget
        if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    rf_barrier();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        re[e] = lds[rf_pos<SWZ>(t + e * T)];
        if (DUAL) im[e] = l2[rf_pos<SWZ>(t + e * T)];
    }
    rf_barrier();
    if constexpr (!DUAL) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int j = t + i * T;
            if (j < nbf) {
#pragma unroll
                for (int q = 0; q < M; ++q) lds[rf_pos<SWZ>(j * M + q)] = oim[i * M + q];
            }
        }
        rf_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e) im[e] = lds[rf_pos<SWZ>(t + e * T)];
        rf_barrier();
    }
}

// Compile-time description of one supported row length.
template <int LEAD_, int K_, bool ALLOW_DUAL = true, int E_ = 0, bool ALLOW_SPLIT = true>
struct RfShape {
    static constexpr int LEAD = LEAD_, K = K_;
    static constexpr int N = LEAD_ << K_, E = E_ > 0 ? E_ : rf_elems(N), T = N / E;
    static constexpr bool DOUBLED = false;
    static constexpr int NSLOT = LEAD_ > 1 ? ((E + LEAD_ - 1) / LEAD_) * LEAD_ : E;  // load-functor slots per row
    static constexpr int NP = rf_npass(K_);
    // LDS layout of the exchanges (see rf_pos / rf_transpose).  Power-of-two shapes from 4096 points take the natural layout
    // too, PADDED per exchange: one spare double per 16 positions in the first exchange (writes of stride 16 -> 17) and 16 per
    // 256 in the second (the 16-lane groups of a write land 256 apart -> 272); both need N / 16 spare doubles per component.
    // Smaller ones (T not a multiple of 256) keep the xor swizzle.
    static constexpr bool SWZ = LEAD_ == 1 && (N < 4096 || E_ > 16);
    static constexpr int RLAST = rf_radix(K_, NP - 1);
    // Two workgroups per CU where they fit (LDS: 2 x N doubles when the row moves one component at a time; threads: 2 T <=
    // RF_TWO_WG_MAXT = 1024, i.e. 4 waves per SIMD and 128 VGPRs): the exchanges are barrier-separated phases in which the whole
    // workgroup writes LDS, reads LDS or computes, so one workgroup leaves the VALU idle during its exchanges and the LDS idle
    // during its butterflies.  Measured (tools/rowfft_probe.py, 4096 rows): N = 8192 0.222 -> 0.201 ms.  With T = 640 (N = 10240)
    // the second workgroup means 5 waves per SIMD and 96 VGPRs: the plain transform fits without spills but gains nothing
    // (0.343 -> 0.350 ms, it is HBM-bound at 4.7 TB/s), the transposing store spills (0.43 -> 0.65 ms) and the fused pad kernel
    // loses its LDS image row (1.90 -> 2.34 ms) -- RF_TWO_WG_MAXT stays at 1024.
    static constexpr int NLX = N + ((LEAD_ == 1 && N >= 4096 && E == 16) ? N / 16 : 0);  // (component length before XPAD is known)
    static constexpr bool TWO_WG_SPLIT = ALLOW_DUAL && ALLOW_SPLIT && E == 16 && 2 * NLX * int(sizeof(double)) <= 160 * 1024 && 2 * T <= RF_TWO_WG_MAXT;
#ifdef RF_NO_DUAL
    static constexpr bool DUAL = false;
#else
    static constexpr bool DUAL = ALLOW_DUAL && E == 16 && NLX * 16 <= 160 * 1024 && !(TWO_WG_SPLIT && 2 * NLX * 16 > 160 * 1024);
#endif
    // unused doubles behind every 16 LEAD positions of the exchange after the first radix-16 pass (see rf_transpose)
    static constexpr int XPAD = (!DUAL && LEAD_ > 1 && K_ >= 8 && E == 16) ? RF_XPAD : 0;
    // padding (doubles per block of R Ns positions) of the exchange behind a radix-R pass at sub-length Ns
    static constexpr int xpad(int R, int Ns)
    {
        if (LEAD_ > 1) return (R == 16 && Ns == LEAD_) ? XPAD : 0;
        if (SWZ) return 0;
        return (R == 16 && Ns == 1) ? 1 : ((R == 16 && Ns == 16) ? 16 : 0);
    }
    static constexpr int NL = N + (LEAD_ > 1 ? XPAD * (N / (16 * LEAD_)) : (SWZ ? 0 : N / 16));  // doubles per LDS component
    static constexpr int LDS_BYTES = (DUAL ? 2 : 1) * NL * int(sizeof(double));
    // (the fused kernels, ALLOW_DUAL = false, add their image row to the LDS: one workgroup per CU)
    static constexpr int WG_PER_CU = (ALLOW_DUAL && 2 * LDS_BYTES <= 160 * 1024 && 2 * T <= RF_TWO_WG_MAXT) ? 2 : 1;
    static constexpr int WAVES_PER_SIMD = (WG_PER_CU * ((T + 63) / 64) + 3) / 4;  // register budget = 512 / this
    // leading odd pass: butterflies whose requests are in flight together (rf_first_odd): all of them at >= 168 VGPRs,
    // about 14 complex words' worth at the 128 of a 1024-thread workgroup
    static constexpr int LOAD_BATCH = WAVES_PER_SIMD <= 3 ? 64 : (LEAD_ >= 14 ? 1 : 14 / (LEAD_ > 1 ? LEAD_ : 1));
    // position of the value left in slot e after the last pass
    static __device__ __forceinline__ int out_pos(int t, int e) { return t + rf_last_slot(RLAST, E, e) * T; }
};

// Doubled shape over S1 (see RF_FOR_SHAPES2): 2 S1::E values per thread, the second half at positions + N1.
template <class S1_>
struct RfShape2 {
    using S1 = S1_;
    static constexpr bool DOUBLED = true;
    static constexpr int LEAD = S1::LEAD, K = S1::K + 1;
    static constexpr int N = 2 * S1::N, E = 2 * S1::E, T = S1::T;
    static constexpr int NSLOT = 2 * S1::NSLOT;
    static constexpr bool DUAL = S1::DUAL;
    static constexpr bool SWZ = S1::SWZ;
    static constexpr int XPAD = S1::XPAD;
    static constexpr int NL = S1::NL;
    static constexpr int LDS_BYTES = S1::LDS_BYTES;
    static constexpr int WG_PER_CU = 1;
    static constexpr int WAVES_PER_SIMD = ((T + 63) / 64 + 3) / 4;
    static constexpr int LOAD_BATCH = 2;  // (of each half transform)
    static __device__ __forceinline__ int out_pos(int t, int e)
    {
        return e < S1::E ? S1::out_pos(t, e) : S1::N + S1::out_pos(t, e - S1::E);
    }
};

// Hooks (round 4): a caller's own work interleaved with the passes of a row -- at<0>() runs when the row's inputs have been
// consumed (after the loads / the leading odd pass), at<P + 1>() after the butterflies of pass P: S::NP + 1 points in all.  A
// persistent kernel uses them to request the NEXT row's inputs in batches and bank them while this row's passes run: one
// workgroup per CU otherwise serialises load phase (a CU pulls ~10 B per clock), passes and store phase.
struct RfNoHook {
    template <int P>
    __device__ __forceinline__ void at() const
    {
    }
};

// Passes P.. of the power-of-two part; w1 holds the (already requested) twiddles of pass P.  The
// twiddles of pass P + 1 are requested BEFORE the LDS transpose of pass P so that their L2 latency
// hides behind it.
template <class S, int P, int NS, class Hook>
__device__ __forceinline__ void rf_passes(double (&re)[S::E], double (&im)[S::E], int t,
                                          const double2 *__restrict__ tw, double *lds,
                                          const double2 (&w1)[S::E / rf_radix(S::K, P)], Hook &hook)
{
    constexpr int R = rf_radix(S::K, P);
    rf_butterflies<R, S::E>(re, im, NS > 1, w1);
    hook.template at<P + 1>();
    if constexpr (P + 1 < S::NP) {
        constexpr int R2 = rf_radix(S::K, P + 1);
        double2 w2[S::E / R2];
        rf_load_twiddles<R2, S::E>(w2, t, S::T, S::N, NS * R, tw);
        rf_transpose<R, S::E, S::DUAL, S::SWZ, S::T, S::N, NS, S::xpad(R, NS), S::NL>(re, im, t, lds);
        rf_passes<S, P + 1, NS * R>(re, im, t, tw, lds, w2, hook);
    }
}
template <class S, int P, int NS>
__device__ __forceinline__ void rf_passes(double (&re)[S::E], double (&im)[S::E], int t,
                                          const double2 *__restrict__ tw, double *lds,
                                          const double2 (&w1)[S::E / rf_radix(S::K, P)])
{
    RfNoHook none;
    rf_passes<S, P, NS>(re, im, t, tw, lds, w1, none);
}

// Makes the thread index opaque to the optimiser again.  Callers put it between a transform and their
// epilogue: otherwise the epilogue's addresses (functions of t) are computed before the transform and
// stay live through it -- tens of VGPRs the passes need.
__device__ __forceinline__ void rf_opaque(int &t) { asm volatile("" : "+v"(t)); }

// The even (PAR = 0) / odd (PAR = 1) samples of a row as a load functor of the half-length transform (doubled shapes)
template <class Load, int PAR, int SLOT0, bool = rf_has_fetch<Load>::value>
struct RfHalfLoad {
    Load &ld;
    __device__ __forceinline__ double2 operator()(int pos, int slot) const { return ld(2 * pos + PAR, slot + SLOT0); }
};
template <class Load, int PAR, int SLOT0>
struct RfHalfLoad<Load, PAR, SLOT0, true> {
    Load &ld;
    __device__ __forceinline__ auto fetch(int pos, int slot) const { return ld.fetch(2 * pos + PAR, slot + SLOT0); }
    __device__ __forceinline__ double2 finish(const typename rf_raw_type<Load>::type &raw, int pos, int slot) const
    {
        return ld.finish(raw, 2 * pos + PAR, slot + SLOT0);
    }
    __device__ __forceinline__ double2 operator()(int pos, int slot) const { return ld(2 * pos + PAR, slot + SLOT0); }
};

// load -> passes; on return slot e holds the transform at position S::out_pos(t, e) as
// (re[e], im[e]) for the forward transform and as (im[e], re[e]) for the (unnormalised) inverse.
// NB: butterflies of the leading odd pass whose requests are in flight together (default: the shape's LOAD_BATCH; the doubled
// kernels, which hold half a transform next to the running one, ask for fewer)
template <class S, class Load, int NB = S::LOAD_BATCH, class Hook = RfNoHook>
__device__ __forceinline__ void rf_row_compute(const double2 *__restrict__ tw, Load &ld, bool inverse, double *lds,
                                               int &t_out, double (&re)[S::E], double (&im)[S::E], Hook &&hook = Hook{})
{
    if constexpr (S::DOUBLED) {
        using S1 = typename S::S1;
        double er[S1::E], ei[S1::E], orr[S1::E], oi[S1::E];
        RfHalfLoad<Load, 0, 0> ld_even{ld};
        RfHalfLoad<Load, 1, S1::NSLOT> ld_odd{ld};
        int t;
        rf_row_compute<S1, decltype(ld_even), S::LOAD_BATCH>(tw, ld_even, inverse, lds, t, er, ei);
        __builtin_amdgcn_sched_barrier(0);
        rf_row_compute<S1, decltype(ld_odd), S::LOAD_BATCH>(tw, ld_odd, inverse, lds, t, orr, oi);  // er / ei wait in registers or scratch
        __builtin_amdgcn_sched_barrier(0);
        rf_opaque(t);
        const double2 *__restrict__ tw2 = tw + S1::N;  // exp(-2 pi i k / N), k < N1
#pragma unroll
        for (int e = 0; e < S1::E; ++e) {
            const double2 w = tw2[S1::out_pos(t, e)];
            const double tr = orr[e] * w.x - oi[e] * w.y, ti = orr[e] * w.y + oi[e] * w.x;
            re[e] = er[e] + tr;
            im[e] = ei[e] + ti;
            re[e + S1::E] = er[e] - tr;
            im[e + S1::E] = ei[e] - ti;
        }
        t_out = t;
        return;
    } else {
    int t = threadIdx.x;
    // Opaque to the optimiser: otherwise every pass's (row-invariant) LDS and global addresses are
    // hoisted out of the caller's loops and kept live -- hundreds of VGPRs of loop invariants.
    asm volatile("" : "+v"(t));
    t_out = t;
    constexpr int R0 = rf_radix(S::K, 0);
    double2 w0[S::E / R0] = {};
    if constexpr (S::LEAD > 1) {
        rf_load_twiddles<R0, S::E>(w0, t, S::T, S::N, S::LEAD, tw);
        rf_first_odd<S::LEAD, S::E, S::DUAL, S::SWZ, S::T, S::N, S::NL, NB>(re, im, t, ld, inverse, lds);
    } else if constexpr (rf_has_fetch<Load>::value) {
        typename rf_raw_type<Load>::type raw[S::E];
#pragma unroll
        for (int e = 0; e < S::E; ++e) raw[e] = ld.fetch(t + e * S::T, e);
#pragma unroll
        for (int e = 0; e < S::E; ++e) {
            const double2 x = ld.finish(raw[e], t + e * S::T, e);
            re[e] = inverse ? x.y : x.x;
            im[e] = inverse ? x.x : x.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < S::E; ++e) {
            double2 x = ld(t + e * S::T, e);
            re[e] = inverse ? x.y : x.x;
            im[e] = inverse ? x.x : x.y;
        }
    }
    hook.template at<0>();
    rf_passes<S, 0, S::LEAD>(re, im, t, tw, lds, w0, hook);
    }
}

// Calls f(pos, slot) for every element the load functor of rf_row_compute<S> will be asked for by
// thread t, in the same order and with the same compile-time slot numbers (< 64); slots of a butterfly past the end of the
// row repeat positions of the thread's first butterfly (what rf_first_odd requests for them).
template <class S, class F>
__device__ __forceinline__ void rf_for_each_load(int t, F &&f)
{
    if constexpr (S::DOUBLED) {
        using S1 = typename S::S1;
        rf_for_each_load<S1>(t, [&](int pos, int slot) { f(2 * pos, slot); });
        rf_for_each_load<S1>(t, [&](int pos, int slot) { f(2 * pos + 1, slot + S1::NSLOT); });
    } else if constexpr (S::LEAD > 1) {
        constexpr int M = S::LEAD, IT = (S::E + M - 1) / M, nbf = S::N / M;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            // (branch-free, like the transform's own requests: a last butterfly past the end repeats the thread's first one)
            const int j = t + i * S::T, jc = (i == IT - 1 && IT * S::T > nbf) ? (j < nbf ? j : t) : j;
#pragma unroll
            for (int q = 0; q < M; ++q) f(jc + q * nbf, i * M + q);
        }
    } else {
#pragma unroll
        for (int e = 0; e < S::E; ++e) f(t + e * S::T, e);
    }
}

// The whole row: load -> passes -> store.
template <class S, class Load, class Store, int NB = S::LOAD_BATCH>
__device__ __forceinline__ void rf_row(const double2 *__restrict__ tw, Load &ld, Store &st, bool inverse, double *lds)
{
    double re[S::E], im[S::E];
    int t;
    rf_row_compute<S, Load, NB>(tw, ld, inverse, lds, t, re, im);
    rf_opaque(t);
    // No store before the last butterfly has consumed its twiddles: loads and stores share the in-order vmcnt, and behind
    // a (conditional) store the compiler waits vmcnt(0) for them -- i.e. for the store's own round trip to memory.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < S::E; ++e)
        st(S::out_pos(t, e), inverse ? make_double2(im[e], re[e]) : make_double2(re[e], im[e]));
}

#endif  // __HIPCC__

}  // namespace pfbhip
