// dev helper (GPU box): the gather's LDS read pattern in isolation -- 1024 threads per workgroup (one per CU, 110 KB of LDS),
// every lane 16 rows x 3 planes of ds_read_b128 per "visibility" with the row walk's addressing (lane b of a 16-lane row reads
// 16-byte slot b of a 48-cell tile row), with and without the 6 DPP-broadcast f64 FMAs per row.  Prints bytes per clock and CU.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.cpp -o tools/ubench_gather && tools/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int LS = 48, LL = LS * LS;

template <int I>
__device__ __forceinline__ void fmac_row_bcast(double &acc, double ku, double cell)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(ku), "v"(cell), "n"(I));
}

// MODE 0: reads only (one v_add per read keeps them alive); 1: reads + the 6 DPP FMAs per row; 2: the FMAs alone (cells in registers)
template <int MODE, int I>
__device__ __forceinline__ void steps(const char *base, double ku, double (&sr)[3], double (&si)[3], double2 (&c)[3])
{
    if constexpr (I < 16) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (MODE != 2) c[k] = *reinterpret_cast<const double2 *>(base + (size_t(k) * LL + size_t(I) * LS) * 16);
            if (MODE == 0) {
                sr[k] += c[k].x;
                si[k] += c[k].y;
            } else {
                fmac_row_bcast<I>(sr[k], ku, c[k].x);
                fmac_row_bcast<I>(si[k], ku, c[k].y);
            }
        }
        steps<MODE, I + 1>(base, ku, sr, si, c);
    }
}

template <int MODE>
__global__ void __launch_bounds__(1024) k(double *out, int rounds, unsigned long long *clk)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 3 * LL * 2; i += 1024) lds[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, b = lane & 15, g = lane >> 4;
    double tot = 0.0;
    unsigned long long c0 = clock64();
    for (int r = 0; r < rounds; ++r) {
        // a different first-tap cell per 16-lane row and round, as in the sorted visibility stream
        const int lu = (r * 7 + g * 5 + (threadIdx.x >> 6)) & 31, lv = (r * 3 + g * 11) & 31;
        const int cb = (b - lv) & 15;
        const char *base = reinterpret_cast<const char *>(lds) + (lu * LS + lv + cb) * 16;
        double ku = 1.0 + 1e-3 * b;
        asm volatile("s_nop 1" : "+v"(ku));
        double sr[3] = {0, 0, 0}, si[3] = {0, 0, 0};
        double2 c[3] = {make_double2(1.0, 2.0), make_double2(3.0, 4.0), make_double2(5.0, 6.0)};
        steps<MODE, 0>(base, ku, sr, si, c);
        tot += sr[0] + si[0] + sr[1] + si[1] + sr[2] + si[2];
    }
    unsigned long long c1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = tot;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int MODE>
void run(const char *name, int wgs, int rounds)
{
    double *out; unsigned long long *clk;
    CK(hipMalloc(&out, size_t(wgs) * 1024 * 8)); CK(hipMalloc(&clk, 16));
    const size_t lds = size_t(3) * LL * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), lds, 0, out, rounds, clk);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), lds, 0, out, rounds, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost));
    const double bytes_cu = MODE == 2 ? 0.0 : double(rounds) * 16 * 3 * 16 * 1024;  // per workgroup = per CU
    printf("%-34s wgs %4d  %8.3f ms  wg0: %9llu clk  %7.1f B/clk/CU (wg0 clock)  %6.1f clk per 64 visibilities\n", name, wgs, ms, h[0],
           bytes_cu / double(h[0]), double(h[0]) / (rounds * 16.0 * 4 / 64.0));
    CK(hipFree(out)); CK(hipFree(clk));
}

int main()
{
    for (int wgs : {256, 1024}) {
        run<0>("ds_read_b128 only", wgs, 2000);
        run<1>("ds_read_b128 + DPP fmac (gather)", wgs, 2000);
        run<2>("DPP fmac only", wgs, 2000);
    }
    return 0;
}
