// dev helper: cycles per f64 MFMA on the GPU box (v_mfma_f64_4x4x4_4b_f64, v_mfma_f64_16x16x4_f64), alone and with
// f64 VALU FMAs issued between them -- the numbers behind DESIGN.md's "why the scatter / gather stay on the VALU".
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma64.cpp -o tools/ubench_mfma64 && tools/ubench_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

// MODE 0: 4x4x4_4b, 8 independent accumulators; 1: 16x16x4, 4 independent accumulators; 2: MODE 0 + 2 VALU fma per MFMA;
// 3: VALU fma only (16 independent); 4: 4x4x4_4b, ONE accumulator (dependent chain: latency)
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, unsigned long long *clk)
{
    const int t = threadIdx.x;
    double a = 1.0 + t * 1e-3, b = 1.0 - t * 1e-3;
    double acc[8];
    double4_t acc4[4];
    double v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = i;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc4[i] = double4_t{double(i), 0.0, 1.0, 2.0};
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = t + i;
    const double m = 1.0000001, c = 1e-9;
    unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[i], 0, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
                v[2 * i] = fma(v[2 * i], m, c);
                v[2 * i + 1] = fma(v[2 * i + 1], m, c);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fma(v[i], m, c);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[0], 0, 0, 0);
        }
    }
    unsigned long long c1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + t] = s;
    if (blockIdx.x == 0 && t == 0) clk[0] = c1 - c0;
}

template <int MODE>
void run(const char *name, int wgs, int threads, int iters, double mfma_per_iter, double macs_per_mfma)
{
    double *out; unsigned long long *clk;
    CK(hipMalloc(&out, size_t(wgs) * 256 * 8)); CK(hipMalloc(&clk, 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, out, iters, clk);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, out, iters, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost));
    const double waves = double(wgs) * threads / 64.0;
    const double flops = waves * iters * mfma_per_iter * macs_per_mfma * 2.0;
    printf("%-44s wgs %5d x %4d thr  %8.3f ms  wave0: %7.1f clk / iter (%5.1f per instr)  %8.2f TFLOP/s\n", name, wgs, threads, ms,
           double(h[0]) / iters, double(h[0]) / iters / mfma_per_iter, flops / (ms * 1e-3) / 1e12);
    CK(hipFree(out)); CK(hipFree(clk));
}

int main()
{
    // one wave per SIMD (256 threads, 1 WG / CU), then 4 waves per SIMD
    for (int wgs : {256, 1024}) {
        run<0>("mfma_f64_4x4x4_4b x8 indep", wgs, 256, 2048, 8, 256);
        run<4>("mfma_f64_4x4x4_4b x8 dependent", wgs, 256, 2048, 8, 256);
        run<1>("mfma_f64_16x16x4 x4 indep", wgs, 256, 2048, 4, 1024);
        run<2>("mfma_f64_4x4x4_4b x8 + 16 v_fma_f64 (mfma flops)", wgs, 256, 2048, 8, 256);
        run<3>("v_fma_f64 x16", wgs, 256, 2048, 16, 64);
    }
    return 0;
}
