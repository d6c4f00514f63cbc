// dev helper: issue rate of v_fma_f64 vs v_pk_fma_f32 vs v_fma_f32 (3 waves per SIMD, 16 independent accumulators per lane),
// and a mix of the two -- the question behind the mixed-precision scatter.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_pk32.cpp -o tools/ubench_pk32 && tools/ubench_pk32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(768) k(double *out, int iters, unsigned long long *clk)
{
    const int t = threadIdx.x;
    double a[12];
    float2_t p[12];
    float f[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { a[i] = t + i; p[i] = float2_t{float(t), float(i)}; f[i] = t * 0.5f + i; }
    const double m = 1.0000001, c = 1e-9;
    const float2_t pm = {1.0000001f, 0.9999999f}, pc = {1e-9f, 2e-9f};
    unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(1.0000001f), "v"(1e-9f));
        } else if (MODE == 3) {  // 6 f64 + 6 pk
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            }
        } else if (MODE == 4) {  // pk with op_sel broadcast of the low half of src1
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(pm), "v"(pc));
        } else if (MODE == 5) {  // v_cvt_f32_f64
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
        }
    }
    unsigned long long c1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += a[i] + p[i].x + p[i].y + f[i];
    out[blockIdx.x * 768 + t] = s;
    if (blockIdx.x == 0 && t == 0) clk[0] = c1 - c0;
}

template <int MODE>
void run(const char *name, int per_iter)
{
    double *out; unsigned long long *clk;
    const int wgs = 256, iters = 20000;
    CK(hipMalloc(&out, size_t(wgs) * 768 * 8)); CK(hipMalloc(&clk, 16));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(768), 0, 0, out, iters, clk);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(768), 0, 0, out, iters, clk);
    CK(hipDeviceSynchronize());
    unsigned long long h;
    CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
    // 3 waves per SIMD share it: cycles per instruction per SIMD = cycles / (iters * per_iter * 3)
    printf("%-34s %6.2f cycles per wave-instruction on a SIMD shared by 3 waves\n", name, double(h) / (double(iters) * per_iter * 3));
    CK(hipFree(out)); CK(hipFree(clk));
}

int main()
{
    run<0>("v_fma_f64", 12);
    run<1>("v_pk_fma_f32", 12);
    run<2>("v_fma_f32", 12);
    run<3>("6 v_fma_f64 + 6 v_pk_fma_f32", 12);
    run<4>("v_pk_fma_f32 op_sel_hi:[0,1,1]", 12);
    run<5>("v_cvt_f32_f64", 12);
    return 0;
}
