// dev helper: issue-rate microbenchmarks on the GPU box (f64 FMA, int add, LDS write / read), chip-wide.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench.cpp -o tools/ubench && tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, unsigned long long *clk)
{
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = t + i;
    int ia[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ia[i] = t * i;
    const double m = 1.0000001, c = 1e-9;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fma(a[i], m, c);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(t));
        } else if (MODE == 2) {  // LDS b64 writes, unit stride
#pragma unroll
            for (int i = 0; i < 16; ++i) lds[t + 256 * i] = a[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 3) {  // LDS b64 reads
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] += lds[t + 256 * ((i + it) & 15)];
        } else if (MODE == 4) {  // LDS b128 writes
#pragma unroll
            for (int i = 0; i < 8; ++i) reinterpret_cast<double2 *>(lds)[t + 256 * i] = make_double2(a[2 * i], a[2 * i + 1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 5) {  // LDS b64 writes, stride 5 doubles (the radix-5 exchange)
#pragma unroll
            for (int i = 0; i < 16; ++i) lds[(t * 5 + i * 1280) % 4096 + (i & 3)] = a[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + ia[i];
    out[blockIdx.x * 256 + t] = s;
    if (blockIdx.x == 0 && t == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int MODE>
void run(const char *name, int wgs, int iters, double units_per_thread_iter, const char *unit)
{
    double *out; unsigned long long *clk;
    CK(hipMalloc(&out, size_t(wgs) * 256 * 8)); CK(hipMalloc(&clk, 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 32768, 0, out, iters, clk);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 32768, 0, out, iters, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double total = double(wgs) * 256 * iters * units_per_thread_iter;
    printf("%-28s wgs %5d  %8.3f ms  %10.2f %s   clock64/wall(100MHz) = %.1f MHz\n", name, wgs, ms, total / (ms * 1e-3) / 1e12, unit,
           double(h[0]) / double(h[1]) * 100.0);
    CK(hipFree(out)); CK(hipFree(clk));
}

int main()
{
    for (int wgs : {1024, 4096}) {
        run<0>("f64 fma (16 indep)", wgs, 4096, 16 * 2, "TFLOP/s");
        run<1>("v_add_u32", wgs, 4096, 16, "Tops/s");
        run<2>("ds_write_b64 unit stride", wgs, 2048, 16 * 8, "TB/s");
        run<3>("ds_read_b64 unit stride", wgs, 2048, 16 * 8, "TB/s");
        run<4>("ds_write_b128 unit stride", wgs, 2048, 16 * 8, "TB/s");
        run<5>("ds_write_b64 stride 5", wgs, 2048, 16 * 8, "TB/s");
    }
    return 0;
}
