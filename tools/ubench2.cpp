// dev helper: do two 640-thread workgroups with 80 KiB of LDS each share a CU?  (time of 512 spinning workgroups vs 256)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void __launch_bounds__(640, 5) spin(double *out, long long ticks)
{
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
    out[blockIdx.x] = lds[(threadIdx.x + 1) % 640];
}
int main()
{
    double *out; CK(hipMalloc(&out, 8 * 4096));
    for (int lds : {81920, 81920 - 1280, 65536, 40960}) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int wgs : {256, 512, 768}) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(640), lds, 0, out, 10000);  // 100 us
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(640), lds, 0, out, 10000);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("lds %6d  wgs %4d  %.3f ms\n", lds, wgs, ms);
        }
    }
    return 0;
}
