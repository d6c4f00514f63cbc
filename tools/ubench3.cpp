// dev helper: do LDS exchange phases and VALU phases of DIFFERENT waves of one workgroup overlap when no barrier aligns them?
//   A: exchange | barrier | exchange-read | barrier | butterflies (all waves in phase, as the row FFT)   B: no barriers, odd waves
//   start with the VALU phase (wave-private LDS regions)   C: VALU only   D: LDS only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(640, 3) k(double *out, int iters)
{
    extern __shared__ double lds[];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    double a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = t + i;
    const double m = 1.0000001, c = 1e-9;
    double *mine = lds + wave * 2048;  // 64 lanes x 32 doubles
    auto valu = [&]() {
#pragma unroll
        for (int r = 0; r < 11; ++r)
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] = fma(a[i], m, c);
    };
    auto xchg = [&](bool barrier) {
        if (barrier) {
#pragma unroll
            for (int i = 0; i < 32; ++i) lds[t + 640 * i] = a[i];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] = lds[(t * 5 + 641 * i) % 20480];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) mine[lane + 64 * i] = a[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] = mine[(lane * 5 + 65 * i) & 2047];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    if (MODE == 1 && (wave & 1)) valu();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { xchg(true); valu(); }
        if (MODE == 1) { xchg(false); valu(); }
        if (MODE == 2) valu();
        if (MODE == 3) xchg(true);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i];
    out[blockIdx.x * 640 + t] = s;
}
template <int MODE>
void run(const char *name)
{
    double *out; CK(hipMalloc(&out, size_t(256) * 640 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(640), 163840, 0, out, 300);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(640), 163840, 0, out, 300);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %8.3f ms  (%.2f us per iteration)\n", name, ms, ms * 1e3 / 300);
    CK(hipFree(out));
}
int main()
{
    run<0>("A exchange + barriers + 352 FMA, in phase");
    run<1>("B no barriers, odd waves half a period off");
    run<2>("C 352 FMA only");
    run<3>("D exchange + barriers only");
    return 0;
}
