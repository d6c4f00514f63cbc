// dev helper (GPU box): what device allocations of a plan's size cost -- hipMalloc / hipFree against the stream-ordered pool
// (hipMallocAsync with the release threshold raised so that freed blocks stay in the pool).
//   hipcc -O2 --offload-arch=gfx950 tools/malloc_probe.cpp -o tools/malloc_probe && tools/malloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (size_t gb : {1, 8, 27}) {
        const size_t n = gb << 30;
        for (int rep = 0; rep < 2; ++rep) {
            void *p = nullptr;
            double t0 = now();
            CK(hipMalloc(&p, n));
            double t1 = now();
            CK(hipMemsetAsync(p, 0, n, st));
            CK(hipStreamSynchronize(st));
            double t2 = now();
            CK(hipFree(p));
            double t3 = now();
            printf("hipMalloc %2zu GB: alloc %8.1f ms  memset %7.1f ms  free %8.1f ms\n", gb, t1 - t0, t2 - t1, t3 - t2);
        }
    }
    hipMemPool_t pool;
    CK(hipDeviceGetDefaultMemPool(&pool, 0));
    unsigned long long thr = ~0ull;
    CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    for (size_t gb : {1, 8, 27}) {
        const size_t n = gb << 30;
        for (int rep = 0; rep < 3; ++rep) {
            void *p = nullptr;
            double t0 = now();
            CK(hipMallocAsync(&p, n, st));
            CK(hipStreamSynchronize(st));
            double t1 = now();
            CK(hipMemsetAsync(p, 0, n, st));
            CK(hipStreamSynchronize(st));
            double t2 = now();
            CK(hipFreeAsync(p, st));
            CK(hipStreamSynchronize(st));
            double t3 = now();
            printf("pool      %2zu GB: alloc %8.1f ms  memset %7.1f ms  free %8.1f ms\n", gb, t1 - t0, t2 - t1, t3 - t2);
        }
    }
    return 0;
}
