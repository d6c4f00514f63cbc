// fftbench.cpp -- rocFFT micro-benchmark for the plane transforms of the gridder (dev tool).
//   hipcc --offload-arch=gfx950 -O3 tools/fftbench.cpp -o tools/fftbench -lrocfft
// Prints ms and effective GB/s (2 passes x (read+write) x 16 B per point for 2-D, 1 pass for 1-D).
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e = (x); if (e != 0) { printf("fail %s -> %d\n", #x, int(e)); exit(1); } } while (0)

static double time_plan(rocfft_plan plan, void *buf, void *out, int reps)
{
    size_t ws = 0;
    CK(rocfft_plan_get_work_buffer_size(plan, &ws));
    rocfft_execution_info info;
    CK(rocfft_execution_info_create(&info));
    void *work = nullptr;
    if (ws) { CK(hipMalloc(&work, ws)); CK(rocfft_execution_info_set_work_buffer(info, work, ws)); }
    void *in[1] = {buf}, *o[1] = {out};
    CK(rocfft_execute(plan, in, out ? o : nullptr, info));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) CK(rocfft_execute(plan, in, out ? o : nullptr, info));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    if (work) CK(hipFree(work));
    rocfft_execution_info_destroy(info);
    printf("   (work buffer %.2f GB)", ws / 1e9);
    return ms / reps;
}

int main(int argc, char **argv)
{
    CK(rocfft_setup());
    std::vector<size_t> sizes = {8192, 9216, 10240, 10368, 10752, 11520, 12288, 12800, 13824, 14336, 16384};
    if (argc > 1) { sizes.clear(); for (int i = 1; i < argc; ++i) sizes.push_back(size_t(atoll(argv[i]))); }
    for (size_t n : sizes) {
        void *buf; CK(hipMalloc(&buf, n * n * 16)); CK(hipMemset(buf, 0, n * n * 16));
        double bytes2d = 4.0 * n * n * 16;
        {   // full 2-D in place
            size_t len[2] = {n, n};
            rocfft_plan p; CK(rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 2, len, 1, nullptr));
            double ms = time_plan(p, buf, nullptr, 5);
            printf(" n=%zu 2D inplace      : %8.3f ms  %7.1f GB/s\n", n, ms, bytes2d / ms / 1e6);
            rocfft_plan_destroy(p);
        }
        {   // rows: n contiguous transforms of length n
            size_t len[1] = {n};
            rocfft_plan p; CK(rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 1, len, n, nullptr));
            double ms = time_plan(p, buf, nullptr, 5);
            printf(" n=%zu 1D rows  x%zu   : %8.3f ms  %7.1f GB/s\n", n, n, ms, 0.5 * bytes2d / ms / 1e6);
            rocfft_plan_destroy(p);
        }
        {   // columns: n strided transforms (stride n, dist 1)
            size_t len[1] = {n};
            rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
            size_t stride[1] = {n};
            CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, stride, 1, 1, stride, 1));
            rocfft_plan p; CK(rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 1, len, n, d));
            double ms = time_plan(p, buf, nullptr, 5);
            printf(" n=%zu 1D cols  x%zu   : %8.3f ms  %7.1f GB/s\n", n, n, ms, 0.5 * bytes2d / ms / 1e6);
            rocfft_plan_destroy(p); rocfft_plan_description_destroy(d);
        }
        CK(hipFree(buf));
    }
    return 0;
}
