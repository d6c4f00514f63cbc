// cbtest.cpp -- does rocFFT's (experimental) load/store callback path work for batched double-complex
// row transforms on this ROCm, and what does it cost?  (dev tool)
//   hipcc --offload-arch=gfx950 -O3 tools/cbtest.cpp -o tools/cbtest -lrocfft
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e = (x); if (e != 0) { printf("fail %s -> %d\n", #x, int(e)); exit(1); } } while (0)

struct CbData { double scale; const double2 *src; double2 *dst; };

__device__ double2 load_cb(double2 *data, size_t offset, void *cbdata, void *)
{
    const CbData *d = static_cast<const CbData *>(cbdata);
    double2 v = d->src[offset];
    v.x *= d->scale; v.y *= d->scale;
    return v;
}
__device__ void store_cb(double2 *data, size_t offset, double2 element, void *cbdata, void *)
{
    const CbData *d = static_cast<const CbData *>(cbdata);
    d->dst[offset] = element;
}
__device__ auto load_cb_ptr = load_cb;
__device__ auto store_cb_ptr = store_cb;

int main(int argc, char **argv)
{
    size_t n = argc > 1 ? atoll(argv[1]) : 1024, batch = argc > 2 ? atoll(argv[2]) : 8;
    CK(rocfft_setup());
    size_t tot = n * batch;
    std::vector<double2> h(tot);
    for (size_t i = 0; i < tot; ++i) h[i] = make_double2(sin(0.37 * i), cos(0.11 * i));
    double2 *in, *src, *dst, *ref;
    CK(hipMalloc(&in, tot * 16)); CK(hipMalloc(&src, tot * 16)); CK(hipMalloc(&dst, tot * 16)); CK(hipMalloc(&ref, tot * 16));
    CK(hipMemcpy(src, h.data(), tot * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(ref, h.data(), tot * 16, hipMemcpyHostToDevice));
    CK(hipMemset(in, 0, tot * 16)); CK(hipMemset(dst, 0, tot * 16));
    size_t len[1] = {n};
    rocfft_plan p; CK(rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 1, len, batch, nullptr));
    size_t ws = 0; CK(rocfft_plan_get_work_buffer_size(p, &ws));
    void *work = nullptr; if (ws) CK(hipMalloc(&work, ws));
    // reference: plain in-place on ref (scale 1)
    rocfft_execution_info i0; CK(rocfft_execution_info_create(&i0)); if (ws) CK(rocfft_execution_info_set_work_buffer(i0, work, ws));
    void *b0[1] = {ref};
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(rocfft_execute(p, b0, nullptr, i0)); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref, h.data(), tot * 16, hipMemcpyHostToDevice));
    CK(hipEventRecord(a, 0)); CK(rocfft_execute(p, b0, nullptr, i0)); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms0; CK(hipEventElapsedTime(&ms0, a, b));
    // callbacks: load reads src*scale, store writes dst
    CbData hd{1.0, src, dst}, *dd; CK(hipMalloc(&dd, sizeof hd)); CK(hipMemcpy(dd, &hd, sizeof hd, hipMemcpyHostToDevice));
    void *lp, *sp;
    CK(hipMemcpyFromSymbol(&lp, HIP_SYMBOL(load_cb_ptr), sizeof lp)); CK(hipMemcpyFromSymbol(&sp, HIP_SYMBOL(store_cb_ptr), sizeof sp));
    rocfft_execution_info i1; CK(rocfft_execution_info_create(&i1)); if (ws) CK(rocfft_execution_info_set_work_buffer(i1, work, ws));
    void *cbd[1] = {dd};
    CK(rocfft_execution_info_set_load_callback(i1, &lp, cbd, 0));
    CK(rocfft_execution_info_set_store_callback(i1, &sp, cbd, 0));
    void *b1[1] = {in};
    CK(rocfft_execute(p, b1, nullptr, i1)); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0)); CK(rocfft_execute(p, b1, nullptr, i1)); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms1; CK(hipEventElapsedTime(&ms1, a, b));
    std::vector<double2> r0(tot), r1(tot);
    CK(hipMemcpy(r0.data(), ref, tot * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), dst, tot * 16, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (size_t i = 0; i < tot; ++i) { err += (r0[i].x - r1[i].x) * (r0[i].x - r1[i].x) + (r0[i].y - r1[i].y) * (r0[i].y - r1[i].y); nrm += r0[i].x * r0[i].x + r0[i].y * r0[i].y; }
    printf("n=%zu batch=%zu work=%zu plain %.3f ms, callbacks %.3f ms, rel err %.2e\n", n, batch, ws, ms0, ms1, sqrt(err / nrm));
    return 0;
}
