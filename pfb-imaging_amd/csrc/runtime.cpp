// runtime.cpp -- device / memory / error plumbing of the C-ABI.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <string>

#include "common.hpp"

namespace pfbhip {

static thread_local std::string t_last_error;

void set_last_error(const std::string &msg) { t_last_error = msg; }

static std::once_flag g_rocfft_once;
void rocfft_setup_once()
{
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
}

static std::atomic<int> g_pool_size{1};

}  // namespace pfbhip

using namespace pfbhip;

extern "C" {

const char *pfbhip_last_error(void) { return t_last_error.c_str(); }

int pfbhip_device_count(int *count)
{
    return guarded([&] {
        PFB_REQUIRE(count, "NULL argument");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            n = 0;
        }
        *count = n;
    });
}

int pfbhip_set_device(int device)
{
    return guarded([&] { PFB_HIP(hipSetDevice(device)); });
}

int pfbhip_get_device(int *device)
{
    return guarded([&] {
        PFB_REQUIRE(device, "NULL argument");
        PFB_HIP(hipGetDevice(device));
    });
}

int pfbhip_device_name(char *buf, size_t buflen)
{
    return guarded([&] {
        PFB_REQUIRE(buf && buflen > 0, "NULL argument");
        int dev = 0;
        PFB_HIP(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        PFB_HIP(hipGetDeviceProperties(&prop, dev));
        std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
        std::strncpy(buf, s.c_str(), buflen - 1);
        buf[buflen - 1] = 0;
    });
}

int pfbhip_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    return guarded([&] {
        size_t f = 0, t = 0;
        PFB_HIP(hipMemGetInfo(&f, &t));
        if (free_bytes) *free_bytes = f;
        if (total_bytes) *total_bytes = t;
    });
}

int pfbhip_resize_thread_pool(int nthreads)
{
    g_pool_size = nthreads > 0 ? nthreads : 1;
    return PFBHIP_OK;
}

int pfbhip_thread_pool_size(void) { return g_pool_size; }

int64_t pfbhip_good_size(int64_t n, int real) { return good_size(n, real != 0); }

int pfbhip_malloc(void **ptr_dev, size_t bytes)
{
    return guarded([&] {
        PFB_REQUIRE(ptr_dev, "NULL argument");
        *ptr_dev = nullptr;
        if (bytes) PFB_HIP(hipMalloc(ptr_dev, bytes));
    });
}

int pfbhip_free(void *ptr_dev)
{
    return guarded([&] {
        if (ptr_dev) PFB_HIP(hipFree(ptr_dev));
    });
}

int pfbhip_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes)
{
    return guarded([&] {
        if (bytes) PFB_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    });
}

int pfbhip_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    return guarded([&] {
        if (bytes) PFB_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    });
}

int pfbhip_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes)
{
    return guarded([&] {
        if (bytes) PFB_HIP(hipMemcpy(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice));
    });
}

int pfbhip_memset(void *dst_dev, int value, size_t bytes)
{
    return guarded([&] {
        if (bytes) PFB_HIP(hipMemset(dst_dev, value, bytes));
    });
}

int pfbhip_synchronize(void)
{
    return guarded([&] { PFB_HIP(hipDeviceSynchronize()); });
}

}  // extern "C"
