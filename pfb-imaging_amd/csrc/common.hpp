// common.hpp -- shared host-side plumbing for libpfbhip (error handling, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pfbhip.h"

namespace pfbhip {

struct InvalidArg : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void set_last_error(const std::string &msg);

inline std::string strprintf(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return std::string(buf);
}

#define PFB_HIP(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw std::runtime_error(pfbhip::strprintf("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                                       __FILE__, __LINE__));                              \
    } while (0)

#define PFB_REQUIRE(cond, ...)                                               \
    do {                                                                     \
        if (!(cond)) throw pfbhip::InvalidArg(pfbhip::strprintf(__VA_ARGS__)); \
    } while (0)

// Runs body(), converting C++ exceptions into C status codes.
void debug_install_signals();  // PFBHIP_BACKTRACE=1: native backtrace on SIGABRT / SIGSEGV (runtime.cpp)

template <class F>
int guarded(F &&body) noexcept
{
    debug_install_signals();
    try {
        body();
        return PFBHIP_OK;
    } catch (const InvalidArg &e) {
        set_last_error(e.what());
        return PFBHIP_ERR_INVALID;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return PFBHIP_ERR_RUNTIME;
    } catch (...) {
        set_last_error("unknown C++ exception");
        return PFBHIP_ERR_RUNTIME;
    }
}

// Device allocations of the handles go through a process-wide cache of released blocks (runtime.cpp): hipMalloc costs
// ~45 ms per GB on this platform (27 GB: 0.7-1.3 s; tools/malloc_probe.cpp) and a C5-size plan holds ~100 GB, so the second
// plan of a geometry -- the next band, the next major cycle -- used to spend 5 s allocating what the first one had just
// freed.  Blocks of >= 32 MiB are kept on release (exact-size reuse, per device, bounded by PFBHIP_DEVCACHE_MB, default
// 40 % of the device's memory; 0 disables) and everything cached is given back when an allocation fails -- one of ours
// (dev_alloc, pfbhip_malloc) or one made inside a library on our behalf (rocFFT plan creation, RCCL communicator
// creation: retry_after_cache_flush below).  pfbhip_mem_info counts cached blocks as free.  Like hipFree, releasing a block waits for
// the device first.  PFBHIP_DEVCACHE_POISON=1 fills every block handed out with 0xFF bytes (NaNs): no kernel may rely on
// fresh memory being zero.
void *dev_alloc(size_t bytes);
void dev_free(void *p, size_t bytes) noexcept;
size_t dev_cache_bytes(bool flush) noexcept;
bool dev_cache_release_for_retry() noexcept;  // flushes; true when something was given back

// Calls that allocate device memory inside a library (rocfft_plan_create, ncclCommInitRank ...) never see the cache, so
// an allocation failure there may be ours to cure: run `call` (returns true on success); on failure give the cached
// blocks back and run it once more.
template <class F>
bool retry_after_cache_flush(F &&call)
{
    if (call()) return true;
    (void)hipGetLastError();
    if (!dev_cache_release_for_retry()) return false;
    return call();
}

// Owning device buffer.
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count)
    {
        release();
        if (count) p = static_cast<T *>(dev_alloc(count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count)
    {
        if (count > n) alloc(count);
    }
    void release()
    {
        if (p) dev_free(p, n * sizeof(T));
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Opt a kernel in to more than 64 KiB of dynamic LDS.  The attribute belongs to the (device, kernel) pair, so it is set
// once per pair (thread-safe), not once per process: a process that builds handles on a second device (pfbhip_set_device,
// one rank per GPU under a multi-process launcher) needs it there too.
void allow_dynamic_lds(const void *kernel, int bytes);

int64_t good_size_2357(int64_t n);
int64_t good_size(int64_t n, bool real);

// Gauss-Legendre nodes/weights on [-1,1].
void gauss_legendre(int n, std::vector<double> &x, std::vector<double> &w);

}  // namespace pfbhip
