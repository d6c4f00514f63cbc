// runtime.cpp -- device / memory / error plumbing of the C-ABI.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <utility>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace {
// PFBHIP_BACKTRACE=1: native backtrace on SIGABRT / SIGSEGV (the Python fault handler only shows the Python frames)
void pfbhip_fatal_signal(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "[pfbhip] fatal signal, native backtrace:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
}  // namespace

namespace pfbhip {
void debug_install_signals()
{
    static const bool on = [] { const char *e = std::getenv("PFBHIP_BACKTRACE"); return e != nullptr && e[0] == '1'; }();
    if (!on) return;
    signal(SIGABRT, pfbhip_fatal_signal);  // (re-armed at every API call: other libraries install handlers of their own)
    signal(SIGSEGV, pfbhip_fatal_signal);
}

static thread_local std::string t_last_error;

void set_last_error(const std::string &msg) { t_last_error = msg; }

static std::once_flag g_rocfft_once;
void rocfft_setup_once()
{
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
}

static std::atomic<int> g_pool_size{1};

static std::mutex g_lds_mutex;
static std::set<std::pair<int, const void *>> g_lds_done;
void allow_dynamic_lds(const void *kernel, int bytes)
{
    int dev = 0;
    PFB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_lds_mutex);
    if (g_lds_done.count({dev, kernel})) return;
    PFB_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_lds_done.insert({dev, kernel});
}

}  // namespace pfbhip

// ---- cache of released device blocks (see common.hpp) ----
namespace pfbhip {
namespace {
struct DevCache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> blocks;  // (device, bytes) -> block
    size_t cached = 0;
    size_t limit = ~size_t(0);   // "not set yet": 40 % of the device's memory at the first release (see bound())
    bool limit_from_env = false;
    bool poison = false;
    static constexpr size_t MIN_BLOCK = size_t(32) << 20;
    DevCache()
    {
        if (const char *e = std::getenv("PFBHIP_DEVCACHE_MB")) {
            limit = size_t(std::max(0ll, std::atoll(e))) << 20;
            limit_from_env = true;
        }
        if (const char *e = std::getenv("PFBHIP_DEVCACHE_POISON")) poison = e[0] == '1';
    }
    // Default bound: 40 % of the device's memory (115 GB of 288) -- one C5-size plan's blocks survive a plan change, and
    // allocators that never see this cache (rocFFT's plan internals, RCCL's buffers, other libraries in the process)
    // keep the larger share whatever has been released.  Called with mu held.
    size_t bound()
    {
        if (limit == ~size_t(0)) {
            size_t f = 0, t = 0;
            limit = hipMemGetInfo(&f, &t) == hipSuccess ? t / 10 * 4 : size_t(65536) << 20;
            (void)hipGetLastError();
        }
        return limit;
    }
    // (never destroyed: blocks still cached at process exit go with the context)
};
DevCache &dev_cache()
{
    static DevCache *c = new DevCache();
    return *c;
}
}  // namespace

size_t dev_cache_bytes(bool flush) noexcept
{
    DevCache &c = dev_cache();
    std::vector<void *> drop;
    size_t was;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        was = c.cached;
        if (flush) {
            for (auto &kv : c.blocks) drop.push_back(kv.second);
            c.blocks.clear();
            c.cached = 0;
        }
    }
    for (void *p : drop) (void)hipFree(p);
    return was;
}

void *dev_alloc(size_t bytes)
{
    DevCache &c = dev_cache();
    void *p = nullptr;
    int dev = 0;
    PFB_HIP(hipGetDevice(&dev));
    if (bytes >= DevCache::MIN_BLOCK) {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.blocks.find(std::make_pair(dev, bytes));
        if (it != c.blocks.end()) {
            p = it->second;
            c.blocks.erase(it);
            c.cached -= bytes;
        }
    }
    if (p == nullptr) {
        hipError_t err = hipMalloc(&p, bytes);
        if (err == hipErrorOutOfMemory) {  // give back what the cache holds and try once more
            (void)hipGetLastError();
            (void)dev_cache_bytes(true);
            err = hipMalloc(&p, bytes);
        }
        PFB_HIP(err);
    }
    if (c.poison) {
        hipError_t e = hipMemset(p, 0xFF, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            (void)hipFree(p);  // the block is neither handed out nor cached: give it back
            PFB_HIP(e);
        }
    }
    return p;
}

bool dev_cache_release_for_retry() noexcept
{
    return dev_cache_bytes(true) > 0;
}

void dev_free(void *p, size_t bytes) noexcept
{
    if (p == nullptr) return;
    DevCache &c = dev_cache();
    int dev = 0;
    if (bytes >= DevCache::MIN_BLOCK && c.limit > 0 && hipGetDevice(&dev) == hipSuccess) {
        hipPointerAttribute_t attr;
        // a block goes back to the cache of the device it lives on; like hipFree, not before the device is done with it
        if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.device == dev && hipDeviceSynchronize() == hipSuccess) {
            std::lock_guard<std::mutex> lk(c.mu);
            if (c.cached + bytes <= c.bound()) {
                c.blocks.emplace(std::make_pair(dev, bytes), p);
                c.cached += bytes;
                return;
            }
        }
        (void)hipGetLastError();
    }
    (void)hipFree(p);
}
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

int pfbhip_device_cache(size_t *cached_bytes, int flush)
{
    return guarded([&] {
        const size_t was = pfbhip::dev_cache_bytes(flush != 0);
        if (cached_bytes) *cached_bytes = was;
    });
}

int pfbhip_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    return guarded([&] {
        size_t f = 0, t = 0;
        PFB_HIP(hipMemGetInfo(&f, &t));
        f += pfbhip::dev_cache_bytes(false);  // released blocks waiting for reuse are available memory, not use
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

// Whole-array content hash of a HOST buffer (plan-cache keys of the stateless ducc0-style calls: the reference's calls
// are stateless, so a cached plan may be reused only for byte-identical inputs).  Position-dependent 64-bit mix of every
// 8-byte word, chunks hashed by a few threads (memory-bandwidth bound: ~2 ms for the 80 MB weights of a 1e7-visibility band).
uint64_t pfbhip_hash64(const void *data_host, size_t nbytes)
{
    const unsigned char *p = static_cast<const unsigned char *>(data_host);
    if (p == nullptr || nbytes == 0) return 0x9E3779B97F4A7C15ull;
    const size_t nwords = nbytes / 8;
    const size_t chunk = size_t(1) << 20;  // words per chunk (8 MiB)
    const size_t nchunks = (nwords + chunk - 1) / chunk;
    std::vector<uint64_t> part(nchunks ? nchunks : 1, 0);
    auto mix = [](uint64_t h) {
        h ^= h >> 33;
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 33;
        h *= 0xc4ceb9fe1a85ec53ull;
        h ^= h >> 33;
        return h;
    };
    auto work = [&](size_t c0, size_t c1) {
        for (size_t c = c0; c < c1; ++c) {
            const size_t w0 = c * chunk, w1 = std::min(nwords, w0 + chunk);
            uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, k = 2 * uint64_t(w0) + 1;
            size_t i = w0;
            for (; i + 4 <= w1; i += 4) {  // four independent lanes; odd multipliers k, k+2, ... make the sum position-dependent
                uint64_t a, b, cc, d;
                std::memcpy(&a, p + 8 * i, 8);
                std::memcpy(&b, p + 8 * i + 8, 8);
                std::memcpy(&cc, p + 8 * i + 16, 8);
                std::memcpy(&d, p + 8 * i + 24, 8);
                s0 += (a ^ 0x9E3779B97F4A7C15ull) * k;
                s1 += (b ^ 0x9E3779B97F4A7C15ull) * (k + 2);
                s2 += (cc ^ 0x9E3779B97F4A7C15ull) * (k + 4);
                s3 += (d ^ 0x9E3779B97F4A7C15ull) * (k + 6);
                k += 8;
            }
            for (; i < w1; ++i) {
                uint64_t a;
                std::memcpy(&a, p + 8 * i, 8);
                s0 += (a ^ 0x9E3779B97F4A7C15ull) * k;
                k += 2;
            }
            part[c] = mix(s0) ^ mix(s1 + 1) ^ mix(s2 + 2) ^ mix(s3 + 3);
        }
    };
    const size_t nthr = std::min<size_t>(std::max<size_t>(nchunks, 1), std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())));
    if (nthr <= 1 || nchunks <= 1) {
        work(0, nchunks);
    } else {
        // (no exception may cross the C ABI: a thread that cannot be created -- std::system_error under a process / thread
        // limit -- leaves its share to this thread)
        std::vector<std::thread> th;
        size_t started = 0;
        try {
            th.reserve(nthr);
            for (; started < nthr; ++started) th.emplace_back(work, nchunks * started / nthr, nchunks * (started + 1) / nthr);
        } catch (...) {
        }
        for (auto &t : th) t.join();
        if (started < nthr) work(nchunks * started / nthr, nchunks);
    }
    uint64_t h = mix(uint64_t(nbytes));
    for (size_t c = 0; c < nchunks; ++c) h = mix(h ^ part[c]) + 0x9E3779B97F4A7C15ull * (c + 1);
    uint64_t tail = 0;
    std::memcpy(&tail, p + 8 * nwords, nbytes - 8 * nwords);
    return mix(h ^ mix(tail + 0x51ull));
}

// Page-locked host memory for the RESULT arrays the Python layer hands back (dirty images, visibilities): a device-to-host
// copy into pageable memory runs at ~12 GB/s on the target host (the runtime stages it through a bounce buffer), into
// pinned memory at the PCIe rate (~55 GB/s).
int pfbhip_host_alloc(void **ptr_host, size_t bytes)
{
    return guarded([&] {
        PFB_REQUIRE(ptr_host, "NULL argument");
        *ptr_host = nullptr;
        if (bytes) PFB_HIP(hipHostMalloc(ptr_host, bytes, hipHostMallocDefault));
    });
}

int pfbhip_host_free(void *ptr_host)
{
    return guarded([&] {
        if (ptr_host) PFB_HIP(hipHostFree(ptr_host));
    });
}

int pfbhip_malloc(void **ptr_dev, size_t bytes)
{
    return guarded([&] {
        PFB_REQUIRE(ptr_dev, "NULL argument");
        *ptr_dev = nullptr;
        if (bytes) {
            hipError_t err = hipMalloc(ptr_dev, bytes);
            if (err == hipErrorOutOfMemory) {  // the handles' block cache gives way to the caller's buffers
                (void)hipGetLastError();
                (void)pfbhip::dev_cache_bytes(true);
                err = hipMalloc(ptr_dev, bytes);
            }
            PFB_HIP(err);
        }
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
        // (the calling thread's own stream, then wait: copies issued by the band pool's threads do not queue behind one
        // another on the device's single null stream, and the two PCIe directions overlap across bands)
        if (bytes) {
            PFB_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, hipStreamPerThread));
            PFB_HIP(hipStreamSynchronize(hipStreamPerThread));
        }
    });
}

int pfbhip_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    return guarded([&] {
        if (bytes) {
            PFB_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, hipStreamPerThread));
            PFB_HIP(hipStreamSynchronize(hipStreamPerThread));
        }
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
