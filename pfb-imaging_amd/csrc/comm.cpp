// comm.cpp -- band reduce over xGMI with RCCL (one process per GPU).
//
// Replaces the driver-side Python sums over bands of the reference
// (/root/reference/src/pfb_imaging/core/grid.py:430-446, core/deconv.py:320-321,
// operators/band_worker.py:305-308): each rank holds its band's (ncorr,nx,ny) f64 image on
// its GPU; sum-to-root is one ncclReduce.  xGMI is point-to-point, so a single large
// reduce (537 MB at 8192^2) lets RCCL spread over all links instead of many small ones.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>

#include "common.hpp"

#define PFB_NCCL(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess)                                                                          \
            throw std::runtime_error(pfbhip::strprintf("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), \
                                                       __FILE__, __LINE__));                            \
    } while (0)

struct pfbhip_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int nranks = 1, rank = 0;
    // persistent device staging of the host-array collectives (grow-only: no hipMalloc per call)
    pfbhip::DevBuf<double> stage_a, stage_b;
    double *barrier_word = nullptr;
    ~pfbhip_comm()
    {
        if (barrier_word) pfbhip::dev_free(barrier_word, sizeof(double));
        if (comm) ncclCommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

using namespace pfbhip;

static_assert(sizeof(ncclUniqueId) <= PFBHIP_UNIQUE_ID_BYTES, "unique id does not fit");

extern "C" {

int pfbhip_comm_unique_id(uint8_t *id)
{
    return guarded([&] {
        PFB_REQUIRE(id, "NULL argument");
        ncclUniqueId uid;
        PFB_NCCL(ncclGetUniqueId(&uid));
        std::memset(id, 0, PFBHIP_UNIQUE_ID_BYTES);
        std::memcpy(id, &uid, sizeof uid);
    });
}

int pfbhip_comm_create(const uint8_t *id, int nranks, int rank, pfbhip_comm **out)
{
    return guarded([&] {
        PFB_REQUIRE(id && out, "NULL argument");
        PFB_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
        std::unique_ptr<pfbhip_comm> c(new pfbhip_comm);
        c->nranks = nranks;
        c->rank = rank;
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        PFB_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        // RCCL allocates its buffers with hipMalloc behind our back and never sees the handles' cache of released blocks;
        // communicator creation is a collective (one rank cannot retry alone) and happens once per run, so the cache is
        // simply given back first
        (void)dev_cache_bytes(true);
        PFB_NCCL(ncclCommInitRank(&c->comm, nranks, uid, rank));
        *out = c.release();
    });
}

int pfbhip_comm_destroy(pfbhip_comm *c)
{
    return guarded([&] { delete c; });
}

int pfbhip_comm_reduce_sum(pfbhip_comm *c, const double *send_dev, double *recv_dev, int64_t count, int root)
{
    return guarded([&] {
        PFB_REQUIRE(c && send_dev && count >= 0, "NULL argument");
        PFB_REQUIRE(root >= 0 && root < c->nranks, "bad root %d", root);
        PFB_REQUIRE(recv_dev || c->rank != root, "root needs a receive buffer");
        // recvbuff is only significant on the root; other ranks may pass NULL (replaced by sendbuff)
        void *rb = recv_dev ? static_cast<void *>(recv_dev) : const_cast<double *>(send_dev);
        PFB_NCCL(ncclReduce(send_dev, rb, size_t(count), ncclDouble, ncclSum, root, c->comm, c->stream));
        PFB_HIP(hipStreamSynchronize(c->stream));
    });
}

int pfbhip_comm_allreduce_sum(pfbhip_comm *c, const double *send_dev, double *recv_dev, int64_t count)
{
    return guarded([&] {
        PFB_REQUIRE(c && send_dev && recv_dev && count >= 0, "NULL argument");
        PFB_NCCL(ncclAllReduce(send_dev, recv_dev, size_t(count), ncclDouble, ncclSum, c->comm, c->stream));
        PFB_HIP(hipStreamSynchronize(c->stream));
    });
}

int pfbhip_comm_allgather(pfbhip_comm *c, const double *send_dev, double *recv_dev, int64_t count)
{
    return guarded([&] {
        PFB_REQUIRE(c && send_dev && recv_dev && count >= 0, "NULL argument");
        PFB_NCCL(ncclAllGather(send_dev, recv_dev, size_t(count), ncclDouble, c->comm, c->stream));
        PFB_HIP(hipStreamSynchronize(c->stream));
    });
}

// Host-array forms (the cube-level methods of the band pool return numpy arrays): upload -> collective on the persistent
// staging buffers -> download, all on the communicator's stream.
int pfbhip_comm_allreduce_sum_host(pfbhip_comm *c, double *inout_host, int64_t count)
{
    return guarded([&] {
        PFB_REQUIRE(c && inout_host && count >= 0, "NULL argument");
        if (count == 0) return;
        c->stage_a.ensure(size_t(count));
        PFB_HIP(hipMemcpyAsync(c->stage_a.p, inout_host, size_t(count) * sizeof(double), hipMemcpyHostToDevice, c->stream));
        PFB_NCCL(ncclAllReduce(c->stage_a.p, c->stage_a.p, size_t(count), ncclDouble, ncclSum, c->comm, c->stream));
        PFB_HIP(hipMemcpyAsync(inout_host, c->stage_a.p, size_t(count) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PFB_HIP(hipStreamSynchronize(c->stream));
    });
}

int pfbhip_comm_reduce_sum_host(pfbhip_comm *c, const double *send_host, double *recv_host, int64_t count, int root)
{
    return guarded([&] {
        PFB_REQUIRE(c && send_host && count >= 0, "NULL argument");
        PFB_REQUIRE(root >= 0 && root < c->nranks, "bad root %d", root);
        PFB_REQUIRE(recv_host || c->rank != root, "root needs a receive buffer");
        if (count == 0) return;
        c->stage_a.ensure(size_t(count));
        PFB_HIP(hipMemcpyAsync(c->stage_a.p, send_host, size_t(count) * sizeof(double), hipMemcpyHostToDevice, c->stream));
        PFB_NCCL(ncclReduce(c->stage_a.p, c->stage_a.p, size_t(count), ncclDouble, ncclSum, root, c->comm, c->stream));
        if (c->rank == root)
            PFB_HIP(hipMemcpyAsync(recv_host, c->stage_a.p, size_t(count) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PFB_HIP(hipStreamSynchronize(c->stream));
    });
}

// recv_host holds nranks blocks of `count` doubles, block r = rank r's send_host
int pfbhip_comm_allgather_host(pfbhip_comm *c, const double *send_host, double *recv_host, int64_t count)
{
    return guarded([&] {
        PFB_REQUIRE(c && send_host && recv_host && count >= 0, "NULL argument");
        if (count == 0) return;
        c->stage_a.ensure(size_t(count));
        c->stage_b.ensure(size_t(count) * size_t(c->nranks));
        PFB_HIP(hipMemcpyAsync(c->stage_a.p, send_host, size_t(count) * sizeof(double), hipMemcpyHostToDevice, c->stream));
        PFB_NCCL(ncclAllGather(c->stage_a.p, c->stage_b.p, size_t(count), ncclDouble, c->comm, c->stream));
        PFB_HIP(hipMemcpyAsync(recv_host, c->stage_b.p, size_t(count) * size_t(c->nranks) * sizeof(double), hipMemcpyDeviceToHost,
                               c->stream));
        PFB_HIP(hipStreamSynchronize(c->stream));
    });
}

int pfbhip_comm_barrier(pfbhip_comm *c)
{
    return guarded([&] {
        PFB_REQUIRE(c, "NULL argument");
        // a 1-element all-reduce on a scratch word is RCCL's barrier
        if (c->barrier_word == nullptr) c->barrier_word = static_cast<double *>(dev_alloc(sizeof(double)));
        double *w = c->barrier_word;
        PFB_HIP(hipMemsetAsync(w, 0, sizeof(double), c->stream));
        PFB_NCCL(ncclAllReduce(w, w, 1, ncclDouble, ncclSum, c->comm, c->stream));
        PFB_HIP(hipStreamSynchronize(c->stream));
    });
}

}  // extern "C"
