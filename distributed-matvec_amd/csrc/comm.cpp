// comm.cpp -- RCCL side of the thin extern-"C" shim (lsk_comm_*): the inter-GPU exchange of the packet path and the
// PRIMME reductions, called by the C host (host.c).  One process per GPU; collectives go over xGMI.
//
// Replaces /root/reference/src/DistributedMatrixVector.chpl:313-449,638-661 (_RemoteBuffer.put: one-sided PUT + remote
// flag store per mailbox) by ONE grouped ncclSend/ncclRecv per round with the exact byte counts of the plan, and
// /root/reference/src/PRIMME.chpl:267-373 (atomic-buffer sum and copy-from-locale-0 broadcast behind three barriers)
// by ncclAllReduce / ncclBroadcast.
//
// librccl is resolved at run time (dlopen): single-GPU users of libls_amd.so do not need it, and inside a PyTorch
// process the already-loaded librccl.so.1 is reused, so the library and torch.distributed share one RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <pthread.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "lsk.h"

static thread_local char g_cerr[512] = "";
extern "C" char const *lsk_comm_last_error(void) { return g_cerr; }

namespace {
struct Api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Api g_api;

template <typename F> bool sym(F &slot, char const *name) {
    slot = reinterpret_cast<F>(dlsym(g_api.handle, name));
    if (!slot) snprintf(g_cerr, sizeof(g_cerr), "librccl: symbol %s not found", name);
    return slot != nullptr;
}

int load_api() {
    if (g_api.handle) return 0;
    char const *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (char const *n : names) {
        g_api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_api.handle) break;
    }
    if (!g_api.handle) {
        snprintf(g_cerr, sizeof(g_cerr), "cannot load librccl (%s)", dlerror());
        return -1;
    }
    bool ok = sym(g_api.GetUniqueId, "ncclGetUniqueId") && sym(g_api.CommInitRank, "ncclCommInitRank") &&
              sym(g_api.CommDestroy, "ncclCommDestroy") && sym(g_api.CommCount, "ncclCommCount") &&
              sym(g_api.GroupStart, "ncclGroupStart") &&
              sym(g_api.GroupEnd, "ncclGroupEnd") && sym(g_api.Send, "ncclSend") && sym(g_api.Recv, "ncclRecv") &&
              sym(g_api.AllReduce, "ncclAllReduce") && sym(g_api.Broadcast, "ncclBroadcast") &&
              sym(g_api.AllGather, "ncclAllGather") && sym(g_api.GetErrorString, "ncclGetErrorString");
    if (!ok) { g_api.handle = nullptr; return -1; }
    return 0;
}
} // namespace

#define NCCL_CHECK(expr)                                                                                   \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) {                                                                           \
            snprintf(g_cerr, sizeof(g_cerr), "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,            \
                     g_api.GetErrorString ? g_api.GetErrorString(r_) : "?");                               \
            return -1;                                                                                     \
        }                                                                                                  \
    } while (0)
#define HIP_CHECK(expr)                                                                                    \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            snprintf(g_cerr, sizeof(g_cerr), "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,            \
                     hipGetErrorString(e_));                                                               \
            return -1;                                                                                     \
        }                                                                                                  \
    } while (0)

// Loop-back group: `size` communicators inside ONE process on ONE device, one per host thread, rendezvousing through a
// barrier and moving the bytes with device-to-device copies.  Same call sequence, same buffers, same counts as the RCCL
// path -- what differs is only the transport.  RCCL refuses two ranks on one device, so this is how the multi-rank logic
// of the C host (set-up collectives, round pipeline, both exchange layouts) runs on a one-GPU box; the reference tests its
// multi-locale code the same way, by oversubscribing one machine.  Test infrastructure: ls_amd_comm_create_local.
struct LocalGroup {
    int size, refs;
    pthread_barrier_t bar;
    struct Post { void const *send; int64_t const *off; int64_t const *bytes; void *buf; } post[LSK_MAX_PARTS];
    double host[LSK_MAX_PARTS][512]; // small reductions (counts, dots, Gram-Schmidt coefficients)
};

struct lsk_comm {
    ncclComm_t comm;
    int size, rank;
    hipStream_t xstream;     // exchange stream: the collectives of round r overlap the kernels of round r +- 1
    hipEvent_t ready[2], done[2];
    LocalGroup *local;       // non-null: loop-back transport
};

extern "C" int lsk_comm_available(void) { return load_api() == 0; }

extern "C" int lsk_comm_unique_id(void *id128) {
    if (load_api() != 0) return -1;
    static_assert(sizeof(ncclUniqueId) == 128, "ls_amd.h promises 128 bytes");
    ncclUniqueId id;
    NCCL_CHECK(g_api.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

// stream + event pairs of a communicator; on failure everything created so far is released again
static int comm_resources(lsk_comm *c) {
    c->xstream = nullptr;
    for (int i = 0; i < 2; ++i) c->ready[i] = c->done[i] = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) return 0;
    snprintf(g_cerr, sizeof(g_cerr), "communicator stream / events: %s", hipGetErrorString(e));
    for (int i = 0; i < 2; ++i) {
        if (c->ready[i]) (void)hipEventDestroy(c->ready[i]);
        if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    }
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    return -1;
}

extern "C" int lsk_comm_create(lsk_comm **out, int size, int rank, void const *id128) {
    *out = nullptr;
    if (load_api() != 0) return -1;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    lsk_comm *c = new lsk_comm();
    c->size = size;
    c->rank = rank;
    c->local = nullptr;
    if (g_api.CommInitRank(&c->comm, size, id, rank) != ncclSuccess) {
        snprintf(g_cerr, sizeof(g_cerr), "ncclCommInitRank(size %d, rank %d) failed", size, rank);
        delete c;
        return -1;
    }
    if (comm_resources(c) != 0) {
        if (g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
        delete c;
        return -1;
    }
    *out = c;
    return 0;
}

extern "C" int lsk_comm_create_local(lsk_comm **out, int size) {
    if (size < 1 || size > LSK_MAX_PARTS) { snprintf(g_cerr, sizeof(g_cerr), "bad group size"); return -1; }
    LocalGroup *g = new LocalGroup();
    g->size = g->refs = size;
    pthread_barrier_init(&g->bar, nullptr, (unsigned)size);
    for (int r = 0; r < size; ++r) {
        lsk_comm *c = new lsk_comm();
        c->comm = nullptr; c->size = size; c->rank = r; c->local = g;
        if (comm_resources(c) != 0) { // unwind: the ranks created so far, the barrier, the group
            delete c;
            for (int q = 0; q < r; ++q) {
                for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(out[q]->ready[i]); (void)hipEventDestroy(out[q]->done[i]); }
                (void)hipStreamDestroy(out[q]->xstream);
                delete out[q];
                out[q] = nullptr;
            }
            pthread_barrier_destroy(&g->bar);
            delete g;
            return -1;
        }
        out[r] = c;
    }
    return 0;
}
static pthread_mutex_t g_local_lock = PTHREAD_MUTEX_INITIALIZER;

extern "C" void lsk_comm_destroy(lsk_comm *c) {
    if (!c) return;
    (void)hipStreamSynchronize(c->xstream);
    if (c->local) {
        pthread_mutex_lock(&g_local_lock);
        const bool last = --c->local->refs == 0;
        pthread_mutex_unlock(&g_local_lock);
        if (last) { pthread_barrier_destroy(&c->local->bar); delete c->local; }
        for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(c->ready[i]); (void)hipEventDestroy(c->done[i]); }
        (void)hipStreamDestroy(c->xstream);
        delete c;
        return;
    }
    if (g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
    for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(c->ready[i]); (void)hipEventDestroy(c->done[i]); }
    (void)hipStreamDestroy(c->xstream);
    delete c;
}
extern "C" int lsk_comm_size(lsk_comm const *c) { return c->size; }
extern "C" int lsk_comm_rank(lsk_comm const *c) { return c->rank; }
// the communicator size as RCCL itself reports it (ncclCommCount); 0 for a loop-back group, -1 on error
extern "C" int lsk_comm_rccl_count(lsk_comm const *c) {
    if (c->local) return 0;
    int n = -1;
    if (!g_api.CommCount || g_api.CommCount(c->comm, &n) != ncclSuccess) return -1;
    return n;
}

// in-place reductions / broadcast / gather on `stream` (device buffers)
extern "C" int lsk_comm_allreduce(lsk_comm *c, void *d_buf, int64_t count, int dtype /* 0 f64, 1 f32, 2 i64 */,
                                  int op /* 0 sum, 1 max */, void *stream) {
    if (c->local) { // every rank stages its (small) buffer on the host, everybody reduces everybody's copy
        LocalGroup *g = c->local;
        const size_t es = dtype == 1 ? 4 : 8;
        if ((size_t)count * es > sizeof(g->host[0])) { snprintf(g_cerr, sizeof(g_cerr), "loop-back all-reduce: at most %zu bytes", sizeof(g->host[0])); return -1; }
        // (a failing copy must not skip a barrier: the other ranks' threads would wait for ever)
        hipError_t e0 = hipStreamSynchronize((hipStream_t)stream);
        if (e0 == hipSuccess) e0 = hipMemcpy(g->host[c->rank], d_buf, (size_t)count * es, hipMemcpyDeviceToHost);
        pthread_barrier_wait(&g->bar);
        if (e0 != hipSuccess) { pthread_barrier_wait(&g->bar); snprintf(g_cerr, sizeof(g_cerr), "loop-back all-reduce: %s", hipGetErrorString(e0)); return -1; }
        double acc[512];
        memcpy(acc, g->host[0], (size_t)count * es);
        for (int r = 1; r < g->size; ++r)
            for (int64_t k = 0; k < count; ++k) {
                if (dtype == 0) { double v = g->host[r][k]; acc[k] = op == 0 ? acc[k] + v : (v > acc[k] ? v : acc[k]); }
                else if (dtype == 1) { float v = ((float *)g->host[r])[k], &a = ((float *)acc)[k]; a = op == 0 ? a + v : (v > a ? v : a); }
                else { int64_t v = ((int64_t *)g->host[r])[k], &a = ((int64_t *)acc)[k]; a = op == 0 ? a + v : (v > a ? v : a); }
            }
        pthread_barrier_wait(&g->bar); // everybody has read every copy
        HIP_CHECK(hipMemcpy(d_buf, acc, (size_t)count * es, hipMemcpyHostToDevice));
        return 0;
    }
    const ncclDataType_t t = dtype == 0 ? ncclDouble : dtype == 1 ? ncclFloat : ncclInt64;
    NCCL_CHECK(g_api.AllReduce(d_buf, d_buf, (size_t)count, t, op == 0 ? ncclSum : ncclMax, c->comm, (hipStream_t)stream));
    return 0;
}
extern "C" int lsk_comm_broadcast(lsk_comm *c, void *d_buf, int64_t bytes, int root, void *stream) {
    if (c->local) {
        LocalGroup *g = c->local;
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        g->post[c->rank].buf = d_buf;
        pthread_barrier_wait(&g->bar);
        hipError_t e0 = hipSuccess;
        if (c->rank != root) e0 = hipMemcpy(d_buf, g->post[root].buf, (size_t)bytes, hipMemcpyDeviceToDevice);
        pthread_barrier_wait(&g->bar);
        if (e0 != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "loop-back broadcast: %s", hipGetErrorString(e0)); return -1; }
        return 0;
    }
    NCCL_CHECK(g_api.Broadcast(d_buf, d_buf, (size_t)bytes, ncclChar, root, c->comm, (hipStream_t)stream));
    return 0;
}
extern "C" int lsk_comm_allgather(lsk_comm *c, void const *d_send, void *d_recv, int64_t bytes_per_rank, void *stream) {
    if (c->local) {
        LocalGroup *g = c->local;
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        g->post[c->rank].send = d_send;
        pthread_barrier_wait(&g->bar);
        hipError_t e0 = hipSuccess;
        for (int r = 0; r < g->size && e0 == hipSuccess; ++r)
            e0 = hipMemcpy((char *)d_recv + (size_t)r * (size_t)bytes_per_rank, g->post[r].send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice);
        pthread_barrier_wait(&g->bar);
        if (e0 != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "loop-back all-gather: %s", hipGetErrorString(e0)); return -1; }
        return 0;
    }
    NCCL_CHECK(g_api.AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclChar, c->comm, (hipStream_t)stream));
    return 0;
}

// all-to-all-v of bytes: segment d of the send buffer goes to rank d, segment s of the receive buffer comes from rank
// s; one grouped send/recv, i.e. every pair of GPUs talks over its own xGMI link at the same time.  Issued on the
// exchange stream between two events: `slot` (0/1) selects the event pair of the double-buffered pipeline.
//   compute stream: ... generate(r) | record ready[slot]                      wait done[slot] | scatter(r) ...
//   exchange stream:                  wait ready[slot] | grouped send/recv | record done[slot]
extern "C" int lsk_comm_exchange_begin(lsk_comm *c, int slot, void *compute_stream) {
    HIP_CHECK(hipEventRecord(c->ready[slot], (hipStream_t)compute_stream));
    HIP_CHECK(hipStreamWaitEvent(c->xstream, c->ready[slot], 0));
    return 0;
}
// K segments per peer in one call: segment k for / from peer p is entry [k * size + p] of the offset / byte arrays (a
// segment of 0 bytes is skipped on both sides; the k-th message of a pair matches the k-th).
extern "C" int lsk_comm_alltoallv_multi_on(lsk_comm *c, void *stream, int K, void const *d_send, int64_t const *send_off,
                                           int64_t const *send_bytes, void *d_recv, int64_t const *recv_off,
                                           int64_t const *recv_bytes) {
    hipStream_t s = (hipStream_t)stream;
    if (K < 1) return 0;
    if (c->local) { // every rank posts its send segments, then copies what is addressed to it out of its peers' buffers
        LocalGroup *g = c->local;
        HIP_CHECK(hipStreamSynchronize(s));
        g->post[c->rank].send = d_send; g->post[c->rank].off = send_off; g->post[c->rank].bytes = send_bytes;
        pthread_barrier_wait(&g->bar);
        for (int k = 0; k < K; ++k)
            for (int src = 0; src < g->size; ++src) {
                const int64_t want = recv_bytes[(size_t)k * g->size + src];
                const int64_t have = src == c->rank ? want : g->post[src].bytes[(size_t)k * g->size + c->rank];
                if (src == c->rank) continue;
                if (have != want) {
                    snprintf(g_cerr, sizeof(g_cerr), "loop-back all-to-all-v: rank %d sends %lld bytes (segment %d) to rank %d, which expects %lld",
                             src, (long long)have, k, c->rank, (long long)want);
                    pthread_barrier_wait(&g->bar);
                    return -1;
                }
                if (want == 0) continue;
                const hipError_t e1 = hipMemcpyAsync((char *)d_recv + recv_off[(size_t)k * g->size + src],
                                                     (char const *)g->post[src].send + g->post[src].off[(size_t)k * g->size + c->rank],
                                                     (size_t)want, hipMemcpyDeviceToDevice, s);
                if (e1 != hipSuccess) { pthread_barrier_wait(&g->bar); snprintf(g_cerr, sizeof(g_cerr), "loop-back all-to-all-v: %s", hipGetErrorString(e1)); return -1; }
            }
        const hipError_t e2 = hipStreamSynchronize(s);
        pthread_barrier_wait(&g->bar); // nobody reuses a send buffer before every peer has read it
        if (e2 != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "loop-back all-to-all-v: %s", hipGetErrorString(e2)); return -1; }
        return 0;
    }
    // one group per segment index: every group is the plain all-to-all-v pattern (at most one send and one receive per peer),
    // segment k of a pair always meets segment k.  (All K x (P - 1) operations in ONE group would overlap the segments of a pair
    // as well, but nothing here can test RCCL with more than one rank, and several operations per peer inside a group is the
    // less travelled path; the K - 1 extra launches cost ~20 us each against milliseconds of exchange.)
    for (int k = 0; k < K; ++k) {
        bool any = false;
        for (int p = 0; p < c->size && !any; ++p)
            any = p != c->rank && (send_bytes[(size_t)k * c->size + p] > 0 || recv_bytes[(size_t)k * c->size + p] > 0);
        if (!any) continue;
        NCCL_CHECK(g_api.GroupStart());
        for (int step = 1; step < c->size; ++step) {
            const int dst = (c->rank + step) % c->size, src = (c->rank - step + c->size) % c->size;
            const size_t ks = (size_t)k * c->size + dst, kr = (size_t)k * c->size + src;
            if (send_bytes[ks] > 0)
                NCCL_CHECK(g_api.Send((char const *)d_send + send_off[ks], (size_t)send_bytes[ks], ncclChar, dst, c->comm, s));
            if (recv_bytes[kr] > 0)
                NCCL_CHECK(g_api.Recv((char *)d_recv + recv_off[kr], (size_t)recv_bytes[kr], ncclChar, src, c->comm, s));
        }
        NCCL_CHECK(g_api.GroupEnd());
    }
    return 0;
}
extern "C" int lsk_comm_alltoallv_on(lsk_comm *c, void *stream, void const *d_send, int64_t const *send_off,
                                     int64_t const *send_bytes, void *d_recv, int64_t const *recv_off,
                                     int64_t const *recv_bytes) {
    return lsk_comm_alltoallv_multi_on(c, stream, 1, d_send, send_off, send_bytes, d_recv, recv_off, recv_bytes);
}
extern "C" int lsk_comm_alltoallv(lsk_comm *c, void const *d_send, int64_t const *send_off, int64_t const *send_bytes,
                                  void *d_recv, int64_t const *recv_off, int64_t const *recv_bytes) {
    return lsk_comm_alltoallv_on(c, (void *)c->xstream, d_send, send_off, send_bytes, d_recv, recv_off, recv_bytes);
}
extern "C" int lsk_comm_exchange_end(lsk_comm *c, int slot) {
    HIP_CHECK(hipEventRecord(c->done[slot], c->xstream));
    return 0;
}
extern "C" int lsk_comm_exchange_wait(lsk_comm *c, int slot, void *compute_stream) {
    HIP_CHECK(hipStreamWaitEvent((hipStream_t)compute_stream, c->done[slot], 0));
    return 0;
}
