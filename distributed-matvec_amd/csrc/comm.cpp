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

#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

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
// ---- never hang (VERDICT r5 #1b) ---------------------------------------------------------------------------------------
// A mismatched send / receive, or a peer that died, makes RCCL wait for ever -- and the host then sits in a stream
// synchronisation for ever.  Two guards, both with the deadline LS_AMD_COMM_WATCHDOG_S (seconds; default 300; 0 = off):
//   * RCCL communicators: a WATCHDOG THREAD per communicator.  Every collective and every exchange leaves an event behind it
//     and arms an item {event, when, what}; the thread queries the armed events twice a second.  An event that has not
//     completed by the deadline ends the process -- message on stderr (rank, what was in flight, for how long), exit code
//     86 -- which is what turns a hung multi-GPU job into a failed one (torch's own NCCL watchdog does the same).
//     lsk_comm_wait() is the polite form for callers that would rather get rc != 0: it polls a stream with the same deadline.
//   * loop-back groups (host threads on one device): the rendezvous is a barrier WITH A DEADLINE; a rank that never shows
//     up makes every waiting rank return an error naming how many arrived, and the group stays broken (later collectives
//     fail at once) instead of leaving threads in pthread_barrier_wait.
static double watchdog_seconds() {
    char const *e = getenv("LS_AMD_COMM_WATCHDOG_S");
    if (!e || !*e) return 300.0;
    const double v = atof(e);
    return v > 0 ? v : 0.0;
}
static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct DeadlineBarrier {
    std::mutex m;
    std::condition_variable cv;
    int size = 1, waiting = 0;
    uint64_t generation = 0;
    bool broken = false;
    // true: all `size` ranks arrived.  false: the deadline passed (or the group was broken before): *arrived = how many were here
    bool wait(int *arrived) {
        std::unique_lock<std::mutex> lk(m);
        if (broken) { *arrived = waiting; return false; }
        const uint64_t gen = generation;
        if (++waiting == size) { waiting = 0; ++generation; cv.notify_all(); return true; }
        const double limit = watchdog_seconds();
        auto pred = [&] { return generation != gen || broken; };
        if (limit > 0) {
            if (!cv.wait_for(lk, std::chrono::duration<double>(limit), pred)) { broken = true; *arrived = waiting; cv.notify_all(); return false; }
        } else cv.wait(lk, pred);
        if (generation == gen) { *arrived = waiting; return false; } // woken by a peer that gave up
        return true;
    }
};

struct LocalGroup {
    int size, refs;
    DeadlineBarrier bar;
    struct Post { void const *send; int64_t const *off; int64_t const *bytes; void *buf; } post[LSK_MAX_PARTS];
    double host[LSK_MAX_PARTS][512]; // small reductions (counts, dots, Gram-Schmidt coefficients)
};
// rendezvous of the loop-back ranks; on a timeout the message names the collective and the ranks that made it
#define LOCAL_BARRIER(g, what)                                                                                         \
    do {                                                                                                               \
        int arrived_ = 0;                                                                                              \
        if (!(g)->bar.wait(&arrived_)) {                                                                               \
            snprintf(g_cerr, sizeof(g_cerr), "loop-back %s: rank %d gave up after %.0f s at the rendezvous: %d of %d ranks arrived "  \
                     "(a rank failed or left; LS_AMD_COMM_WATCHDOG_S)", what, c->rank, watchdog_seconds(), arrived_, (g)->size);       \
            return -1;                                                                                                 \
        }                                                                                                              \
    } while (0)

struct Watchdog {
    struct Item { hipEvent_t ev = nullptr; bool armed = false; double since = 0; uint64_t seq = 0; char what[200] = ""; };
    enum { N_ITEMS = 3 }; // exchange slots 0 / 1, collectives on the caller's stream
    Item item[N_ITEMS];
    std::mutex m;
    std::condition_variable cv;
    std::thread th;
    bool stop = false;
    int device = 0, rank = 0, size = 1;
    double limit = 0;
    std::atomic<uint64_t> fired{0};
};

struct lsk_comm {
    ncclComm_t comm;
    int size, rank;
    hipStream_t xstream;     // exchange stream: the collectives of round r overlap the kernels of round r +- 1
    hipEvent_t ready[2], done[2];
    hipEvent_t xt0[2], xt1[2]; // timing pair around the exchange of a slot, on the exchange stream (lsk_comm_exchange_ms)
    bool xt_recorded[2];
    LocalGroup *local;       // non-null: loop-back transport
    Watchdog *wd;            // RCCL communicators with a deadline; else null
    char tag[160];           // what the next exchange is (set by the C host: round, bytes), for the watchdog's message
};

static void watchdog_loop(Watchdog *w) {
    (void)hipSetDevice(w->device);
    std::unique_lock<std::mutex> lk(w->m);
    while (!w->stop) {
        w->cv.wait_for(lk, std::chrono::milliseconds(500));
        if (w->stop) break;
        for (auto &it : w->item) {
            if (!it.armed) continue;
            const hipError_t q = hipEventQuery(it.ev);
            if (q == hipSuccess) { it.armed = false; continue; }
            if (q != hipErrorNotReady) (void)hipGetLastError();
            const double waited = now_s() - it.since;
            if (q == hipErrorNotReady && waited < w->limit) continue;
            fprintf(stderr, "\nlibls_amd watchdog: rank %d of %d: %s -- issued %.1f s ago and not complete (%s; deadline LS_AMD_COMM_WATCHDOG_S = %.0f s).\n"
                            "A peer has died or the send / receive byte counts of two ranks disagree; RCCL would wait for ever.  Ending the process (exit code 86).\n",
                    w->rank, w->size, it.what, waited, q == hipErrorNotReady ? "still pending" : hipGetErrorString(q), w->limit);
            fflush(stderr);
            w->fired++;
            _exit(86);
        }
    }
}
static void watchdog_start(lsk_comm *c) {
    c->wd = nullptr;
    const double limit = watchdog_seconds();
    if (limit <= 0 || c->local) return;
    Watchdog *w = new Watchdog();
    w->limit = limit; w->rank = c->rank; w->size = c->size;
    (void)hipGetDevice(&w->device);
    w->item[0].ev = c->done[0];
    w->item[1].ev = c->done[1];
    if (hipEventCreateWithFlags(&w->item[2].ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); delete w; return; }
    w->th = std::thread(watchdog_loop, w);
    c->wd = w;
}
static void watchdog_stop(lsk_comm *c) {
    Watchdog *w = c->wd;
    if (!w) return;
    { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    (void)hipEventDestroy(w->item[2].ev);
    delete w;
    c->wd = nullptr;
}
// the event of item `i` has just been recorded behind `what`
static void watchdog_arm(lsk_comm *c, int i, char const *what) {
    Watchdog *w = c->wd;
    if (!w) return;
    std::lock_guard<std::mutex> lk(w->m);
    Watchdog::Item &it = w->item[i];
    it.armed = true; it.since = now_s(); ++it.seq;
    snprintf(it.what, sizeof(it.what), "%s", what);
}
// collectives on the caller's stream: one more event behind them (item 2)
static int watchdog_after_collective(lsk_comm *c, hipStream_t s, char const *what) {
    Watchdog *w = c->wd;
    if (!w) return 0;
    {   // (the event is re-recorded: the deadline then counts from the latest collective, which completes after the earlier ones)
        std::lock_guard<std::mutex> lk(w->m);
        if (hipEventRecord(w->item[2].ev, s) != hipSuccess) { (void)hipGetLastError(); return 0; }
    }
    watchdog_arm(c, 2, what);
    return 0;
}
extern "C" void lsk_comm_set_tag(lsk_comm *c, char const *what) { snprintf(c->tag, sizeof(c->tag), "%s", what ? what : ""); }
// test hook: the event of exchange slot 0 is recorded behind a kernel-free stall of `seconds` on the exchange stream (a host
// callback that sleeps), armed like an exchange -- the watchdog must end the process when the deadline is shorter
static void stall_cb(void *arg) { usleep((useconds_t)(uintptr_t)arg); }
extern "C" int lsk_comm_test_stall(lsk_comm *c, double seconds) {
    if (!c->wd) { snprintf(g_cerr, sizeof(g_cerr), "no watchdog on this communicator (loop-back, or LS_AMD_COMM_WATCHDOG_S=0)"); return -1; }
    HIP_CHECK(hipLaunchHostFunc(c->xstream, stall_cb, (void *)(uintptr_t)(seconds * 1e6)));
    HIP_CHECK(hipEventRecord(c->done[0], c->xstream));
    watchdog_arm(c, 0, "test stall on the exchange stream (lsk_comm_test_stall)");
    return 0;
}
// polite form: wait until `stream` AND the exchange stream have drained, or the deadline passes (rc -1, message says what
// was in flight).  timeout_s <= 0: LS_AMD_COMM_WATCHDOG_S.
extern "C" int lsk_comm_wait(lsk_comm *c, void *stream, double timeout_s) {
    const double limit = timeout_s > 0 ? timeout_s : watchdog_seconds();
    const double t0 = now_s();
    hipStream_t ss[2] = {(hipStream_t)stream, c->xstream};
    for (int i = 0; i < 2; ++i) {
        for (;;) {
            const hipError_t q = hipStreamQuery(ss[i]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { snprintf(g_cerr, sizeof(g_cerr), "lsk_comm_wait: %s", hipGetErrorString(q)); (void)hipGetLastError(); return -1; }
            if (limit > 0 && now_s() - t0 > limit) {
                snprintf(g_cerr, sizeof(g_cerr), "rank %d of %d: the %s stream has not drained after %.0f s (last exchange: %s): a peer died or send / receive "
                         "counts disagree", c->rank, c->size, i == 0 ? "compute" : "exchange", limit, c->tag[0] ? c->tag : "none tagged");
                return -1;
            }
            usleep(50);
        }
    }
    return 0;
}

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
    for (int i = 0; i < 2; ++i) { c->ready[i] = c->done[i] = c->xt0[i] = c->xt1[i] = nullptr; c->xt_recorded[i] = false; }
    hipError_t e = hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&c->xt0[i]);
        if (e == hipSuccess) e = hipEventCreate(&c->xt1[i]);
    }
    if (e == hipSuccess) return 0;
    snprintf(g_cerr, sizeof(g_cerr), "communicator stream / events: %s", hipGetErrorString(e));
    for (int i = 0; i < 2; ++i) {
        if (c->ready[i]) (void)hipEventDestroy(c->ready[i]);
        if (c->done[i]) (void)hipEventDestroy(c->done[i]);
        if (c->xt0[i]) (void)hipEventDestroy(c->xt0[i]);
        if (c->xt1[i]) (void)hipEventDestroy(c->xt1[i]);
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
    c->tag[0] = 0;
    watchdog_start(c);
    *out = c;
    return 0;
}

extern "C" int lsk_comm_create_local(lsk_comm **out, int size) {
    if (size < 1 || size > LSK_MAX_PARTS) { snprintf(g_cerr, sizeof(g_cerr), "bad group size"); return -1; }
    LocalGroup *g = new LocalGroup();
    g->size = g->refs = size;
    g->bar.size = size;
    for (int r = 0; r < size; ++r) {
        lsk_comm *c = new lsk_comm();
        c->comm = nullptr; c->size = size; c->rank = r; c->local = g; c->wd = nullptr; c->tag[0] = 0;
        if (comm_resources(c) != 0) { // unwind: the ranks created so far, the barrier, the group
            delete c;
            for (int q = 0; q < r; ++q) {
                for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(out[q]->ready[i]); (void)hipEventDestroy(out[q]->done[i]); (void)hipEventDestroy(out[q]->xt0[i]); (void)hipEventDestroy(out[q]->xt1[i]); }
                (void)hipStreamDestroy(out[q]->xstream);
                delete out[q];
                out[q] = nullptr;
            }
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
        if (last) delete c->local;
        for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(c->ready[i]); (void)hipEventDestroy(c->done[i]); (void)hipEventDestroy(c->xt0[i]); (void)hipEventDestroy(c->xt1[i]); }
        (void)hipStreamDestroy(c->xstream);
        delete c;
        return;
    }
    watchdog_stop(c);
    if (g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
    for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(c->ready[i]); (void)hipEventDestroy(c->done[i]); (void)hipEventDestroy(c->xt0[i]); (void)hipEventDestroy(c->xt1[i]); }
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
        LOCAL_BARRIER(g, "all-reduce");
        if (e0 != hipSuccess) { int a_ = 0; (void)g->bar.wait(&a_); snprintf(g_cerr, sizeof(g_cerr), "loop-back all-reduce: %s", hipGetErrorString(e0)); return -1; }
        double acc[512];
        memcpy(acc, g->host[0], (size_t)count * es);
        for (int r = 1; r < g->size; ++r)
            for (int64_t k = 0; k < count; ++k) {
                if (dtype == 0) { double v = g->host[r][k]; acc[k] = op == 0 ? acc[k] + v : (v > acc[k] ? v : acc[k]); }
                else if (dtype == 1) { float v = ((float *)g->host[r])[k], &a = ((float *)acc)[k]; a = op == 0 ? a + v : (v > a ? v : a); }
                else { int64_t v = ((int64_t *)g->host[r])[k], &a = ((int64_t *)acc)[k]; a = op == 0 ? a + v : (v > a ? v : a); }
            }
        LOCAL_BARRIER(g, "all-reduce"); // everybody has read every copy
        HIP_CHECK(hipMemcpy(d_buf, acc, (size_t)count * es, hipMemcpyHostToDevice));
        return 0;
    }
    const ncclDataType_t t = dtype == 0 ? ncclDouble : dtype == 1 ? ncclFloat : ncclInt64;
    NCCL_CHECK(g_api.AllReduce(d_buf, d_buf, (size_t)count, t, op == 0 ? ncclSum : ncclMax, c->comm, (hipStream_t)stream));
    return watchdog_after_collective(c, (hipStream_t)stream, "ncclAllReduce (set-up agreement / PRIMME reduction)");
}
extern "C" int lsk_comm_broadcast(lsk_comm *c, void *d_buf, int64_t bytes, int root, void *stream) {
    if (c->local) {
        LocalGroup *g = c->local;
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        g->post[c->rank].buf = d_buf;
        LOCAL_BARRIER(g, "broadcast");
        hipError_t e0 = hipSuccess;
        if (c->rank != root) e0 = hipMemcpy(d_buf, g->post[root].buf, (size_t)bytes, hipMemcpyDeviceToDevice);
        LOCAL_BARRIER(g, "broadcast");
        if (e0 != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "loop-back broadcast: %s", hipGetErrorString(e0)); return -1; }
        return 0;
    }
    NCCL_CHECK(g_api.Broadcast(d_buf, d_buf, (size_t)bytes, ncclChar, root, c->comm, (hipStream_t)stream));
    return watchdog_after_collective(c, (hipStream_t)stream, "ncclBroadcast");
}
extern "C" int lsk_comm_allgather(lsk_comm *c, void const *d_send, void *d_recv, int64_t bytes_per_rank, void *stream) {
    if (c->local) {
        LocalGroup *g = c->local;
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        g->post[c->rank].send = d_send;
        LOCAL_BARRIER(g, "all-gather");
        hipError_t e0 = hipSuccess;
        for (int r = 0; r < g->size && e0 == hipSuccess; ++r)
            e0 = hipMemcpy((char *)d_recv + (size_t)r * (size_t)bytes_per_rank, g->post[r].send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice);
        LOCAL_BARRIER(g, "all-gather");
        if (e0 != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "loop-back all-gather: %s", hipGetErrorString(e0)); return -1; }
        return 0;
    }
    NCCL_CHECK(g_api.AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclChar, c->comm, (hipStream_t)stream));
    return watchdog_after_collective(c, (hipStream_t)stream, "ncclAllGather (set-up: counts / layouts)");
}

// all-to-all-v of bytes: segment d of the send buffer goes to rank d, segment s of the receive buffer comes from rank
// s; one grouped send/recv, i.e. every pair of GPUs talks over its own xGMI link at the same time.  Issued on the
// exchange stream between two events: `slot` (0/1) selects the event pair of the double-buffered pipeline.
//   compute stream: ... generate(r) | record ready[slot]                      wait done[slot] | scatter(r) ...
//   exchange stream:                  wait ready[slot] | grouped send/recv | record done[slot]
extern "C" int lsk_comm_exchange_begin(lsk_comm *c, int slot, void *compute_stream) {
    HIP_CHECK(hipEventRecord(c->ready[slot], (hipStream_t)compute_stream));
    HIP_CHECK(hipStreamWaitEvent(c->xstream, c->ready[slot], 0));
    HIP_CHECK(hipEventRecord(c->xt0[slot], c->xstream)); // the exchange could start here: everything behind it is wire (and peers)
    return 0;
}
// how long the last completed exchange of `slot` took on the exchange stream, from "this rank was ready" to "everything arrived":
// 0 = *ms is valid, 1 = not recorded yet / still running, -1 = error.  Never blocks.
extern "C" int lsk_comm_exchange_ms(lsk_comm *c, int slot, float *ms) {
    if (!c->xt_recorded[slot]) return 1;
    const hipError_t q = hipEventQuery(c->xt1[slot]);
    if (q == hipErrorNotReady) return 1;
    if (q != hipSuccess || hipEventElapsedTime(ms, c->xt0[slot], c->xt1[slot]) != hipSuccess) { (void)hipGetLastError(); return -1; }
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
        LOCAL_BARRIER(g, "all-to-all-v");
        for (int k = 0; k < K; ++k)
            for (int src = 0; src < g->size; ++src) {
                const int64_t want = recv_bytes[(size_t)k * g->size + src];
                const int64_t have = src == c->rank ? want : g->post[src].bytes[(size_t)k * g->size + c->rank];
                if (src == c->rank) continue;
                if (have != want) {
                    int a_ = 0;
                    (void)g->bar.wait(&a_);
                    snprintf(g_cerr, sizeof(g_cerr), "loop-back all-to-all-v: rank %d sends %lld bytes (segment %d) to rank %d, which expects %lld",
                             src, (long long)have, k, c->rank, (long long)want);
                    return -1;
                }
                if (want == 0) continue;
                const hipError_t e1 = hipMemcpyAsync((char *)d_recv + recv_off[(size_t)k * g->size + src],
                                                     (char const *)g->post[src].send + g->post[src].off[(size_t)k * g->size + c->rank],
                                                     (size_t)want, hipMemcpyDeviceToDevice, s);
                if (e1 != hipSuccess) { int a_ = 0; (void)g->bar.wait(&a_); snprintf(g_cerr, sizeof(g_cerr), "loop-back all-to-all-v: %s", hipGetErrorString(e1)); return -1; }
            }
        const hipError_t e2 = hipStreamSynchronize(s);
        LOCAL_BARRIER(g, "all-to-all-v"); // nobody reuses a send buffer before every peer has read it
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
        // (an error inside the group must not leave it open: every later RCCL call of this thread would be queued into it and
        // nothing would ever be issued -- the group is closed on every path, and the first failure is what gets reported)
        ncclResult_t bad = ncclSuccess;
        int bad_peer = -1;
        char const *bad_call = "";
        for (int step = 1; step < c->size && bad == ncclSuccess; ++step) {
            const int dst = (c->rank + step) % c->size, src = (c->rank - step + c->size) % c->size;
            const size_t ks = (size_t)k * c->size + dst, kr = (size_t)k * c->size + src;
            if (send_bytes[ks] > 0) {
                bad = g_api.Send((char const *)d_send + send_off[ks], (size_t)send_bytes[ks], ncclChar, dst, c->comm, s);
                if (bad != ncclSuccess) { bad_peer = dst; bad_call = "ncclSend to"; break; }
            }
            if (recv_bytes[kr] > 0) {
                bad = g_api.Recv((char *)d_recv + recv_off[kr], (size_t)recv_bytes[kr], ncclChar, src, c->comm, s);
                if (bad != ncclSuccess) { bad_peer = src; bad_call = "ncclRecv from"; }
            }
        }
        const ncclResult_t end = g_api.GroupEnd();
        if (bad != ncclSuccess || end != ncclSuccess) {
            if (bad != ncclSuccess)
                snprintf(g_cerr, sizeof(g_cerr), "all-to-all-v (rank %d, segment %d%s%s): %s rank %d failed: %s", c->rank, k, c->tag[0] ? ", " : "", c->tag,
                         bad_call, bad_peer, g_api.GetErrorString(bad));
            else
                snprintf(g_cerr, sizeof(g_cerr), "all-to-all-v (rank %d, segment %d%s%s): ncclGroupEnd failed: %s", c->rank, k, c->tag[0] ? ", " : "", c->tag,
                         g_api.GetErrorString(end));
            return -1;
        }
    }
    if (s != c->xstream) { // exchanges on the caller's stream (the replicated-x driver's un-overlapped ones): their own watch item
        char what[200];
        snprintf(what, sizeof(what), "grouped ncclSend/ncclRecv on the compute stream (%s)", c->tag[0] ? c->tag : "all-to-all-v");
        return watchdog_after_collective(c, s, what);
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
    if (c->wd) { // (recorded under the watchdog's lock: its thread queries the same event)
        std::lock_guard<std::mutex> lk(c->wd->m);
        HIP_CHECK(hipEventRecord(c->done[slot], c->xstream));
    } else HIP_CHECK(hipEventRecord(c->done[slot], c->xstream));
    HIP_CHECK(hipEventRecord(c->xt1[slot], c->xstream));
    c->xt_recorded[slot] = true;
    char what[200];
    snprintf(what, sizeof(what), "grouped ncclSend/ncclRecv on the exchange stream (%s)", c->tag[0] ? c->tag : "all-to-all-v");
    watchdog_arm(c, slot, what);
    return 0;
}

// Set-up cross-check of an exchange layout (VERDICT r5 #1b), collective: every rank contributes its [K][size] send and receive
// byte counts; everybody gathers everybody's and checks EVERY (segment, source, destination) pair -- what s sends to d must be
// what d expects from s -- so that all ranks reach the same verdict and a disagreement ends the set-up with a message (which
// segment, which two ranks, both byte counts) instead of a hang or misplaced data in the first matvec.
extern "C" int lsk_comm_check_counts(lsk_comm *c, int K, int64_t const *send_bytes, int64_t const *recv_bytes, char const *what, void *stream) {
    const int P = c->size;
    if (P < 2 || K < 1) return 0;
    const size_t per = (size_t)2 * (size_t)K * (size_t)P; // int64 entries per rank: sends, then receives
    int64_t *h = (int64_t *)malloc(sizeof(int64_t) * per * (size_t)(P + 1));
    void *d = nullptr;
    memcpy(h, send_bytes, sizeof(int64_t) * per / 2);
    memcpy(h + per / 2, recv_bytes, sizeof(int64_t) * per / 2);
    int rc = 0;
    if (hipMalloc(&d, sizeof(int64_t) * per * (size_t)(P + 1)) != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "count check: no device memory"); (void)hipGetLastError(); d = nullptr; rc = -1; }
    if (rc == 0 && hipMemcpy(d, h, sizeof(int64_t) * per, hipMemcpyHostToDevice) != hipSuccess) { snprintf(g_cerr, sizeof(g_cerr), "count check: copy failed"); rc = -1; }
    // (a rank without memory still enters the collective with zeros: its peers must not wait for it; it reports its own error)
    void *dd = d;
    if (!dd && hipMalloc(&dd, sizeof(int64_t) * per * (size_t)(P + 1)) != hipSuccess) { free(h); return -1; }
    if (lsk_comm_allgather(c, dd, (char *)dd + sizeof(int64_t) * per, (int64_t)(sizeof(int64_t) * per), stream) != 0) rc = -1;
    if (rc == 0 && (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
                    hipMemcpy(h + per, (char *)dd + sizeof(int64_t) * per, sizeof(int64_t) * per * (size_t)P, hipMemcpyDeviceToHost) != hipSuccess)) {
        snprintf(g_cerr, sizeof(g_cerr), "count check: copy back failed"); (void)hipGetLastError(); rc = -1;
    }
    (void)hipFree(dd);
    if (rc == 0) {
        for (int k = 0; k < K && rc == 0; ++k)
            for (int src = 0; src < P && rc == 0; ++src)
                for (int dst = 0; dst < P; ++dst) {
                    if (src == dst) continue;
                    const int64_t sends = h[per * (size_t)(1 + src) + (size_t)k * P + dst];
                    const int64_t expects = h[per * (size_t)(1 + dst) + per / 2 + (size_t)k * P + src];
                    if (sends != expects) {
                        snprintf(g_cerr, sizeof(g_cerr), "%s: exchange layouts disagree: rank %d sends %lld bytes to rank %d in segment %d, which expects %lld "
                                 "(checked at set-up on every rank: no exchange was started)", what, src, (long long)sends, dst, k, (long long)expects);
                        rc = -1;
                        break;
                    }
                }
    }
    free(h);
    return rc;
}
extern "C" int lsk_comm_exchange_wait(lsk_comm *c, int slot, void *compute_stream) {
    HIP_CHECK(hipStreamWaitEvent((hipStream_t)compute_stream, c->done[slot], 0));
    return 0;
}
