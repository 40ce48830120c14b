// stage.cpp -- host <-> HBM staging of the host-pointer entry points (ls_chpl_matrix_vector_product, ls_chpl_primme_matvec:
// /root/reference/src/DistributedMatrixVector.chpl:1095-1110, src/Diagonalize.chpl:134-162).  The reference hands `double *`
// to its kernels because they run on the host; here the vectors have to cross PCIe, and how they cross decides what the
// drop-in entry costs (the kernel is 7.6 ms on chain_32, the vectors are 2 x 4.8 GB):
//   * a pointer that already IS device memory (hipMalloc / torch) is used in place: zero copies (lsk_pointer_kind);
//   * pinned host memory (hipHostMalloc, or registered once with ls_amd_host_register -- PRIMME reuses its workspace) goes
//     through ONE asynchronous DMA per direction;
//   * pageable memory is moved through two pinned bounce buffers per direction: a small pool of host threads copies chunk
//     i + 1 into its buffer while the DMA engine moves chunk i, uploads and downloads run at the same time (PCIe is full
//     duplex): the upload of column k + 1 of a PRIMME block overlaps the download of column k.
// Host-only C++ (no kernels); plain C interface in lsk.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "lsk.h"

static thread_local char g_serr[512] = "";
extern "C" char const *lsk_stage_last_error(void) { return g_serr; }

#define ST_CHECK(expr)                                                                                     \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            snprintf(g_serr, sizeof(g_serr), "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,            \
                     hipGetErrorString(e_));                                                               \
            (void)hipGetLastError();                                                                       \
            return -1;                                                                                     \
        }                                                                                                  \
    } while (0)

// 0 = pageable (or unknown) host memory, 1 = pinned / registered host memory, 2 = device memory (hipMalloc: coarse-grained),
// 3 = managed memory (hipMallocManaged).  Managed memory is fine-grained unless the caller advised otherwise, and the push
// kernels' hardware f64 atomics (-munsafe-fp-atomics -> global_atomic_add_f64) are specified for coarse-grained memory only: a
// managed y used in place could lose updates silently.  It is therefore never used in place: the boundary
// stages it through the plan's own hipMalloc vectors with one DMA per direction (hipMemcpyDefault), like pinned memory.
extern "C" int lsk_pointer_kind(void const *p) {
    if (!p) return LSK_PTR_PAGEABLE;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { // an ordinary malloc'ed pointer: "invalid value" on older runtimes
        (void)hipGetLastError();
        return LSK_PTR_PAGEABLE;
    }
    if (a.type == hipMemoryTypeManaged || a.isManaged) return LSK_PTR_MANAGED;
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeArray) return LSK_PTR_DEVICE;
    if (a.type == hipMemoryTypeHost) return LSK_PTR_PINNED;
    return LSK_PTR_PAGEABLE; // hipMemoryTypeUnregistered
}
extern "C" int lsk_host_register(void *p, size_t bytes) { ST_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault)); return 0; }
extern "C" int lsk_host_unregister(void *p) { ST_CHECK(hipHostUnregister(p)); return 0; }

namespace {
// A handful of host threads that copy one buffer together (a single core moves ~10 GB/s, a PCIe 5.0 x16 link ~55 GB/s each way).
struct CopyPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable go, done;
    uint64_t generation = 0;
    int pending = 0;
    char *dst = nullptr;
    char const *src = nullptr;
    size_t bytes = 0;
    size_t par_min = (size_t)1 << 20; // below this one thread copies
    int T = 1;

    static void slice(char *d, char const *s, size_t n, int i, int T) {
        const size_t per = ((n + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
        const size_t lo = std::min(n, per * (size_t)i), hi = std::min(n, lo + per);
        if (hi > lo) memcpy(d + lo, s + lo, hi - lo);
    }
    void worker(int i) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            go.wait(lk, [&] { return generation != seen; });
            seen = generation;
            char *d = dst;
            char const *s = src;
            const size_t n = bytes;
            lk.unlock();
            slice(d, s, n, i, T);
            lk.lock();
            if (--pending == 0) done.notify_one();
        }
    }
    explicit CopyPool(int t) : T(t < 1 ? 1 : t) {
        for (int i = 1; i < T; ++i) threads.emplace_back([this, i] { worker(i); });
        for (auto &th : threads) th.detach(); // the pool lives as long as the process (never destroyed: no join at exit)
    }
    std::mutex use; // one copy at a time (two stagers, or two host threads in the boundary, share the pool)
    void copy(void *d, void const *s, size_t n) {
        if (T == 1 || n < par_min) { memcpy(d, s, n); return; }
        std::lock_guard<std::mutex> only(use);
        {
            std::lock_guard<std::mutex> lk(m);
            dst = (char *)d; src = (char const *)s; bytes = n; pending = T - 1; ++generation;
        }
        go.notify_all();
        slice((char *)d, (char const *)s, n, 0, T);
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return pending == 0; });
    }
};

struct Direction {
    hipStream_t stream = nullptr;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
};
} // namespace

struct lsk_stager {
    size_t chunk;
    Direction up, down;
    CopyPool *pool;
    std::mutex lock; // one transfer pair at a time per stager
};
// one pool per process, created with the first stager and never destroyed (its detached threads sleep on a condition variable)
static CopyPool *g_pool = nullptr;
static std::mutex g_pool_lock;

static int direction_init(Direction &d, size_t chunk) {
    ST_CHECK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        ST_CHECK(hipHostMalloc(&d.pin[i], chunk, hipHostMallocDefault));
        ST_CHECK(hipEventCreateWithFlags(&d.ev[i], hipEventDisableTiming));
    }
    return 0;
}
static void direction_free(Direction &d) {
    for (int i = 0; i < 2; ++i) {
        if (d.pin[i]) (void)hipHostFree(d.pin[i]);
        if (d.ev[i]) (void)hipEventDestroy(d.ev[i]);
    }
    if (d.stream) (void)hipStreamDestroy(d.stream);
}

extern "C" int lsk_stager_create(lsk_stager **out, size_t chunk_bytes, int threads) {
    *out = nullptr;
    lsk_stager *st = new lsk_stager();
    st->chunk = chunk_bytes < 4096 ? 4096 : (chunk_bytes & ~(size_t)4095);
    if (direction_init(st->up, st->chunk) != 0 || direction_init(st->down, st->chunk) != 0) {
        direction_free(st->up); direction_free(st->down);
        delete st;
        return -1;
    }
    {
        std::lock_guard<std::mutex> guard(g_pool_lock);
        if (!g_pool) {
            if (threads < 1) {
                const unsigned hc = std::thread::hardware_concurrency();
                threads = (int)std::min(16u, std::max(1u, hc / 4));
            }
            g_pool = new CopyPool(threads);
        }
        char const *pm = getenv("LS_AMD_STAGE_PAR_MIN_KB"); // (tests: small vectors through the thread pool)
        g_pool->par_min = pm && atoi(pm) > 0 ? (size_t)atoi(pm) << 10 : (size_t)1 << 20;
    }
    st->pool = g_pool;
    *out = st;
    return 0;
}
extern "C" void lsk_stager_destroy(lsk_stager *st) {
    if (!st) return;
    (void)hipStreamSynchronize(st->up.stream);
    (void)hipStreamSynchronize(st->down.stream);
    direction_free(st->up);
    direction_free(st->down);
    delete st; // (the pool stays)
}
extern "C" int lsk_stager_threads(lsk_stager const *st) { return st->pool->T; }
extern "C" size_t lsk_stager_chunk(lsk_stager const *st) { return st->chunk; }

// (The runtime's own copies from two host threads, one per direction, do NOT overlap on this stack: a block of four chain_32
// columns took 716 ms either way, against 555 ms through the bounce buffers below -- profiles/r5_bench_default.json.)
// One upload (host -> device) and one download (device -> host) at the same time; either may be empty (bytes == 0).
// host_kind: LSK_PTR_PAGEABLE -> bounce buffers, LSK_PTR_PINNED / LSK_PTR_MANAGED -> one DMA.  Returns when both are complete.
extern "C" int lsk_stage_run(lsk_stager *st, void *d_up, void const *h_up, size_t up_bytes, int up_kind,
                             void *h_down, void const *d_down, size_t down_bytes, int down_kind) {
    std::lock_guard<std::mutex> guard(st->lock);
    const size_t C = st->chunk;
    const bool up_direct = up_kind != LSK_PTR_PAGEABLE, down_direct = down_kind != LSK_PTR_PAGEABLE;
    // (hipMemcpyDefault: the "host" side may be managed memory, wherever its pages live)
    if (up_bytes && up_direct) ST_CHECK(hipMemcpyAsync(d_up, h_up, up_bytes, hipMemcpyDefault, st->up.stream));
    if (down_bytes && down_direct) ST_CHECK(hipMemcpyAsync(h_down, d_down, down_bytes, hipMemcpyDefault, st->down.stream));
    const size_t nu = (up_bytes && !up_direct) ? (up_bytes + C - 1) / C : 0;
    const size_t nd = (down_bytes && !down_direct) ? (down_bytes + C - 1) / C : 0;
    // software pipeline over chunk index i: upload chunk i is copied into its bounce buffer and sent; download chunk i is
    // requested, and download chunk i - 1 (whose DMA ran meanwhile) is copied out to the caller's buffer
    for (size_t i = 0; i < std::max(nu, nd + 1); ++i) {
        if (i < nd) {
            const size_t off = i * C, len = std::min(C, down_bytes - off);
            // buffer (i & 1) was emptied by the host copy of chunk i - 2 (synchronous, below)
            ST_CHECK(hipMemcpyAsync(st->down.pin[i & 1], (char const *)d_down + off, len, hipMemcpyDeviceToHost, st->down.stream));
            ST_CHECK(hipEventRecord(st->down.ev[i & 1], st->down.stream));
        }
        if (i < nu) {
            const size_t off = i * C, len = std::min(C, up_bytes - off);
            if (i >= 2) ST_CHECK(hipEventSynchronize(st->up.ev[i & 1])); // the DMA of chunk i - 2 has left this buffer
            st->pool->copy(st->up.pin[i & 1], (char const *)h_up + off, len);
            ST_CHECK(hipMemcpyAsync((char *)d_up + off, st->up.pin[i & 1], len, hipMemcpyHostToDevice, st->up.stream));
            ST_CHECK(hipEventRecord(st->up.ev[i & 1], st->up.stream));
        }
        if (i >= 1 && i - 1 < nd) {
            const size_t j = i - 1, off = j * C, len = std::min(C, down_bytes - off);
            ST_CHECK(hipEventSynchronize(st->down.ev[j & 1]));
            st->pool->copy((char *)h_down + off, st->down.pin[j & 1], len);
        }
    }
    if (up_bytes) ST_CHECK(hipStreamSynchronize(st->up.stream));
    if (down_bytes) ST_CHECK(hipStreamSynchronize(st->down.stream));
    return 0;
}
