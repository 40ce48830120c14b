// kernels.hip -- hand-written HIP (gfx950 / CDNA4) kernels of the matrix-free y <- H x hot path and
// the thin extern-"C" shim (lsk_*) the C host side calls.  No MFMA anywhere: this path is irregular
// integer/bit work plus gather/scatter, bounded by HBM and (for symmetry-projected bases) integer ALU.
//
// Reference call sites replaced:
//   K1 localDiagonalBatch            /root/reference/src/DistributedMatrixVector.chpl:36-53
//   K2 computeOffDiag                /root/reference/src/BatchedOperator.chpl:82-116
//   K3 spin-inversion canonicalise   /root/reference/src/BatchedOperator.chpl:139-153
//   K4 symmetry projection           /root/reference/src/BatchedOperator.chpl:163-203
//   K5 hash64_01 % numLocales        /root/reference/src/StatesEnumeration.chpl:122-136
//   K6 radixOneStep                  /root/reference/src/DistributedMatrixVector.chpl:265-311
//   K7 ls_hs_state_index             /root/reference/src/DistributedMatrixVector.chpl:96-103
//   K8 ConcurrentAccessor.localAdd   /root/reference/src/ConcurrentAccessor.chpl:48-54
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "lsk.h"

// ---------------------------------------------------------------------------------------------
// runtime shim
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

#define LSK_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            snprintf(g_err, sizeof(g_err), "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,    \
                     hipGetErrorString(e_));                                                     \
            return -1;                                                                           \
        }                                                                                        \
    } while (0)

#define LSK_LAUNCH_CHECK() LSK_CHECK(hipGetLastError())

extern "C" char const *lsk_last_error(void) { return g_err; }
extern "C" char *lsk_error_buffer(size_t *capacity) { *capacity = sizeof(g_err); return g_err; } // the other translation units report through it
extern "C" int lsk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int lsk_set_device(int device) { LSK_CHECK(hipSetDevice(device)); return 0; }
extern "C" int lsk_malloc(void **p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // an allocation failure is recoverable: do not leave it for the next launch check
        snprintf(g_err, sizeof(g_err), "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        *p = nullptr;
        return -1;
    }
    return 0;
}
extern "C" int lsk_free(void *p) { if (p) LSK_CHECK(hipFree(p)); return 0; }
extern "C" int lsk_mem_info(size_t *free_bytes, size_t *total_bytes) { LSK_CHECK(hipMemGetInfo(free_bytes, total_bytes)); return 0; }
extern "C" int lsk_h2d(void *dst, void const *src, size_t bytes) {
    if (bytes) LSK_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int lsk_d2h(void *dst, void const *src, size_t bytes) {
    if (bytes) LSK_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int lsk_d2d_async(void *dst, void const *src, size_t bytes, void *stream) {
    if (bytes) LSK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
extern "C" int lsk_memset_async(void *p, int value, size_t bytes, void *stream) {
    if (bytes) LSK_CHECK(hipMemsetAsync(p, value, bytes, (hipStream_t)stream));
    return 0;
}
extern "C" int lsk_sync(void *stream) { LSK_CHECK(hipStreamSynchronize((hipStream_t)stream)); return 0; }
extern "C" int lsk_device_sync(void) { LSK_CHECK(hipDeviceSynchronize()); return 0; }

extern "C" int lsk_event_create(void **ev) { hipEvent_t e; LSK_CHECK(hipEventCreate(&e)); *ev = (void *)e; return 0; }
extern "C" int lsk_event_destroy(void *ev) { if (ev) LSK_CHECK(hipEventDestroy((hipEvent_t)ev)); return 0; }
extern "C" int lsk_event_record(void *ev, void *stream) { LSK_CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return 0; }
extern "C" int lsk_event_elapsed_ms(void *start, void *stop, float *ms) {
    LSK_CHECK(hipEventSynchronize((hipEvent_t)stop));
    LSK_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
constexpr int kBlock = 256;
constexpr int kMaxGrid = 256 * 8; // 256 CUs x 8 resident blocks: grid-stride beyond this

// Persistent (grid-stride) launches must not exceed what is co-resident, or the surplus blocks run as a
// second, mostly idle wave (measured: +33 % on the row kernel when 7 instead of 8 blocks fit per CU).
// resident_grid() asks the runtime once per kernel; note that on ROCm 7.2 the answer is one block per
// CU too high for 256-thread kernels with 81..96 SGPRs (MI355X_MICROARCH.md), so the hot kernels are
// kept at <= 80 SGPRs (asserted in tests/test_host_tables.py::test_hot_kernel_register_budget).
#include <map>
#include <mutex>
#include <vector>
static int g_num_cus = 0;
template <typename K>
static int resident_grid(K kernel, int64_t work_blocks, size_t dyn_lds = 0, int block = kBlock) {
    static std::map<std::pair<void const *, size_t>, int> cache;
    static std::mutex lock; // plans may be created / launched from several host threads (loop-back communicators)
    std::lock_guard<std::mutex> guard(lock);
    if (g_num_cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    const std::pair<void const *, size_t> key((void const *)kernel, dyn_lds);
    auto it = cache.find(key);
    int per_cu;
    if (it == cache.end()) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, dyn_lds) != hipSuccess || nb < 1) nb = 1;
        if (nb > 8) nb = 8;
        cache[key] = nb;
        per_cu = nb;
    } else per_cu = it->second;
    int64_t g = (int64_t)g_num_cus * per_cu;
    if (work_blocks < g) g = work_blocks;
    if (g < 1) g = 1;
    return (int)g;
}

// Grid of the staged (tile) kernels: one block per tile up to the whole tile count.  They walk rows with a plain grid
// stride (no XCD tile lists), so nothing needs them to be persistent, and their 106 SGPRs put them where the occupancy
// API over-reports the resident blocks by one (MI355X_MICROARCH.md): a CUs x API-answer grid runs a straggler round
// with one block per CU (measured r2: k_tile_pull on chain_36_symm 34.8 -> 24.5 ms with the plain grid).
template <typename K>
static int tile_grid(K, int64_t work_blocks) {
    if (work_blocks < 1) work_blocks = 1;
    if (work_blocks > (int64_t)1 << 30) work_blocks = (int64_t)1 << 30;
    return (int)work_blocks;
}

// LS_AMD_ABLATE (lsk_basis.debug_ablate) switches stages of the tile / pull kernels off to price them --
// profiling builds only (make ABLATE=1): the shipped kernels carry none of these branches.
#ifndef LSK_ABLATE
#define LSK_ABLATE 0
#endif
constexpr bool kAblate = LSK_ABLATE != 0;
extern "C" int lsk_ablate_mask(void) {
    if (!kAblate) return 0;
    char const *e = getenv("LS_AMD_ABLATE");
    return e ? atoi(e) : 0;
}

static inline int grid_for(int64_t n, int per_block = kBlock) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kMaxGrid) b = kMaxGrid;
    return (int)b;
}

// K5: splitmix64 finaliser (StatesEnumeration.chpl:122-127)
__device__ __forceinline__ uint64_t hash64_01(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    x = x ^ (x >> 31);
    return x;
}

// owner = hash % P with 32-bit arithmetic only (a true modulo: P = 3 must work, not just 2^k)
struct Owner {
    uint32_t P;
    uint32_t pmask;    // P - 1 when P is a power of two, else 0xffffffff
    uint32_t two32mod; // 2^32 mod P
};
static inline Owner make_owner(int P) {
    Owner o;
    o.P = (uint32_t)P;
    o.pmask = ((P & (P - 1)) == 0) ? (uint32_t)(P - 1) : 0xffffffffu;
    o.two32mod = (uint32_t)((1ULL << 32) % (uint64_t)P);
    return o;
}
__device__ __forceinline__ int owner_of(uint64_t s, Owner o) {
    if (o.P == 1) return 0; // one locale: nothing to hash (wave-uniform)
    uint64_t h = hash64_01(s);
    if (o.pmask != 0xffffffffu) return (int)((uint32_t)h & o.pmask);
    uint32_t hi = (uint32_t)(h >> 32), lo = (uint32_t)h;
    uint32_t r = hi % o.P;
    return (int)((r * o.two32mod + lo % o.P) % o.P);
}

// K8: relaxed, agent-scope f64 add.  unsafeAtomicAdd lowers to global_atomic_add_f64 on gfx950 for
// coarse-grained (hipMalloc) memory -- checked in the ISA dump, see DESIGN.md.
__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

template <bool REAL>
__device__ __forceinline__ void term_sum(lsk_term const *__restrict__ terms, int b, int e, uint64_t a,
                                         double &cr, double &ci) {
    cr = 0.0;
    ci = 0.0;
    for (int t = b; t < e; ++t) {
        lsk_term T = terms[t];
        if ((a & T.m) == T.r) {
            bool neg = __popcll(a & T.s) & 1;
            cr += neg ? -T.v_re : T.v_re;
            if (!REAL) ci += neg ? -T.v_im : T.v_im;
        }
    }
}

// K2: coefficient of one flip-mask group on state a
template <bool REAL>
__device__ __forceinline__ void group_coeff(lsk_group const &G, lsk_term const *__restrict__ off,
                                            uint64_t a, double &cr, double &ci) {
    if (G.fast == LSK_GROUP_EXCHANGE) {
        bool act = __popcll(a & G.x) == 1;
        cr = act ? G.v_re : 0.0;
        ci = (!REAL && act) ? G.v_im : 0.0;
        return;
    }
    term_sum<REAL>(off, G.begin, G.end, a, cr, ci);
}

// combinadic rank among equal-popcount integers (ls_hs_fixed_hamming_state_to_index, FFI.chpl:165)
__device__ __forceinline__ int64_t rank_combinadic(uint64_t s, uint64_t const *binom) {
    int64_t idx = 0;
    int k = 1;
    while (s) {
        int p = __ffsll((unsigned long long)s) - 1;
        idx += (int64_t)binom[p * LSK_BINOM_K + k];
        ++k;
        s &= s - 1;
    }
    return idx;
}
__device__ __forceinline__ uint64_t unrank_combinadic(int64_t idx, int hamming, uint64_t const *binom) {
    uint64_t s = 0;
    int p = 63;
    for (int k = hamming; k >= 1; --k) {
        while (p > k - 1 && (int64_t)binom[p * LSK_BINOM_K + k] > idx) --p;
        // p is now the largest position with C(p, k) <= idx
        s |= 1ULL << p;
        idx -= (int64_t)binom[p * LSK_BINOM_K + k];
        --p;
    }
    return s;
}
// Gosper's hack (StatesEnumeration.chpl:31-34)
__device__ __forceinline__ uint64_t next_fixed_hamming(uint64_t v) {
    uint64_t t = v | (v - 1);
    return (t + 1) | (((~t & (t + 1)) - 1) >> (__ffsll((unsigned long long)v)));
}

// K7: prefix-bucket table + binary search in the ascending representatives
__device__ __forceinline__ int64_t search_index(lsk_index const &ix, uint64_t s) {
    uint64_t b = s >> ix.shift;
    uint32_t lo = ix.table[b], hi = ix.table[b + 1];
    const uint32_t end = hi;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (ix.reps[mid] < s) lo = mid + 1; else hi = mid;
    }
    return (lo < end && ix.reps[lo] == s) ? (int64_t)lo : -1;
}

// Rank directory (lsk_rankdir, lsk.h): LDS copy of the binomials a rank needs -- C(p, k), p < sites, k <= weight -- and the look-up
__device__ __forceinline__ int rankdir_lds_entries(lsk_index const &ix) { return ix.dir ? ix.dir_sites * (ix.dir_weight + 1) : 0; }
__device__ __forceinline__ void rankdir_load(lsk_index const &ix, uint64_t *s_db) { // (the caller synchronises the block)
    const int kc = ix.dir_weight + 1;
    for (int i = threadIdx.x; i < ix.dir_sites * kc; i += blockDim.x) s_db[i] = ix.binom[(i / kc) * LSK_BINOM_K + (i % kc)];
}
__device__ __forceinline__ int64_t rankdir_index(lsk_index const &ix, uint64_t s, uint64_t const *s_db) {
    const int kc = ix.dir_weight + 1;
    if (__popcll(s) != ix.dir_weight || (ix.dir_sites < 64 && (s >> ix.dir_sites) != 0)) return -1;
    uint64_t g = 0;
    int k = 1;
    while (s) {
        const int p = __ffsll((unsigned long long)s) - 1;
        g += s_db[p * kc + k];
        ++k;
        s &= s - 1;
    }
    const ulonglong2 e = *reinterpret_cast<ulonglong2 const *>(ix.dir + (g >> 6));
    const uint64_t bit = 1ULL << (g & 63);
    if (!(e.x & bit)) return -1;
    return (int64_t)(uint32_t)e.y + __popcll(e.x & (bit - 1));
}

// All-destinations directory (lsk_gdir, lsk.h): index of state s inside the block of partition d, or -1 (s_db as above: the
// binomials C(p, k), p < sites, k <= weight)
__device__ __forceinline__ int64_t gdir_index(lsk_gdir const &gd, uint64_t s, int d, uint64_t const *s_db) {
    const int kc = gd.weight + 1;
    if (__popcll(s) != gd.weight || (gd.sites < 64 && (s >> gd.sites) != 0)) return -1;
    uint64_t g = 0;
    int k = 1;
    while (s) {
        const int p = __ffsll((unsigned long long)s) - 1;
        g += s_db[p * kc + k];
        ++k;
        s &= s - 1;
    }
    if ((int64_t)g >= gd.n_ranks) return -1;
    const ulonglong2 e = *reinterpret_cast<ulonglong2 const *>(gd.entries + (g >> 6) * (uint64_t)gd.P + (uint64_t)d);
    const uint64_t bit = 1ULL << (g & 63);
    if (!(e.x & bit)) return -1;
    return (int64_t)(uint32_t)e.y + __popcll(e.x & (bit - 1));
}
// the same look-up for a state whose global rank g is already known
constexpr uint32_t kNoRank = 0xffffffffu;
__device__ __forceinline__ int64_t gdir_index_of_rank(lsk_gdir const &gd, uint64_t g, int d) {
    if ((int64_t)g >= gd.n_ranks) return -1;
    const ulonglong2 e = *reinterpret_cast<ulonglong2 const *>(gd.entries + (g >> 6) * (uint64_t)gd.P + (uint64_t)d);
    const uint64_t bit = 1ULL << (g & 63);
    if (!(e.x & bit)) return -1;
    return (int64_t)(uint32_t)e.y + __popcll(e.x & (bit - 1));
}
__device__ __forceinline__ void gdir_load(lsk_gdir const &gd, uint64_t const *__restrict__ g_binom, uint64_t *s_db) { // (the caller synchronises)
    const int kc = gd.weight + 1;
    for (int i = threadIdx.x; i < gd.sites * kc; i += blockDim.x) s_db[i] = g_binom[(i / kc) * LSK_BINOM_K + (i % kc)];
}

// Open-addressing hash table {representative -> x * norm(rep)} used by the staged pull kernel.  The
// uncoalesced per-lane loads of a search (table + ~5 probes + value = 8 line requests per packet) were what
// bounded k_tile_pull (L1/TA issue: one line per lane per cycle); a hit in the home slot costs ONE 16-byte
// request.  Keys are inserted once per plan (linear probing, load factor <= 0.5), values are refreshed
// every matvec through slot_of[i].  Entry = {key, re[, im, pad]}: 2 (f64) or 4 (c128) u64 words.
constexpr uint64_t kHashEmpty = ~0ULL;
__device__ __forceinline__ uint64_t hash_slot(uint64_t key, int bits) {
    return (key * 0x9E3779B97F4A7C15ULL) >> (64 - bits);
}
template <int ES>
__device__ __forceinline__ bool hash_lookup(uint64_t const *__restrict__ tab, int bits, uint64_t key, double &vr,
                                            double &vi) {
    const uint64_t mask = (1ULL << bits) - 1;
    uint64_t slot = hash_slot(key, bits);
    for (;;) {
        if (ES == 2) {
            const ulonglong2 e = *(ulonglong2 const *)(tab + slot * 2);
            if (e.x == key) { vr = __longlong_as_double((long long)e.y); vi = 0.0; return true; }
            if (e.x == kHashEmpty) return false;
        } else {
            const ulonglong2 e = *(ulonglong2 const *)(tab + slot * 4);
            if (e.x == key) {
                vr = __longlong_as_double((long long)e.y);
                vi = __longlong_as_double((long long)tab[slot * 4 + 2]);
                return true;
            }
            if (e.x == kHashEmpty) return false;
        }
        slot = (slot + 1) & mask;
    }
}

// one symmetry-group element applied to a state
__device__ __forceinline__ uint64_t delta_swap(uint64_t x, uint64_t m, int d) {
    uint64_t t = ((x >> d) ^ x) & m;
    return x ^ t ^ (t << d);
}
__device__ __forceinline__ uint64_t apply_elem(lsk_group_elem const &e, uint64_t x, int L, uint64_t mask) {
    if (e.kind == LSK_ELEM_BENES) {
        if (e.masks[0]) x = delta_swap(x, e.masks[0], 32);
        if (e.masks[1]) x = delta_swap(x, e.masks[1], 16);
        if (e.masks[2]) x = delta_swap(x, e.masks[2], 8);
        if (e.masks[3]) x = delta_swap(x, e.masks[3], 4);
        if (e.masks[4]) x = delta_swap(x, e.masks[4], 2);
        if (e.masks[5]) x = delta_swap(x, e.masks[5], 1);
        if (e.masks[6]) x = delta_swap(x, e.masks[6], 2);
        if (e.masks[7]) x = delta_swap(x, e.masks[7], 4);
        if (e.masks[8]) x = delta_swap(x, e.masks[8], 8);
        if (e.masks[9]) x = delta_swap(x, e.masks[9], 16);
        if (e.masks[10]) x = delta_swap(x, e.masks[10], 32);
        return x;
    }
    if (e.kind == LSK_ELEM_REVROT) x = __brevll(x) >> (64 - L);
    int k = e.k;
    if (k == 0) return x;
    return ((x >> k) | (x << (L - k))) & mask;
}

// 32-bit variant of apply_elem for bases with <= 32 sites: permutations only move the low 32 bits, so
// the distance-32 Benes stages are empty and the remaining masks live in the low words.
__device__ __forceinline__ uint32_t delta_swap32(uint32_t x, uint32_t m, int d) {
    uint32_t t = ((x >> d) ^ x) & m;
    return x ^ t ^ (t << d);
}
__device__ __forceinline__ uint32_t apply_elem32(lsk_group_elem const &e, uint32_t x, int L, uint32_t mask) {
    if (e.kind == LSK_ELEM_BENES) {
        // only the low words of the masks are read (4-byte scalar loads: half the scalar registers of the 8-byte ones)
        uint32_t const *const m32 = reinterpret_cast<uint32_t const *>(e.masks);
        if (m32[2]) x = delta_swap32(x, m32[2], 16);
        if (m32[4]) x = delta_swap32(x, m32[4], 8);
        if (m32[6]) x = delta_swap32(x, m32[6], 4);
        if (m32[8]) x = delta_swap32(x, m32[8], 2);
        if (m32[10]) x = delta_swap32(x, m32[10], 1);
        if (m32[12]) x = delta_swap32(x, m32[12], 2);
        if (m32[14]) x = delta_swap32(x, m32[14], 4);
        if (m32[16]) x = delta_swap32(x, m32[16], 8);
        if (m32[18]) x = delta_swap32(x, m32[18], 16);
        return x;
    }
    if (e.kind == LSK_ELEM_REVROT) x = __brev(x) >> (32 - L);
    int k = e.k;
    if (k == 0) return x;
    return ((x >> k) | (x << (L - k))) & mask;
}
template <typename W> __device__ __forceinline__ W apply_elem_w(lsk_group_elem const &e, W x, int L, W mask);
template <> __device__ __forceinline__ uint64_t apply_elem_w<uint64_t>(lsk_group_elem const &e, uint64_t x, int L, uint64_t mask) {
    return apply_elem(e, x, L, mask);
}
template <> __device__ __forceinline__ uint32_t apply_elem_w<uint32_t>(lsk_group_elem const &e, uint32_t x, int L, uint32_t mask) {
    return apply_elem32(e, x, L, mask);
}

// K4: ls_hs_state_info -- orbit minimum, conj(character) of a minimising element, stabiliser sum.
// One pass over the permutations; the optional global spin flip is folded in by canonicalising every
// image to "top site bit clear" (t ^ mask < t iff the top bit of t is set), which halves the work.
// The stabiliser sum needs no `g(a) == a` tests: the elements that map a onto its representative
// form the coset g0 Stab(a), so  sum_{s in Stab(a)} chi(s) = conj(chi(g0)) * sum_{g: g(a) = rep} chi(g),
// i.e. it is accumulated over the ties with the running minimum.
// PM1: every character (and the inversion character) is +-1 -> integer accumulation.
template <typename W, bool PM1>
__device__ __forceinline__ void state_info_w(lsk_basis const &bs, lsk_group_elem const *__restrict__ elems,
                                             W a, W &rep, double &chr, double &chi, double &stab) {
    W best = ~(W)0;
    int info = 0;
    int si = 0;
    double sr = 0.0, sim = 0.0;
    const int inv = bs.spin_inversion;
    const int L = bs.number_sites;
    const W mask = (W)bs.site_mask;
    for (int g = 0; g < bs.n_elems; ++g) {
        lsk_group_elem const &e = elems[g];
        W t = apply_elem_w<W>(e, a, L, mask);
        int top = 0;
        if (inv != 0) {
            top = (int)((t >> (L - 1)) & 1);
            t = top ? (W)(t ^ mask) : t;
        }
        const bool less = t < best, eq = t == best;
        if (PM1) {
            int ch = (int)e.ch_re;
            ch = top ? ch * inv : ch;
            si = less ? ch : (eq ? si + ch : si);
        } else {
            double cr = e.ch_re, ci = e.ch_im;
            if (top) { cr *= (double)inv; ci *= (double)inv; }
            sr = less ? cr : (eq ? sr + cr : sr);
            sim = less ? ci : (eq ? sim + ci : sim);
        }
        best = less ? t : best;
        info = less ? (2 * g + top) : info;
    }
    rep = best;
    lsk_group_elem const &e0 = elems[info >> 1];
    double c0r = e0.ch_re, c0i = e0.ch_im;
    if (info & 1) { c0r *= (double)inv; c0i *= (double)inv; }
    chr = c0r;
    chi = -c0i;
    if (PM1) stab = c0r * (double)si;
    else stab = c0r * sr + c0i * sim; // Re(conj(chi0) * S)
}
// K4, trivial sector, cyclic / dihedral group (mode 3): orbit minimum WITHOUT visiting every rotation.
// The smallest rotation (as an integer, site L-1 = MSB) starts with the longest cyclic run of zeros, so
//   1. R <- start positions (MSB ends) of the longest zero runs: R_1 = z, R_{j+1} = R_j & rotl(z, j)
//      with z = ~a; the loop runs (longest run) times -- ~5-8 on typical states instead of L;
//   2. only those start positions are candidates (usually 1-2): rotate each to the top and take the min;
//   3. reflections: the runs of rev(a) are the mirrored runs of a, so their candidates come from R
//      by one rotate + bit-reverse, no second search;
//   4. global spin flip: the flipped images start with a run of *ones* of a, so only the family whose
//      longest run is longer (both on a tie) can contain the minimum.
// (host-callable as well: tests/test_host_tables.py checks it against the brute-force orbit minimum through
// lsk_test_rep_trivial_dihedral)
__host__ __device__ __forceinline__ int k4_ctz32(uint32_t v) { return __builtin_ctz(v); }
__host__ __device__ __forceinline__ int k4_ctz64(uint64_t v) { return __builtin_ctzll(v); }
__host__ __device__ __forceinline__ uint32_t k4_brev32(uint32_t v) { return __builtin_bitreverse32(v); }
__host__ __device__ __forceinline__ uint64_t k4_brev64(uint64_t v) { return __builtin_bitreverse64(v); }
template <typename W>
__host__ __device__ __forceinline__ W rotl_sites(W x, int s, int L, W mask) {
    return s == 0 ? x : (W)(((x << s) | (x >> (L - s))) & mask);
}
// rotl_sites without the final mask: for the run searches, where the result only meets words inside the mask
template <typename W>
__host__ __device__ __forceinline__ W rotl_raw(W x, int s, int L) {
    return s == 0 ? x : (W)((x << s) | (x >> (L - s)));
}
template <typename W>
__host__ __device__ __forceinline__ W rev_sites(W x, int L) {
    if (sizeof(W) == 4) return (W)(k4_brev32((uint32_t)x) >> (32 - L));
    return (W)(k4_brev64((uint64_t)x) >> (64 - L));
}
// Start positions (MSB ends) of the longest cyclic runs of set bits of z, and their length.  The run length is found by
// doubling and refining instead of one rotation per unit of length: R2 = z & rot(z,1), R4 = R2 & rot(R2,2), R8 = R4 &
// rot(R4,4) hold the starts of runs >= 2, 4, 8; below 8 two more rotations settle the exact length (R6 = R4 & rot(R2,4), then
// one step of 1) -- five rotations, no data-dependent trip count, where the step-by-step loop makes every lane of a wave wait
// for the longest run among 64 packets (8.5 steps on half-filled 36-site states against 5.3 on average).  Runs >= 8
// continue step by step from R8.
template <typename W>
__host__ __device__ __forceinline__ W longest_runs(W z, int L, W mask, int &len) {
    if (z == 0) { len = 0; return (W)1; }          // no zero site at all: every rotation is the same word
    if (z == mask) { len = L; return (W)1; }        // all sites zero
#ifdef LSK_K4_STEPWISE
    const bool stepwise = true; // A/B builds: the round-2 loop, one rotation per unit of run length
#else
    const bool stepwise = false;
#endif
    if (stepwise || L < 9) { // tiny rings: the doubling steps would wrap around the ring
        W R = z;
        int s = 1;
        for (;;) {
            const W T = R & rotl_raw<W>(z, s, L);
            if (T == 0) break;
            R = T;
            ++s;
        }
        len = s;
        return R;
    }
    const W R2 = z & rotl_raw<W>(z, 1, L);
    const W R4 = R2 & rotl_raw<W>(R2, 2, L);
    const W R8 = R4 & rotl_raw<W>(R4, 4, L);
    W R;
    int s;
    if (R8 != 0) { // rare per packet; the tail of the old loop
        R = R8;
        s = 8;
        for (;;) {
            const W T = R & rotl_raw<W>(z, s, L);
            if (T == 0 || s + 1 >= L) break;
            R = T;
            ++s;
        }
        len = s;
        return R;
    }
    const bool c4 = R4 != 0, c2 = R2 != 0;
    R = c4 ? R4 : (c2 ? R2 : z);
    s = c4 ? 4 : (c2 ? 2 : 1);
    const W R6 = R4 & rotl_raw<W>(R2, 4, L);
    if (R6 != 0) { R = R6; s = 6; }
    const W T = R & rotl_raw<W>(z, s, L);
    if (T != 0) { R = T; ++s; }
    len = s;
    return R;
}
// minimum over the rotations that put the MSB end of a longest run on top and, with reflections, over the mirrored
// words that start with the same run: the run whose MSB end is p has the candidate c = rotl(word, L-1-p); mirrored, the
// run leads again when its LSB end is on top of rev(word), and that word is rev(rotl(c, len)) -- one more rotation of c
// and a bit reversal, inside the same loop (round 2 ran a second loop over rev(word) and the mirrored end positions).
template <typename W>
__host__ __device__ __forceinline__ W family_min(W word, W R, int len, int L, W mask, bool reflect, W best) {
    const int lr = len >= L ? 0 : len; // len == L only for the all-equal words, whose rotations coincide
    while (R) {
        const int p = sizeof(W) == 4 ? k4_ctz32((uint32_t)R) : k4_ctz64((uint64_t)R);
        R &= R - 1;
        const W c = rotl_sites<W>(word, L - 1 - p, L, mask); // site p becomes the top site
        best = c < best ? c : best;
        if (reflect) {
            const W m = rev_sites<W>(rotl_sites<W>(c, lr, L, mask), L);
            best = m < best ? m : best;
        }
    }
    return best;
}
// (Both families through ONE loop -- a lane walking the starts of the first, then of the second, with the second's
// candidates as complements of the rotations of a -- measured slower: 15.7 vs 14.4 ms for the packets of chain_36_symm,
// scripts/k4_rate.py; the selects per iteration cost more than the shorter trip count saves.)
template <typename W>
__host__ __device__ __forceinline__ W rep_trivial_dihedral(W a, int L, W mask, bool inv, bool reflect) {
    if (!inv) {
        int len0;
        const W R0 = longest_runs<W>((W)(~a & mask), L, mask, len0); // zero runs of a
        return family_min<W>(a, R0, len0, L, mask, reflect, ~(W)0);
    }
    // With the global spin flip the minimum starts with the longest run of EQUAL bits of a, zeros or ones (a run of ones
    // leads the flipped word).  Those runs are the zero runs of the transition word t = a ^ rotl(a, 1) (t_p = 1 where
    // a changes between p-1 and p), one site shorter and with the same MSB ends -- so ONE run search finds the longest
    // runs of both kinds, ties between the kinds included, and ONE loop walks their starts; the kind of a run is the top bit
    // of the rotated word.  (Until late round 3: a search for zeros, one for ones, and two candidate passes that nearly
    // every wave entered both of; K4 alone 7.0 ms for the packets of chain_36_symm.)
    const W zt = (W)(~(a ^ rotl_sites<W>(a, 1, L, mask)) & mask);
    W R;
    int ell; // length of the longest run of equal bits
    if (zt == 0) { R = mask; ell = 1; }            // a alternates: every site starts a run of one
    else if (zt == mask) { R = (W)1; ell = L; }    // all sites equal
    else { int lt; R = longest_runs<W>(zt, L, mask, lt); ell = lt + 1; }
    const int lr = ell >= L ? 0 : ell;
    W best = ~(W)0;
    while (R) {
        const int p = sizeof(W) == 4 ? k4_ctz32((uint32_t)R) : k4_ctz64((uint64_t)R);
        R &= R - 1;
        const W r = rotl_sites<W>(a, L - 1 - p, L, mask); // site p becomes the top site
        const W flip = ((r >> (L - 1)) & 1) ? mask : (W)0;  // a run of ones: its flipped image competes
        const W c = r ^ flip;
        best = c < best ? c : best;
        if (reflect) {
            const W m = rev_sites<W>(rotl_sites<W>(r, lr, L, mask), L) ^ flip;
            best = m < best ? m : best;
        }
    }
    return best;
}
// host test hook: mode-3 orbit minimum of `a` on a ring of L sites (32-bit words for L <= 32, as the kernels choose)
extern "C" uint64_t lsk_test_rep_trivial_dihedral(uint64_t a, int L, int inv, int reflect) {
    const uint64_t mask = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    if (L <= 32) return (uint64_t)rep_trivial_dihedral<uint32_t>((uint32_t)a, L, (uint32_t)mask, inv != 0, reflect != 0);
    return rep_trivial_dihedral<uint64_t>(a, L, mask, inv != 0, reflect != 0);
}

// Profiling entry (scripts/k4_rate.py): K4 alone over the packets of a ring -- every row's state with each adjacent pair
// (b, b + 1 mod L) flipped, all lanes busy -- to price it outside the tile kernels.  variant 0: the whole orbit minimum;
// 1: the two run searches only; 2: the packets only (loop and flip, no K4).
template <typename W>
__global__ __launch_bounds__(kBlock) void k_bench_k4(int L, int inv, int reflect, int variant, int64_t n,
                                                     uint64_t const *__restrict__ reps, uint64_t *__restrict__ out) {
    const W mask = (W)(L >= 64 ? ~0ULL : ((1ULL << L) - 1));
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const W a = (W)reps[i];
        W acc = 0;
        for (int b = 0; b < L; ++b) {
            const W pm = (W)(((W)1 << b) | ((W)1 << (b + 1 == L ? 0 : b + 1)));
            const W beta = a ^ pm;
            if (variant == 0) acc ^= rep_trivial_dihedral<W>(beta, L, mask, inv != 0, reflect != 0);
            else if (variant == 1) {
                int l0, l1;
                const W r0 = longest_runs<W>((W)(~beta & mask), L, mask, l0);
                const W r1 = longest_runs<W>(beta, L, mask, l1);
                acc ^= r0 + r1 + (W)(l0 + 64 * l1);
            } else acc ^= beta + (W)b;
        }
        out[i] = (uint64_t)acc;
    }
}
extern "C" int lsk_bench_k4(int L, int inv, int reflect, int variant, int64_t n, uint64_t const *reps, uint64_t *out, void *stream) {
    if (n <= 0 || L < 2 || L > 64) return 0;
    const dim3 g((unsigned)grid_for(n)), b(kBlock);
    if (L <= 32) hipLaunchKernelGGL(k_bench_k4<uint32_t>, g, b, 0, (hipStream_t)stream, L, inv, reflect, variant, n, reps, out);
    else hipLaunchKernelGGL(k_bench_k4<uint64_t>, g, b, 0, (hipStream_t)stream, L, inv, reflect, variant, n, reps, out);
    LSK_LAUNCH_CHECK();
    return 0;
}

// Minimum of a word over the tw * th translations of a tw x th torus (site = y tw + x) and, with `inv`, of its complement:
// the rows are the digits of the word, so the minimum puts on top the smallest value ANY row takes under ANY rotation inside
// the row.  rowtab[r] (made by the host, lsk_torus_rowtab) holds for a row value r: bits 0-7 the minimum over its rotations,
// 8-15 the set of rotation amounts that reach it, 16-23 the maximum, 24-31 the amounts that reach that (the rows of the
// complemented word are the complements, so its row minima are the complements of the maxima).  One table load per row, then
// only the (row, amount) pairs that put the overall row minimum on top are built and compared -- 1-2 of the 2 tw th
// candidates on a half-filled 6 x 6 lattice -- and none at all when an earlier coset already has a smaller top row.
// (tw <= 8.  Until mid round 4: tw * th steps of "rotate the rows by one site / the word by one row" per coset.)
template <typename W>
__host__ __device__ __forceinline__ W torus_min(W v, int L, int tw, W mask, W col0, bool inv, uint32_t const *__restrict__ rowtab, W best) {
    const int th = L / tw;
    const uint32_t rmask = (1u << tw) - 1u;
    uint32_t mstar = 0xffffffffu;
    for (int k = 0; k < th; ++k) {
        const uint32_t e = rowtab[(uint32_t)(v >> (k * tw)) & rmask];
        uint32_t m = e & 0xffu;
        if (inv) { const uint32_t m2 = ~(e >> 16) & rmask; m = m2 < m ? m2 : m; }
        mstar = m < mstar ? m : mstar;
    }
    if ((W)mstar > (W)(best >> (L - tw))) return best; // (best == ~0 at the start: never true)
    // the candidates as site masks: bit k tw + i of cand[fam] = "rotate the rows by i, then row k to the top".  Collected
    // first and then popped ONE PER LANE AND ITERATION: the lanes of a wave hold different words, and a loop over
    // (row, family) with the construction inside its body made every wave run that body for nearly all 2 th combinations
    // (some lane always matches) -- 62.8 ms per matvec on heisenberg_square_6x6 -- instead of for the 1-3 its lanes need.
    const W nv = (W)(~v & mask);
    W cand0 = 0, cand1 = 0;
    for (int k = 0; k < th; ++k) {
        const uint32_t e = rowtab[(uint32_t)(v >> (k * tw)) & rmask];
        if ((e & 0xffu) == mstar) cand0 |= (W)((e >> 8) & 0xffu) << (k * tw);
        if (inv && (~(e >> 16) & rmask) == mstar) cand1 |= (W)(e >> 24) << (k * tw);
    }
    const uint32_t inv_tw = 65536u / (uint32_t)tw + 1u; // p / tw for p < 64, tw <= 8
    while (cand0 | cand1) {
        const bool first = cand0 != 0;
        const W cm = first ? cand0 : cand1;
        const int p = sizeof(W) == 4 ? k4_ctz32((uint32_t)cm) : k4_ctz64((uint64_t)cm);
        if (first) cand0 &= cand0 - 1; else cand1 &= cand1 - 1;
        const int k = (int)(((uint32_t)p * inv_tw) >> 16), i = p - k * tw;
        const W word = first ? v : nv;
        const W lo = (W)(col0 * (W)((1u << i) - 1u)); // columns 0 .. i-1 of every row (no carries: 2^i - 1 < 2^tw)
        W c = (W)((((W)(word << i)) & (W)~lo & mask) | ((W)(word >> (tw - i)) & lo)); // (i == 0: lo == 0, the first term is the word)
        c = rotl_sites<W>(c, tw * (th - 1 - k), L, mask);
        best = c < best ? c : best;
    }
    return best;
}
// ---- K4 mode 5: the point group of a rectangular / square torus, factorised -------------------------------------
// The cosets T g of such a lattice group are (modulo translations) the elements of D2 = {1, r, o, r o} -- r reverses every row
// (x -> tw-1-x), o reverses the order of the rows (y -> th-1-y), r o is the reversal of the whole word -- and, on a square
// torus, those times the transpose: D4.  Mode 4 sends the word through ONE compiled 11-stage network PER coset (8 x ~110 VALU
// instructions on 64-bit words) and runs torus_min on each image (8 x ~350).  Here
//   * only the transpose is a network; r is floor(tw / 2) delta swaps, r o one bit reversal, o = (r o) r;
//   * the four images of a base word are made of TWO row alphabets: {rows of v} for v and o(v), {reversed rows} for r(v) and
//     r o (v); rowtab2[row] carries the torus_min fields of the row (low half) AND of the reversed row (high half), so ONE pass
//     over the rows finds the smallest top row any of the four images can reach (and of their complements under the spin flip);
//   * only the alphabets that reach it build candidates, for their two words each.
// `present`: bit 0 = identity, 1 = r, 2 = o, 3 = r o (the images that belong to the group).
template <typename W>
__host__ __device__ __forceinline__ W rowrev_w(W v, int tw, W col0) {
    for (int j = 0; 2 * j + 1 < tw; ++j) { // swap columns j and tw-1-j of every row
        const int d = tw - 1 - 2 * j;
        const W t = (W)(((v >> d) ^ v) & (W)(col0 << j));
        v ^= (W)(t | (W)(t << d));
    }
    return v;
}
template <typename W>
__host__ __device__ __forceinline__ W torus_candidate(W word, int i, int kpos, int L, int tw, int th, W mask, W col0) {
    const W lo = (W)((W)(col0 << i) - col0); // columns 0 .. i-1 of every row: 2^i - 1 per row, no borrow between rows (i < tw)
    const W c = (W)((((W)(word << i)) & (W)~lo & mask) | ((W)(word >> (tw - i)) & lo)); // every row rotated by i
    return rotl_sites<W>(c, tw * (th - 1 - kpos), L, mask);                              // row kpos to the top
}
template <typename W>
__host__ __device__ __forceinline__ W torus_min_d2(W v, int L, int tw, W mask, W col0, bool inv, int present,
                                                   uint64_t const *__restrict__ rowtab2, W best) {
    const int th = L / tw;
    const uint32_t rmask = (1u << tw) - 1u;
    const bool needA = present & 5, needB = present & 10;
    // pass 1: the smallest top row of each alphabet (A: rows as they are, B: rows reversed; 1: of the complemented word)
    uint32_t mA0 = 0xffffffffu, mA1 = 0xffffffffu, mB0 = 0xffffffffu, mB1 = 0xffffffffu;
    for (int k = 0; k < th; ++k) {
        const uint64_t e = rowtab2[(uint32_t)(v >> (k * tw)) & rmask];
        const uint32_t ea = (uint32_t)e, eb = (uint32_t)(e >> 32);
        uint32_t m = ea & 0xffu; mA0 = m < mA0 ? m : mA0;
        m = ~(ea >> 16) & rmask;  mA1 = m < mA1 ? m : mA1;
        m = eb & 0xffu;           mB0 = m < mB0 ? m : mB0;
        m = ~(eb >> 16) & rmask;  mB1 = m < mB1 ? m : mB1;
    }
    if (!needA) mA0 = mA1 = 0xffffffffu;
    if (!needB) mB0 = mB1 = 0xffffffffu;
    if (!inv) mA1 = mB1 = 0xffffffffu;
    uint32_t mstar = mA0 < mA1 ? mA0 : mA1;
    mstar = mB0 < mstar ? mB0 : mstar;
    mstar = mB1 < mstar ? mB1 : mstar;
    if ((W)mstar > (W)(best >> (L - tw))) return best;
    // pass 2, ONE loop over the rows for all four alphabets (a loop per alphabet made every wave walk the rows four times: some
    // lane always needs each of them): bit k tw + i of a mask = "rotate the rows by i, then row k (base order) to the top"
    W cA0 = 0, cA1 = 0, cB0 = 0, cB1 = 0;
    for (int k = 0; k < th; ++k) {
        const uint64_t e = rowtab2[(uint32_t)(v >> (k * tw)) & rmask];
        const uint32_t ea = (uint32_t)e, eb = (uint32_t)(e >> 32);
        const int sh = k * tw;
        if ((ea & 0xffu) == mstar && mA0 == mstar) cA0 |= (W)((ea >> 8) & 0xffu) << sh;
        if ((~(ea >> 16) & rmask) == mstar && mA1 == mstar) cA1 |= (W)(ea >> 24) << sh;
        if ((eb & 0xffu) == mstar && mB0 == mstar) cB0 |= (W)((eb >> 8) & 0xffu) << sh;
        if ((~(eb >> 16) & rmask) == mstar && mB1 == mstar) cB1 |= (W)(eb >> 24) << sh;
    }
    // the words: rows in the base order (v | r(v)) and in reversed order (o(v) = rev(r(v)) | r o (v) = rev(v))
    const W rv = rowrev_w<W>(v, tw, col0);
    const W brv = rev_sites<W>(rv, L), bv = rev_sites<W>(v, L);
    const bool fwdA = present & 1, fwdB = present & 2, bwdA = present & 4, bwdB = present & 8;
    const uint32_t inv_tw = 65536u / (uint32_t)tw + 1u; // p / tw for p < 64, tw <= 8
    // ONE pop loop over the candidates of all alphabets (its trip count is the largest number of candidates a lane of the wave holds)
    while (cA0 | cA1 | cB0 | cB1) {
        const bool a0 = cA0 != 0, a1 = !a0 && cA1 != 0, b0 = !a0 && !a1 && cB0 != 0;
        const bool isB = !a0 && !a1, cpl = a1 || (isB && !b0);
        const W cm = a0 ? cA0 : (a1 ? cA1 : (b0 ? cB0 : cB1));
        const int p = sizeof(W) == 4 ? k4_ctz32((uint32_t)cm) : k4_ctz64((uint64_t)cm);
        const W rest = (W)(cm & (cm - 1));
        if (a0) cA0 = rest; else if (a1) cA1 = rest; else if (b0) cB0 = rest; else cB1 = rest;
        const int k = (int)(((uint32_t)p * inv_tw) >> 16), i = p - k * tw;
        const W flip = cpl ? mask : (W)0;
        if (isB ? fwdB : fwdA) { const W c = torus_candidate<W>((W)((isB ? rv : v) ^ flip), i, k, L, tw, th, mask, col0); best = c < best ? c : best; }
        if (isB ? bwdB : bwdA) { const W c = torus_candidate<W>((W)((isB ? bv : brv) ^ flip), i, th - 1 - k, L, tw, th, mask, col0); best = c < best ? c : best; }
    }
    return best;
}
// rowtab2[r] = rowtab[r] | rowtab[rev_tw(r)] << 32 (tw <= 8): out[2^tw]
extern "C" int lsk_torus_rowtab(int tw, uint32_t *out);
extern "C" int lsk_torus_rowtab2(int tw, uint64_t *out) {
    if (tw < 1 || tw > 8) return -1;
    uint32_t t[256];
    lsk_torus_rowtab(tw, t);
    for (uint32_t r = 0; r < (1u << tw); ++r) {
        uint32_t q = 0;
        for (int i = 0; i < tw; ++i) if (r & (1u << i)) q |= 1u << (tw - 1 - i);
        out[r] = (uint64_t)t[r] | ((uint64_t)t[q] << 32);
    }
    return 0;
}
// host test hook: the factorised minimum over the images `present` of one base word (64-bit arithmetic)
extern "C" uint64_t lsk_test_torus_min_d2(uint64_t v, int L, int tw, int inv, int present, uint64_t const *rowtab2, uint64_t best) {
    const uint64_t mask = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    uint64_t col0 = 0;
    for (int y = 0; y < L / tw; ++y) col0 |= 1ULL << (y * tw);
    return torus_min_d2<uint64_t>(v, L, tw, mask, col0, inv != 0, present, rowtab2, best);
}

// the row table of torus_min for rows of tw <= 8 bits: out[2^tw]
extern "C" int lsk_torus_rowtab(int tw, uint32_t *out) {
    if (tw < 1 || tw > 8) return -1;
    const uint32_t rmask = (1u << tw) - 1u;
    for (uint32_t r = 0; r <= rmask; ++r) {
        uint32_t mn = r, mx = r, amn = 0, amx = 0;
        for (int i = 1; i < tw; ++i) {
            const uint32_t q = ((r << i) | (r >> (tw - i))) & rmask;
            mn = q < mn ? q : mn;
            mx = q > mx ? q : mx;
        }
        for (int i = 0; i < tw; ++i) {
            const uint32_t q = i == 0 ? r : (((r << i) | (r >> (tw - i))) & rmask);
            if (q == mn) amn |= 1u << i;
            if (q == mx) amx |= 1u << i;
        }
        out[r] = mn | (amn << 8) | (mx << 16) | (amx << 24);
    }
    return 0;
}
// host test hook: torus_min of one word (64-bit arithmetic)
extern "C" uint64_t lsk_test_torus_min(uint64_t v, int L, int tw, int inv, uint32_t const *rowtab, uint64_t best) {
    const uint64_t mask = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    uint64_t col0 = 0;
    for (int y = 0; y < L / tw; ++y) col0 |= 1ULL << (y * tw);
    return torus_min<uint64_t>(v, L, tw, mask, col0, inv != 0, rowtab, best);
}

// K4, trivial sector: only the orbit minimum.  mode 2 generates the L rotations incrementally
// (rotr by one site = shift + move bit 0 to bit L-1) for a and, with reflections, for rev(a).
template <typename W>
__device__ __forceinline__ W rep_trivial(lsk_basis const &bs, lsk_group_elem const *__restrict__ elems, W a) {
    const int L = bs.number_sites;
    const W mask = (W)bs.site_mask;
    const bool inv = bs.spin_inversion != 0;
    W best = ~(W)0;
    if (bs.k4_mode == 3) return rep_trivial_dihedral<W>(a, L, mask, inv, bs.reflect != 0);
    if (bs.k4_mode == 5) { // D2 / D4 point group of a torus, factorised: one network (the transpose) instead of one per coset
        const W col0 = (W)bs.tcol0;
        best = torus_min_d2<W>(a, L, bs.tw, mask, col0, inv, bs.d4_mask & 15, bs.trow2, best);
        if (bs.d4_mask >> 4) best = torus_min_d2<W>(apply_elem_w<W>(bs.cosets[0], a, L, mask), L, bs.tw, mask, col0, inv, bs.d4_mask >> 4, bs.trow2, best);
        return best;
    }
    if (bs.k4_mode == 4) {
        // translations of a tw x th torus as a subgroup: one compiled network per coset representative (the point group), then
        // tw * th cheap steps -- rotate every row by one site; after tw of them the word is back, rotate it by one row
        const int tw = bs.tw, th = L / tw;
        const W col0 = (W)bs.tcol0, ncol0 = (W)(~col0 & mask);
        if (bs.trow) { // tw <= 8: the row table picks the few translations that can be minimal (torus_min)
            for (int r = 0; r < bs.n_cosets; ++r)
                best = torus_min<W>(apply_elem_w<W>(bs.cosets[r], a, L, mask), L, tw, mask, col0, inv, bs.trow, best);
            return best;
        }
        for (int r = 0; r < bs.n_cosets; ++r) {
            W b = apply_elem_w<W>(bs.cosets[r], a, L, mask);
            for (int j = 0; j < th; ++j) {
                for (int i = 0; i < tw; ++i) {
                    W c = b;
                    if (inv) c = ((b >> (L - 1)) & 1) ? (W)(b ^ mask) : b;
                    best = c < best ? c : best;
                    b = (W)(((W)(b << 1) & ncol0) | ((W)(b >> (tw - 1)) & col0));
                }
                b = rotl_sites<W>(b, tw, L, mask);
            }
        }
        return best;
    }
    if (bs.k4_mode == 2) {
        W r = a;
        for (int pass = 0; pass <= bs.reflect; ++pass) {
#pragma unroll 4
            for (int k = 0; k < L; ++k) {
                W c = r;
                if (inv) c = ((r >> (L - 1)) & 1) ? (W)(r ^ mask) : r;
                best = c < best ? c : best;
                r = (W)(r >> 1) | (W)((r & 1) << (L - 1));
            }
            if (sizeof(W) == 4) r = (W)(__brev((uint32_t)a) >> (32 - L));
            else r = (W)(__brevll((uint64_t)a) >> (64 - L));
        }
        return best;
    }
    for (int g = 0; g < bs.n_elems; ++g) {
        W t = apply_elem_w<W>(elems[g], a, L, mask);
        if (inv) t = ((t >> (L - 1)) & 1) ? (W)(t ^ mask) : t;
        best = t < best ? t : best;
    }
    return best;
}

__device__ __forceinline__ void state_info(lsk_basis const &bs, lsk_group_elem const *__restrict__ elems,
                                           uint64_t a, uint64_t &rep, double &chr, double &chi, double &stab) {
    if (bs.chars_pm1) state_info_w<uint64_t, true>(bs, elems, a, rep, chr, chi, stab);
    else state_info_w<uint64_t, false>(bs, elems, a, rep, chr, chi, stab);
}

// ls_hs_is_representative with early exit: false as soon as some element maps below a
__device__ __forceinline__ bool is_representative(lsk_basis const &bs, lsk_group_elem const *__restrict__ elems,
                                                  uint64_t a) {
    double st = 0.0;
    const int inv = bs.spin_inversion;
    for (int g = 0; g < bs.n_elems; ++g) {
        lsk_group_elem const &e = elems[g];
        uint64_t t = apply_elem(e, a, bs.number_sites, bs.site_mask);
        if (t < a) return false;
        if (t == a) st += e.ch_re;
        if (inv != 0) {
            uint64_t tf = t ^ bs.site_mask;
            if (tf < a) return false;
            if (tf == a) st += e.ch_re * (double)inv;
        }
    }
    return st * bs.inv_order > 1e-12;
}

__device__ __forceinline__ void load_binom(uint64_t *s_binom, uint64_t const *__restrict__ g_binom) {
    for (int i = threadIdx.x; i < 64 * LSK_BINOM_K; i += blockDim.x) s_binom[i] = g_binom[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// word-width helpers: bases with <= 32 sites run the row kernels on 32-bit states (half the VALU work)
// ---------------------------------------------------------------------------------------------
template <typename W> struct WordTraits;
template <> struct WordTraits<uint32_t> {
    typedef uint32_t binom_t;
    static __device__ __forceinline__ int popc(uint32_t v) { return __popc(v); }
    static __device__ __forceinline__ int ctz(uint32_t v) { return __ffs((int)v) - 1; }
};
template <> struct WordTraits<uint64_t> {
    typedef uint64_t binom_t;
    static __device__ __forceinline__ int popc(uint64_t v) { return __popcll(v); }
    static __device__ __forceinline__ int ctz(uint64_t v) { return __ffsll((unsigned long long)v) - 1; }
};
template <typename W, typename BT>
__device__ __forceinline__ int64_t rank_combinadic_w(W s, BT const *binom) {
    int64_t idx = 0;
    int k = 1;
    while (s) {
        int p = WordTraits<W>::ctz(s);
        idx += (int64_t)binom[p * LSK_BINOM_K + k];
        ++k;
        s &= s - 1;
    }
    return idx;
}

// diagonal coefficient with the zz-run shortcut: sum_b v (-1)^{[bits b, b+1 differ]} = v (cnt - 2 #differ)
template <typename W, bool REAL>
__device__ __forceinline__ void diag_coeff(lsk_runs const &runs, int n_diag, lsk_term const *__restrict__ diag,
                                           W a, double &dr, double &di) {
    dr = 0.0;
    di = 0.0;
    if (runs.n_zz > 0) {
        const W t = a ^ (a >> 1);
        for (int r = 0; r < runs.n_zz; ++r) {
            const W m = (W)(((uint64_t)1 << runs.zz_cnt[r]) - 1) << runs.zz_lo0[r];
            dr += runs.zz_v[r] * (double)(runs.zz_cnt[r] - 2 * WordTraits<W>::popc(t & m));
        }
    }
    if (runs.n_zz_terms < n_diag) {
        double gr, gi;
        term_sum<REAL>(diag, runs.n_zz_terms, n_diag, (uint64_t)a, gr, gi);
        dr += gr;
        di += gi;
    }
}

// ---------------------------------------------------------------------------------------------
// K1: diagonal pass  y[i] = d(sigma_i) x[i]
// ---------------------------------------------------------------------------------------------
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_diag(lsk_runs runs, int n_diag, lsk_term const *__restrict__ diag,
                                                 int64_t n, uint64_t const *__restrict__ reps,
                                                 double const *__restrict__ x, double *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t a = reps[i];
        double dr, di;
        diag_coeff<uint64_t, false>(runs, n_diag, diag, a, dr, di);
        if (CPLX) {
            double xr = x[2 * i], xi = x[2 * i + 1];
            y[2 * i] = dr * xr - di * xi;
            y[2 * i + 1] = dr * xi + di * xr;
        } else {
            y[i] = dr * x[i];
        }
    }
}

extern "C" int lsk_diag(lsk_operator op, int cplx, int64_t n, uint64_t const *reps, void const *x, void *y,
                        void *stream) {
    if (n == 0 || op.n_diag == 0) return 0;
    dim3 g(grid_for(n)), b(kBlock);
    if (cplx) hipLaunchKernelGGL(k_diag<true>, g, b, 0, (hipStream_t)stream, op.runs, op.n_diag, op.diag, n, reps, (double const *)x, (double *)y);
    else hipLaunchKernelGGL(k_diag<false>, g, b, 0, (hipStream_t)stream, op.runs, op.n_diag, op.diag, n, reps, (double const *)x, (double *)y);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Direct fused kernel: one partition, no permutation symmetries.  One row per lane, uniform loop
// over flip-mask groups.  Consecutive lanes hold consecutive basis states, so for a given group
// the active lanes' targets are (piecewise) consecutive as well: the scatter / gather coalesces.
//   PUSH: y[idx(beta)] += c x[i]                 (K2 + K3 + K7 + K8 fused; y holds the diagonal part)
//   PULL: y[i] = d x[i] + sum conj(c) x[idx(beta)]   (Hermitian operators; no atomics, y written once)
// Exchange runs (adjacent transpositions, e.g. the open bonds of a chain) take a branch-free inner
// loop: the rank of the target differs from the row's own rank by +-C(lo, k) with k = number of set
// bits below lo, which is carried incrementally; inactive lanes gather their own x and add 0, so
// the compiler can unroll and keep several gathers in flight.
// Row tiles come from a host-built tile map (lsk_tile_entry, lsk.h): block b runs on XCD b % 8 and
// walks that XCD's list of tiles, so the traversal order -- which decides what the XCD's L2 can
// reuse -- is data, not code.
// ---------------------------------------------------------------------------------------------
template <typename W, bool CPLX, int INDEX, bool INV, bool PULL, bool REAL>
__global__ __launch_bounds__(kBlock) void k_direct(lsk_runs runs, int n_groups, lsk_group const *__restrict__ groups,
                                                   lsk_term const *__restrict__ off, int n_diag,
                                                   lsk_term const *__restrict__ diag, lsk_basis bs,
                                                   lsk_index ix, uint64_t const *__restrict__ tilemap,
                                                   int64_t slots_per_xcd, uint64_t const *__restrict__ reps,
                                                   double const *__restrict__ x, double *y, int *err, int gx,
                                                   int64_t const *__restrict__ row_gidx) {
    typedef typename WordTraits<W>::binom_t BT;
    typedef WordTraits<W> WT;
    __shared__ BT s_binom[INDEX == LSK_INDEX_COMBINADIC ? 64 * LSK_BINOM_K : 1];
    if (INDEX == LSK_INDEX_COMBINADIC) {
        for (int k = threadIdx.x; k < 64 * LSK_BINOM_K; k += blockDim.x) s_binom[k] = (BT)ix.binom[k];
        __syncthreads();
    }
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3; // grid is a multiple of 8
    const W site_mask = (W)bs.site_mask;
    tilemap += (int64_t)xcd * slots_per_xcd;
    for (int64_t t = blockIdx.x >> 3; t < slots_per_xcd; t += blocks_per_xcd) {
        const uint64_t slot = tilemap[t]; // (first row, number of rows <= kBlock): lsk_tile_entry
        if ((uint64_t)threadIdx.x >= (slot >> 48)) continue;
        const int64_t i = (int64_t)(slot & 0xffffffffffffULL) + threadIdx.x;
        const W a = (W)__builtin_nontemporal_load(reps + i);
        // replicated-x mode (gx): rows are one hash partition, x is the whole vector in global ascending
        // order; ig = global index of this row (closed form, or precomputed for searched bases)
        int64_t ig = i;
        if (gx & 1) {
            if (INDEX == LSK_INDEX_COMBINADIC) ig = rank_combinadic_w<W, BT>(a, s_binom);
            else if (INDEX == LSK_INDEX_IDENTITY) ig = (int64_t)a;
            else ig = row_gidx[i];
        }
        double xr, xi = 0.0;
        if (CPLX) { xr = x[2 * ig]; xi = x[2 * ig + 1]; } else xr = x[ig];
        double accr = 0.0, acci = 0.0;
        if (PULL && n_diag == 0) { // no diagonal pass in the reference either: y is accumulated into (DMV:1062-1063)
            if (CPLX) { accr = y[2 * i]; acci = y[2 * i + 1]; } else accr = y[i];
        }
        if (PULL && n_diag > 0) {
            double dr, di;
            diag_coeff<W, REAL>(runs, n_diag, diag, a, dr, di);
            accr = dr * xr - (CPLX ? di * xi : 0.0);
            if (CPLX) acci = dr * xi + di * xr;
        }
        int g_begin = 0;
        if (INDEX == LSK_INDEX_COMBINADIC) {
            // ---- exchange runs: branch-free ------------------------------------------------------
            g_begin = runs.n_run_groups;
            const W tdiff = a ^ (a >> 1);
            for (int r = 0; r < runs.n_runs; ++r) {
                const int lo0 = runs.lo0[r], cnt = runs.cnt[r];
                const double vr = runs.v_re[r], vi = REAL ? 0.0 : runs.v_im[r];
                int k = WT::popc(a & (W)(((uint64_t)1 << lo0) - 1));
                int lo_begin = lo0, lo_end = lo0 + cnt;
                if (sizeof(W) == 4 && PULL && REAL && !(gx & 1)) {
                    // Far pairs (lo >= hb): the 64 consecutive states of a wave nearly always agree on every
                    // bit >= hb, so such a pair is anti-aligned for the whole wave or for none of it.  The
                    // test, the bit count below the pair and the rank shift are then wave-uniform (scalar
                    // unit), an aligned pair issues no gather at all, and an anti-aligned one costs an
                    // add, an address and an fma per lane.
                    const int hb = (gx >> 24) & 63;
                    const int split = hb == 0 ? lo_end : (hb < lo0 ? lo0 : (hb > lo_end ? lo_end : hb));
                    const uint32_t a0 = __builtin_amdgcn_readfirstlane((uint32_t)a);
                    const bool uni = split < lo_end &&
                                     __builtin_amdgcn_ballot_w64((((uint32_t)a ^ a0) >> split) != 0) == 0;
                    if (uni) {
                        uint32_t m = (a0 ^ (a0 >> 1)) & (uint32_t)((((uint64_t)1 << lo_end) - 1) & ~(((uint64_t)1 << split) - 1));
                        const uint32_t i32 = (uint32_t)ig;
                        while (m) {
                            double xv[4], xw[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                xv[u] = 0.0;
                                xw[u] = 0.0;
                                if (m) {
                                    const int lo = __builtin_ctz(m);
                                    m &= m - 1;
                                    const int kk = bs.hamming_weight - __popc(a0 >> lo); // set bits below lo
                                    const uint32_t d = (uint32_t)s_binom[lo * LSK_BINOM_K + kk];
                                    const uint32_t idx = ((a0 >> lo) & 1) ? i32 + d : i32 - d;
                                    if (CPLX) {
                                        const double2 q = reinterpret_cast<double2 const *>(x)[idx];
                                        xv[u] = q.x;
                                        xw[u] = q.y;
                                    } else xv[u] = x[idx];
                                }
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                accr = fma(vr, xv[u], accr);
                                if (CPLX) acci = fma(vr, xw[u], acci);
                            }
                        }
                        lo_end = split;
                    }
                }
#pragma unroll 4
                for (int lo = lo_begin; lo < lo_end; ++lo) {
                    const bool bit = (a >> lo) & 1;
                    const bool act = (tdiff >> lo) & 1;
                    const BT d = s_binom[lo * LSK_BINOM_K + k];
                    k += bit ? 1 : 0;
                    if (sizeof(W) == 4) {
                        const uint32_t i32 = (uint32_t)ig;
                        uint32_t idx = bit ? i32 + (uint32_t)d : i32 - (uint32_t)d;
                        if (PULL) {
                            idx = act ? idx : i32;
                            if (CPLX) {
                                double yr = x[2 * (size_t)idx], yi = x[2 * (size_t)idx + 1];
                                // conj(v) * x[idx]
                                accr += act ? (vr * yr + vi * yi) : 0.0;
                                acci += act ? (vr * yi - vi * yr) : 0.0;
                            } else {
                                double yv = x[idx];
                                accr = fma(act ? vr : 0.0, yv, accr);
                            }
                        } else if (act) {
                            if (CPLX) {
                                atomic_add_f64(y + 2 * (size_t)idx, vr * xr - vi * xi);
                                atomic_add_f64(y + 2 * (size_t)idx + 1, vr * xi + vi * xr);
                            } else atomic_add_f64(y + idx, vr * xr);
                        }
                    } else {
                        int64_t idx = bit ? ig + (int64_t)d : ig - (int64_t)d;
                        if (PULL) {
                            idx = act ? idx : ig;
                            if (CPLX) {
                                double yr = x[2 * idx], yi = x[2 * idx + 1];
                                accr += act ? (vr * yr + vi * yi) : 0.0;
                                acci += act ? (vr * yi - vi * yr) : 0.0;
                            } else {
                                double yv = x[idx];
                                accr = fma(act ? vr : 0.0, yv, accr);
                            }
                        } else if (act) {
                            if (CPLX) {
                                atomic_add_f64(y + 2 * idx, vr * xr - vi * xi);
                                atomic_add_f64(y + 2 * idx + 1, vr * xi + vi * xr);
                            } else atomic_add_f64(y + idx, vr * xr);
                        }
                    }
                }
            }
        }
        // ---- everything else: generic groups ----------------------------------------------------
        for (int g = g_begin; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            group_coeff<REAL>(G, off, (uint64_t)a, cr, ci);
            if (cr == 0.0 && (REAL || ci == 0.0)) continue;
            W beta = a ^ (W)G.x;
            bool flipped = false;
            if (INV) { // K3
                W f = beta ^ site_mask;
                if (f < beta) { beta = f; flipped = true; cr *= (double)bs.spin_inversion; ci *= (double)bs.spin_inversion; }
            }
            int64_t idx;
            if (INDEX == LSK_INDEX_IDENTITY) idx = (int64_t)beta;
            else if (INDEX == LSK_INDEX_COMBINADIC) {
                if (G.adj >= 0 && !flipped && WT::popc(a & (W)G.x) == 1) {
                    // adjacent transposition: rank changes by C(lo, #set bits below lo)
                    int k = WT::popc(a & (W)(((uint64_t)1 << G.adj) - 1));
                    int64_t d = (int64_t)s_binom[G.adj * LSK_BINOM_K + k];
                    idx = ((a >> G.adj) & 1) ? ig + d : ig - d;
                } else {
                    // a state of another Hamming weight is outside the basis: ls_hs_state_index would
                    // return a negative index and the reference halts (DMV:115-118)
                    if (WT::popc(beta) != bs.hamming_weight) { atomicExch(err, 1); continue; }
                    idx = rank_combinadic_w<W, BT>(beta, s_binom);
                }
            } else {
                idx = search_index(ix, (uint64_t)beta);
                if (idx < 0) { atomicExch(err, 1); continue; } // DMV:115-118
            }
            if (PULL) {
                // conj(c) * x[idx]
                if (CPLX) {
                    double yr = x[2 * idx], yi = x[2 * idx + 1];
                    accr += cr * yr + ci * yi;
                    acci += cr * yi - ci * yr;
                } else accr += cr * x[idx];
            } else {
                if (CPLX) {
                    atomic_add_f64(y + 2 * idx, cr * xr - ci * xi);
                    atomic_add_f64(y + 2 * idx + 1, cr * xi + ci * xr);
                } else atomic_add_f64(y + idx, cr * xr);
            }
        }
        if (PULL) {
            if (CPLX) { y[2 * i] = accr; y[2 * i + 1] = acci; } else __builtin_nontemporal_store(accr, y + i);
        }
    }
}

// first pair index handled wave-uniformly by the 32-bit pull row kernels.  Measured on chain_32: 14 is best for k_direct,
// 12 (= every pair outside the LDS window) for k_chain_t.
constexpr int kDirectHighPair = 14;
template <typename W, bool CPLX, int INDEX, bool INV, bool PULL>
static int launch_direct3(lsk_operator op, lsk_basis bs, lsk_index ix, lsk_tilemap tm, uint64_t const *reps,
                          void const *x, void *y, int *d_err, void *stream, int gx, int64_t const *row_gidx) {
    int64_t gb = tm.slots_per_xcd * 8;
    // (f64 vectors only ever meet real operators: the plan refuses the other combination, so it is not instantiated)
    constexpr bool kCplxOp = CPLX;
    int64_t cap;
    if constexpr (kCplxOp) cap = op.is_real ? resident_grid(k_direct<W, CPLX, INDEX, INV, PULL, true>, gb) : resident_grid(k_direct<W, CPLX, INDEX, INV, PULL, false>, gb);
    else cap = resident_grid(k_direct<W, CPLX, INDEX, INV, PULL, true>, gb);
    cap &= ~(int64_t)7; // XCD dealing needs a multiple of 8
    if (cap < 8) cap = 8;
    if (gb > cap) gb = cap; // persistent: one block per 256-row tile costs more than it gains here (13.3 -> 15.5 ms on chain_32)
    dim3 g((unsigned)gb), b(kBlock);
    gx = (gx & 1) | (kDirectHighPair << 24);
    if (op.is_real || !kCplxOp)
        hipLaunchKernelGGL((k_direct<W, CPLX, INDEX, INV, PULL, true>), g, b, 0, (hipStream_t)stream, op.runs,
                           op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs, ix, tm.entries, tm.slots_per_xcd, reps,
                           (double const *)x, (double *)y, d_err, gx, row_gidx);
    else if constexpr (kCplxOp)
        hipLaunchKernelGGL((k_direct<W, CPLX, INDEX, INV, PULL, false>), g, b, 0, (hipStream_t)stream, op.runs,
                           op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs, ix, tm.entries, tm.slots_per_xcd, reps,
                           (double const *)x, (double *)y, d_err, gx, row_gidx);
    LSK_LAUNCH_CHECK();
    return 0;
}
template <typename W, bool CPLX, int INDEX>
static int launch_direct2(lsk_operator op, lsk_basis bs, lsk_index ix, int pull, lsk_tilemap n,
                          uint64_t const *reps, void const *x, void *y, int *d_err, void *stream, int gx,
                          int64_t const *row_gidx) {
    const bool inv = bs.proj == LSK_PROJ_INVERSION;
    if (inv) {
        if (pull) return launch_direct3<W, CPLX, INDEX, true, true>(op, bs, ix, n, reps, x, y, d_err, stream, gx, row_gidx);
        return launch_direct3<W, CPLX, INDEX, true, false>(op, bs, ix, n, reps, x, y, d_err, stream, gx, row_gidx);
    }
    if (pull) return launch_direct3<W, CPLX, INDEX, false, true>(op, bs, ix, n, reps, x, y, d_err, stream, gx, row_gidx);
    return launch_direct3<W, CPLX, INDEX, false, false>(op, bs, ix, n, reps, x, y, d_err, stream, gx, row_gidx);
}
template <bool CPLX, int INDEX>
static int launch_direct1(lsk_operator op, lsk_basis bs, lsk_index ix, int pull, lsk_tilemap n,
                          uint64_t const *reps, void const *x, void *y, int *d_err, void *stream, int gx,
                          int64_t const *row_gidx) {
    // 32-bit states: every site, and every rank, fits 32 bits (C(32, 16) < 2^31)
    if constexpr (INDEX == LSK_INDEX_COMBINADIC)
        if (bs.number_sites <= 32) return launch_direct2<uint32_t, CPLX, INDEX>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
    return launch_direct2<uint64_t, CPLX, INDEX>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
}
static int direct_dispatch(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int pull, lsk_tilemap n,
                           uint64_t const *reps, void const *x, void *y, int *d_err, void *stream, int gx,
                           int64_t const *row_gidx) {
    if (n.slots_per_xcd == 0) return 0;
    if (!n.entries) { snprintf(g_err, sizeof(g_err), "lsk_direct: no tile map"); return -1; }
    if (bs.proj == LSK_PROJ_FULL) { snprintf(g_err, sizeof(g_err), "lsk_direct: basis needs projection"); return -1; }
    switch (ix.kind) {
    case LSK_INDEX_IDENTITY:
        return cplx ? launch_direct1<true, LSK_INDEX_IDENTITY>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx)
                    : launch_direct1<false, LSK_INDEX_IDENTITY>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
    case LSK_INDEX_COMBINADIC:
        return cplx ? launch_direct1<true, LSK_INDEX_COMBINADIC>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx)
                    : launch_direct1<false, LSK_INDEX_COMBINADIC>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
    default:
        return cplx ? launch_direct1<true, LSK_INDEX_SEARCH>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx)
                    : launch_direct1<false, LSK_INDEX_SEARCH>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
    }
}
extern "C" int lsk_direct(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int pull, lsk_tilemap tm,
                          uint64_t const *reps, void const *x, void *y, int *d_err, void *stream) {
    return direct_dispatch(op, bs, ix, cplx, pull, tm, reps, x, y, d_err, stream, 0, nullptr);
}
// replicated-x pull: `reps` = the n rows of one partition, `x` = whole vector in global order, `ix` = index
// of the GLOBAL basis, row_gidx[i] = global index of row i (only read for SEARCH indices)
extern "C" int lsk_direct_gx(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, lsk_tilemap tm, uint64_t const *reps,
                             int64_t const *row_gidx, void const *x_global, void *y, int *d_err, void *stream) {
    return direct_dispatch(op, bs, ix, cplx, 1, tm, reps, x_global, y, d_err, stream, 1, row_gidx);
}

// ---------------------------------------------------------------------------------------------
// Staged row kernel for chain-like operators (pull): the full fixed-Hamming-weight basis (row i = i-th state), exchange
// runs of adjacent pairs plus at most two other exchange pairs whose partner ranks the plan caches (anything else stays
// with k_direct).
//
// The generic row kernel above is bound by the vector-memory address unit (TA busy > 80 %: every gather is a wave
// instruction whatever it hits).  An adjacent pair (lo, lo + 1) moves a state by C(lo, k) <= C(11, 5) = 462 ranks when
// lo < 12, so those twelve gathers stay inside a window of the block's own rows +- 512: the block loads the window into
// LDS once (coalesced 16-byte loads -- the price of the old own-x load) and reads the near partners from LDS.  Pairs
// >= 12: the 64 consecutive states of a wave agree on every bit >= 12 in 92 % of the waves, so the anti-alignment test,
// the bit count below the pair and the rank shift are the same for the whole wave: an aligned far pair issues nothing,
// an anti-aligned one costs add + address + fma.  Waves that straddle two high parts take the per-lane loop.
// Measured on chain_32 f64 (gpurun_out/r2/ablate_sweep.log, pairs dropped from the top): streaming part 2.85 ms, the 12
// LDS pairs +2.2 ms, pairs 12..19 +1.25 ms (mostly L2 hits), pairs 20..30 +3.1-3.75 ms (every gather misses the L2:
// 27 GB at the fabric rate), the cached ring-closing pair +1.0-1.6 ms.
// ---------------------------------------------------------------------------------------------
constexpr int kChainHalo = 512;    // >= C(11, 5)
constexpr int kChainLdsPairs = 12; // pairs lo < 12 are served from the LDS window
constexpr int kChainFar = 12;      // far-pair gathers in flight per row before the first wait (8 / 10 / 12: 7.67 / 7.66 / 7.63 ms)

// W = state word (u32 up to 32 sites, u64 up to 64), R = rank type (u32 while the basis has < 2^32 - 1 states, else u64),
// CPLX = complex128 vectors (real operator; the window holds double2, gathers are 16 bytes per lane), TILE rows per
// block iteration (1024 for f64, 512 for c128: 5 blocks per CU).
// Far pairs (>= hb) of a wave-row are priced lane-parallel: lane l looks at pair split + l of the wave-uniform state a0
// (anti-aligned?, bits below, binomial from LDS, sign), and the loop over the anti-aligned ones only does ballot-mask
// ctz + v_readlane + add + gather.  (The round-1 kernel did that arithmetic on the scalar unit, ~340 scalar
// instructions per 64 rows; moving it to the vector lanes changed nothing measurable: 10.70 vs 10.71 ms.)
template <typename W, typename R> struct ChainTraits;
template <> struct ChainTraits<uint32_t, uint32_t> { static constexpr int NB = 32; };
template <> struct ChainTraits<uint64_t, uint32_t> { static constexpr int NB = 64; };
template <> struct ChainTraits<uint64_t, uint64_t> { static constexpr int NB = 64; };

template <typename T> __device__ __forceinline__ T readlane_t(T v, int lane);
template <> __device__ __forceinline__ uint32_t readlane_t<uint32_t>(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
template <> __device__ __forceinline__ uint64_t readlane_t<uint64_t>(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <typename T> __device__ __forceinline__ T readfirstlane_t(T v);
template <> __device__ __forceinline__ uint32_t readfirstlane_t<uint32_t>(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
template <> __device__ __forceinline__ uint64_t readfirstlane_t<uint64_t>(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <bool CPLX> struct ChainX { typedef double type; };
template <> struct ChainX<true> { typedef double2 type; };
__device__ __forceinline__ void cx_fma(double c, double v, double &acc) { acc = fma(c, v, acc); }
__device__ __forceinline__ void cx_fma(double c, double2 v, double2 &acc) { acc.x = fma(c, v.x, acc.x); acc.y = fma(c, v.y, acc.y); }
__device__ __forceinline__ double cx_scale(double c, double v) { return c * v; }
__device__ __forceinline__ double2 cx_scale(double c, double2 v) { return make_double2(c * v.x, c * v.y); }
template <typename X> __device__ __forceinline__ X cx_zero();
template <> __device__ __forceinline__ double cx_zero<double>() { return 0.0; }
template <> __device__ __forceinline__ double2 cx_zero<double2>() { return make_double2(0.0, 0.0); }
__device__ __forceinline__ void cx_store_nt(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void cx_store_nt(double2 *p, double2 v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
}

constexpr int kChainFarC = 6; // complex: 6 x 16 bytes in flight per lane

// launch bounds, second argument = waves per SIMD the register allocation must allow.  256-thread blocks are admitted per CU
// up to floor(800 / (ceil(sgpr / 16) * 16 + 16)): at 98 SGPRs the 7th block does not fit while the occupancy API still
// answers 7 -- a straggler round of blocks, measured +24 % (13.1 vs 10.7 ms on chain_32)
// (f64: 7 blocks = what 22.5 KB of LDS admit; c128: bounds 4 / 5 / 6 measure 14.56 / 14.55 / 14.57 ms; the 64-bit f64
// instantiations -- 33..64 sites, no in-tree config -- need 6: at 7 they spill 20-40 bytes per lane to scratch)
template <typename W, typename R, bool CPLX, int TILE, bool REC>
__global__ __launch_bounds__(kBlock, (CPLX || sizeof(R) == 8 ? 5 : (sizeof(W) == 8 ? 6 : 7))) void k_chain_t(lsk_runs runs, int n_diag, lsk_term const *__restrict__ diag,
                                                    int hamming_weight, uint4 const *__restrict__ g_img, int img16, int kc,
                                                    int near_off,
                                                    uint64_t const *__restrict__ tilemap, int64_t slots_per_xcd, int64_t n,
                                                    uint64_t const *__restrict__ reps, void const *__restrict__ x_v,
                                                    void *__restrict__ y_v, int hb, int n_cached,
                                                    R const *__restrict__ cache, double cv0, double cv1, int64_t row0,
                                                    int64_t n_x) {
    typedef typename ChainX<CPLX>::type X;
    typedef WordTraits<W> WT;
    constexpr int NB = ChainTraits<W, R>::NB;
    // complex vectors: 11 LDS pairs and a 256-row halo (>= C(10, 5)) keep the block at 20.7 KB like the f64 one, i.e. more
    // resident blocks per CU; pair 11 then gathers from global memory (its partners are <= 462 rows away: L1 / L2 hits)
    constexpr int HALO = CPLX ? 256 : kChainHalo;
    constexpr int LDSP = CPLX ? 11 : kChainLdsPairs;
    constexpr int WINDOW = TILE + 2 * HALO + 2;
    constexpr int FAR = CPLX ? kChainFarC : kChainFar;
    constexpr R kNone = ~(R)0;
    X const *__restrict__ x = (X const *)x_v;
    X *__restrict__ y = (X *)y_v;
    // profiling builds only (make ablate, LS_AMD_ABLATE & 128): ONE MORE 8-byte stream per row, prefetched like the records -- what a
    // byte per row costs this kernel, i.e. what computing sigma instead of loading it could save at best (scripts/chain_stream_cost.py)
    const bool extra_stream = kAblate && (n_cached & 0x100);
    if (kAblate) n_cached &= 0xff;
    uint64_t ex_next = 0;
    // LDS image made once by the host (chain_lds_image): the binomial table in the rank type, NB rows of kc = weight + 2
    // columns, then the near-pair table (below); copied with 16-byte loads -- one block per tile means once per 1024 rows
    extern __shared__ uint4 s_img[];
    R const *const s_binom = reinterpret_cast<R const *>(s_img);
    uint2 const *const s_near = reinterpret_cast<uint2 const *>(reinterpret_cast<char const *>(s_img) + near_off);
    __shared__ X s_x[WINDOW + 1]; // last slot: 0, read by the lanes whose near pair is aligned
    for (int k = threadIdx.x; k < img16; k += kBlock) s_img[k] = g_img[k];
    (void)NB;
    if (threadIdx.x == 0) s_x[WINDOW] = cx_zero<X>();
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3;
    const int lane = threadIdx.x & 63;
    tilemap += (int64_t)xcd * slots_per_xcd;
    uint32_t const *__restrict__ reps32 = reinterpret_cast<uint32_t const *>(reps);
    for (int64_t t = blockIdx.x >> 3; t < slots_per_xcd; t += blocks_per_xcd) {
        const uint64_t slot = tilemap[t];
        const int cnt = (int)(slot >> 48);
        if (cnt == 0) continue; // block-uniform
        const int64_t i0 = (int64_t)(slot & 0xffffffffffffULL);
        const int64_t w0 = (row0 + i0 - HALO) & ~(int64_t)1; // first row of the window (even; may be < 0)
        W a_next = 0;
        R t0_next = kNone, t1_next = kNone;
        // REC: `reps` is the plan's fused record array, row -> sigma (low word) | partner rank of the first cached pair
        // (high word, ~0 = none): one 8-byte load per row instead of a 4-byte state and a 4-byte cache load
        auto load_row = [&](int64_t row, W &a_out, R &t0_out, R &t1_out) {
            if (REC) {
                const uint64_t rec = __builtin_nontemporal_load(reps + row);
                a_out = (W)(uint32_t)rec;
                t0_out = (R)(uint32_t)(rec >> 32);
                if (n_cached == 0) t0_out = kNone;
            } else {
                if (sizeof(W) == 4) a_out = (W)__builtin_nontemporal_load(reps32 + 2 * row); // low word only
                else a_out = (W)__builtin_nontemporal_load(reps + row);
                if (n_cached > 0) t0_out = __builtin_nontemporal_load(cache + row);
            }
            if (n_cached > 1) t1_out = __builtin_nontemporal_load(cache + (size_t)n + (size_t)row);
            if (kAblate && extra_stream) ex_next = __builtin_nontemporal_load(reps + (row < (n >> 1) ? row + (n >> 1) : row - (n >> 1)));
        };
        // Lanes past the end of a partial tile stay ACTIVE as ghosts of the tile's last row (they recompute it and
        // store nothing): the far pairs are priced lane-parallel, which needs every lane of a live wave.
        const int wave0 = (int)(threadIdx.x & ~63u);
        if (wave0 < cnt) { // first row of this thread: requested before the window is staged
            const int64_t rr = i0 + ((int)threadIdx.x < cnt ? (int)threadIdx.x : cnt - 1);
            load_row(rr, a_next, t0_next, t1_next);
        }
        __syncthreads(); // every wave is done with the previous window (and s_binom is loaded)
        if (CPLX) {
            for (int j = threadIdx.x; j < WINDOW; j += kBlock) {
                const int64_t row = w0 + j;
                s_x[j] = (row >= 0 && row < n_x) ? x[row] : cx_zero<X>();
            }
        } else {
            double const *xd = (double const *)x_v;
            double *sd = (double *)s_x;
            for (int j = 2 * threadIdx.x; j < WINDOW; j += 2 * kBlock) {
                const int64_t row = w0 + j;
                double2 v;
                if (row >= 0 && row + 1 < n_x) v = *reinterpret_cast<double2 const *>(xd + row);
                else { v.x = (row >= 0 && row < n_x) ? xd[row] : 0.0; v.y = (row + 1 >= 0 && row + 1 < n_x) ? xd[row + 1] : 0.0; }
                sd[j] = v.x;
                sd[j + 1] = v.y;
            }
        }
        __syncthreads();
        const int own0 = (int)(row0 + i0 - w0);
        X y_pending = cx_zero<X>(); // stored one iteration late (see k_chain)
        int64_t i_pending = -1;
#pragma unroll 1
        for (int sub = 0; sub < TILE / kBlock; ++sub) {
            const int r = sub * kBlock + threadIdx.x;
            if (i_pending >= 0) cx_store_nt(y + i_pending, y_pending);
            i_pending = -1;
            const W a = a_next;
            const R t0 = t0_next, t1 = t1_next;
            const uint64_t ex = ex_next;
            if (sub + 1 < TILE / kBlock && (sub + 1) * kBlock + wave0 < cnt) {
                const int64_t in = i0 + (r + kBlock < cnt ? r + kBlock : cnt - 1);
                load_row(in, a_next, t0_next, t1_next);
            }
            if (sub * kBlock + wave0 >= cnt) continue; // wave-uniform: the whole wave is past the end
            const bool ghost = r >= cnt;
            const int64_t i = i0 + (ghost ? cnt - 1 : r);
            const R ig = (R)(row0 + i);
            X g0 = cx_zero<X>(), g1 = cx_zero<X>();
            if (n_cached > 0) g0 = x[t0 != kNone ? t0 : ig];
            if (n_cached > 1) g1 = x[t1 != kNone ? t1 : ig];
            const int jr = own0 + (int)(i - i0);
            const X xr = s_x[jr];
            double dr, di;
            diag_coeff<W, true>(runs, n_diag, diag, a, dr, di);
            X acc = cx_scale(dr, xr);
            const W tdiff = a ^ (a >> 1);
            // The rows of a wave ascend, so every lane shares the common prefix of the first and the last state: pairs at or
            // above `ubit` (one past the highest bit on which those two differ) are wave-uniform.  92 % of the waves of
            // chain_32 have ubit <= 12; the others used to fall back to the per-lane loop for ALL their far pairs and now
            // only walk the pairs below ubit per lane.
            const W a0 = readfirstlane_t<W>(a);
            const W adiff = a0 ^ readlane_t<W>(a, 63);
            const int ubit = adiff == 0 ? 0 : (int)(8 * sizeof(W)) - (sizeof(W) == 4 ? __clz((int)(uint32_t)adiff) : __clzll((long long)(uint64_t)adiff));
            const R ig0 = readfirstlane_t<R>(ig);
            const uint32_t dl = (uint32_t)(ig - ig0); // 0..63: the far gathers address x as (uniform base) + dl
            for (int q = 0; q < runs.n_runs; ++q) {
                const int lo0 = runs.lo0[q];
                int lo_end = lo0 + runs.cnt[q];
                const double vr = runs.v_re[q];
                int k = WT::popc(a & (W)(((uint64_t)1 << lo0) - 1));
                int lo = lo0;
                const int e1 = lo_end < LDSP ? lo_end : LDSP;
                const int near_end = lo0 > e1 ? lo0 : e1;
                int split = hb == 0 ? lo_end : (hb < near_end ? near_end : (hb > lo_end ? lo_end : hb));
                if (hb != 0 && split < ubit) split = ubit < lo_end ? ubit : lo_end;
                // ---- far pairs: the 64 states of the wave agree on every bit >= split ------------------------------
                const bool uni = split < lo_end;
                unsigned long long m = 0;
                R off = 0;
                X xv[FAR];
#pragma unroll
                for (int u = 0; u < FAR; ++u) xv[u] = cx_zero<X>();
                if (uni) {
                    // lane l prices pair p = split + l of the common state
                    const int p = split + lane;
                    const bool in_run = p < lo_end;
                    const int ps = in_run ? p : 0;
                    const W hi = a0 >> ps;
                    const bool bit = hi & 1;
                    const bool act = in_run && (((hi >> 1) & 1) != (W)bit);
                    const int kk = hamming_weight - WT::popc(hi); // set bits below p
                    const R d = s_binom[ps * kc + (kk < 0 ? 0 : kk)];
                    off = bit ? d : (R)(0 - d);
                    m = __builtin_amdgcn_ballot_w64(act);
#pragma unroll
                    for (int u = 0; u < FAR; ++u) {
                        if (m) {
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            xv[u] = (x + (size_t)(R)(ig0 + readlane_t<R>(off, l)))[dl];
                        }
                    }
                    lo_end = split;
                }
                // ---- near pairs: partner inside the LDS window ----------------------------------------
                if (lo0 == 0 && e1 == LDSP) {
                    // The usual case, the run covers every LDS pair: the byte displacements of four pairs at a time come
                    // from a table indexed by the five state bits they touch and the number of set bits below them
                    // (chain_lds_image); an aligned pair holds a displacement that clamps to the zero slot.  3-4 VALU
                    // instructions per pair instead of 12 (two bit tests, binomial address, sign, select, scale).
                    const uint32_t al = (uint32_t)a;
                    const uint32_t jb = (uint32_t)jr * (uint32_t)sizeof(X);
                    constexpr uint32_t ZOFF = (uint32_t)WINDOW * (uint32_t)sizeof(X);
                    char const *const sb = reinterpret_cast<char const *>(s_x);
                    const uint2 q0 = s_near[al & 31u];
                    const uint2 q1 = s_near[32 + 32 * __popc(al & 15u) + ((al >> 4) & 31u)];
                    const uint2 q2 = s_near[192 + 32 * __popc(al & 255u) + ((al >> 8) & 31u)];
                    const uint32_t wq[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
                    // LDS reads in flight before their fmas (pair order kept).  Measured on chain_32 f64: 1 / 2 / 3 / 4 in flight
                    // 7.64 / 7.63 / 7.93 / 7.93 ms -- the registers of a deeper batch cost more than its latency hiding buys;
                    // c128 is indifferent (14.55-14.67 ms with or without the table)
                    constexpr int NBATCH = 2;
#pragma unroll
                    for (int p0 = 0; p0 < LDSP; p0 += NBATCH) {
                        X nv[NBATCH];
#pragma unroll
                        for (int u = 0; u < NBATCH; ++u) {
                            const int p = p0 + u;
                            const uint32_t w = wq[(p < LDSP ? p : 0) >> 1];
                            const int32_t d = (p & 1) ? ((int32_t)w >> 16) : (int32_t)(int16_t)(w & 0xffffu);
                            uint32_t o = jb + (uint32_t)d;
                            o = o < ZOFF ? o : ZOFF;
                            nv[u] = p < LDSP ? *reinterpret_cast<X const *>(sb + o) : cx_zero<X>();
                        }
#pragma unroll
                        for (int u = 0; u < NBATCH; ++u)
                            if (p0 + u < LDSP) cx_fma(vr, nv[u], acc);
                    }
                    lo = LDSP;
                    k = __popc(al & ((1u << LDSP) - 1u));
                }
#pragma unroll 4
                for (; lo < e1; ++lo) {
                    const bool bit = (a >> lo) & 1;
                    const bool act = (tdiff >> lo) & 1;
                    const int d = (int)s_binom[lo * kc + k];
                    k += bit ? 1 : 0;
                    const int j = bit ? jr + d : jr - d;
                    cx_fma(vr, s_x[act ? j : WINDOW], acc);
                }
#pragma unroll
                for (int u = 0; u < FAR; ++u) cx_fma(vr, xv[u], acc); // zero-filled slots included: counting the gathers and
                                                                      // branching around idle fmas measured slower (7.72 vs 7.59 ms)
                while (m) { // more than FAR anti-aligned far pairs
                    X xw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xw[u] = cx_zero<X>();
                        if (m) {
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            xw[u] = (x + (size_t)(R)(ig0 + readlane_t<R>(off, l)))[dl];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) cx_fma(vr, xw[u], acc);
                }
                // ---- middle pairs (and far pairs of a wave that straddles two high parts) --------------
#pragma unroll 4
                for (; lo < lo_end; ++lo) {
                    const bool bit = (a >> lo) & 1;
                    const bool act = (tdiff >> lo) & 1;
                    const R d = s_binom[lo * kc + k];
                    k += bit ? 1 : 0;
                    R idx = bit ? (R)(ig + d) : (R)(ig - d);
                    idx = act ? idx : ig;
                    cx_fma(act ? vr : 0.0, x[idx], acc);
                }
            }
            cx_fma(t0 != kNone ? cv0 : 0.0, g0, acc);
            cx_fma(t1 != kNone ? cv1 : 0.0, g1, acc);
            if (kAblate && extra_stream && ex == 0x0123456789abcdefULL) cx_fma(1.0, xr, acc); // (practically never: keeps the load alive)
            y_pending = acc;
            i_pending = ghost ? -1 : i;
        }
        if (i_pending >= 0) cx_store_nt(y + i_pending, y_pending);
    }
}

// cache[i] = rank of reps[i] ^ xmask when exactly one of the two bits of xmask is set in reps[i], else ~0;
// *flag is raised when a partner falls outside the basis (the caller then does not use the cache, and
// the generic path reports the error at run time as the reference does)
template <typename W, typename R>
__global__ __launch_bounds__(kBlock) void k_chain_cache(int64_t n, uint64_t const *__restrict__ reps, uint64_t xmask,
                                                        int hamming_weight, uint64_t const *__restrict__ g_binom,
                                                        R *__restrict__ out, int *__restrict__ flag) {
    constexpr int NB = ChainTraits<W, R>::NB;
    __shared__ R s_binom[NB * LSK_BINOM_K];
    for (int k = threadIdx.x; k < NB * LSK_BINOM_K; k += blockDim.x) s_binom[k] = (R)g_binom[k];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const W a = (W)reps[i];
        R t = ~(R)0;
        if (WordTraits<W>::popc(a & (W)xmask) == 1) {
            const W beta = a ^ (W)xmask;
            if (WordTraits<W>::popc(beta) != hamming_weight) atomicExch(flag, 1);
            else t = (R)rank_combinadic_w<W, R>(beta, s_binom);
        }
        out[i] = t;
    }
}
extern "C" int lsk_chain_cache(lsk_basis bs, lsk_index ix, int64_t n, uint64_t const *reps, uint64_t xmask, void *out,
                               int wide_ranks, int *d_flag, void *stream) {
    if (n == 0) return 0;
    dim3 g(grid_for(n)), b(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (bs.number_sites <= 32 && !wide_ranks)
        hipLaunchKernelGGL((k_chain_cache<uint32_t, uint32_t>), g, b, 0, s, n, reps, xmask, bs.hamming_weight, ix.binom, (uint32_t *)out, d_flag);
    else if (!wide_ranks)
        hipLaunchKernelGGL((k_chain_cache<uint64_t, uint32_t>), g, b, 0, s, n, reps, xmask, bs.hamming_weight, ix.binom, (uint32_t *)out, d_flag);
    else
        hipLaunchKernelGGL((k_chain_cache<uint64_t, uint64_t>), g, b, 0, s, n, reps, xmask, bs.hamming_weight, ix.binom, (uint64_t *)out, d_flag);
    LSK_LAUNCH_CHECK();
    return 0;
}

// fused per-row record of the 32-bit instantiation: sigma | partner rank << 32 (see k_chain_t, REC)
__global__ __launch_bounds__(kBlock) void k_chain_pack(int64_t n, uint64_t const *__restrict__ reps,
                                                       uint32_t const *__restrict__ cache, uint64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = (uint64_t)(uint32_t)reps[i] | ((uint64_t)(cache ? cache[i] : 0xffffffffu) << 32);
}
extern "C" int lsk_chain_pack(int64_t n, uint64_t const *reps, void const *cache, uint64_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_chain_pack, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, reps, (uint32_t const *)cache, out);
    LSK_LAUNCH_CHECK();
    return 0;
}

// binomial table in the rank type of the staged kernel: the u64 table as it is, or a u32 copy made once per device table
__global__ void k_binom_narrow(uint64_t const *__restrict__ in, uint32_t *__restrict__ out, int n) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) out[k] = (uint32_t)in[k];
}
template <typename R> static R const *chain_binom(uint64_t const *g_binom, hipStream_t stream);
template <> uint64_t const *chain_binom<uint64_t>(uint64_t const *g_binom, hipStream_t) { return g_binom; }
template <> uint32_t const *chain_binom<uint32_t>(uint64_t const *g_binom, hipStream_t stream) {
    static uint64_t const *src = nullptr;
    static uint32_t *narrow = nullptr;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    if (src != g_binom || !narrow) {
        if (!narrow && hipMalloc((void **)&narrow, sizeof(uint32_t) * 64 * LSK_BINOM_K) != hipSuccess) return nullptr;
        hipLaunchKernelGGL(k_binom_narrow, dim3(4), dim3(kBlock), 0, stream, g_binom, narrow, 64 * LSK_BINOM_K);
        src = g_binom;
    }
    return narrow;
}

// LDS image of k_chain_t, made once per (rank type, rows, weight, vector type) and kept on the device:
//   [rows][kc] binomials C(n, k), k < kc = weight + 2, in the rank type (low 32 bits for 32-bit ranks), padded to 16 bytes;
//   near-pair table: for g = 0..2 (pairs 4g..4g+3), kidx = number of set bits below bit 4g (0..4g), pat = state bits
//   4g..4g+4: four int16 = signed byte displacement of the partner inside the LDS window of x (elem bytes per row,
//   +C(lo, k) rows when the lower bit of the pair is set, -C(lo, k) when the upper one is), 0x7000 for an aligned pair
//   or a pair >= ldsp (added to any row offset it lands past the window and clamps to the zero slot).
//   Entry index = {0, 32, 192}[g] + 32 kidx + pat.
static int chain_near_fill(uint64_t const (*C)[LSK_BINOM_K], int elem, int ldsp, int16_t *near) {
    int const base[3] = {0, 32, 192};
    for (int g = 0; g < 3; ++g)
        for (int kidx = 0; kidx <= 4 * g; ++kidx)
            for (int pat = 0; pat < 32; ++pat)
                for (int p = 0; p < 4; ++p) {
                    int const lo = 4 * g + p;
                    int const bit = (pat >> p) & 1, nxt = (pat >> (p + 1)) & 1;
                    int const k = kidx + __builtin_popcount(pat & ((1 << p) - 1));
                    int64_t const d = (int64_t)C[lo][k < LSK_BINOM_K ? k : 0] * elem;
                    int16_t v = 0x7000;
                    if (lo < ldsp && bit != nxt && k <= lo) {
                        if (d >= 0x7000) { snprintf(g_err, sizeof(g_err), "chain_near_fill: displacement out of range"); return -1; }
                        v = (int16_t)(bit ? d : -d);
                    }
                    near[4 * (base[g] + 32 * kidx + pat) + p] = v;
                }
    return 0;
}
static void chain_binomials(uint64_t (*C)[LSK_BINOM_K]) {
    for (int n = 0; n < 64; ++n)
        for (int k = 0; k < LSK_BINOM_K; ++k) C[n][k] = k == 0 ? 1 : (n == 0 ? 0 : C[n - 1][k - 1] + C[n - 1][k]);
}
// host test hook: the near-pair table alone (480 entries of four int16)
extern "C" int lsk_test_chain_near_table(int elem, int ldsp, int16_t *out) {
    uint64_t C[64][LSK_BINOM_K];
    chain_binomials(C);
    return chain_near_fill(C, elem, ldsp, out);
}
struct ChainImage { int rsize, rows, kc, elem, ldsp, device; uint4 *dev; int bytes, near_off; };
static int chain_lds_image(int rsize, int rows, int weight, int elem, int ldsp, ChainImage *out) {
    static std::vector<ChainImage> cache;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    int kc = (weight < 0 ? rows : weight) + 2;
    if (kc > LSK_BINOM_K) kc = LSK_BINOM_K;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0; // the image lives in the memory of the device it was made on
    for (ChainImage const &c : cache)
        if (c.rsize == rsize && c.rows == rows && c.kc == kc && c.elem == elem && c.ldsp == ldsp && c.device == device) { *out = c; return 0; }
    uint64_t C[64][LSK_BINOM_K];
    chain_binomials(C);
    int const near_off = (rows * kc * rsize + 15) & ~15;
    int const bytes = near_off + 480 * 8;
    std::vector<unsigned char> img((size_t)bytes, 0);
    for (int n = 0; n < rows; ++n)
        for (int k = 0; k < kc; ++k) {
            if (rsize == 4) reinterpret_cast<uint32_t *>(img.data())[n * kc + k] = (uint32_t)C[n][k];
            else reinterpret_cast<uint64_t *>(img.data())[n * kc + k] = C[n][k];
        }
    if (chain_near_fill(C, elem, ldsp, reinterpret_cast<int16_t *>(img.data() + near_off)) != 0) return -1;
    ChainImage c = {rsize, rows, kc, elem, ldsp, device, nullptr, bytes, near_off};
    if (hipMalloc((void **)&c.dev, (size_t)bytes) != hipSuccess) { snprintf(g_err, sizeof(g_err), "chain_lds_image: no device memory"); return -1; }
    if (hipMemcpy(c.dev, img.data(), (size_t)bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(c.dev); snprintf(g_err, sizeof(g_err), "chain_lds_image: copy failed"); return -1; }
    cache.push_back(c);
    *out = c;
    return 0;
}

template <typename W, typename R, bool CPLX, int TILE, bool REC>
static int launch_chain(lsk_operator op, lsk_basis bs, lsk_index ix, lsk_tilemap tm, int64_t n, uint64_t const *reps,
                        int64_t row0, int64_t n_x, void const *x, void *y, int n_cached, void const *cache, double cv0,
                        double cv1, void *stream) {
    int64_t gb = tm.slots_per_xcd * 8;
    ChainImage img;
    if (chain_lds_image((int)sizeof(R), ChainTraits<W, R>::NB, bs.hamming_weight, CPLX ? 16 : 8, CPLX ? 11 : kChainLdsPairs, &img) != 0) return -1;
    // One block per tile (block b -> tile b / 8 of XCD list b % 8), NOT a persistent grid: measured on chain_32 8.47 vs 10.85 ms
    // (f64) and 15.7 vs 20.8 ms (c128).  Persistent blocks start together and stay phase-locked (window load, LDS pairs, far
    // gathers), so the phases' costs add up; blocks dispatched one by one as others retire drift apart and the memory phases of
    // some overlap the LDS / ALU phases of others.
    hipLaunchKernelGGL((k_chain_t<W, R, CPLX, TILE, REC>), dim3((unsigned)gb), dim3(kBlock), (size_t)img.bytes, (hipStream_t)stream, op.runs,
                       op.n_diag, op.diag, bs.hamming_weight, img.dev, img.bytes / 16, img.kc, img.near_off, tm.entries, tm.slots_per_xcd, n, reps, x, y,
                       kChainLdsPairs, n_cached | ((kAblate && (bs.debug_ablate & 128)) ? 0x100 : 0), (R const *)cache, cv0, cv1, row0, n_x);
    LSK_LAUNCH_CHECK();
    return 0;
}

// rows per tile: 1024 (f64) / 512 (c128).  Measured r2 on chain_32: doubling them (fewer blocks, 1.5x instead of 2x window
// loads, but 5 instead of 7 blocks per CU) is slower, 8.72 vs 8.26 ms (f64), 15.5 vs 15.4 ms (c128).
extern "C" int lsk_chain_tile_rows(int cplx) { return cplx ? 512 : 1024; }

extern "C" int lsk_chain(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int wide_ranks, int fused_records, lsk_tilemap tm,
                         int64_t n, uint64_t const *reps, int64_t row0, int64_t n_x, void const *x, void *y, int n_cached,
                         void const *cache, double cv0, double cv1, void *stream) {
    if (n == 0 || tm.slots_per_xcd == 0) return 0;
    const bool narrow = bs.number_sites <= 32 && !wide_ranks;
#define LSK_CHAIN_ARGS op, bs, ix, tm, n, reps, row0, n_x, x, y, n_cached, cache, cv0, cv1, stream
    if (fused_records) { // `reps` is the record array made by lsk_chain_pack (32-bit states and ranks only)
        if (!narrow || cplx) { snprintf(g_err, sizeof(g_err), "lsk_chain: fused records need 32-bit states and ranks and f64 vectors"); return -1; }
        return launch_chain<uint32_t, uint32_t, false, 1024, true>(LSK_CHAIN_ARGS);
    }
    if (narrow) {
        if (!cplx) { snprintf(g_err, sizeof(g_err), "lsk_chain: 32-bit states and ranks with f64 vectors run on fused records (lsk_chain_pack)"); return -1; }
        return launch_chain<uint32_t, uint32_t, true, 512, false>(LSK_CHAIN_ARGS);
    }
    if (!wide_ranks) return cplx ? launch_chain<uint64_t, uint32_t, true, 512, false>(LSK_CHAIN_ARGS) : launch_chain<uint64_t, uint32_t, false, 1024, false>(LSK_CHAIN_ARGS);
    return cplx ? launch_chain<uint64_t, uint64_t, true, 512, false>(LSK_CHAIN_ARGS) : launch_chain<uint64_t, uint64_t, false, 1024, false>(LSK_CHAIN_ARGS);
#undef LSK_CHAIN_ARGS
}

// ---------------------------------------------------------------------------------------------
// Staged row kernel for ANY set of exchange pairs (k_pairs_t): real Hermitian operators whose off-diagonal part is a sum of
// v_p (|01><10| + |10><01|) over arbitrary site pairs (i_p, j_p) and whose diagonal is a sum of vz_p s_i s_j over the same
// pairs -- the Heisenberg / XXZ model on any lattice (square, kagome, J1-J2 rings, ...), on the full fixed-weight basis of
// <= 32 sites, pull form.  k_chain_t is the special case "adjacent pairs", which is faster still for rings; everything that is
// not a ring used to run the generic row kernel (k_direct: one full combinadic re-ranking per non-adjacent non-zero).
//
// A state is high | low with low = bits 0..10.  In the ascending order all 11-bit words of weight kl under one `high` are a
// contiguous block of C(11, kl) <= 462 rows (kl = weight - popcount(high)), and rank = blockstart(high) + rank_low[low]
// (rank_low: 2048 x u16 in LDS).  The 64 rows of a wave are consecutive, so they almost always lie in ONE block; a wave that
// straddles blocks is processed block segment by block segment (same code, partial lane mask).  Per segment the pairs fall
// into three classes:
//   NEAR      both sites in `low`: the partner is in the same block, at rank_low[low ^ m] - rank_low[low] rows: LDS window
//             of x (+-512 rows), two LDS reads and a handful of VALU instructions per pair and lane;
//   HIGH      both sites in `high`: anti-alignment and the rank shift are the same for every lane of the segment: priced once,
//             lane-parallel (lane l <-> pair l: the shift is a sum over the set bits between the two sites), then a loop over the
//             anti-aligned pairs only -- readlane, add, gather of 64 CONSECUTIVE elements of x, fma;
//   STRADDLE  i in `low`, j in `high`: the partner block (high ^ bit j, kl -+ 1) is the same for every lane: its start is priced
//             once, lane-parallel; per lane the partner is blockstart + rank_low[low ^ bit i] (a gather inside a <= 3.7 KB block).
// The diagonal comes out of the same loops: d(a) = sum_p vz_p - 2 sum_{p anti-aligned} vz_p.
// ---------------------------------------------------------------------------------------------
constexpr int kPairLowBits = 11;
constexpr int kPairHalo = 512; // >= C(11, 5)
constexpr int kPairFar = 8;    // gathers of HIGH pairs in flight per lane
enum { PAIR_NEAR = 0, PAIR_STRADDLE = 1, PAIR_HIGH = 2 };

template <bool CPLX, int TILE>
__global__ __launch_bounds__(kBlock, (CPLX ? 4 : 6)) void k_pairs_t(lsk_pairplan pp, int hamming_weight, uint64_t const *__restrict__ tilemap,
                                                              int64_t slots_per_xcd, int64_t n, void const *__restrict__ x_v,
                                                              void *__restrict__ y_v) {
    typedef typename ChainX<CPLX>::type X;
    constexpr int HALO = kPairHalo;
    constexpr int WINDOW = TILE + 2 * HALO + 2;
    constexpr int LOWMASK = (1 << kPairLowBits) - 1;
    X const *__restrict__ x = (X const *)x_v;
    X *__restrict__ y = (X *)y_v;
    __shared__ X s_x[WINDOW + 1]; // last slot: 0
    __shared__ uint16_t s_rl[1 << kPairLowBits];
    __shared__ uint32_t s_binom[32 * LSK_PAIR_KC];
    __shared__ lsk_pair s_pairs[LSK_MAX_PAIRS];
    const int n_near = pp.n_near, n_str = pp.n_str, n_high = pp.n_high;
    const int kc = LSK_PAIR_KC;
    for (int k = threadIdx.x; k < (1 << kPairLowBits); k += kBlock) s_rl[k] = pp.rank_low[k];
    for (int k = threadIdx.x; k < 32 * kc; k += kBlock) s_binom[k] = pp.binom[k];
    for (int k = threadIdx.x; k < n_near + n_str + n_high; k += kBlock) s_pairs[k] = pp.pairs[k];
    if (threadIdx.x == 0) s_x[WINDOW] = cx_zero<X>();
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3;
    const int lane = threadIdx.x & 63;
    tilemap += (int64_t)xcd * slots_per_xcd;
    for (int64_t t = blockIdx.x >> 3; t < slots_per_xcd; t += blocks_per_xcd) {
        const uint64_t slot = tilemap[t];
        const int cnt = (int)(slot >> 48);
        if (cnt == 0) continue;
        const int64_t i0 = (int64_t)(slot & 0xffffffffffffULL);
        const int64_t w0 = (i0 - HALO) & ~(int64_t)1;
        __syncthreads(); // every wave is done with the previous window (and the tables are loaded)
        if (CPLX) {
            for (int j = threadIdx.x; j < WINDOW; j += kBlock) {
                const int64_t row = w0 + j;
                s_x[j] = (row >= 0 && row < n) ? x[row] : cx_zero<X>();
            }
        } else {
            double const *xd = (double const *)x_v;
            double *sd = (double *)s_x;
            for (int j = 2 * threadIdx.x; j < WINDOW; j += 2 * kBlock) {
                const int64_t row = w0 + j;
                double2 v;
                if (row >= 0 && row + 1 < n) v = *reinterpret_cast<double2 const *>(xd + row);
                else { v.x = (row >= 0 && row < n) ? xd[row] : 0.0; v.y = (row + 1 >= 0 && row + 1 < n) ? xd[row + 1] : 0.0; }
                sd[j] = v.x;
                sd[j + 1] = v.y;
            }
        }
        __syncthreads();
        const int own0 = (int)(i0 - w0);
        const int wave0 = (int)(threadIdx.x & ~63u);
#pragma unroll 1
        for (int sub = 0; sub < TILE / kBlock; ++sub) {
            if (sub * kBlock + wave0 >= cnt) break; // wave-uniform: the whole wave is past the end
            const int r = sub * kBlock + threadIdx.x;
            const bool ghost = r >= cnt; // lanes past the end stay active as copies of the last row (they store nothing)
            const int64_t i = i0 + (ghost ? cnt - 1 : r);
            const uint32_t a = __builtin_nontemporal_load(pp.states + i);
            const uint32_t ig = (uint32_t)i;
            const uint32_t low = a & LOWMASK, hi = a >> kPairLowBits;
            const int jr = own0 + (int)(i - i0);
            const X xr = s_x[jr];
            X acc = cx_zero<X>();
            double dsub = 0.0; // sum of vz over this row's anti-aligned pairs
            // ---- NEAR pairs ---------------------------------------------------------------------------------------------
            const int rl0 = (int)s_rl[low];
            for (int p = 0; p < n_near; ++p) {
                lsk_pair const P = s_pairs[p];
                const uint32_t m = (1u << P.i) | (1u << P.j);
                const bool act = __popc(low & m) == 1;
                const int r1 = (int)s_rl[low ^ m];
                const int o = jr + (r1 - rl0);
                cx_fma(P.v, s_x[act ? o : WINDOW], acc);
                dsub += act ? P.vz : 0.0;
            }
            // ---- block segments: lanes that share `hi` -----------------------------------------------------------------------
            unsigned long long pending = __builtin_amdgcn_ballot_w64(true);
            while (pending) {
                const int l0 = __builtin_ctzll(pending);
                const uint32_t href = (uint32_t)__builtin_amdgcn_readlane((int)hi, l0);
                const bool inseg = hi == href;
                pending &= ~__builtin_amdgcn_ballot_w64(inseg);
                const int kl = hamming_weight - __popc(href); // set bits of `low`, the same for every lane of the segment
                double dz_u = 0.0;
                // ---- HIGH pairs: priced once, lane l <-> pair pass + l -------------------------------------------------------
                for (int pass = 0; pass < n_high; pass += 64) {
                    const int q = pass + lane;
                    const bool in = q < n_high;
                    lsk_pair const P = s_pairs[n_near + n_str + (in ? q : 0)];
                    const int pi = P.i - kPairLowBits, pj = P.j - kPairLowBits; // positions inside `hi`
                    const uint32_t bi = (href >> pi) & 1u, bj = (href >> pj) & 1u;
                    const bool act = in && bi != bj;
                    // rank of the configuration "bit at i" minus rank of "bit at j": only the set bits at or above i matter
                    int k = kl + __popc(href & ((1u << pi) - 1u)); // set bits of the state below site i
                    uint32_t between = href & ((1u << pj) - 1u) & ~((2u << pi) - 1u);
                    int64_t lowcfg = (int64_t)s_binom[P.i * kc + min(k + 1, kc - 1)], highcfg = 0;
                    int tt = 0;
                    while (between) {
                        const int b = __builtin_ctz(between) + kPairLowBits;
                        between &= between - 1;
                        ++tt;
                        lowcfg += (int64_t)s_binom[b * kc + min(k + 1 + tt, kc - 1)];
                        highcfg += (int64_t)s_binom[b * kc + min(k + tt, kc - 1)];
                    }
                    highcfg += (int64_t)s_binom[P.j * kc + min(k + tt + 1, kc - 1)];
                    const int32_t delta = (int32_t)(bi ? highcfg - lowcfg : lowcfg - highcfg); // partner rank - own rank
                    unsigned long long m = __builtin_amdgcn_ballot_w64(act);
                    while (m) {
                        X xv[kPairFar];
                        double vv[kPairFar];
#pragma unroll
                        for (int u = 0; u < kPairFar; ++u) {
                            xv[u] = cx_zero<X>();
                            vv[u] = 0.0;
                            if (m) {
                                const int l = __builtin_ctzll(m);
                                m &= m - 1;
                                const int32_t d = __builtin_amdgcn_readlane(delta, l);
                                vv[u] = readlane_f64(P.v, l);
                                dz_u += readlane_f64(P.vz, l);
                                xv[u] = x[inseg ? (uint32_t)(ig + (uint32_t)d) : ig];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < kPairFar; ++u) cx_fma(inseg ? vv[u] : 0.0, xv[u], acc);
                    }
                }
                // ---- STRADDLE pairs: the partner block is priced once, the place inside it per lane ----------------------------
                for (int pass = 0; pass < n_str; pass += 64) {
                    const int q = pass + lane;
                    const bool in = q < n_str;
                    lsk_pair const P = s_pairs[n_near + (in ? q : 0)];
                    const int pj = P.j - kPairLowBits;
                    const uint32_t bj = (href >> pj) & 1u;
                    const uint32_t h2 = href ^ (1u << pj);
                    const int kl2 = bj ? kl + 1 : kl - 1; // a bit comes down into `low`, or leaves it
                    uint32_t base = 0;
                    {
                        uint32_t hb = h2;
                        int idx = kl2;
                        while (hb) {
                            const int b = __builtin_ctz(hb) + kPairLowBits;
                            hb &= hb - 1;
                            ++idx;
                            base += s_binom[b * kc + min(max(idx, 0), kc - 1)];
                        }
                    }
                    const int np = min(64, n_str - pass);
                    for (int l = 0; l < np; ++l) {
                        const int pi = __builtin_amdgcn_readlane((int)P.i, l);
                        const uint32_t sbj = (uint32_t)__builtin_amdgcn_readlane((int)bj, l);
                        const uint32_t sbase = (uint32_t)__builtin_amdgcn_readlane((int)base, l);
                        const double sv = readlane_f64(P.v, l), svz = readlane_f64(P.vz, l);
                        const bool act = inseg && ((low >> pi) & 1u) != sbj;
                        const uint32_t idx = act ? sbase + (uint32_t)s_rl[low ^ (1u << pi)] : ig;
                        cx_fma(act ? sv : 0.0, x[idx], acc);
                        dsub += act ? svz : 0.0;
                    }
                }
                dsub += inseg ? dz_u : 0.0;
            }
            cx_fma(pp.dsum - 2.0 * dsub, xr, acc);
            if (!ghost) cx_store_nt(y + i, acc);
        }
    }
}

extern "C" int lsk_pairs_tile_rows(int cplx) { return cplx ? 512 : 1024; }
extern "C" int lsk_pairs(lsk_pairplan pp, int hamming_weight, int cplx, lsk_tilemap tm, int64_t n, void const *x, void *y, void *stream) {
    if (n == 0 || tm.slots_per_xcd == 0) return 0;
    if (pp.n_near + pp.n_str + pp.n_high > LSK_MAX_PAIRS || hamming_weight + 2 > LSK_PAIR_KC) { snprintf(g_err, sizeof(g_err), "lsk_pairs: plan out of range"); return -1; }
    const int64_t gb = tm.slots_per_xcd * 8; // one block per tile
    if (cplx) hipLaunchKernelGGL((k_pairs_t<true, 512>), dim3((unsigned)gb), dim3(kBlock), 0, (hipStream_t)stream, pp, hamming_weight, tm.entries, tm.slots_per_xcd, n, x, y);
    else hipLaunchKernelGGL((k_pairs_t<false, 1024>), dim3((unsigned)gb), dim3(kBlock), 0, (hipStream_t)stream, pp, hamming_weight, tm.entries, tm.slots_per_xcd, n, x, y);
    LSK_LAUNCH_CHECK();
    return 0;
}
// states[i] = (u32)reps[i]: the 4-byte state array the kernel streams
__global__ __launch_bounds__(kBlock) void k_narrow_states(int64_t n, uint64_t const *__restrict__ reps, uint32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (uint32_t)reps[i];
}
extern "C" int lsk_narrow_states(int64_t n, uint64_t const *reps, uint32_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_narrow_states, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, reps, out);
    LSK_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Staged ("tile") kernel: symmetry projection and/or hash-partitioned output.
// A 256-row tile expands kGC flip-mask groups at a time into an LDS term list (stage A, K2), the
// list is then processed densely, one packet per lane (stage B): K3/K4 projection, K5 owner hash,
// and either K7+K8 (own partition) or a rank inside the (tile, destination) bucket.  Buckets are
// reserved in the send buffer with ONE global atomic per (tile-chunk, destination) and written out
// from LDS (K6: the radix partition by destination happens here, in LDS).
// ---------------------------------------------------------------------------------------------

constexpr uint32_t kDead = 0xffffffffu;

// GC = flip-mask groups expanded per LDS list: 8 for cheap packets (fewer barriers: chain_28, P = 8: 11.0 vs 14.0 ms with 4), 4 for
// symmetry-projected bases (20 instead of 40 KB of LDS per block: twice the blocks per CU to hide K4 and the index look-ups:
// chain_36_symm push 46.1 -> 31.8 ms)
template <typename W, bool PM1, bool CPLX, bool REAL, int GC>
__global__ __launch_bounds__(kBlock) void k_tile(int n_groups, lsk_group const *__restrict__ groups,
                                                 lsk_term const *__restrict__ off, lsk_basis bs,
                                                 lsk_group_elem const *__restrict__ elems, lsk_index ix,
                                                 int count_only, Owner owner, int me, int64_t row0, int64_t row1,
                                                 uint64_t const *__restrict__ reps,
                                                 double const *__restrict__ norms,
                                                 double const *__restrict__ x, double *y,
                                                 unsigned long long *cursors,
                                                 lsk_round_layout const *__restrict__ layout, char *send,
                                                 unsigned long long *counts, int *err) {
    constexpr int kCap = kBlock * GC;
    __shared__ uint64_t s_beta[kCap];
    __shared__ double s_val[kCap * (CPLX ? 2 : 1)];
    __shared__ uint32_t s_meta[kCap];
    __shared__ unsigned s_cnt[LSK_MAX_PARTS];
    __shared__ unsigned long long s_base[LSK_MAX_PARTS];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int P = (int)owner.P;

    for (int64_t t0 = row0 + (int64_t)blockIdx.x * kBlock; t0 < row1; t0 += (int64_t)gridDim.x * kBlock) {
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        uint64_t a = 0;
        double xr = 0.0, xi = 0.0;
        if (valid) {
            a = reps[i];
            if (count_only) xr = 1.0; // the packet set must not depend on x (exact send counts)
            else if (CPLX) { xr = x[2 * i]; xi = x[2 * i + 1]; } else xr = x[i];
            if (!count_only && bs.proj == LSK_PROJ_FULL) { // fold 1 / norm(alpha) into x  (BatchedOperator.chpl:198-202)
                double na = norms[i];
                double s = na > 0.0 ? 1.0 / na : 0.0;
                xr *= s;
                xi *= s;
            }
        }
        for (int g0 = 0; g0 < n_groups; g0 += GC) {
            if (tid == 0) s_n = 0;
            for (int d = tid; d < P; d += kBlock) s_cnt[d] = 0;
            __syncthreads();
            // ---- stage A: expand terms of kGC groups into the LDS list --------------------------
            const int g1 = min(g0 + GC, n_groups);
            for (int g = g0; g < g1; ++g) {
                lsk_group const G = groups[g];
                double cr = 0.0, ci = 0.0;
                if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                const bool act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
                const unsigned long long ball = __ballot(act);
                int base = 0;
                if (lane == 0 && ball) base = atomicAdd(&s_n, __popcll(ball));
                base = __shfl(base, 0);
                if (act) {
                    const int slot = base + __popcll(ball & ((1ULL << lane) - 1));
                    s_beta[slot] = a ^ G.x;
                    if (CPLX) {
                        s_val[2 * slot] = cr * xr - ci * xi;
                        s_val[2 * slot + 1] = cr * xi + ci * xr;
                    } else s_val[slot] = cr * xr;
                }
            }
            __syncthreads();
            const int n = (kAblate && (bs.debug_ablate & 1)) ? 0 : s_n; // LS_AMD_ABLATE (profiling only): 1 no stage B, 8 drop own packets, 16 no packet writes
            // ---- stage B: project, hash, scatter locally or rank into a destination bucket --------
            for (int e = tid; e < n; e += kBlock) {
                uint64_t beta = s_beta[e];
                double vr, vi = 0.0;
                if (CPLX) { vr = s_val[2 * e]; vi = s_val[2 * e + 1]; } else vr = s_val[e];
                bool dead = false;
                if (bs.proj == LSK_PROJ_INVERSION) {
                    uint64_t f = beta ^ bs.site_mask;
                    if (f < beta) { beta = f; vr *= (double)bs.spin_inversion; vi *= (double)bs.spin_inversion; }
                } else if (bs.proj == LSK_PROJ_FULL && bs.k4_mode != 0) {
                    beta = (uint64_t)rep_trivial<W>(bs, elems, (W)beta); // norm(rep) applied at index time
                } else if (bs.proj == LSK_PROJ_FULL) {
                    W rep; double chr, chi, stab;
                    state_info_w<W, PM1>(bs, elems, (W)beta, rep, chr, chi, stab);
                    double n2 = stab * bs.inv_order;
                    if (n2 > 1e-12) {
                        double nb = sqrt(n2);
                        beta = (uint64_t)rep;
                        if (CPLX) {
                            double tr = (vr * chr - vi * chi) * nb, ti = (vr * chi + vi * chr) * nb;
                            vr = tr; vi = ti;
                        } else vr = vr * chr * nb;
                    } else dead = true; // zero-norm orbit: c == 0 => skipped (DMV:110)
                }
                uint32_t meta = kDead;
                if (!dead) {
                    const int dest = owner_of(beta, owner);
                    if (count_only) {
                        atomicAdd(&s_cnt[dest], 1u);
                    } else if (dest == me) {
                        if (kAblate && (bs.debug_ablate & 8)) { s_meta[e] = kDead; continue; }
                        int64_t idx = search_index(ix, beta);
                        if (idx < 0) atomicExch(err, 1);
                        else {
                            if (bs.proj == LSK_PROJ_FULL && bs.k4_mode != 0) { double nb = norms[idx]; vr *= nb; vi *= nb; }
                            if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
                            else atomic_add_f64(y + idx, vr);
                        }
                    } else {
                        unsigned rank = atomicAdd(&s_cnt[dest], 1u);
                        meta = ((uint32_t)dest << 16) | rank;
                        s_beta[e] = beta;
                        if (CPLX) { s_val[2 * e] = vr; s_val[2 * e + 1] = vi; } else s_val[e] = vr;
                    }
                }
                s_meta[e] = meta;
            }
            __syncthreads();
            if (count_only) {
                for (int d = tid; d < P; d += kBlock)
                    if (s_cnt[d]) atomicAdd(&counts[d], (unsigned long long)s_cnt[d]);
            } else if (P > 1) {
                for (int d = tid; d < P; d += kBlock)
                    if (s_cnt[d]) s_base[d] = atomicAdd(&cursors[d], (unsigned long long)s_cnt[d]);
                __syncthreads();
                for (int e = tid; e < n; e += kBlock) {
                    const uint32_t meta = s_meta[e];
                    if (meta == kDead || (kAblate && (bs.debug_ablate & 16))) continue;
                    const int dest = (int)(meta >> 16);
                    const unsigned long long pos = s_base[dest] + (meta & 0xffffu);
                    uint64_t *ob = (uint64_t *)(send + layout->beta_off[dest]);
                    double *ov = (double *)(send + layout->val_off[dest]);
                    ob[pos] = s_beta[e];
                    if (CPLX) { ov[2 * pos] = s_val[2 * e]; ov[2 * pos + 1] = s_val[2 * e + 1]; }
                    else ov[pos] = s_val[e];
                }
            }
            __syncthreads();
        }
    }
}

extern "C" int lsk_tile(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int count_only, int P, int me,
                        int64_t row0, int64_t row1, uint64_t const *reps, double const *norms, void const *x,
                        void *y, unsigned long long *d_cursors, lsk_round_layout const *d_layout, void *d_send,
                        unsigned long long *d_counts, int *d_err, void *stream) {
    if (row1 <= row0 || op.n_groups == 0) return 0;
    if (P > LSK_MAX_PARTS || P < 1) { snprintf(g_err, sizeof(g_err), "lsk_tile: bad partition count %d", P); return -1; }
    if (!count_only && ix.kind != LSK_INDEX_SEARCH) { snprintf(g_err, sizeof(g_err), "lsk_tile needs a SEARCH index"); return -1; }
    Owner ow = make_owner(P);
    dim3 g(1), b(kBlock);
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    hipStream_t s = (hipStream_t)stream;
#define LSK_TILE_ARGS op.n_groups, op.groups, op.off, bs, bs.elems, ix, count_only, ow, me, row0, row1, reps, norms, \
        (double const *)x, (double *)y, d_cursors, d_layout, (char *)d_send, d_counts, d_err
#define LSK_TILE_LAUNCH(W, PM1, GC)                                                                            \
    do {                                                                                                   \
        if (cplx) {                                                                                        \
            if (op.is_real) { g.x = tile_grid(k_tile<W, PM1, true, true, GC>, work_blocks); hipLaunchKernelGGL((k_tile<W, PM1, true, true, GC>), g, b, 0, s, LSK_TILE_ARGS); } \
            else { g.x = tile_grid(k_tile<W, PM1, true, false, GC>, work_blocks); hipLaunchKernelGGL((k_tile<W, PM1, true, false, GC>), g, b, 0, s, LSK_TILE_ARGS); } \
        } else { /* f64 vectors: real operators only (the plan refuses the rest) */                        \
            g.x = tile_grid(k_tile<W, PM1, false, true, GC>, work_blocks); hipLaunchKernelGGL((k_tile<W, PM1, false, true, GC>), g, b, 0, s, LSK_TILE_ARGS); \
        }                                                                                                  \
    } while (0)
    const bool narrow = bs.number_sites <= 32 && bs.proj == LSK_PROJ_FULL;
    if (narrow) { if (bs.chars_pm1) LSK_TILE_LAUNCH(uint32_t, true, 4); else LSK_TILE_LAUNCH(uint32_t, false, 4); }
    else if (bs.proj == LSK_PROJ_FULL) { if (bs.chars_pm1) LSK_TILE_LAUNCH(uint64_t, true, 4); else LSK_TILE_LAUNCH(uint64_t, false, 4); }
    else LSK_TILE_LAUNCH(uint64_t, true, 8); // no projection: PM1 is irrelevant
#undef LSK_TILE_LAUNCH
#undef LSK_TILE_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Packet producer with per-WAVE packet rings and a DETERMINISTIC send layout (k_tile_wv; P <= 64).
//
// k_tile above synchronises its four waves three times per list (stage A | stage B | bucket reservation | write-out), ranks
// the packets of a list with 64-lane LDS atomics on P addresses and reserves the buckets with global atomics on P cursors:
// 75 % of its wave cycles wait (profiles/r2_packets_chain28_P8_sq_counters.txt).  Here a wave owns its 64 rows, a 256-slot
// ring of the LDS list and -- in lane d -- the cursor of destination d inside the round's send segment:
//   * the plan's count pass (COUNT) leaves the number of packets of every (wave, destination) in wtab[wave][P]; the host
//     turns them into exclusive offsets along the waves of a round, so the position of every packet in the send buffer is
//     fixed by the plan: no cursor atomics, no bucket reservation, and the packet order (hence the order in which the
//     consumer's atomics arrive) no longer depends on the block schedule;
//   * stage A appends the packets of three flip-mask groups to the ring, stage B takes chunks of 64 out of it: projection
//     (inversion | orbit minimum | state_info), owner hash, then a loop over the DISTINCT destinations of the chunk:
//     ballot + mbcnt = rank inside the chunk, readlane of the destination's cursor and segment offsets, one store of beta
//     and one of the value straight into the send segment (consecutive chunks continue the same run of every segment).
//   No block barrier anywhere; LDS holds only the rings (16 KB f64, 24 KB c128 per block).
// ---------------------------------------------------------------------------------------------
constexpr int kTwRing = 256;
constexpr int kTwGroups = 3;
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int lane) {
    return (int64_t)readlane_t<uint64_t>((uint64_t)v, lane);
}
// PK12 (pre-indexed packets; unprojected fixed-weight bases): the local index of EVERY packet at its destination -- the own
// partition included -- is read off the all-destinations directory gd (one 16-byte load), remote packets leave as (u32 index,
// value) and the consumer neither ranks nor searches.
template <typename W, bool PM1, bool CPLX, bool REAL, bool COUNT, bool PK12>
__global__ __launch_bounds__(kBlock) void k_tile_wv(int n_groups, lsk_group const *__restrict__ groups,
                                                    lsk_term const *__restrict__ off, lsk_basis bs,
                                                    lsk_group_elem const *__restrict__ elems, lsk_index ix, lsk_gdir gd, Owner owner,
                                                    int me, int64_t row0, int64_t row1, uint64_t const *__restrict__ reps,
                                                    double const *__restrict__ norms, double const *__restrict__ x, double *y,
                                                    uint32_t *__restrict__ wtab, lsk_round_layout const *__restrict__ layout,
                                                    char *send, int *err) {
    constexpr int kCap = (kBlock / 64) * kTwRing;
    __shared__ uint64_t s_beta[kCap];
    __shared__ double s_val[COUNT ? 1 : kCap * (CPLX ? 2 : 1)];
    // PK12: colex rank of beta when it is one binomial away from alpha's (exchange on adjacent sites: rank(alpha) +- C(lo, k)), so that
    // the rank sum of the directory look-up runs once per row, not once per packet; kNoRank: the full sum (see k_tile_st)
    __shared__ uint32_t s_rank[(!COUNT && PK12) ? kCap : 1];
    extern __shared__ uint64_t s_db[]; // rank directory of the own partition: binomials of the closed-form rank (0 bytes without one)
    if (!COUNT && PK12) { gdir_load(gd, ix.binom, s_db); __syncthreads(); }
    else if (!COUNT && ix.dir) { rankdir_load(ix, s_db); __syncthreads(); }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rb = wave * kTwRing;
    const int P = (int)owner.P;
    // lane d keeps what the wave knows about destination d: segment offsets of the round and the running cursor
    int64_t seg_b = 0, seg_v = 0;
    if (!COUNT && lane < P) { seg_b = layout->beta_off[lane]; seg_v = layout->val_off[lane]; }
    for (int64_t t0 = row0 + (int64_t)blockIdx.x * kBlock; t0 < row1; t0 += (int64_t)gridDim.x * kBlock) {
        if (t0 + (wave << 6) >= row1) continue; // wave-uniform: nothing below synchronises the block
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        uint64_t a = 0;
        double xr = 0.0, xi = 0.0;
        if (valid) {
            a = reps[i];
            if (COUNT) xr = 1.0; // the packet set must not depend on x (exact send counts)
            else {
                if (CPLX) { xr = x[2 * i]; xi = x[2 * i + 1]; } else xr = x[i];
                if (bs.proj == LSK_PROJ_FULL) { // fold 1 / norm(alpha) into x  (BatchedOperator.chpl:198-202)
                    const double na = norms[i];
                    const double s = na > 0.0 ? 1.0 / na : 0.0;
                    xr *= s;
                    xi *= s;
                }
            }
        }
        const bool narrow_ranks = !COUNT && PK12 && bs.proj == LSK_PROJ_NONE && gd.n_ranks <= 0xffffffffLL;
        uint64_t ga = 0; // colex rank of alpha
        if (narrow_ranks && valid) {
            const int kc = gd.weight + 1;
            uint64_t t = a;
            int k = 1;
            while (t && k < kc) { ga += s_db[(__ffsll((unsigned long long)t) - 1) * kc + k]; ++k; t &= t - 1; }
        }
        const int64_t wg = ((t0 - row0) >> 6) + wave; // this wave's 64 rows inside the round
        uint32_t cur = 0;                             // lane d: packets so far (COUNT) | next position in segment d
        if (!COUNT && lane < P) cur = wtab[wg * P + lane];
        int head = 0, cnt = 0; // wave-uniform: the ring holds [head, head + cnt) mod kTwRing
        auto chunk = [&](int m) {
            bool live = lane < m;
            const int e = rb + ((head + lane) & (kTwRing - 1));
            uint64_t beta = live ? s_beta[e] : 0;
            double vr = 0.0, vi = 0.0;
            if (!COUNT) { if (CPLX) { vr = s_val[2 * e]; vi = s_val[2 * e + 1]; } else vr = s_val[e]; }
            if (bs.proj == LSK_PROJ_INVERSION) {
                const uint64_t f = beta ^ bs.site_mask;
                if (f < beta) { beta = f; vr *= (double)bs.spin_inversion; vi *= (double)bs.spin_inversion; }
            } else if (bs.proj == LSK_PROJ_FULL && bs.k4_mode != 0) {
                beta = (uint64_t)rep_trivial<W>(bs, elems, (W)beta); // norm(rep) applied at index time
            } else if (bs.proj == LSK_PROJ_FULL) {
                if (live) {
                    W rep; double chr, chi, stab;
                    state_info_w<W, PM1>(bs, elems, (W)beta, rep, chr, chi, stab);
                    const double n2 = stab * bs.inv_order;
                    if (n2 > 1e-12) {
                        const double nb = sqrt(n2);
                        beta = (uint64_t)rep;
                        if (CPLX) { const double tr = (vr * chr - vi * chi) * nb, ti = (vr * chi + vi * chr) * nb; vr = tr; vi = ti; }
                        else vr = vr * chr * nb;
                    } else live = false; // zero-norm orbit: c == 0 => skipped (DMV:110)
                }
            }
            const int dest = live ? owner_of(beta, owner) : -1;
            bool remote = live;
            uint32_t pidx = 0; // PK12: the packet's index inside its destination's block
            if (!COUNT && PK12) {
                if (live) {
                    const uint32_t rk = s_rank[e];
                    const int64_t idx = rk != kNoRank ? gdir_index_of_rank(gd, (uint64_t)rk, dest) : gdir_index(gd, beta, dest, s_db);
                    if (idx < 0) { // not a basis state (DMV:115-118): the flag halts the matvec; the slot the count pass reserved for the
                        atomicExch(err, 1); // packet is still filled -- with (index 0, value 0) -- so that no consumer meets a stale key
                        if (dest == me) remote = false;
                        else { vr = 0.0; vi = 0.0; }
                    } else if (dest == me) {
                        remote = false;
                        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
                        else atomic_add_f64(y + idx, vr);
                    } else pidx = (uint32_t)idx;
                }
            } else if (!COUNT) {
                if (live && dest == me) {
                    remote = false;
                    const int64_t idx = ix.dir ? rankdir_index(ix, beta, s_db) : search_index(ix, beta);
                    if (idx < 0) atomicExch(err, 1);
                    else {
                        if (bs.proj == LSK_PROJ_FULL && bs.k4_mode != 0) { const double nb = norms[idx]; vr *= nb; vi *= nb; }
                        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
                        else atomic_add_f64(y + idx, vr);
                    }
                }
            }
            // one pass per distinct destination of the chunk (<= min(P, 64))
            unsigned long long rem = __ballot(remote);
            while (rem) {
                const int l = __builtin_ctzll(rem);
                const int d = __builtin_amdgcn_readlane(dest, l);
                const bool mine = remote && dest == d;
                const unsigned long long mm = __ballot(mine);
                if (!COUNT) {
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)cur, d);
                    const int64_t ob = readlane_i64(seg_b, d), ov = readlane_i64(seg_v, d);
                    if (mine) {
                        const size_t pos = (size_t)base + (size_t)__popcll(mm & ((1ULL << lane) - 1));
                        if (PK12) reinterpret_cast<uint32_t *>(send + ob)[pos] = pidx;
                        else reinterpret_cast<uint64_t *>(send + ob)[pos] = beta;
                        double *pv = reinterpret_cast<double *>(send + ov);
                        if (CPLX) { pv[2 * pos] = vr; pv[2 * pos + 1] = vi; } else pv[pos] = vr;
                    }
                }
                if (lane == d) cur += (uint32_t)__popcll(mm);
                rem &= ~mm;
            }
        };
        for (int g0 = 0; g0 < n_groups; g0 += kTwGroups) {
            const int g1 = min(g0 + kTwGroups, n_groups);
            for (int g = g0; g < g1; ++g) { // stage A: append
                lsk_group const G = groups[g];
                double cr = 0.0, ci = 0.0;
                if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                const bool act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
                const unsigned long long ball = __ballot(act);
                if (act) {
                    const int slot = rb + ((head + cnt + __popcll(ball & ((1ULL << lane) - 1))) & (kTwRing - 1));
                    s_beta[slot] = a ^ G.x;
                    if (!COUNT) {
                        if (CPLX) { s_val[2 * slot] = cr * xr - ci * xi; s_val[2 * slot + 1] = cr * xi + ci * xr; }
                        else s_val[slot] = cr * xr;
                        if (PK12) {
                            uint32_t rk = kNoRank;
                            if (narrow_ranks && G.fast == LSK_GROUP_EXCHANGE && G.adj >= 0) {
                                const uint64_t c = s_db[G.adj * (gd.weight + 1) + __popcll(a & ((1ULL << G.adj) - 1))];
                                rk = (uint32_t)(((a >> G.adj) & 1ULL) ? ga + c : ga - c); // the lower site's bit moves up | the upper one's down
                            }
                            s_rank[slot] = rk;
                        }
                    }
                }
                cnt += __popcll(ball);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            while (cnt >= 64) {
                chunk(64);
                head = (head + 64) & (kTwRing - 1);
                cnt -= 64;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (cnt > 0) chunk(cnt);
        if (COUNT && lane < P) wtab[wg * P + lane] = cur;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int lsk_tile_wv_max_parts(void) { return 64; }
// rows [row0, row1) of partition `me` (row0 = first row of the round: wave w of the round owns rows row0 + 64 w ...).
// count_only: wtab[w][P] <- packets of wave w per destination (the own partition included); otherwise wtab holds the
// exclusive offsets of every (wave, destination) inside the round's segments and the packets are written to d_send.
extern "C" int lsk_tile_wv(lsk_operator op, lsk_basis bs, lsk_index ix, lsk_gdir gd, int cplx, int count_only, int P, int me,
                           int64_t row0, int64_t row1, uint64_t const *reps, double const *norms, void const *x, void *y,
                           uint32_t *d_wtab, lsk_round_layout const *d_layout, void *d_send, int *d_err, void *stream) {
    if (row1 <= row0 || op.n_groups == 0) return 0;
    if (P > lsk_tile_wv_max_parts() || P < 1 || !d_wtab) { snprintf(g_err, sizeof(g_err), "lsk_tile_wv: bad partition count %d or no wave table", P); return -1; }
    const bool pk12 = !count_only && gd.entries != nullptr;
    if (!count_only && !pk12 && ix.kind != LSK_INDEX_SEARCH) { snprintf(g_err, sizeof(g_err), "lsk_tile_wv needs a SEARCH index"); return -1; }
    if (pk12 && (bs.proj == LSK_PROJ_FULL || gd.P != P || !ix.binom)) { snprintf(g_err, sizeof(g_err), "lsk_tile_wv: pre-indexed packets need an unprojected basis and a directory over %d partitions", P); return -1; }
    Owner ow = make_owner(P);
    dim3 g(1), b(kBlock);
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    hipStream_t s = (hipStream_t)stream;
    const size_t dyn_db = pk12 ? sizeof(uint64_t) * (size_t)gd.sites * (size_t)(gd.weight + 1)
                               : (ix.dir ? sizeof(uint64_t) * (size_t)ix.dir_sites * (size_t)(ix.dir_weight + 1) : 0);
#define LSK_TW_ARGS op.n_groups, op.groups, op.off, bs, bs.elems, ix, gd, ow, me, row0, row1, reps, norms, (double const *)x, (double *)y, \
        d_wtab, d_layout, (char *)d_send, d_err
#define LSK_TW_ONE(W, PM1, CPLX, REAL)                                                                                           \
    do {                                                                                                                         \
        if (count_only) { g.x = tile_grid(k_tile_wv<W, PM1, CPLX, REAL, true, false>, work_blocks); hipLaunchKernelGGL((k_tile_wv<W, PM1, CPLX, REAL, true, false>), g, b, 0, s, LSK_TW_ARGS); } \
        else { g.x = tile_grid(k_tile_wv<W, PM1, CPLX, REAL, false, false>, work_blocks); hipLaunchKernelGGL((k_tile_wv<W, PM1, CPLX, REAL, false, false>), g, b, dyn_db, s, LSK_TW_ARGS); } \
    } while (0)
#define LSK_TW_LAUNCH(W, PM1)                                                                                  \
    do {                                                                                                       \
        if (cplx) { if (op.is_real) LSK_TW_ONE(W, PM1, true, true); else LSK_TW_ONE(W, PM1, true, false); }    \
        else LSK_TW_ONE(W, PM1, false, true); /* f64 vectors: real operators only (the plan refuses the rest) */ \
    } while (0)
    if (pk12) { // unprojected bases only: W / PM1 are irrelevant
        if (cplx) {
            if (op.is_real) { g.x = tile_grid(k_tile_wv<uint64_t, true, true, true, false, true>, work_blocks); hipLaunchKernelGGL((k_tile_wv<uint64_t, true, true, true, false, true>), g, b, dyn_db, s, LSK_TW_ARGS); }
            else { g.x = tile_grid(k_tile_wv<uint64_t, true, true, false, false, true>, work_blocks); hipLaunchKernelGGL((k_tile_wv<uint64_t, true, true, false, false, true>), g, b, dyn_db, s, LSK_TW_ARGS); }
        } else { g.x = tile_grid(k_tile_wv<uint64_t, true, false, true, false, true>, work_blocks); hipLaunchKernelGGL((k_tile_wv<uint64_t, true, false, true, false, true>), g, b, dyn_db, s, LSK_TW_ARGS); }
        LSK_LAUNCH_CHECK();
        return 0;
    }
    const bool narrow = bs.number_sites <= 32 && bs.proj == LSK_PROJ_FULL;
    if (narrow) { if (bs.chars_pm1) LSK_TW_LAUNCH(uint32_t, true); else LSK_TW_LAUNCH(uint32_t, false); }
    else if (bs.proj == LSK_PROJ_FULL) { if (bs.chars_pm1) LSK_TW_LAUNCH(uint64_t, true); else LSK_TW_LAUNCH(uint64_t, false); }
    else LSK_TW_LAUNCH(uint64_t, true); // no projection: PM1 is irrelevant
#undef LSK_TW_LAUNCH
#undef LSK_TW_ONE
#undef LSK_TW_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Staged PULL kernel for symmetry-projected bases (Hermitian operators):
//   y[r] = d(r) x[r] + sum_j conj(H~[r'_j, r]) x[r'_j],   H~[r', r] = c conj(chi0) n(r') / n(r)
// Same stage A as k_tile (LDS term list per 256-row tile), stage B projects every packet, looks the
// representative up in the GLOBAL basis, gathers x there and accumulates into a per-tile LDS copy of
// y (ds_add_f64) -- no global atomics, y written once.  With one partition "global" == "local"; with
// one partition per GPU x is the all-gathered vector in global ascending order (replicated-x mode).
// ---------------------------------------------------------------------------------------------
constexpr int kGCPull = 4;
constexpr int kCapPull = kBlock * kGCPull;
// Near window of the pull kernel.  In the sorted array of representatives the partners of a row cluster around the row
// itself: on the symmetric chains 48 % of all projected states |rep(beta)> lie within 512 entries of the tile that
// generated them (measured with the oracle, tests/test_partner_locality.py).  The tile therefore
// stages the representatives [tile - 512, tile + 256 + 512) in LDS as 32-bit offsets from the first of them and
// resolves those partners with a binary search there; their values come from the index-ordered (prescaled) x, whose
// lines are shared by the whole neighbourhood in L1/L2.  Only the others pay the random 16-byte request into the hash
// table, which is what bounds this kernel (one fabric request per probe).
constexpr int kPullHalo = 512;
constexpr int kPullWin = kBlock + 2 * kPullHalo;
constexpr uint32_t kWinAbsent = 0xffffffffu; // offsets >= 2^32 - 1 are treated as "not in the window" (hash path)
// position of offset d in the ascending window w[0, n), or -1.  n <= 2047.
__host__ __device__ __forceinline__ int window_find(uint32_t const *w, int n, uint32_t d) {
    int pos = 0; // lower bound: first entry >= d
#pragma unroll
    for (int step = 1024; step >= 1; step >>= 1)
        if (pos + step <= n && w[pos + step - 1] < d) pos += step;
    return (pos < n && w[pos] == d) ? pos : -1;
}
__host__ __device__ __forceinline__ uint32_t window_offset(uint64_t rep, uint64_t v0) {
    const uint64_t d = rep - v0; // rep >= v0 inside the window (ascending)
    return d >= (uint64_t)kWinAbsent ? kWinAbsent : (uint32_t)d;
}

extern "C" int lsk_test_window_find(uint64_t const *reps, int n, uint64_t key) {
    if (n < 1 || n > kPullWin) return -2;
    uint32_t w[kPullWin];
    for (int i = 0; i < n; ++i) w[i] = window_offset(reps[i], reps[0]);
    if (key < reps[0]) return -1;
    const uint32_t d = window_offset(key, reps[0]);
    return d == kWinAbsent ? -1 : window_find(w, n, d);
}

template <typename W, bool PM1, bool CPLX, bool REAL>
__global__ __launch_bounds__(kBlock) void k_tile_pull(lsk_runs runs, int n_groups, lsk_group const *__restrict__ groups,
                                                      lsk_term const *__restrict__ off, int n_diag,
                                                      lsk_term const *__restrict__ diag, lsk_basis bs,
                                                      lsk_group_elem const *__restrict__ elems, lsk_index ixg,
                                                      int64_t row0, int64_t row1,
                                                      uint64_t const *__restrict__ reps,
                                                      double const *__restrict__ norms_local,
                                                      double const *__restrict__ norms_global,
                                                      int64_t const *__restrict__ row_gidx,
                                                      uint64_t const *__restrict__ tab, int tab_bits,
                                                      uint64_t const *__restrict__ greps, int64_t n_global,
                                                      double const *__restrict__ xs, int halo,
                                                      double const *__restrict__ x, double *__restrict__ y, int *err) {
    constexpr int ES = CPLX ? 4 : 2; // u64 words per hash entry
    __shared__ uint32_t s_win[kPullWin];
    __shared__ uint64_t s_beta[kCapPull];
    constexpr bool RC = REAL && PM1; // conj(H~) stays real: real coefficients and +-1 characters
    __shared__ double s_coef[kCapPull * (RC ? 1 : 2)];
    __shared__ uint16_t s_row[kCapPull];
    __shared__ double s_acc[kBlock * (CPLX ? 2 : 1)];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    for (int64_t t0 = row0 + (int64_t)blockIdx.x * kBlock; t0 < row1; t0 += (int64_t)gridDim.x * kBlock) {
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        uint64_t a = 0;
        double inv_na = 0.0;
        if (valid) {
            a = reps[i];
            double na = norms_local[i];
            inv_na = na > 0.0 ? 1.0 / na : 0.0;
        }
        if (CPLX) { s_acc[2 * tid] = 0.0; s_acc[2 * tid + 1] = 0.0; } else s_acc[tid] = 0.0;
        // near window: global indices [gbase, gbase + wn) around the tile (the global index of its first row)
        int64_t gbase = 0;
        int wn = 0;
        uint64_t v0 = 0;
        if (halo > 0) {
            const int64_t ig0 = row_gidx ? row_gidx[t0] : t0;
            gbase = ig0 > halo ? ig0 - halo : 0;
            const int64_t left = n_global - gbase;
            wn = (int)(left < (int64_t)(kBlock + 2 * halo) ? left : (int64_t)(kBlock + 2 * halo));
            v0 = greps[gbase];
            for (int w = tid; w < wn; w += kBlock) s_win[w] = window_offset(greps[gbase + w], v0);
        }
        for (int g0 = 0; g0 < n_groups; g0 += kGCPull) {
            if (tid == 0) s_n = 0;
            __syncthreads();
            const int g1 = min(g0 + kGCPull, n_groups);
            for (int g = g0; g < g1; ++g) {
                lsk_group const G = groups[g];
                double cr = 0.0, ci = 0.0;
                if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                const bool act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
                const unsigned long long ball = __ballot(act);
                int base = 0;
                if (lane == 0 && ball) base = atomicAdd(&s_n, __popcll(ball));
                base = __shfl(base, 0);
                if (act) {
                    const int slot = base + __popcll(ball & ((1ULL << lane) - 1));
                    s_beta[slot] = a ^ G.x;
                    s_row[slot] = (uint16_t)tid;
                    // conj(c) / n(alpha)
                    if (RC) s_coef[slot] = cr * inv_na;
                    else { s_coef[2 * slot] = cr * inv_na; s_coef[2 * slot + 1] = -ci * inv_na; }
                }
            }
            __syncthreads();
            const int n = (kAblate && (bs.debug_ablate & 1)) ? 0 : s_n;
            // ---- stage B1: K4 on every packet; representative and conj(H~) go back into the list ---------
            for (int e = tid; e < n; e += kBlock) {
                uint64_t beta = s_beta[e];
                double hr, hi = 0.0; // conj(H~) so far
                if (RC) hr = s_coef[e]; else { hr = s_coef[2 * e]; hi = s_coef[2 * e + 1]; }
                if (kAblate && (bs.debug_ablate & 4)) {
                    beta = a; // a key that exists (this thread's own row)
                } else if (bs.k4_mode != 0) {
                    beta = (uint64_t)rep_trivial<W>(bs, elems, (W)beta); // x is pre-multiplied by norm(rep)
                } else {
                    W rep; double chr, chi, stab;
                    state_info_w<W, PM1>(bs, elems, (W)beta, rep, chr, chi, stab);
                    double n2 = stab * bs.inv_order;
                    if (!(n2 > 1e-12)) { s_row[e] = 0xffff; continue; } // zero-norm orbit: contributes nothing (DMV:110)
                    const double nb = sqrt(n2);
                    beta = (uint64_t)rep;
                    // times chi0 = conj(conj(chi0)) = (chr, -chi), times norm(rep)
                    double tr = (hr * chr + hi * chi) * nb, ti = (hi * chr - hr * chi) * nb;
                    hr = tr; hi = ti;
                    if (RC) s_coef[e] = hr; else { s_coef[2 * e] = hr; s_coef[2 * e + 1] = hi; }
                }
                s_beta[e] = beta;
            }
            // ---- stage B2: gathers.  A thread's packets are independent: issue all home-slot loads first
            // (kGCPull requests in flight per lane), then resolve and accumulate -----------------------------
            if (!(kAblate && (bs.debug_ablate & 2))) {
                const uint64_t hmask = (1ULL << tab_bits) - 1;
                uint64_t key[kGCPull], slot[kGCPull];
                ulonglong2 first[kGCPull];
                double im0[kGCPull];
                bool live[kGCPull];
                int pos[kGCPull];
#pragma unroll
                for (int k = 0; k < kGCPull; ++k) { // the (independent) window searches first: LDS only
                    const int e = tid + k * kBlock;
                    live[k] = e < n && s_row[e] != 0xffff;
                    key[k] = 0; slot[k] = 0; im0[k] = 0.0; pos[k] = -1;
                    first[k] = make_ulonglong2(0, 0);
                    if (live[k]) {
                        key[k] = s_beta[e];
                        if (wn > 0 && key[k] >= v0) {
                            const uint32_t d = window_offset(key[k], v0);
                            if (d != kWinAbsent) pos[k] = window_find(s_win, wn, d);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < kGCPull; ++k) { // then every global load of this thread's packets
                    if (!live[k]) continue;
                    if (pos[k] >= 0) { // near partner: value from the index-ordered vector; looks like a home-slot hit below
                        const int64_t j = gbase + pos[k];
                        first[k].x = key[k];
                        if (CPLX) { first[k].y = (unsigned long long)__double_as_longlong(xs[2 * j]); im0[k] = xs[2 * j + 1]; }
                        else first[k].y = (unsigned long long)__double_as_longlong(xs[j]);
                    } else {
                        slot[k] = hash_slot(key[k], tab_bits);
                        first[k] = *(ulonglong2 const *)(tab + slot[k] * ES);
                        if (CPLX) im0[k] = __longlong_as_double((long long)tab[slot[k] * ES + 2]);
                    }
                }
#pragma unroll
                for (int k = 0; k < kGCPull; ++k) {
                    if (!live[k]) continue;
                    const int e = tid + k * kBlock;
                    double xr, xi = 0.0;
                    if (first[k].x == key[k]) {
                        xr = __longlong_as_double((long long)first[k].y);
                        if (CPLX) xi = im0[k];
                    } else {
                        // collision (load factor <= 0.5: ~1 in 4): continue the probe sequence
                        bool found = false;
                        uint64_t sl = slot[k];
                        uint64_t cur = first[k].x;
                        while (cur != kHashEmpty) {
                            sl = (sl + 1) & hmask;
                            const ulonglong2 en = *(ulonglong2 const *)(tab + sl * ES);
                            cur = en.x;
                            if (cur == key[k]) {
                                xr = __longlong_as_double((long long)en.y);
                                if (CPLX) xi = __longlong_as_double((long long)tab[sl * ES + 2]);
                                found = true;
                                break;
                            }
                        }
                        if (!found) { atomicExch(err, 1); continue; }
                    }
                    double hr, hi = 0.0;
                    if (RC) hr = s_coef[e]; else { hr = s_coef[2 * e]; hi = s_coef[2 * e + 1]; }
                    const int r = s_row[e];
                    if (CPLX) {
                        atomicAdd(&s_acc[2 * r], hr * xr - hi * xi);
                        atomicAdd(&s_acc[2 * r + 1], hr * xi + hi * xr);
                    } else {
                        atomicAdd(&s_acc[r], hr * xr);
                    }
                }
            }
            __syncthreads();
        }
        if (valid) {
            const int64_t ig = row_gidx ? row_gidx[i] : i;
            double dr = 0.0, di = 0.0;
            if (n_diag > 0) diag_coeff<uint64_t, REAL>(runs, n_diag, diag, a, dr, di);
            if (CPLX) {
                const double xr = x[2 * ig], xi = x[2 * ig + 1];
                double yr = dr * xr - di * xi + s_acc[2 * tid], yi = dr * xi + di * xr + s_acc[2 * tid + 1];
                if (n_diag == 0) { yr += y[2 * i]; yi += y[2 * i + 1]; } // accumulate, DMV:1062-1063
                y[2 * i] = yr; y[2 * i + 1] = yi;
            } else {
                double yr = dr * x[ig] + s_acc[tid];
                if (n_diag == 0) yr += y[i];
                y[i] = yr;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void k_hash_insert(int64_t n, uint64_t const *__restrict__ reps, int bits,
                                                        int es, uint64_t *tab, uint32_t *__restrict__ slot_of) {
    const uint64_t mask = (1ULL << bits) - 1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t key = reps[i];
        uint64_t slot = hash_slot(key, bits);
        for (;;) {
            unsigned long long old = atomicCAS((unsigned long long *)(tab + slot * es), (unsigned long long)kHashEmpty,
                                               (unsigned long long)key);
            if (old == kHashEmpty || old == key) break;
            slot = (slot + 1) & mask;
        }
        slot_of[i] = (uint32_t)slot;
    }
}
__global__ __launch_bounds__(kBlock) void k_hash_clear(int64_t entries, int es, uint64_t *__restrict__ tab) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < entries; i += (int64_t)gridDim.x * kBlock)
        tab[i * es] = kHashEmpty;
}
// values: tab[slot_of[i]] <- x[i] * norms[i]   (norms == NULL: unscaled)
// xs (may be NULL): the same scaled values in index order, for the near window of k_tile_pull
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_hash_fill(int64_t n, uint32_t const *__restrict__ slot_of,
                                                      double const *__restrict__ x, double const *__restrict__ norms,
                                                      uint64_t *__restrict__ tab, double *__restrict__ xs) {
    constexpr int ES = CPLX ? 4 : 2;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double nb = norms ? norms[i] : 1.0;
        double *val = (double *)(tab + (size_t)slot_of[i] * ES + 1);
        if (CPLX) {
            const double vr = x[2 * i] * nb, vi = x[2 * i + 1] * nb;
            val[0] = vr; val[1] = vi;
            if (xs) { xs[2 * i] = vr; xs[2 * i + 1] = vi; }
        } else {
            const double v = x[i] * nb;
            val[0] = v;
            if (xs) xs[i] = v;
        }
    }
}
extern "C" int lsk_hash_build(int cplx, int64_t n, uint64_t const *reps, int bits, void *tab, uint32_t *slot_of,
                              void *stream) {
    const int es = cplx ? 4 : 2;
    const int64_t entries = (int64_t)1 << bits;
    hipLaunchKernelGGL(k_hash_clear, dim3(grid_for(entries)), dim3(kBlock), 0, (hipStream_t)stream, entries, es, (uint64_t *)tab);
    LSK_LAUNCH_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(k_hash_insert, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, reps, bits, es, (uint64_t *)tab, slot_of);
        LSK_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int lsk_hash_fill(int cplx, int64_t n, uint32_t const *slot_of, void const *x, double const *norms, void *tab,
                             void *xs, void *stream) {
    if (n == 0) return 0;
    if (cplx) hipLaunchKernelGGL(k_hash_fill<true>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, slot_of, (double const *)x, norms, (uint64_t *)tab, (double *)xs);
    else hipLaunchKernelGGL(k_hash_fill<false>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, slot_of, (double const *)x, norms, (uint64_t *)tab, (double *)xs);
    LSK_LAUNCH_CHECK();
    return 0;
}

extern "C" int lsk_tile_pull(lsk_operator op, lsk_basis bs, lsk_index ix_global, int cplx, int64_t row0, int64_t row1,
                             uint64_t const *reps, double const *norms_local, double const *norms_global,
                             int64_t const *row_gidx, void const *tab, int tab_bits, uint64_t const *reps_global,
                             int64_t n_global, void const *xs_global, int halo, void const *x_global, void *y,
                             int *d_err, void *stream) {
    if (row1 <= row0) return 0;
    if (bs.proj != LSK_PROJ_FULL) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull is for projected bases"); return -1; }
    if (halo < 0 || halo > kPullHalo || (halo > 0 && (!reps_global || !xs_global || n_global <= 0))) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull: bad near window"); return -1; }
    dim3 g(1), b(kBlock);
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    hipStream_t s = (hipStream_t)stream;
#define LSK_TP_ARGS op.runs, op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs, bs.elems, ix_global, row0, row1, reps, \
        norms_local, norms_global, row_gidx, (uint64_t const *)tab, tab_bits, reps_global, n_global, (double const *)xs_global, halo, \
        (double const *)x_global, (double *)y, d_err
#define LSK_TP_LAUNCH(W, PM1)                                                                                   \
    do {                                                                                                        \
        if (cplx) {                                                                                             \
            if (op.is_real) { g.x = tile_grid(k_tile_pull<W, PM1, true, true>, work_blocks); hipLaunchKernelGGL((k_tile_pull<W, PM1, true, true>), g, b, 0, s, LSK_TP_ARGS); } \
            else { g.x = tile_grid(k_tile_pull<W, PM1, true, false>, work_blocks); hipLaunchKernelGGL((k_tile_pull<W, PM1, true, false>), g, b, 0, s, LSK_TP_ARGS); } \
        } else {                                                                                                \
            g.x = tile_grid(k_tile_pull<W, PM1, false, true>, work_blocks); hipLaunchKernelGGL((k_tile_pull<W, PM1, false, true>), g, b, 0, s, LSK_TP_ARGS); /* f64: real operators only */ \
        }                                                                                                       \
    } while (0)
    if (bs.number_sites <= 32) { if (bs.chars_pm1) LSK_TP_LAUNCH(uint32_t, true); else LSK_TP_LAUNCH(uint32_t, false); }
    else { if (bs.chars_pm1) LSK_TP_LAUNCH(uint64_t, true); else LSK_TP_LAUNCH(uint64_t, false); }
#undef LSK_TP_LAUNCH
#undef LSK_TP_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Static index table {representative -> 32-bit payload} (lsk_gtab, lsk.h) and the INDEXED mode of the staged pull
// kernel.  The value table above costs one request per far partner but must be rewritten every matvec -- a pass of N
// random 16-byte writes on EVERY rank (chain_40_symm: 43 of 373 ms on one GPU, and undivided by P in the replicated-x
// exchange).  The index table is built once: a far partner costs two dependent requests (bucket, then x[slot]), nothing is
// refreshed, and x is read wherever it already lies -- index order on one device, or the blocks of the replicated-x
// exchange as they arrive from their owners (slot = owner * max_count + local index), which removes the hashed -> block
// permutation pass as well.  Per-rank work then shrinks with P.
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kGtEmpty = ~0ULL;
constexpr int kGtMaxDist = 255;
// an L-bit bijection (odd multiplications mod 2^L and xor-shifts): bucket and tag together identify the key
__host__ __device__ __forceinline__ uint64_t gt_mix(uint64_t k, int L) {
    const uint64_t m = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    const int s = (L + 1) >> 1;
    k = (k * 0x9E3779B97F4A7C15ULL) & m;
    k ^= k >> s;
    k = (k * 0xD6E8FEB86659FD93ULL) & m;
    k ^= k >> s;
    return k;
}
__host__ __device__ __forceinline__ void gt_split(lsk_gtab const &t, uint64_t key, uint64_t &bucket, uint32_t &tag) {
    const uint64_t h = gt_mix(key, t.L);
    bucket = h >> t.tbits;
    tag = (uint32_t)(h & ((1ULL << t.tbits) - 1));
}
// upper word of an entry: tag << 8 | displacement
__host__ __device__ __forceinline__ uint32_t gt_hi(uint32_t tag, int dist) { return (tag << 8) | (uint32_t)dist; }

__global__ __launch_bounds__(kBlock) void k_gtab_clear(int64_t entries, uint64_t *__restrict__ tab) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < entries; i += (int64_t)gridDim.x * kBlock) tab[i] = kGtEmpty;
}
__global__ __launch_bounds__(kBlock) void k_gtab_insert(lsk_gtab t, uint64_t *tab, int64_t n, uint64_t const *__restrict__ reps,
                                                        uint32_t const *__restrict__ payload, int *flag) {
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t b;
        uint32_t tag;
        gt_split(t, reps[i], b, tag);
        const uint32_t pay = payload ? payload[i] : (uint32_t)i;
        bool placed = false;
        for (int d = 0; d <= kGtMaxDist && !placed; ++d) {
            const unsigned long long e = ((unsigned long long)gt_hi(tag, d) << 32) | pay;
            for (int sl = 0; sl < 2 && !placed; ++sl)
                placed = atomicCAS((unsigned long long *)(tab + 2 * b + sl), (unsigned long long)kGtEmpty, e) == kGtEmpty;
            b = (b + 1) & bmask;
        }
        if (!placed) atomicExch(flag, 1);
    }
}
// payload of `key`, or 0xffffffff; `first` is the home bucket when the caller has already loaded it
__device__ __forceinline__ uint32_t gt_resolve(lsk_gtab const &t, uint64_t const *__restrict__ tab, uint64_t b, uint32_t tag,
                                               ulonglong2 cur) {
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int d = 0;; ++d) {
        const uint32_t want = gt_hi(tag, d);
        if ((uint32_t)(cur.x >> 32) == want && cur.x != kGtEmpty) return (uint32_t)cur.x;
        if ((uint32_t)(cur.y >> 32) == want && cur.y != kGtEmpty) return (uint32_t)cur.y;
        if (cur.x == kGtEmpty || cur.y == kGtEmpty || d == kGtMaxDist) return 0xffffffffu; // inserts never skip an empty slot
        b = (b + 1) & bmask;
        cur = *(ulonglong2 const *)(tab + 2 * b);
    }
}
__global__ __launch_bounds__(kBlock) void k_gtab_lookup(lsk_gtab t, int64_t n, uint64_t const *__restrict__ keys,
                                                        uint32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t b;
        uint32_t tag;
        gt_split(t, keys[i], b, tag);
        out[i] = gt_resolve(t, t.entries, b, tag, *(ulonglong2 const *)(t.entries + 2 * b));
    }
}
extern "C" int lsk_gtab_bits(int L, int64_t n, int64_t max_bytes) {
    if (L < 1 || L > 64 || n < 0) return -1;
    int bb = 2;
    while (((int64_t)2 << bb) < 2 * n) ++bb; // two entries per bucket, load factor <= 0.5
    if (bb < L - 24) bb = L - 24;            // tag (L - bbits bits) + displacement (8) + payload (32) must fit 64 bits
    if (bb > L) bb = L;
    if (bb > 40 || ((int64_t)16 << bb) > max_bytes) return -1;
    return bb;
}
extern "C" int lsk_gtab_build(lsk_gtab t, uint64_t *entries, int64_t n, uint64_t const *reps, uint32_t const *payload,
                              int *d_flag, void *stream) {
    if (t.tbits != t.L - t.bbits || t.tbits < 0 || t.tbits > 24 || n >= 0xffffffffLL) { snprintf(g_err, sizeof(g_err), "lsk_gtab_build: bad table shape"); return -1; }
    const int64_t ne = (int64_t)2 << t.bbits;
    hipLaunchKernelGGL(k_gtab_clear, dim3(grid_for(ne)), dim3(kBlock), 0, (hipStream_t)stream, ne, entries);
    LSK_LAUNCH_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(k_gtab_insert, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, t, entries, n, reps, payload, d_flag);
        LSK_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int lsk_gtab_lookup(lsk_gtab t, int64_t n, uint64_t const *keys, uint32_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gtab_lookup, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, t, n, keys, out);
    LSK_LAUNCH_CHECK();
    return 0;
}
// host-side sequential build with the placement rule of k_gtab_insert (tests: no device needed); -1 when a key does not fit
extern "C" int lsk_test_gtab_build_host(lsk_gtab t, uint64_t *h, int64_t n, uint64_t const *reps, uint32_t const *payload) {
    const int64_t ne = (int64_t)2 << t.bbits;
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int64_t i = 0; i < ne; ++i) h[i] = kGtEmpty;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t b;
        uint32_t tag;
        gt_split(t, reps[i], b, tag);
        bool placed = false;
        for (int d = 0; d <= kGtMaxDist && !placed; ++d) {
            for (int sl = 0; sl < 2 && !placed; ++sl)
                if (h[2 * b + sl] == kGtEmpty) { h[2 * b + sl] = ((uint64_t)gt_hi(tag, d) << 32) | (payload ? payload[i] : (uint32_t)i); placed = true; }
            b = (b + 1) & bmask;
        }
        if (!placed) return -1;
    }
    return 0;
}
extern "C" int64_t lsk_test_gtab_find(lsk_gtab t, uint64_t const *h, uint64_t key) {
    uint64_t b;
    uint32_t tag;
    gt_split(t, key, b, tag);
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int d = 0; d <= kGtMaxDist; ++d) {
        const uint32_t want = gt_hi(tag, d);
        for (int sl = 0; sl < 2; ++sl) {
            const uint64_t e = h[2 * b + sl];
            if (e == kGtEmpty) return -1;
            if ((uint32_t)(e >> 32) == want) return (int64_t)(uint32_t)e;
        }
        b = (b + 1) & bmask;
    }
    return -1;
}

template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scale(int64_t n, double const *__restrict__ x, double const *__restrict__ norms,
                                                  double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double nb = norms[i];
        if (CPLX) { out[2 * i] = x[2 * i] * nb; out[2 * i + 1] = x[2 * i + 1] * nb; } else out[i] = x[i] * nb;
    }
}
extern "C" int lsk_scale(int cplx, int64_t n, void const *x, double const *norms, void *out, void *stream) {
    if (n == 0) return 0;
    if (cplx) hipLaunchKernelGGL(k_scale<true>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, (double const *)x, norms, (double *)out);
    else hipLaunchKernelGGL(k_scale<false>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, (double const *)x, norms, (double *)out);
    LSK_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_scatter_owned(int64_t n, uint32_t const *__restrict__ perm, int64_t base, int64_t count,
                                                          uint64_t const *__restrict__ src, uint64_t *__restrict__ dst) {
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < n; g += (int64_t)gridDim.x * kBlock) {
        const int64_t j = (int64_t)perm[g] - base;
        if (j >= 0 && j < count) dst[j] = src[g];
    }
}
extern "C" int lsk_scatter_owned(int64_t n, uint32_t const *perm, int64_t base, int64_t count, uint64_t const *src, uint64_t *dst,
                                 void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_scatter_owned, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, perm, base, count, src, dst);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// INDEXED pull kernels of the projected bases (k_pull_t / k_pull_gather).
//
// One block per 256-row tile, the packet list PER WAVE: each wave owns a 256-slot ring of the LDS list and the rows of its
// 64 lanes.  Stage A appends the packets of three flip-mask groups (<= 192) behind what is left in the ring, stage B takes
// full chunks of 64 packets out of it -- K4 with every lane busy -- and leaves the remainder (< 64) for the next round; the
// tile ends with one partial chunk.  No block barrier inside a tile except around the shared near window, so the four waves
// of a block drift apart and the ALU phase (K4) of one overlaps the look-ups of another.
//
// Stage B per packet: K4 (orbit minimum [+ character, norm]) -> SLOT of the representative:
//   near partners: the tile stages the sorted representatives [tile - halo, tile + 256 + halo) as a two-way hash set in LDS
//     (nw_*, below): one ds_read_b64 instead of the 11-step binary search of round 3; global index -> slot (perm[g] or g);
//   far partners: ONE 16-byte bucket of the static index table (lsk_gtab) -> slot.
// What happens with the slot is the SINK:
//   SINK_FUSED   : value = xsrc[slot], ds_add_f64 into the tile's LDS copy of y, y written once (the one-GPU default);
//   SINK_RESOLVE : the slot (and, unless every packet has the same real amplitude, its coefficient) is written to the
//                  per-wave packet stream of lsk_pullbuf and NOTHING of x is read -- this half of the matvec runs while the
//                  blocks of x are still on the wire (ls_amd_repl_matvec, dist.c); k_pull_gather then streams the slots,
//                  gathers x and accumulates.  The stream is recomputed every matvec: the path stays matrix-free.
// ---------------------------------------------------------------------------------------------
constexpr int kWvRing = 256; // slots per wave: < 64 left over + 3 groups x 64 lanes
constexpr int kWvGroups = 3;
enum { K4_TRIVIAL = 0, K4_PM1 = 1, K4_GENERAL = 2 };     // what K4 has to deliver (lsk_basis.k4_mode != 0 -> TRIVIAL)
enum { COEF_UNI = 0, COEF_REAL = 1, COEF_CPLX = 2 };      // per-packet coefficient: none (one real amplitude), f64, 2 x f64
enum { SINK_FUSED = 0, SINK_RESOLVE = 1 };
constexpr uint32_t kNoSlot = 0xffffffffu;

// Near window as a hash set in LDS: kNwSets sets of two 4-byte entries.  h = d * odd constant is a bijection of the 32-bit
// offset d = rep - v0, set = top 10 bits, entry = (low 22 bits of h) << 10 | position in the window (< 1024) -- so set and
// tag together identify d and a match cannot be a false positive.  A set that is already full DROPS the third arrival: the
// window is only an accelerator, whatever it does not answer goes through the static index table (which holds every
// representative).  At <= 768 staged entries ~2 % are dropped.
constexpr int kNwSets = 1024;
constexpr int kNwMaxWin = 1024;
constexpr uint32_t kNwEmpty = 0xffffffffu;
__host__ __device__ __forceinline__ uint32_t nw_mix(uint32_t d) { return d * 0x9E3779B1u; }
__host__ __device__ __forceinline__ uint32_t nw_entry(uint32_t h, int pos) { return (h << 10) | (uint32_t)pos; }
__device__ __forceinline__ void nw_insert(uint32_t *tab, uint32_t d, int pos) {
    const uint32_t h = nw_mix(d), e = nw_entry(h, pos);
    uint32_t *s = tab + 2 * (h >> 22);
    if (atomicCAS(s, kNwEmpty, e) != kNwEmpty) (void)atomicCAS(s + 1, kNwEmpty, e);
}
__host__ __device__ __forceinline__ int nw_match(uint32_t e0, uint32_t e1, uint32_t h) {
    const uint32_t want = h << 10;
    if (((e0 ^ want) >> 10) == 0 && e0 != kNwEmpty) return (int)(e0 & 1023u);
    if (((e1 ^ want) >> 10) == 0 && e1 != kNwEmpty) return (int)(e1 & 1023u);
    return -1;
}
__device__ __forceinline__ int nw_find(uint32_t const *tab, uint32_t d) {
    const uint32_t h = nw_mix(d);
    const uint2 e = *reinterpret_cast<uint2 const *>(tab + 2 * (h >> 22));
    return nw_match(e.x, e.y, h);
}
// host mirror (tests, no device): stage reps[0, n) (ascending, n <= 1024) with the rule of the kernel, sequentially, then look
// `key` up: its position, -1 when the window does not answer (absent, or dropped from a full set), -2 on bad arguments
extern "C" int lsk_test_nw_find(uint64_t const *reps, int n, uint64_t key) {
    if (n < 1 || n > kNwMaxWin) return -2;
    static thread_local uint32_t tab[2 * kNwSets];
    for (int i = 0; i < 2 * kNwSets; ++i) tab[i] = kNwEmpty;
    for (int i = 0; i < n; ++i) {
        const uint32_t d = window_offset(reps[i], reps[0]);
        if (d == kWinAbsent) continue;
        const uint32_t h = nw_mix(d), e = nw_entry(h, i);
        uint32_t *s = tab + 2 * (h >> 22);
        if (s[0] == kNwEmpty) s[0] = e; else if (s[1] == kNwEmpty) s[1] = e;
    }
    if (key < reps[0]) return -1;
    const uint32_t d = window_offset(key, reps[0]);
    if (d == kWinAbsent) return -1;
    const uint32_t h = nw_mix(d);
    return nw_match(tab[2 * (h >> 22)], tab[2 * (h >> 22) + 1], h);
}

// block -> tile of the pull kernels.  Blocks b = x (mod 8) run on XCD x; a plain grid therefore deals every XCD every eighth
// tile, and each of the eight L2s fetches its own copy of the partner sectors that neighbouring tiles share.  With a chunk of C
// tiles per XCD the blocks of one XCD walk C consecutive tiles before they jump by 8 C.  (The last, incomplete round of
// chunks keeps the identity.)
__host__ __device__ __forceinline__ int64_t pull_tile_of_block(int64_t b, int64_t n_tiles, int C) {
    if (C <= 1) return b;
    const int64_t round = 8 * (int64_t)C, full = n_tiles / round * round;
    if (b >= full) return b;
    const int64_t x = b & 7, j = b >> 3, q = j / C, r = j - q * C;
    return (q * 8 + x) * C + r;
}
// Measured (profiles/r4_pull_xcd_chunk_ab.txt; C = 0 / 16 / 64 / 256 / 1024 / 4096): chain_36_symm cached gather 4.18 / 3.88 / 3.76 /
// 3.67 / 3.65 / 3.88 ms, fused 18.08 / 17.58 / 17.24 / 17.38 / 17.28 / 18.08 ms; chain_40_symm cached 65.7 / 63.6 / 59.9 / 60.1 / 60.0 /
// 61.5 ms, fused 282.5 / 285.0 / 280.9 / 287.3 / 279.5 / 280.1 ms.
extern "C" int64_t lsk_test_pull_tile_of_block(int64_t b, int64_t n_tiles, int C) { return pull_tile_of_block(b, n_tiles, C); }
constexpr int kPullXcdChunk = 256;
static int pull_xcd_chunk() { return kPullXcdChunk; }

template <typename W, int K4M, int COEF, bool CPLX, int SINK>
__global__ __launch_bounds__(kBlock, (COEF == COEF_CPLX ? 4 : 6)) void k_pull_t(lsk_runs runs, int n_groups, lsk_group const *__restrict__ groups,
                                                   lsk_term const *__restrict__ off, int n_diag,
                                                   lsk_term const *__restrict__ diag, lsk_basis bs,
                                                   lsk_group_elem const *__restrict__ elems, int64_t row0, int64_t row1,
                                                   uint64_t const *__restrict__ reps,
                                                   double const *__restrict__ norms_local, lsk_pullidx ix,
                                                   uint64_t const *__restrict__ greps, int64_t n_global,
                                                   double const *__restrict__ xsrc, int halo, double uni_v,
                                                   double *__restrict__ y, lsk_pullbuf buf, int *err, int xcd_chunk) {
    typedef typename ChainX<CPLX>::type X;
    constexpr bool REAL = COEF != COEF_CPLX;
    constexpr bool FUSED = SINK == SINK_FUSED;
    constexpr int NC = COEF == COEF_UNI ? 0 : (COEF == COEF_REAL ? 1 : 2);
    X const *__restrict__ xv = (X const *)xsrc;
    constexpr int kCap = (kBlock / 64) * kWvRing;
    __shared__ uint32_t s_nw[2 * kNwSets];
    extern __shared__ uint32_t s_nwslot[]; // [kNwMaxWin] when ix.perm != NULL (launch-time size): slot of every window entry
    __shared__ W s_beta[kCap];
    __shared__ double s_coef[NC ? kCap * NC : 1];
    __shared__ uint8_t s_row[kCap]; // row inside the wave (0..63)
    __shared__ double s_acc[FUSED ? kBlock * (CPLX ? 2 : 1) : 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rb = wave * kWvRing; // this wave's ring
    uint64_t const *__restrict__ tab = ix.tab.entries;
    const int64_t n_tiles = (row1 - row0 + kBlock - 1) / kBlock;
    for (int64_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
        const int64_t t0 = row0 + pull_tile_of_block(tb, n_tiles, gridDim.x >= n_tiles ? xcd_chunk : 0) * kBlock;
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        uint64_t a = 0;
        double inv_na = 0.0;
        if (valid) {
            a = reps[i];
            const double na = norms_local[i];
            inv_na = na > 0.0 ? 1.0 / na : 0.0;
        }
        if (FUSED) { if (CPLX) { s_acc[2 * tid] = 0.0; s_acc[2 * tid + 1] = 0.0; } else s_acc[tid] = 0.0; }
        // the diagonal coefficient now, not in the epilogue: the run tables then do not stay in scalar registers across stage B
        double dr = 0.0, di = 0.0;
        if (FUSED && valid && n_diag > 0) diag_coeff<uint64_t, REAL>(runs, n_diag, diag, a, dr, di);
        int64_t gbase = 0;
        int wn = 0;
        uint64_t v0 = 0;
        if (halo > 0) {
            const int64_t ig0 = ix.row_g0 + t0;
            gbase = ig0 > halo ? ig0 - halo : 0;
            const int64_t left = n_global - gbase;
            wn = (int)(left < (int64_t)(kBlock + 2 * halo) ? left : (int64_t)(kBlock + 2 * halo));
            v0 = greps[gbase];
            uint4 *const z = reinterpret_cast<uint4 *>(s_nw);
            for (int w = tid; w < 2 * kNwSets / 4; w += kBlock) z[w] = make_uint4(kNwEmpty, kNwEmpty, kNwEmpty, kNwEmpty);
            __syncthreads();
            for (int w = tid; w < wn; w += kBlock) {
                const uint32_t d = window_offset(greps[gbase + w], v0);
                if (d != kWinAbsent) nw_insert(s_nw, d, w);
                // replicated-x exchange: the slot of a near partner comes out of LDS (a coalesced load per window entry here)
                // instead of one dependent, uncoalesced load of perm[] per near packet
                if (ix.perm) s_nwslot[w] = ix.perm[gbase + w];
            }
        }
        __syncthreads(); // the window is staged
        int head = 0, cnt = 0; // wave-uniform: the ring holds [head, head + cnt) mod kWvRing
        // packet stream of this wave's 64 rows (SINK_RESOLVE)
        const int64_t wg = ((t0 - buf.row0) >> 6) + wave;
        const int64_t sbase = buf.offs ? buf.offs[wg] : wg * buf.cap; // exact layout (slot cache) | `cap` packets of room each
        int emitted = 0;
        // K chunks at once: the packets at ring positions head + 64 k + lane (the last chunk holds m <= 64 of them):
        // K4 -> slot [-> value -> ds_add_f64], the loads of the K packets of a lane issued together
        auto chunks = [&](auto KC, int m) {
            constexpr int K = decltype(KC)::value;
            uint64_t beta[K], bkt[K];
            double hr[K], hi[K];
            int r[K], pos[K];
            bool live[K];
            uint32_t tag[K], slot[K];
            ulonglong2 first[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                live[k] = k + 1 < K || lane < m;
                const int e = rb + ((head + 64 * k + lane) & (kWvRing - 1));
                beta[k] = live[k] ? (uint64_t)s_beta[e] : 0;
                hr[k] = 1.0; hi[k] = 0.0;
                if (NC == 1) hr[k] = s_coef[e];
                if (NC == 2) { hr[k] = s_coef[2 * e]; hi[k] = s_coef[2 * e + 1]; }
                r[k] = (int)s_row[e];
            }
            // the three x-independent steps of a chunk: K4, near window (LDS), first-level load (perm entry | home bucket)
            auto step_k4 = [&](int k) {
                if (kAblate && (bs.debug_ablate & 4)) return; // profiling builds: no K4 (the look-ups then mostly miss)
                if (K4M == K4_TRIVIAL) {
                    beta[k] = (uint64_t)rep_trivial<W>(bs, elems, (W)beta[k]); // xsrc is pre-multiplied by norm(rep)
                } else if (live[k]) {
                    W rep; double chr, chi, stab;
                    state_info_w<W, K4M == K4_PM1>(bs, elems, (W)beta[k], rep, chr, chi, stab);
                    const double n2 = stab * bs.inv_order;
                    if (!(n2 > 1e-12)) live[k] = false; // zero-norm orbit: contributes nothing (DMV:110)
                    else {
                        const double nb = sqrt(n2);
                        beta[k] = (uint64_t)rep;
                        const double tr = (hr[k] * chr + hi[k] * chi) * nb, ti = (hi[k] * chr - hr[k] * chi) * nb;
                        hr[k] = tr; hi[k] = ti;
                    }
                }
            };
            auto step_window = [&](int k) { // near window: LDS only
                pos[k] = -1; bkt[k] = 0; tag[k] = 0; slot[k] = kNoSlot;
                first[k] = make_ulonglong2(0, 0);
                if (kAblate && (bs.debug_ablate & 2)) { // profiling builds: K4 kept alive, no look-up, no accumulation
                    if (beta[k] == 0x123456789abcdefULL) atomicExch(err, 2);
                    live[k] = false;
                }
                if (live[k]) {
                    if (wn > 0 && beta[k] >= v0 && !(kAblate && (bs.debug_ablate & 32))) {
                        const uint32_t d = window_offset(beta[k], v0);
                        if (d != kWinAbsent) pos[k] = nw_find(s_nw, d);
                    }
                    if (pos[k] < 0) gt_split(ix.tab, beta[k], bkt[k], tag[k]);
                }
            };
            auto step_first = [&](int k) { // first-level loads: perm entry (near) or home bucket (far)
                if (!live[k]) return;
                if (pos[k] >= 0) slot[k] = ix.perm ? s_nwslot[pos[k]] : (uint32_t)(gbase + pos[k]);
                else first[k] = *(ulonglong2 const *)(tab + 2 * bkt[k]);
            };
            // (step by step over the chunks.  Chunk by chunk instead -- the home-bucket load of chunk k in flight while chunk k + 1 runs
            // its K4 -- measured no different: chain_36_symm 17.61 vs 17.72 ms, chain_40_symm 276.5 vs 277.9 ms,
            // profiles/r5_pull_skew_ab.txt: the kernel is at the fabric's random-request rate, not at a latency it could hide)
#pragma unroll
            for (int k = 0; k < K; ++k) step_k4(k);
#pragma unroll
            for (int k = 0; k < K; ++k) step_window(k);
#pragma unroll
            for (int k = 0; k < K; ++k) step_first(k);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (!live[k] || pos[k] >= 0) continue;
                slot[k] = gt_resolve(ix.tab, tab, bkt[k], tag[k], first[k]);
                if (slot[k] == kNoSlot) { atomicExch(err, 1); live[k] = false; }
            }
            if constexpr (FUSED) {
                X val[K];
#pragma unroll
                for (int k = 0; k < K; ++k) val[k] = (live[k] && !(kAblate && (bs.debug_ablate & 64))) ? xv[slot[k]] : cx_zero<X>();
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (!live[k]) continue;
                    const int ra = (wave << 6) + r[k];
                    if constexpr (CPLX) {
                        atomicAdd(&s_acc[2 * ra], hr[k] * val[k].x - hi[k] * val[k].y);
                        atomicAdd(&s_acc[2 * ra + 1], hr[k] * val[k].y + hi[k] * val[k].x);
                    } else if constexpr (NC == 0) {
                        atomicAdd(&s_acc[ra], val[k]);
                    } else {
                        atomicAdd(&s_acc[ra], hr[k] * val[k]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k + 1 == K && lane >= m) continue;
                    const int64_t o = sbase + emitted + 64 * k + lane;
                    __builtin_nontemporal_store(live[k] ? slot[k] : kNoSlot, buf.slots + o);
                    __builtin_nontemporal_store((uint8_t)r[k], buf.rows + o);
                    if (NC == 1) __builtin_nontemporal_store(hr[k], buf.coefs + o);
                    if (NC == 2) { __builtin_nontemporal_store(hr[k], buf.coefs + 2 * o); __builtin_nontemporal_store(hi[k], buf.coefs + 2 * o + 1); }
                }
                emitted += 64 * (K - 1) + m;
            }
        };
        const W tdiff = (W)a ^ (W)((W)a >> 1); // bit b set: sites b, b + 1 differ (adjacent exchange groups)
        for (int g0 = 0; g0 < n_groups; g0 += kWvGroups) {
            const int g1 = min(g0 + kWvGroups, n_groups);
            for (int g = g0; g < g1; ++g) { // stage A: append
                lsk_group const G = groups[g];
                double cr = 0.0, ci = 0.0;
                bool act;
                if (NC == 0) { // every group is an exchange pair with the amplitude uni_v
                    act = valid && (G.adj >= 0 ? (bool)((tdiff >> G.adj) & 1) : WordTraits<W>::popc((W)a & (W)G.x) == 1);
                } else {
                    if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                    act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
                }
                const unsigned long long ball = __ballot(act);
                if (act) {
                    const int slot = rb + ((head + cnt + __popcll(ball & ((1ULL << lane) - 1))) & (kWvRing - 1));
                    s_beta[slot] = (W)(a ^ G.x);
                    s_row[slot] = (uint8_t)lane;
                    if (NC == 1) s_coef[slot] = cr * inv_na;
                    if (NC == 2) { s_coef[2 * slot] = cr * inv_na; s_coef[2 * slot + 1] = -ci * inv_na; }
                }
                cnt += __popcll(ball);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (kAblate && (bs.debug_ablate & 1)) { head = (head + cnt) & (kWvRing - 1); cnt = 0; } // profiling builds: stage A only
            // stage B on full chunks: two at a time while the ring has them (trivial sectors; the element loops of the other
            // sectors are long enough by themselves, and two chunks of their state do not fit the scalar registers)
            constexpr int KMAX = K4M == K4_TRIVIAL ? 2 : 1;
            while (cnt >= 64 * KMAX) {
                chunks(std::integral_constant<int, KMAX>(), 64);
                head = (head + 64 * KMAX) & (kWvRing - 1);
                cnt -= 64 * KMAX;
            }
            if constexpr (KMAX == 2)
                if (cnt >= 64) {
                    chunks(std::integral_constant<int, 1>(), 64);
                    head = (head + 64) & (kWvRing - 1);
                    cnt -= 64;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (cnt > 0) chunks(std::integral_constant<int, 1>(), cnt);
        if constexpr (FUSED) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const int64_t ig = ix.row_g0 + i;
                const uint32_t own = ix.perm ? ix.perm[ig] : (uint32_t)ig;
                const double back = K4M == K4_TRIVIAL ? inv_na : 1.0; // xsrc holds x * norm(rep) in the prescaling K4 modes
                const double sc = NC == 0 ? uni_v * inv_na : 1.0;     // one amplitude for every packet: applied once per row
                if constexpr (CPLX) {
                    const X xo = xv[own];
                    const double xr = xo.x * back, xi = xo.y * back;
                    double yr = dr * xr - di * xi + sc * s_acc[2 * tid], yi = dr * xi + di * xr + sc * s_acc[2 * tid + 1];
                    if (n_diag == 0) { yr += y[2 * i]; yi += y[2 * i + 1]; } // accumulate, DMV:1062-1063
                    y[2 * i] = yr; y[2 * i + 1] = yi;
                } else {
                    double yr = n_diag > 0 ? dr * (xv[own] * back) + sc * s_acc[tid] : sc * s_acc[tid];
                    if (n_diag == 0) yr += y[i];
                    y[i] = yr;
                }
            }
        } else if (lane == 0) buf.counts[wg] = (uint32_t)emitted;
        __syncthreads(); // every wave is done with the window
    }
}

// Second half of the split matvec: the packet stream of k_pull_t<..., SINK_RESOLVE> -> x[slot] -> y.  One wave per 64 rows,
// no barrier at all: a wave only touches the LDS accumulators of its own rows.  GU chunks of 64 packets in flight per wave.
template <int COEF, bool CPLX>
__global__ __launch_bounds__(kBlock) void k_pull_gather(lsk_runs runs, int n_diag, lsk_term const *__restrict__ diag, int k4_mode,
                                                        int64_t row0, int64_t row1, uint64_t const *__restrict__ reps,
                                                        double const *__restrict__ norms_local, lsk_pullidx ix,
                                                        double const *__restrict__ xsrc, double uni_v, double *__restrict__ y,
                                                        lsk_pullbuf buf, int xcd_chunk) {
    typedef typename ChainX<CPLX>::type X;
    constexpr bool REAL = COEF != COEF_CPLX;
    constexpr int NC = COEF == COEF_UNI ? 0 : (COEF == COEF_REAL ? 1 : 2);
    constexpr int GU = 4;
    X const *__restrict__ xv = (X const *)xsrc;
    __shared__ double s_acc[kBlock * (CPLX ? 2 : 1)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t n_tiles = (row1 - row0 + kBlock - 1) / kBlock;
    for (int64_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
        const int64_t t0 = row0 + pull_tile_of_block(tb, n_tiles, gridDim.x >= n_tiles ? xcd_chunk : 0) * kBlock;
        if ((t0 + (wave << 6)) >= row1) continue; // wave-uniform
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        if (CPLX) { s_acc[2 * tid] = 0.0; s_acc[2 * tid + 1] = 0.0; } else s_acc[tid] = 0.0;
        const int64_t wg = ((t0 - buf.row0) >> 6) + wave;
        const int64_t sbase = buf.offs ? buf.offs[wg] : wg * buf.cap;
        const int n = (int)buf.counts[wg];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < n; c += 64 * GU) {
            uint32_t slot[GU];
            int r[GU];
            double hr[GU], hi[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int p = c + 64 * u + lane;
                slot[u] = kNoSlot; r[u] = 0; hr[u] = 1.0; hi[u] = 0.0;
                if (p < n) {
                    const int64_t o = sbase + p;
                    slot[u] = __builtin_nontemporal_load(buf.slots + o);
                    r[u] = (int)__builtin_nontemporal_load(buf.rows + o);
                    if (NC == 1) hr[u] = __builtin_nontemporal_load(buf.coefs + o);
                    if (NC == 2) { hr[u] = __builtin_nontemporal_load(buf.coefs + 2 * o); hi[u] = __builtin_nontemporal_load(buf.coefs + 2 * o + 1); }
                }
            }
            X val[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) val[u] = slot[u] != kNoSlot ? xv[slot[u]] : cx_zero<X>();
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                if (slot[u] == kNoSlot) continue;
                const int ra = (wave << 6) + r[u];
                if constexpr (CPLX) {
                    atomicAdd(&s_acc[2 * ra], hr[u] * val[u].x - hi[u] * val[u].y);
                    atomicAdd(&s_acc[2 * ra + 1], hr[u] * val[u].y + hi[u] * val[u].x);
                } else if constexpr (NC == 0) {
                    atomicAdd(&s_acc[ra], val[u]);
                } else {
                    atomicAdd(&s_acc[ra], hr[u] * val[u]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint64_t a = reps[i];
            const double na = norms_local[i];
            const double inv_na = na > 0.0 ? 1.0 / na : 0.0;
            const int64_t ig = ix.row_g0 + i;
            const uint32_t own = ix.perm ? ix.perm[ig] : (uint32_t)ig;
            const double back = k4_mode != 0 ? inv_na : 1.0;
            const double sc = NC == 0 ? uni_v * inv_na : 1.0;
            double dr = 0.0, di = 0.0;
            if (n_diag > 0) diag_coeff<uint64_t, REAL>(runs, n_diag, diag, a, dr, di);
            if constexpr (CPLX) {
                const X xo = xv[own];
                const double xr = xo.x * back, xi = xo.y * back;
                double yr = dr * xr - di * xi + sc * s_acc[2 * tid], yi = dr * xi + di * xr + sc * s_acc[2 * tid + 1];
                if (n_diag == 0) { yr += y[2 * i]; yi += y[2 * i + 1]; }
                y[2 * i] = yr; y[2 * i + 1] = yi;
            } else {
                double yr = n_diag > 0 ? dr * (xv[own] * back) + sc * s_acc[tid] : sc * s_acc[tid];
                if (n_diag == 0) yr += y[i];
                y[i] = yr;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // this wave's accumulators are free again
        __builtin_amdgcn_wave_barrier();
    }
}

// kind of K4 work and of per-packet coefficient for (operator, basis) -- one decision for the fused, the resolve and the
// gather kernel
static void pull_kinds(lsk_operator const &op, lsk_basis const &bs, int &k4m, int &coef) {
    if (bs.k4_mode != 0) { k4m = K4_TRIVIAL; coef = !op.is_real ? COEF_CPLX : (op.uni ? COEF_UNI : COEF_REAL); }
    else if (bs.chars_pm1) { k4m = K4_PM1; coef = op.is_real ? COEF_REAL : COEF_CPLX; }
    else { k4m = K4_GENERAL; coef = COEF_CPLX; }
}
extern "C" int64_t lsk_pullbuf_cap(lsk_operator op) { return (int64_t)64 * (op.n_groups > 0 ? op.n_groups : 1); }
extern "C" int lsk_pullbuf_coef_doubles(lsk_operator op, lsk_basis bs) {
    int k4m, coef;
    pull_kinds(op, bs, k4m, coef);
    return coef == COEF_UNI ? 0 : (coef == COEF_REAL ? 1 : 2);
}

template <typename W, int K4M, int COEF, bool CPLX, int SINK>
static void launch_pull_t(lsk_operator const &op, lsk_basis const &bs, int64_t row0, int64_t row1, uint64_t const *reps,
                          double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global, void const *xsrc,
                          int halo, void *y, lsk_pullbuf buf, int *d_err, hipStream_t s) {
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    dim3 g((unsigned)tile_grid(k_pull_t<W, K4M, COEF, CPLX, SINK>, work_blocks)), b(kBlock);
    const size_t dyn = ix.perm ? sizeof(uint32_t) * kNwMaxWin : 0;
    hipLaunchKernelGGL((k_pull_t<W, K4M, COEF, CPLX, SINK>), g, b, dyn, s, op.runs, op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs,
                       bs.elems, row0, row1, reps, norms_local, ix, reps_global, n_global, (double const *)xsrc, halo, op.uni_v,
                       (double *)y, buf, d_err, pull_xcd_chunk());
}
template <typename W, bool CPLX, int SINK>
static int dispatch_pull_t(lsk_operator const &op, lsk_basis const &bs, int64_t row0, int64_t row1, uint64_t const *reps,
                           double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global,
                           void const *xsrc, int halo, void *y, lsk_pullbuf buf, int *d_err, hipStream_t s) {
    int k4m, coef;
    pull_kinds(op, bs, k4m, coef);
#define LSK_PT(K4M, COEF) launch_pull_t<W, K4M, COEF, CPLX, SINK>(op, bs, row0, row1, reps, norms_local, ix, reps_global, n_global, xsrc, halo, y, buf, d_err, s)
    if (coef == COEF_CPLX) {
        if constexpr (!CPLX && SINK == SINK_FUSED) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull: complex coefficients need c128 vectors"); return -1; }
        else { if (k4m == K4_TRIVIAL) LSK_PT(K4_TRIVIAL, COEF_CPLX); else if (k4m == K4_PM1) LSK_PT(K4_PM1, COEF_CPLX); else LSK_PT(K4_GENERAL, COEF_CPLX); }
    } else if (coef == COEF_REAL) { if (k4m == K4_TRIVIAL) LSK_PT(K4_TRIVIAL, COEF_REAL); else LSK_PT(K4_PM1, COEF_REAL); }
    else LSK_PT(K4_TRIVIAL, COEF_UNI);
#undef LSK_PT
    return 0;
}
static int pull_args_ok(lsk_basis const &bs, int halo, uint64_t const *reps_global, int64_t n_global, char const *who) {
    if (bs.proj != LSK_PROJ_FULL) { snprintf(g_err, sizeof(g_err), "%s is for projected bases", who); return -1; }
    if (halo < 0 || kBlock + 2 * halo > kNwMaxWin || (halo > 0 && (!reps_global || n_global <= 0))) { snprintf(g_err, sizeof(g_err), "%s: bad near window", who); return -1; }
    return 0;
}
extern "C" int lsk_pull_max_halo(void) { return (kNwMaxWin - kBlock) / 2; }

extern "C" int lsk_tile_pull_idx(lsk_operator op, lsk_basis bs, int cplx, int64_t row0, int64_t row1, uint64_t const *reps,
                                 double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global,
                                 void const *xsrc, int halo, void *y, int *d_err, void *stream) {
    if (row1 <= row0) return 0;
    if (pull_args_ok(bs, halo, reps_global, n_global, "lsk_tile_pull_idx") != 0) return -1;
    lsk_pullbuf none;
    memset(&none, 0, sizeof(none));
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define LSK_PA op, bs, row0, row1, reps, norms_local, ix, reps_global, n_global, xsrc, halo, y, none, d_err, s
    if (bs.number_sites <= 32) rc = cplx ? dispatch_pull_t<uint32_t, true, SINK_FUSED>(LSK_PA) : dispatch_pull_t<uint32_t, false, SINK_FUSED>(LSK_PA);
    else rc = cplx ? dispatch_pull_t<uint64_t, true, SINK_FUSED>(LSK_PA) : dispatch_pull_t<uint64_t, false, SINK_FUSED>(LSK_PA);
#undef LSK_PA
    if (rc != 0) return -1;
    LSK_LAUNCH_CHECK();
    return 0;
}
static int exclusive_scan_i64(int64_t n, int64_t const *in, int64_t *out, hipStream_t s);
// Packets that stage A of k_pull_t generates for every 64 rows (the streams' exact lengths): out[w] for the rows
// [row0 + 64 w, row0 + 64 w + 64).  Same activity test as stage A -- dead packets (zero-norm orbits) keep their place in a
// stream, so this is what the resolve kernel emits.
template <int COEF>
__global__ __launch_bounds__(kBlock) void k_pull_count(int n_groups, lsk_group const *__restrict__ groups, lsk_term const *__restrict__ off,
                                                       int64_t row0, int64_t row1, uint64_t const *__restrict__ reps,
                                                       int64_t *__restrict__ out) {
    constexpr bool REAL = COEF != COEF_CPLX;
    const int lane = threadIdx.x & 63;
    for (int64_t t0 = row0 + (int64_t)blockIdx.x * kBlock; t0 < row1; t0 += (int64_t)gridDim.x * kBlock) {
        const int64_t w0 = t0 + (threadIdx.x & ~63u);
        if (w0 >= row1) continue;
        const int64_t i = t0 + threadIdx.x;
        const bool valid = i < row1;
        const uint64_t a = valid ? reps[i] : 0;
        const uint64_t tdiff = a ^ (a >> 1);
        int cnt = 0;
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            bool act;
            if (COEF == COEF_UNI) act = valid && (G.adj >= 0 ? (bool)((tdiff >> G.adj) & 1) : __popcll(a & G.x) == 1);
            else {
                double cr = 0.0, ci = 0.0;
                if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
            }
            cnt += __popcll(__ballot(act));
        }
        if (lane == 0) out[(w0 - row0) >> 6] = cnt;
    }
}
// out[0, streams] <- exclusive offsets of the packet streams of rows [row0, row1) (streams = ceil(rows / 64); out[streams] =
// total); `out` has ((streams + 3) & ~3) + 1 entries and the entries behind out[streams] repeat the total: the last 256-row
// tile of the resolve / gather kernels runs four waves whatever the row count, and each reads its offset.  Synchronises the stream
extern "C" int lsk_tile_pull_stream_offsets(lsk_operator op, lsk_basis bs, int64_t row0, int64_t row1, uint64_t const *reps,
                                            int64_t *out, void *stream) {
    if (row1 <= row0) return 0;
    const int64_t streams = (row1 - row0 + 63) / 64;
    int k4m, coef;
    pull_kinds(op, bs, k4m, coef);
    hipStream_t s = (hipStream_t)stream;
    const int64_t padded = (streams + 3) & ~(int64_t)3;
    LSK_CHECK(hipMemsetAsync(out, 0, 8 * (size_t)(padded + 1), s));
    const dim3 g((unsigned)grid_for(row1 - row0)), b(kBlock);
    if (coef == COEF_UNI) hipLaunchKernelGGL(k_pull_count<COEF_UNI>, g, b, 0, s, op.n_groups, op.groups, op.off, row0, row1, reps, out);
    else if (coef == COEF_REAL) hipLaunchKernelGGL(k_pull_count<COEF_REAL>, g, b, 0, s, op.n_groups, op.groups, op.off, row0, row1, reps, out);
    else hipLaunchKernelGGL(k_pull_count<COEF_CPLX>, g, b, 0, s, op.n_groups, op.groups, op.off, row0, row1, reps, out);
    LSK_LAUNCH_CHECK();
    return exclusive_scan_i64(padded + 1, out, out, s);
}

// first half of the split matvec: rows [row0, row1) -> packet stream in `buf` (buf.row0 = the row that owns stream 0; a
// multiple of 64 rows below row0).  Reads neither x nor y.
extern "C" int lsk_tile_pull_resolve(lsk_operator op, lsk_basis bs, int64_t row0, int64_t row1, uint64_t const *reps,
                                     double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global,
                                     int halo, lsk_pullbuf buf, int *d_err, void *stream) {
    if (row1 <= row0) return 0;
    if (pull_args_ok(bs, halo, reps_global, n_global, "lsk_tile_pull_resolve") != 0) return -1;
    if (!buf.slots || !buf.rows || !buf.counts || (!buf.offs && buf.cap < lsk_pullbuf_cap(op)) || ((row0 - buf.row0) & 255) != 0 || row0 < buf.row0 ||
        (lsk_pullbuf_coef_doubles(op, bs) > 0 && !buf.coefs)) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull_resolve: bad packet buffer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define LSK_PA op, bs, row0, row1, reps, norms_local, ix, reps_global, n_global, nullptr, halo, nullptr, buf, d_err, s
    if (bs.number_sites <= 32) rc = dispatch_pull_t<uint32_t, false, SINK_RESOLVE>(LSK_PA);
    else rc = dispatch_pull_t<uint64_t, false, SINK_RESOLVE>(LSK_PA);
#undef LSK_PA
    if (rc != 0) return -1;
    LSK_LAUNCH_CHECK();
    return 0;
}
// second half: y[row0, row1) from the packet stream and xsrc
extern "C" int lsk_tile_pull_gather(lsk_operator op, lsk_basis bs, int cplx, int64_t row0, int64_t row1, uint64_t const *reps,
                                    double const *norms_local, lsk_pullidx ix, void const *xsrc, lsk_pullbuf buf, void *y,
                                    void *stream) {
    if (row1 <= row0) return 0;
    int k4m, coef;
    pull_kinds(op, bs, k4m, coef);
    if (coef == COEF_CPLX && !cplx) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull_gather: complex coefficients need c128 vectors"); return -1; }
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(1), b(kBlock);
#define LSK_PG(COEF, CPLX)                                                                                              \
    do {                                                                                                                \
        g.x = (unsigned)tile_grid(k_pull_gather<COEF, CPLX>, work_blocks);                                              \
        hipLaunchKernelGGL((k_pull_gather<COEF, CPLX>), g, b, 0, s, op.runs, op.n_diag, op.diag, bs.k4_mode, row0, row1, reps, \
                           norms_local, ix, (double const *)xsrc, op.uni_v, (double *)y, buf, pull_xcd_chunk());        \
    } while (0)
    if (coef == COEF_CPLX) LSK_PG(COEF_CPLX, true);
    else if (coef == COEF_REAL) { if (cplx) LSK_PG(COEF_REAL, true); else LSK_PG(COEF_REAL, false); }
    else { if (cplx) LSK_PG(COEF_UNI, true); else LSK_PG(COEF_UNI, false); }
#undef LSK_PG
    LSK_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Consumer side (K7 + K8): received packets -> local index -> atomic add
// ---------------------------------------------------------------------------------------------
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scatter(lsk_index ix, int64_t n, uint64_t const *__restrict__ betas,
                                                    double const *__restrict__ vals, double *y,
                                                    double const *__restrict__ norms, int *err, int xcd_chunk) {
    // XCD-chunked block -> packets map (pull_tile_of_block): the look-ups of neighbouring packet blocks read neighbouring table /
    // representative lines, which then meet in ONE L2 instead of eight
    extern __shared__ uint64_t s_db[]; // rank directory: the binomials of the closed-form rank (launch-time size, 0 without one)
    if (ix.dir) { rankdir_load(ix, s_db); __syncthreads(); }
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    for (int64_t kb = blockIdx.x; kb < n_blocks; kb += gridDim.x) {
        const int64_t k = pull_tile_of_block(kb, n_blocks, gridDim.x >= n_blocks ? xcd_chunk : 0) * kBlock + threadIdx.x;
        if (k >= n) continue;
        double vr, vi = 0.0;
        if (CPLX) { vr = vals[2 * k]; vi = vals[2 * k + 1]; } else vr = vals[k];
        if (vr == 0.0 && vi == 0.0) continue; // DMV:110
        int64_t idx = ix.kind == LSK_INDEX_IDENTITY ? (int64_t)betas[k] : (ix.dir ? rankdir_index(ix, betas[k], s_db) : search_index(ix, betas[k]));
        if (idx < 0) { atomicExch(err, 1); continue; }
        if (norms) { double nb = norms[idx]; vr *= nb; vi *= nb; }
        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
        else atomic_add_f64(y + idx, vr);
    }
}
extern "C" int lsk_scatter(lsk_index ix, int cplx, int64_t n, uint64_t const *betas, void const *vals, void *y,
                           double const *norms, int *d_err, void *stream) {
    if (n == 0) return 0;
    if (ix.kind == LSK_INDEX_COMBINADIC) { snprintf(g_err, sizeof(g_err), "lsk_scatter: SEARCH/IDENTITY index only"); return -1; }
    dim3 g(grid_for(n)), b(kBlock);
    // one block per 256 packets, 64 consecutive blocks per XCD (chain_28 x 8 partitions: consumers 14.85 -> 14.37 ms; chunks of
    // 1 / 16 / 256 / 1024: 14.65 / 14.42 / 14.36 / 14.39 -- profiles/r4_scatter_xcd_chunk_ab.txt)
    constexpr int chunk = 64;
    { const int64_t nb = (n + kBlock - 1) / kBlock; g.x = (unsigned)(nb < ((int64_t)1 << 30) ? nb : ((int64_t)1 << 30)); }
    const size_t dyn = ix.dir ? sizeof(uint64_t) * (size_t)ix.dir_sites * (size_t)(ix.dir_weight + 1) : 0;
    if (cplx) hipLaunchKernelGGL(k_scatter<true>, g, b, dyn, (hipStream_t)stream, ix, n, betas, (double const *)vals, (double *)y, norms, d_err, chunk);
    else hipLaunchKernelGGL(k_scatter<false>, g, b, dyn, (hipStream_t)stream, ix, n, betas, (double const *)vals, (double *)y, norms, d_err, chunk);
    LSK_LAUNCH_CHECK();
    return 0;
}

// Fused consumers: ONE launch over all segments of a round's receive buffer (chain_28 x 8 partitions ran 56 launches of 0.26 ms
// per matvec, each with its own ramp and tail).  Block b takes packets [256 b, 256 b + 256) of the concatenated count space;
// the segment of a packet is found in an LDS copy of the (<= 65) segment starts.
__device__ __forceinline__ int seg_of(int64_t const *s_start, int n, int64_t k) {
    int lo = 0, hi = n; // largest s with start[s] <= k
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_start[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scatter_idx(lsk_segs segs, char const *__restrict__ base, int xcd_chunk) {
    __shared__ int64_t s_start[LSK_MAX_SEGS + 1];
    for (int i = threadIdx.x; i <= segs.n; i += kBlock) s_start[i] = segs.start[i];
    __syncthreads();
    const int64_t n = segs.start[segs.n];
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    for (int64_t kb = blockIdx.x; kb < n_blocks; kb += gridDim.x) {
        const int64_t k = pull_tile_of_block(kb, n_blocks, gridDim.x >= n_blocks ? xcd_chunk : 0) * kBlock + threadIdx.x;
        if (k >= n) continue;
        const int sg = seg_of(s_start, segs.n, k);
        const int64_t j = k - s_start[sg];
        const uint32_t idx = reinterpret_cast<uint32_t const *>(base + segs.key_off[sg])[j];
        double const *vals = reinterpret_cast<double const *>(base + segs.val_off[sg]);
        double *y = reinterpret_cast<double *>(segs.y[sg]);
        if (CPLX) {
            const double vr = vals[2 * j], vi = vals[2 * j + 1];
            if (vr == 0.0 && vi == 0.0) continue; // DMV:110
            atomic_add_f64(y + 2 * (size_t)idx, vr);
            atomic_add_f64(y + 2 * (size_t)idx + 1, vi);
        } else {
            const double vr = vals[j];
            if (vr == 0.0) continue;
            atomic_add_f64(y + idx, vr);
        }
    }
}
extern "C" int lsk_scatter_idx(int cplx, lsk_segs const *segs, void const *base, void *stream) {
    if (segs->n < 1 || segs->n > LSK_MAX_SEGS) { snprintf(g_err, sizeof(g_err), "lsk_scatter_idx: %d segments", segs->n); return -1; }
    const int64_t n = segs->start[segs->n];
    if (n <= 0) return 0;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    dim3 g((unsigned)(nb < ((int64_t)1 << 30) ? nb : ((int64_t)1 << 30))), b(kBlock);
    if (cplx) hipLaunchKernelGGL(k_scatter_idx<true>, g, b, 0, (hipStream_t)stream, *segs, (char const *)base, 64);
    else hipLaunchKernelGGL(k_scatter_idx<false>, g, b, 0, (hipStream_t)stream, *segs, (char const *)base, 64);
    LSK_LAUNCH_CHECK();
    return 0;
}
// ... and for packets that carry the state (projected bases, or no room for the all-destinations directory): one index, one y
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scatter_segs(lsk_index ix, lsk_segs segs, char const *__restrict__ base,
                                                         double const *__restrict__ norms, int *err, int xcd_chunk) {
    __shared__ int64_t s_start[LSK_MAX_SEGS + 1];
    extern __shared__ uint64_t s_db[];
    for (int i = threadIdx.x; i <= segs.n; i += kBlock) s_start[i] = segs.start[i];
    if (ix.dir) rankdir_load(ix, s_db);
    __syncthreads();
    const int64_t n = segs.start[segs.n];
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    double *y = reinterpret_cast<double *>(segs.y[0]);
    for (int64_t kb = blockIdx.x; kb < n_blocks; kb += gridDim.x) {
        const int64_t k = pull_tile_of_block(kb, n_blocks, gridDim.x >= n_blocks ? xcd_chunk : 0) * kBlock + threadIdx.x;
        if (k >= n) continue;
        const int sg = seg_of(s_start, segs.n, k);
        const int64_t j = k - s_start[sg];
        double const *vals = reinterpret_cast<double const *>(base + segs.val_off[sg]);
        double vr, vi = 0.0;
        if (CPLX) { vr = vals[2 * j]; vi = vals[2 * j + 1]; } else vr = vals[j];
        if (vr == 0.0 && vi == 0.0) continue; // DMV:110
        const uint64_t beta = reinterpret_cast<uint64_t const *>(base + segs.key_off[sg])[j];
        const int64_t idx = ix.kind == LSK_INDEX_IDENTITY ? (int64_t)beta : (ix.dir ? rankdir_index(ix, beta, s_db) : search_index(ix, beta));
        if (idx < 0) { atomicExch(err, 1); continue; }
        if (norms) { const double nb = norms[idx]; vr *= nb; vi *= nb; }
        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
        else atomic_add_f64(y + idx, vr);
    }
}
extern "C" int lsk_scatter_segs(lsk_index ix, int cplx, lsk_segs const *segs, void const *base, double const *norms, int *d_err, void *stream) {
    if (segs->n < 1 || segs->n > LSK_MAX_SEGS) { snprintf(g_err, sizeof(g_err), "lsk_scatter_segs: %d segments", segs->n); return -1; }
    if (ix.kind == LSK_INDEX_COMBINADIC) { snprintf(g_err, sizeof(g_err), "lsk_scatter_segs: SEARCH/IDENTITY index only"); return -1; }
    const int64_t n = segs->start[segs->n];
    if (n <= 0) return 0;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    dim3 g((unsigned)(nb < ((int64_t)1 << 30) ? nb : ((int64_t)1 << 30))), b(kBlock);
    const size_t dyn = ix.dir ? sizeof(uint64_t) * (size_t)ix.dir_sites * (size_t)(ix.dir_weight + 1) : 0;
    if (cplx) hipLaunchKernelGGL(k_scatter_segs<true>, g, b, dyn, (hipStream_t)stream, ix, *segs, (char const *)base, norms, d_err, 64);
    else hipLaunchKernelGGL(k_scatter_segs<false>, g, b, dyn, (hipStream_t)stream, ix, *segs, (char const *)base, norms, d_err, 64);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ... and with the segments of one producer going to DIFFERENT partitions of this process (P logical partitions on one device: the
// "exchange" is a pointer hand-off): index and norms of a segment come from a device array of per-partition contexts
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scatter_parts(lsk_part_ctx const *__restrict__ parts, lsk_index any, lsk_segs segs,
                                                          char const *__restrict__ base, int *err, int xcd_chunk) {
    __shared__ int64_t s_start[LSK_MAX_SEGS + 1];
    extern __shared__ uint64_t s_db[];
    for (int i = threadIdx.x; i <= segs.n; i += kBlock) s_start[i] = segs.start[i];
    if (any.dir) rankdir_load(any, s_db); // (sites, weight and the binomials are those of the basis: the same for every partition)
    __syncthreads();
    const int64_t n = segs.start[segs.n];
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    for (int64_t kb = blockIdx.x; kb < n_blocks; kb += gridDim.x) {
        const int64_t k = pull_tile_of_block(kb, n_blocks, gridDim.x >= n_blocks ? xcd_chunk : 0) * kBlock + threadIdx.x;
        if (k >= n) continue;
        const int sg = seg_of(s_start, segs.n, k);
        const int64_t j = k - s_start[sg];
        double const *vals = reinterpret_cast<double const *>(base + segs.val_off[sg]);
        double vr, vi = 0.0;
        if (CPLX) { vr = vals[2 * j]; vi = vals[2 * j + 1]; } else vr = vals[j];
        if (vr == 0.0 && vi == 0.0) continue; // DMV:110
        const uint64_t beta = reinterpret_cast<uint64_t const *>(base + segs.key_off[sg])[j];
        lsk_part_ctx const *pc = parts + segs.part[sg];
        lsk_index ix = any; // kind, binom, dir_sites, dir_weight: common; the rest per partition
        ix.shift = pc->ix.shift; ix.count = pc->ix.count; ix.reps = pc->ix.reps; ix.table = pc->ix.table; ix.dir = pc->ix.dir; ix.kind = pc->ix.kind;
        const int64_t idx = ix.kind == LSK_INDEX_IDENTITY ? (int64_t)beta : (ix.dir ? rankdir_index(ix, beta, s_db) : search_index(ix, beta));
        if (idx < 0) { atomicExch(err, 1); continue; }
        double const *norms = pc->norms;
        if (norms) { const double nb = norms[idx]; vr *= nb; vi *= nb; }
        double *y = reinterpret_cast<double *>(segs.y[sg]);
        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
        else atomic_add_f64(y + idx, vr);
    }
}
extern "C" int lsk_scatter_parts(lsk_part_ctx const *d_parts, lsk_index any, int cplx, lsk_segs const *segs, void const *base, int *d_err, void *stream) {
    if (segs->n < 1 || segs->n > LSK_MAX_SEGS) { snprintf(g_err, sizeof(g_err), "lsk_scatter_parts: %d segments", segs->n); return -1; }
    if (any.kind == LSK_INDEX_COMBINADIC) { snprintf(g_err, sizeof(g_err), "lsk_scatter_parts: SEARCH/IDENTITY index only"); return -1; }
    const int64_t n = segs->start[segs->n];
    if (n <= 0) return 0;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    dim3 g((unsigned)(nb < ((int64_t)1 << 30) ? nb : ((int64_t)1 << 30))), b(kBlock);
    const size_t dyn = any.dir ? sizeof(uint64_t) * (size_t)any.dir_sites * (size_t)(any.dir_weight + 1) : 0;
    if (cplx) hipLaunchKernelGGL(k_scatter_parts<true>, g, b, dyn, (hipStream_t)stream, d_parts, any, *segs, (char const *)base, d_err, 64);
    else hipLaunchKernelGGL(k_scatter_parts<false>, g, b, dyn, (hipStream_t)stream, d_parts, any, *segs, (char const *)base, d_err, 64);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Packets in SORTED STREAMS: consumers without atomics (unprojected fixed-weight bases, exchange operators).
//
// What bounds every consumer above is one fabric atomic per packet (~40 G/s: chain_28 x 8 partitions 13 of 22.6 ms).  But the
// packets of an exchange pair are not random: for a fixed pair (i, j) and a fixed pattern of alpha on it (01 or 10) the map
// alpha -> beta = alpha ^ x adds a CONSTANT, so it is monotone; the rows of a producer ascend, the states of a destination
// ascend, hence the destination indices of the packets of one STREAM = (pair, pattern) ascend along the producer's rows.
// A producer that writes every (destination, stream) as its own run of the send segment -- in row order -- hands the consumer
// 2 n_groups SORTED runs per source.  The consumer (k_window) then owns a WINDOW of W consecutive rows of y: it finds the
// sub-run of every stream that falls into the window by binary search on the keys, streams those packets (coalesced 12-byte
// reads) into an LDS copy of the window (ds_add_f64) and adds the window to y once: no global atomics, no fabric request per
// packet, and the packet order inside y's window no longer matters.
//
// Producer (k_tile_st): k_tile_wv's wave rings, but a wave walks a TILE of tile_rows rows (64 at a time, the ring carried over)
// and keeps one cursor per CLASS = (destination, stream) in LDS, initialised from the plan's table ttab[tile][class] = absolute
// position of the tile's first packet of that class inside the destination's segment (count pass + host scan, like wtab).  The
// rank of a packet among the packets of its class inside a chunk of 64 comes from one ballot per class BIT (9 ballots for
// 8 destinations x 56 streams) instead of one pass per destination; ring order = (group, lane) order, so the packets of one
// class leave in row order: every stream is EXACTLY sorted.  Own-partition packets take the same way (no atomics here either).
// ---------------------------------------------------------------------------------------------
constexpr int kStRing = 256;
constexpr int kStMaxClasses = 1024; // LDS: 4 waves x classes x 4 bytes of cursors
#define LSK_WAVE_SYNC()                                              \
    do {                                                             \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
        __builtin_amdgcn_wave_barrier();                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
    } while (0)
template <bool CPLX, bool REAL, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_tile_st(int n_groups, lsk_group const *__restrict__ groups,
                                                    lsk_term const *__restrict__ off, lsk_gdir gd,
                                                    uint64_t const *__restrict__ g_binom, Owner owner, int S, int cbits,
                                                    int tile_rows, int64_t row0, int64_t row1, int64_t n_tiles,
                                                    uint64_t const *__restrict__ reps, double const *__restrict__ x,
                                                    uint32_t *__restrict__ ttab, lsk_round_layout const *__restrict__ layout,
                                                    char *send, int *err, int xcd_chunk) {
    constexpr int kWaves = kBlock / 64;
    constexpr int kCap = kWaves * kStRing;
    __shared__ uint64_t s_beta[kCap];
    __shared__ double s_val[COUNT ? 1 : kCap * (CPLX ? 2 : 1)];
    __shared__ uint8_t s_sid[kCap];
    // global (colex) rank of beta when it is one binomial away from alpha's: an exchange on ADJACENT sites (lo, lo + 1) moves the
    // (k + 1)-th set bit by one place, rank(beta) = rank(alpha) +- C(lo, k), k = set bits of alpha below lo -- the 14-step rank
    // sum of the directory look-up then runs once per row instead of once per packet (kNoRank: the full sum, e.g. the bond
    // that closes a ring; bases with >= 2^32 states always take it)
    __shared__ uint32_t s_rank[COUNT ? 1 : kCap];
    extern __shared__ uint64_t s_dyn[]; // [binomials of the directory][key offsets P][value offsets P][cursors: waves x classes u32]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rb = wave * kStRing;
    const int P = (int)owner.P, C = P * S;
    const int ndb = COUNT ? 0 : gd.sites * (gd.weight + 1);
    uint64_t *s_db = s_dyn;
    int64_t *s_koff = reinterpret_cast<int64_t *>(s_dyn + ndb);
    int64_t *s_voff = s_koff + (COUNT ? 0 : P);
    uint32_t *s_cur = reinterpret_cast<uint32_t *>(s_voff + (COUNT ? 0 : P)) + wave * C;
    if (!COUNT) {
        gdir_load(gd, g_binom, s_db);
        for (int d = tid; d < P; d += kBlock) { s_koff[d] = layout->beta_off[d]; s_voff[d] = layout->val_off[d]; }
        __syncthreads();
    }
    const int64_t n_work = (n_tiles + kWaves - 1) / kWaves; // a block takes kWaves consecutive tiles, one per wave
    for (int64_t wb = blockIdx.x; wb < n_work; wb += gridDim.x) {
        // (consecutive tiles on ONE XCD: they append to the same lines of every stream, which then fill up inside one L2)
        const int64_t tile = pull_tile_of_block(wb, n_work, gridDim.x >= n_work ? xcd_chunk : 0) * kWaves + wave;
        if (tile >= n_tiles) continue; // wave-uniform: nothing below synchronises the block
        const int64_t t0 = row0 + tile * tile_rows;
        const int64_t t1 = t0 + tile_rows < row1 ? t0 + tile_rows : row1;
        uint32_t *trow = ttab + (size_t)tile * C;
        for (int c = lane; c < C; c += 64) s_cur[c] = COUNT ? 0u : trow[c];
        LSK_WAVE_SYNC();
        int head = 0, cnt = 0; // wave-uniform: the ring holds [head, head + cnt) mod kStRing
        auto chunk = [&](int m) {
            const bool live = lane < m;
            const int e = rb + ((head + lane) & (kStRing - 1));
            const uint64_t beta = live ? s_beta[e] : 0;
            const int dest = live ? owner_of(beta, owner) : 0;
            const uint32_t cls = (uint32_t)dest * (uint32_t)S + (live ? (uint32_t)s_sid[e] : 0u);
            // rank among the packets of the same class in this chunk: lanes that agree with me on every class bit
            unsigned long long same = __ballot(live);
            for (int b = 0; b < cbits; ++b) {
                const bool bit = (cls >> b) & 1u;
                const unsigned long long bb = __ballot(live && bit);
                same &= bit ? bb : ~bb;
            }
            const uint32_t rank = (uint32_t)__popcll(same & ((1ULL << lane) - 1));
            const uint32_t n_same = (uint32_t)__popcll(same);
            const uint32_t base = live ? s_cur[cls] : 0u;
            LSK_WAVE_SYNC(); // every lane has read its cursor before the last lane of a class moves it
            if (live && rank + 1 == n_same) s_cur[cls] = base + n_same;
            if (!COUNT && live) {
                double vr, vi = 0.0;
                if (CPLX) { vr = s_val[2 * e]; vi = s_val[2 * e + 1]; } else vr = s_val[e];
                const uint32_t rk = s_rank[e];
                int64_t idx = rk != kNoRank ? gdir_index_of_rank(gd, (uint64_t)rk, dest) : gdir_index(gd, beta, dest, s_db);
                if (idx < 0) { // not a basis state (DMV:115-118): the flag halts the matvec; the slot the count pass reserved is still
                    atomicExch(err, 1); // filled -- (index 0, value 0) -- so that no consumer meets a stale key
                    idx = 0; vr = 0.0; vi = 0.0;
                }
                const size_t pos = (size_t)base + rank;
                reinterpret_cast<uint32_t *>(send + s_koff[dest])[pos] = (uint32_t)idx;
                double *pv = reinterpret_cast<double *>(send + s_voff[dest]);
                if (CPLX) { pv[2 * pos] = vr; pv[2 * pos + 1] = vi; } else pv[pos] = vr;
            }
            LSK_WAVE_SYNC();
        };
        for (int64_t r0 = t0; r0 < t1; r0 += 64) {
            const int64_t i = r0 + lane;
            const bool valid = i < t1;
            uint64_t a = 0;
            double xr = 0.0, xi = 0.0;
            if (valid) {
                a = reps[i];
                if (COUNT) xr = 1.0; // the packet set must not depend on x (exact send counts)
                else if (CPLX) { xr = x[2 * i]; xi = x[2 * i + 1]; }
                else xr = x[i];
            }
            uint64_t ga = 0; // colex rank of alpha (states of another weight never reach the look-up: their rows have no packets)
            const bool narrow_ranks = !COUNT && gd.n_ranks <= 0xffffffffLL;
            if (narrow_ranks && valid) {
                const int kc = gd.weight + 1;
                uint64_t t = a;
                int k = 1;
                while (t && k < kc) { ga += s_db[(__ffsll((unsigned long long)t) - 1) * kc + k]; ++k; t &= t - 1; }
            }
            for (int g0 = 0; g0 < n_groups; g0 += kTwGroups) {
                const int g1 = min(g0 + kTwGroups, n_groups);
                for (int g = g0; g < g1; ++g) { // stage A: append (beta, value, stream)
                    lsk_group const G = groups[g];
                    double cr = 0.0, ci = 0.0;
                    if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                    // (every group is an exchange pair: a packet exists iff alpha is anti-aligned on it, whatever its amplitude)
                    const bool act = valid && __popcll(a & G.x) == 1;
                    const unsigned long long ball = __ballot(act);
                    if (act) {
                        const int slot = rb + ((head + cnt + __popcll(ball & ((1ULL << lane) - 1))) & (kStRing - 1));
                        const int up = (int)((a >> (__ffsll((unsigned long long)G.x) - 1)) & 1ULL); // the lower site's bit moves up
                        s_beta[slot] = a ^ G.x;
                        s_sid[slot] = (uint8_t)(2 * g + up);
                        if (!COUNT) {
                            if (CPLX) { s_val[2 * slot] = cr * xr - ci * xi; s_val[2 * slot + 1] = cr * xi + ci * xr; }
                            else s_val[slot] = cr * xr;
                            uint32_t rk = kNoRank;
                            if (narrow_ranks && G.adj >= 0) {
                                const uint64_t c = s_db[G.adj * (gd.weight + 1) + __popcll(a & ((1ULL << G.adj) - 1))];
                                rk = (uint32_t)(up ? ga + c : ga - c);
                            }
                            s_rank[slot] = rk;
                        }
                    }
                    cnt += __popcll(ball);
                }
                LSK_WAVE_SYNC();
                while (cnt >= 64) {
                    chunk(64);
                    head = (head + 64) & (kStRing - 1);
                    cnt -= 64;
                }
            }
        }
        if (cnt > 0) chunk(cnt);
        if (COUNT) for (int c = lane; c < C; c += 64) trow[c] = s_cur[c];
        LSK_WAVE_SYNC();
    }
}

extern "C" int lsk_tile_st_max_classes(void) { return kStMaxClasses; }
// rows [row0, row1) of one partition, tile t = rows [row0 + t tile_rows, ...).  count_only: d_ttab[tile][P * S] <- packets of every
// (tile, class = destination * S + stream), stream = 2 * group + (bit of alpha at the pair's lower site); otherwise d_ttab holds
// the position of the tile's first packet of every class inside the destination's segment of *d_layout, and the packets --
// (u32 index at the destination, value), the own partition's included -- are written to d_send.
extern "C" int lsk_tile_st(lsk_operator op, lsk_gdir gd, uint64_t const *d_binom, int cplx, int count_only, int P, int S,
                           int tile_rows, int64_t row0, int64_t row1, uint64_t const *reps, void const *x, uint32_t *d_ttab,
                           lsk_round_layout const *d_layout, void *d_send, int *d_err, void *stream) {
    if (row1 <= row0 || op.n_groups == 0) return 0;
    if (S != 2 * op.n_groups || S > 256 || P < 1 || P * S > kStMaxClasses || tile_rows < 64 || (tile_rows & 63) || !d_ttab ||
        (!count_only && (!gd.entries || gd.P != P || !d_binom))) {
        snprintf(g_err, sizeof(g_err), "lsk_tile_st: bad arguments (P = %d, S = %d, tile_rows = %d)", P, S, tile_rows);
        return -1;
    }
    const int C = P * S;
    int cbits = 0;
    while ((1 << cbits) < C) ++cbits;
    Owner ow = make_owner(P);
    const int64_t n_tiles = (row1 - row0 + tile_rows - 1) / tile_rows;
    const int64_t n_work = (n_tiles + kBlock / 64 - 1) / (kBlock / 64);
    dim3 g(1), b(kBlock);
    hipStream_t s = (hipStream_t)stream;
    const size_t dyn = sizeof(uint32_t) * (size_t)(kBlock / 64) * (size_t)C +
                       (count_only ? 0 : sizeof(uint64_t) * ((size_t)gd.sites * (size_t)(gd.weight + 1) + 2 * (size_t)P));
#define LSK_ST_ARGS op.n_groups, op.groups, op.off, gd, d_binom, ow, S, cbits, tile_rows, row0, row1, n_tiles, reps, (double const *)x, \
        d_ttab, d_layout, (char *)d_send, d_err, 64
#define LSK_ST_ONE(CPLX, REAL)                                                                                                    \
    do {                                                                                                                          \
        if (count_only) { g.x = tile_grid(k_tile_st<CPLX, REAL, true>, n_work); hipLaunchKernelGGL((k_tile_st<CPLX, REAL, true>), g, b, dyn, s, LSK_ST_ARGS); } \
        else { g.x = tile_grid(k_tile_st<CPLX, REAL, false>, n_work); hipLaunchKernelGGL((k_tile_st<CPLX, REAL, false>), g, b, dyn, s, LSK_ST_ARGS); } \
    } while (0)
    if (cplx) { if (op.is_real) LSK_ST_ONE(true, true); else LSK_ST_ONE(true, false); }
    else LSK_ST_ONE(false, true); // f64 vectors: real operators only (the plan refuses the rest)
#undef LSK_ST_ONE
#undef LSK_ST_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}

// Consumer of the sorted streams.  Block -> (destination partition, wpb consecutive windows of W rows of its y).  n_src source
// segments per destination, S streams each: soff[s] .. soff[s + 1] = packets of stream s inside the segment, keys ascending.
constexpr int kWinRows = 2048;    // doubles of one window's accumulator (c128: 1024 rows)
constexpr int kWinStreams = 512;  // run bounds kept in LDS per pass over the streams
constexpr int kWinRuns = 4;       // runs a wave has in flight
__device__ __forceinline__ uint32_t lower_bound_u32(uint32_t const *__restrict__ k, uint32_t lo, uint32_t hi, uint32_t v) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (k[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first position >= lo whose key is >= v (the window's end is a few dozen packets on: gallop, then search)
__device__ __forceinline__ uint32_t gallop_u32(uint32_t const *__restrict__ k, uint32_t lo, uint32_t end, uint32_t v) {
    uint32_t step = 64;
    while ((uint64_t)lo + step <= (uint64_t)end && k[lo + step - 1] < v) { lo += step; step <<= 1; }
    const uint32_t hi = (uint64_t)lo + step < (uint64_t)end ? lo + step : end;
    return lower_bound_u32(k, lo, hi, v);
}
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_window(lsk_wdests dests, lsk_wsrc const *__restrict__ srcs, int n_src, int S, int wpb) {
    constexpr int W = CPLX ? kWinRows / 2 : kWinRows;
    __shared__ double s_acc[kWinRows];
    __shared__ uint32_t s_lo[kWinStreams];
    __shared__ uint16_t s_len[kWinStreams]; // (the keys of a stream are distinct: a window holds <= W of them)
    __shared__ uint32_t const *s_keys[LSK_MAX_SEGS];
    __shared__ double const *s_vals[LSK_MAX_SEGS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int d = 0;
    while (d + 1 < dests.n && (int64_t)blockIdx.x >= dests.first_block[d + 1]) ++d; // block-uniform
    const int64_t n = dests.count[d];
    double *__restrict__ y = reinterpret_cast<double *>(dests.y[d]);
    const int64_t wb = (int64_t)blockIdx.x - dests.first_block[d];
    const int T = n_src * S;
    lsk_wsrc const *__restrict__ segs = srcs + (size_t)d * n_src;
    for (int q = tid; q < n_src; q += kBlock) { s_keys[q] = segs[q].keys; s_vals[q] = segs[q].vals; }
    for (int win = 0; win < wpb; ++win) {
        const int64_t w0 = (wb * wpb + win) * W;
        if (w0 >= n) break;
        const int64_t w1 = w0 + W < n ? w0 + W : n;
        for (int i = tid; i < kWinRows; i += kBlock) s_acc[i] = 0.0;
        const bool carry = win > 0 && T <= kWinStreams; // the end of the previous window's run is the start of this one's
        for (int t0 = 0; t0 < T; t0 += kWinStreams) {
            const int tn = T - t0 < kWinStreams ? T - t0 : kWinStreams;
            for (int t = tid; t < tn; t += kBlock) {
                const int q = (t0 + t) / S, s = (t0 + t) - q * S;
                uint32_t const *__restrict__ keys = segs[q].keys;
                uint32_t const *__restrict__ soff = segs[q].soff;
                const uint32_t e = soff[s + 1];
                const uint32_t lo = carry ? s_lo[t] + s_len[t] : lower_bound_u32(keys, soff[s], e, (uint32_t)w0);
                uint32_t hi = w1 >= n ? e : gallop_u32(keys, lo, e, (uint32_t)w1);
                if (hi - lo > (uint32_t)W) hi = lo + (uint32_t)W; // (only after a failed directory look-up: the flag is up anyway)
                s_lo[t] = lo;
                s_len[t] = (uint16_t)(hi - lo);
            }
            __syncthreads();
            // a wave takes the runs t = wave, wave + 4, ..., kWinRuns at a time: all their loads are issued before the first add
            for (int t = wave; t < tn; t += 4 * kWinRuns) {
                uint32_t const *kp[kWinRuns];
                double const *vp[kWinRuns];
                uint32_t len[kWinRuns], longest = 0;
#pragma unroll
                for (int u = 0; u < kWinRuns; ++u) {
                    const int tu = t + 4 * u;
                    const bool has = tu < tn;
                    const int q = has ? (t0 + tu) / S : 0;
                    const uint32_t lo = has ? s_lo[tu] : 0u;
                    len[u] = has ? (uint32_t)s_len[tu] : 0u;
                    kp[u] = s_keys[q] + lo;
                    vp[u] = s_vals[q] + (size_t)lo * (CPLX ? 2 : 1);
                    longest = len[u] > longest ? len[u] : longest;
                }
                for (uint32_t it = (uint32_t)lane; it < longest + (uint32_t)lane; it += 64) { // (wave-uniform trip count)
                    uint32_t key[kWinRuns];
                    double vr[kWinRuns], vi[kWinRuns];
#pragma unroll
                    for (int u = 0; u < kWinRuns; ++u) {
                        key[u] = 0xffffffffu; vr[u] = 0.0; vi[u] = 0.0;
                        if (it < len[u]) {
                            key[u] = kp[u][it];
                            if (CPLX) { vr[u] = vp[u][2 * (size_t)it]; vi[u] = vp[u][2 * (size_t)it + 1]; } else vr[u] = vp[u][it];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kWinRuns; ++u) {
                        const uint32_t o = key[u] - (uint32_t)w0; // (a key outside the window -- only after a failed look-up -- is dropped)
                        if (it < len[u] && o < (uint32_t)W) {
                            if (CPLX) { atomicAdd(&s_acc[2 * o], vr[u]); atomicAdd(&s_acc[2 * o + 1], vi[u]); } else atomicAdd(&s_acc[o], vr[u]);
                        }
                    }
                }
            }
            __syncthreads();
        }
        const int64_t m = (w1 - w0) * (CPLX ? 2 : 1);
        double *__restrict__ yw = y + w0 * (CPLX ? 2 : 1);
        for (int64_t i = tid; i < m; i += kBlock) yw[i] += s_acc[i];
        __syncthreads();
    }
}
extern "C" int lsk_window_rows(int cplx) { return cplx ? kWinRows / 2 : kWinRows; }
// y[d][key] += value for every packet of every stream of every source segment: dests (by value) names the destination vectors
// and the first block of each (ceil(ceil(count / rows) / wpb) blocks per destination); d_srcs is [dests.n][n_src].
extern "C" int lsk_window(int cplx, lsk_wdests const *dests, lsk_wsrc const *d_srcs, int n_src, int S, int wpb, void *stream) {
    if (dests->n < 1 || dests->n > LSK_MAX_SEGS || n_src < 1 || S < 1 || wpb < 1) { snprintf(g_err, sizeof(g_err), "lsk_window: bad arguments"); return -1; }
    const int64_t nb = dests->first_block[dests->n];
    if (nb <= 0) return 0;
    if (nb > ((int64_t)1 << 31) - 1) { snprintf(g_err, sizeof(g_err), "lsk_window: too many windows"); return -1; }
    dim3 g((unsigned)nb), b(kBlock);
    if (cplx) hipLaunchKernelGGL(k_window<true>, g, b, 0, (hipStream_t)stream, *dests, d_srcs, n_src, S, wpb);
    else hipLaunchKernelGGL(k_window<false>, g, b, 0, (hipStream_t)stream, *dests, d_srcs, n_src, S, wpb);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// out[i] = src[perm[i]]: the hashed -> block permutation of the replicated-x exchange (P ascending streams interleaved).
// Two outputs per thread so that f64 results leave as 16-byte stores; perm is read with 8- / 16-byte loads.
// ---------------------------------------------------------------------------------------------
template <typename I, typename T>
__global__ __launch_bounds__(kBlock) void k_gather_perm(int64_t n, I const *__restrict__ perm, T const *__restrict__ src,
                                                        T *__restrict__ out) {
    const int64_t pairs = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < pairs; k += (int64_t)gridDim.x * kBlock) {
        const I p0 = __builtin_nontemporal_load(perm + 2 * k), p1 = __builtin_nontemporal_load(perm + 2 * k + 1);
        const T a = src[p0], b = src[p1];
        out[2 * k] = a;
        out[2 * k + 1] = b;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = src[perm[n - 1]];
}
__global__ __launch_bounds__(kBlock) void k_iota(int64_t n, int64_t base, int64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = base + i;
}
extern "C" int lsk_iota_i64(int64_t n, int64_t base, int64_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, base, out);
    LSK_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_narrow_i32(int64_t n, int64_t const *__restrict__ in, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (int32_t)in[i];
}
extern "C" int lsk_narrow_i32(int64_t n, int64_t const *in, int32_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_narrow_i32, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, in, out);
    LSK_LAUNCH_CHECK();
    return 0;
}
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_axpy1(int64_t n, double const *__restrict__ a, double *__restrict__ y) {
    const int64_t m = CPLX ? 2 * n : n;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) y[i] += a[i];
}
// y += a (the accumulate semantics of operators without diagonal terms, DMV:1062-1063, in the replicated-x driver)
extern "C" int lsk_add_into(int cplx, int64_t n, void const *a, void *y, void *stream) {
    if (n == 0) return 0;
    if (cplx) hipLaunchKernelGGL(k_axpy1<true>, dim3(grid_for(2 * n)), dim3(kBlock), 0, (hipStream_t)stream, n, (double const *)a, (double *)y);
    else hipLaunchKernelGGL(k_axpy1<false>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, (double const *)a, (double *)y);
    LSK_LAUNCH_CHECK();
    return 0;
}
extern "C" int lsk_gather_perm(int64_t n, void const *perm, int perm_is_64, int elt_size, void const *src, void *out, void *stream) {
    if (n == 0) return 0;
    const int64_t blocks = (n / 2 + kBlock - 1) / kBlock;
    dim3 g((unsigned)(blocks < 1 ? 1 : (blocks > (1 << 20) ? (1 << 20) : blocks))), b(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (elt_size == 8) {
        if (perm_is_64) hipLaunchKernelGGL((k_gather_perm<int64_t, double>), g, b, 0, s, n, (int64_t const *)perm, (double const *)src, (double *)out);
        else hipLaunchKernelGGL((k_gather_perm<int32_t, double>), g, b, 0, s, n, (int32_t const *)perm, (double const *)src, (double *)out);
    } else if (elt_size == 16) {
        if (perm_is_64) hipLaunchKernelGGL((k_gather_perm<int64_t, double2>), g, b, 0, s, n, (int64_t const *)perm, (double2 const *)src, (double2 *)out);
        else hipLaunchKernelGGL((k_gather_perm<int32_t, double2>), g, b, 0, s, n, (int32_t const *)perm, (double2 const *)src, (double2 *)out);
    } else { snprintf(g_err, sizeof(g_err), "lsk_gather_perm: element size %d", elt_size); return -1; }
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// plan-time helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_norms(lsk_basis bs, lsk_group_elem const *__restrict__ elems, int64_t n,
                                                  uint64_t const *__restrict__ reps, double *__restrict__ norms) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t rep; double chr, chi, stab;
        state_info(bs, elems, reps[i], rep, chr, chi, stab);
        double n2 = stab * bs.inv_order;
        norms[i] = n2 > 1e-12 ? sqrt(n2) : 0.0;
    }
}
extern "C" int lsk_norms(lsk_basis bs, int64_t n, uint64_t const *reps, double *norms, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_norms, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, bs, bs.elems, n, reps, norms);
    LSK_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_check_combinadic(uint64_t const *__restrict__ g_binom, int64_t n,
                                                             uint64_t const *__restrict__ reps, int *flag) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (rank_combinadic(reps[i], s_binom) != i) atomicExch(flag, 1);
}
extern "C" int lsk_check_combinadic(lsk_index ix, int hamming_weight, int64_t n, uint64_t const *reps, int *d_flag,
                                    void *stream) {
    (void)hamming_weight;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_check_combinadic, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, ix.binom, n, reps, d_flag);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---- rank directory (lsk_rankdir): build + self-check -----------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_rankdir_mark(int64_t n, uint64_t const *__restrict__ reps, int sites, int weight,
                                                         uint64_t const *__restrict__ g_binom, lsk_rankdir *__restrict__ dir, int *flag) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t s = reps[i];
        if (__popcll(s) != weight || (sites < 64 && (s >> sites) != 0)) { atomicExch(flag, 1); continue; }
        const uint64_t g = (uint64_t)rank_combinadic(s, s_binom);
        atomicOr((unsigned long long *)&dir[g >> 6].bits, 1ULL << (g & 63));
    }
}
struct ScanDirPopcIn {
    lsk_rankdir const *dir;
    __device__ int64_t operator()(int64_t i) const { return (int64_t)__popcll(dir[i].bits); }
};
__global__ __launch_bounds__(kBlock) void k_rankdir_prefix(int64_t entries, int64_t const *__restrict__ pre, lsk_rankdir *__restrict__ dir) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < entries; i += (int64_t)gridDim.x * kBlock) {
        dir[i].prefix = (uint32_t)pre[i];
        dir[i].pad = 0;
    }
}
__global__ __launch_bounds__(kBlock) void k_rankdir_check(lsk_index ix, int64_t n, uint64_t const *__restrict__ reps, int *flag) {
    extern __shared__ uint64_t s_db[];
    rankdir_load(ix, s_db);
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (rankdir_index(ix, reps[i], s_db) != i) atomicExch(flag, 1);
}

// table[b] = first i with (reps[i] >> shift) >= b.  Element i owns the buckets (bucket(i - 1), bucket(i)]; the representatives of
// a projected basis are far from uniform over the top bits (92 % of the buckets of chain_32_symm are empty, in runs of millions),
// so a run longer than 64 buckets is filled by the WHOLE WAVE of its owner instead of one lane (chain_40_symm: 331 -> 23 ms per table).
__global__ __launch_bounds__(kBlock) void k_build_table(int64_t n, uint64_t const *__restrict__ reps, int shift,
                                                        int64_t nbuckets, uint32_t *__restrict__ table) {
    const int lane = threadIdx.x & 63;
    const int64_t total = n + 1, rounded = (total + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < rounded; i += (int64_t)gridDim.x * kBlock) {
        int64_t lo = 0, hi = -1; // (an empty range for the lanes past the end: they still take part in the wave's long runs)
        if (i < total) {
            lo = (i == 0) ? 0 : (int64_t)(reps[i - 1] >> shift) + 1;
            hi = (i == n) ? nbuckets : (int64_t)(reps[i] >> shift);
        }
        const bool is_long = hi - lo >= 64;
        if (!is_long) for (int64_t b = lo; b <= hi; ++b) table[b] = (uint32_t)i;
        unsigned long long m = __ballot(is_long);
        while (m) { // wave-uniform: every lane helps to fill the long runs of the wave, one after the other
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t rlo = (int64_t)readlane_t<uint64_t>((uint64_t)lo, l), rhi = (int64_t)readlane_t<uint64_t>((uint64_t)hi, l);
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)i, l);
            for (int64_t b = rlo + lane; b <= rhi; b += 64) table[b] = v;
        }
    }
}
extern "C" int lsk_build_table(int64_t n, uint64_t const *reps, int shift, int64_t nbuckets, uint32_t *table,
                               void *stream) {
    hipLaunchKernelGGL(k_build_table, dim3(grid_for(n + 1)), dim3(kBlock), 0, (hipStream_t)stream, n, reps, shift, nbuckets, table);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// batched externs on device pointers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_state_info(lsk_basis bs, lsk_group_elem const *__restrict__ elems,
                                                       int64_t n, uint64_t const *__restrict__ alphas,
                                                       uint64_t *__restrict__ betas, double *__restrict__ chars,
                                                       double *__restrict__ norms) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t rep; double chr, chi, stab;
        uint64_t a = alphas[i];
        if (bs.proj == LSK_PROJ_NONE) { rep = a; chr = 1.0; chi = 0.0; stab = 1.0 / bs.inv_order; }
        else state_info(bs, elems, a, rep, chr, chi, stab);
        double n2 = stab * bs.inv_order;
        betas[i] = rep;
        chars[2 * i] = chr;
        chars[2 * i + 1] = chi;
        norms[i] = n2 > 1e-12 ? sqrt(n2) : 0.0;
    }
}
extern "C" int lsk_state_info(lsk_basis bs, int64_t n, uint64_t const *alphas, uint64_t *betas, double *characters,
                              double *norms, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_state_info, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, bs, bs.elems, n, alphas, betas, characters, norms);
    LSK_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_state_index(lsk_index ix, int64_t n, uint64_t const *__restrict__ spins,
                                                        int64_t *__restrict__ indices) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    if (ix.kind == LSK_INDEX_COMBINADIC) load_binom(s_binom, ix.binom);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t s = spins[i];
        int64_t idx;
        if (ix.kind == LSK_INDEX_IDENTITY) idx = (int64_t)s < ix.count ? (int64_t)s : -1;
        else if (ix.kind == LSK_INDEX_COMBINADIC) {
            idx = rank_combinadic(s, s_binom);
            // membership: the basis is the first `count` states of one popcount class
            if (idx >= ix.count || __popcll(s) != __popcll(ix.reps[0])) idx = -1;
        } else idx = search_index(ix, s);
        indices[i] = idx;
    }
}
extern "C" int lsk_state_index(lsk_index ix, int64_t n, uint64_t const *spins, int64_t *indices, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_state_index, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, ix, n, spins, indices);
    LSK_LAUNCH_CHECK();
    return 0;
}
// Plan time, replicated-x exchange of unprojected bases: which blocks of 2^shift rows of the GLOBAL vector do the rows
// alphas[0, n) read?  Every non-zero off-diagonal group of a row -> partner state -> index in the global basis -> one bit.
// (The rows' own neighbourhood -- the LDS windows of the staged kernels -- is added by the host.)
__global__ __launch_bounds__(kBlock) void k_reach_blocks(int n_groups, lsk_group const *__restrict__ groups, lsk_term const *__restrict__ off,
                                                         lsk_index ix, int64_t n, uint64_t const *__restrict__ alphas, int shift,
                                                         uint32_t *__restrict__ bitmap) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    if (ix.kind == LSK_INDEX_COMBINADIC) load_binom(s_binom, ix.binom);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t a = alphas[i];
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            group_coeff<false>(G, off, a, cr, ci);
            if (cr == 0.0 && ci == 0.0) continue;
            const uint64_t s = a ^ G.x;
            int64_t idx;
            if (ix.kind == LSK_INDEX_IDENTITY) idx = (int64_t)s < ix.count ? (int64_t)s : -1;
            else if (ix.kind == LSK_INDEX_COMBINADIC) {
                idx = rank_combinadic(s, s_binom);
                if (idx >= ix.count || __popcll(s) != __popcll(ix.reps[0])) idx = -1;
            } else idx = search_index(ix, s);
            if (idx < 0) continue; // outside the basis: the matvec reports it (DMV:115-118)
            const int64_t b = idx >> shift;
            const uint32_t bit = 1u << (b & 31);
            if (!(bitmap[b >> 5] & bit)) atomicOr(bitmap + (b >> 5), bit); // plain read first: nearly every block is marked early
        }
    }
}
extern "C" int lsk_reach_blocks(lsk_operator op, lsk_index ix_global, int64_t n, uint64_t const *alphas, int shift, uint32_t *bitmap,
                                void *stream) {
    if (n == 0 || op.n_groups == 0) return 0;
    hipLaunchKernelGGL(k_reach_blocks, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, op.n_groups, op.groups, op.off, ix_global, n,
                       alphas, shift, bitmap);
    LSK_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_offdiag_counts(int n_groups, lsk_group const *__restrict__ groups,
                                                           lsk_term const *__restrict__ off, int64_t n,
                                                           uint64_t const *__restrict__ alphas,
                                                           int64_t *__restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t a = alphas[i];
        int c = 0;
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            group_coeff<false>(G, off, a, cr, ci);
            c += (cr != 0.0 || ci != 0.0);
        }
        counts[i] = c;
    }
}
extern "C" int lsk_offdiag_counts(lsk_operator op, int64_t n, uint64_t const *alphas, int64_t *counts, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_offdiag_counts, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, op.n_groups, op.groups, op.off, n, alphas, counts);
    LSK_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_offdiag_fill(int n_groups, lsk_group const *__restrict__ groups,
                                                         lsk_term const *__restrict__ off, int64_t n,
                                                         uint64_t const *__restrict__ alphas,
                                                         int64_t const *__restrict__ offsets,
                                                         uint64_t *__restrict__ betas, double *__restrict__ coeffs,
                                                         double const *__restrict__ xs) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t a = alphas[i];
        int64_t o = offsets[i];
        double xv = xs ? xs[i] : 1.0;
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            group_coeff<false>(G, off, a, cr, ci);
            if (cr != 0.0 || ci != 0.0) {
                betas[o] = a ^ G.x;
                coeffs[2 * o] = cr * xv;
                coeffs[2 * o + 1] = ci * xv;
                ++o;
            }
        }
    }
}
extern "C" int lsk_offdiag_fill(lsk_operator op, int64_t n, uint64_t const *alphas, int64_t const *offsets,
                                uint64_t *betas, double *coeffs, double const *xs, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_offdiag_fill, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, op.n_groups, op.groups, op.off, n, alphas, offsets, betas, coeffs, xs);
    LSK_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_diag_coeffs(int n_diag, lsk_term const *__restrict__ diag, int64_t n,
                                                        uint64_t const *__restrict__ alphas, double *__restrict__ ys,
                                                        double const *__restrict__ xs) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double dr, di;
        term_sum<true>(diag, 0, n_diag, alphas[i], dr, di);
        ys[i] = xs ? dr * xs[i] : dr;
    }
}
extern "C" int lsk_diag_coeffs(lsk_operator op, int64_t n, uint64_t const *alphas, double *ys, double const *xs,
                               void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_diag_coeffs, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, op.n_diag, op.diag, n, alphas, ys, xs);
    LSK_LAUNCH_CHECK();
    return 0;
}

// Exclusive prefix sum of int64 values (the layout converters, the enumeration, ls_chpl_operator_apply_off_diag): three small
// kernels per level -- per-block scan of kScanPer elements + block totals, the totals scanned recursively, the offsets added
// back.  (Hand-written: hipCUB's DeviceScan brought 225 trampoline kernels into the library for these three call sites.)
constexpr int kScanItems = 8;
constexpr int kScanPer = kBlock * kScanItems;
struct ScanArrayIn {
    int64_t const *v;
    __device__ __forceinline__ int64_t operator()(int64_t i) const { return v[i]; }
};
struct ScanMaskIn { // 1 where masks[i] == p
    uint8_t const *masks;
    uint8_t p;
    __device__ __forceinline__ int64_t operator()(int64_t i) const { return masks[i] == p ? 1 : 0; }
};
template <typename In>
__global__ __launch_bounds__(kBlock) void k_scan_block(In in, int64_t n, int64_t *__restrict__ out, int64_t *__restrict__ totals) {
    __shared__ int64_t s_wave[kBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanPer + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems], sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { v[k] = base + k < n ? in(base + k) : 0; sum += v[k]; }
    // inclusive scan of the per-thread sums inside the wave, then across the four waves
    int64_t inc = sum;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    int64_t wave_off = 0, total = 0;
    for (int w = 0; w < kBlock / 64; ++w) { if (w < (int)(threadIdx.x >> 6)) wave_off += s_wave[w]; total += s_wave[w]; }
    int64_t run = wave_off + inc - sum;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}
__global__ __launch_bounds__(kBlock) void k_scan_add(int64_t n, int64_t *__restrict__ out, int64_t const *__restrict__ offsets) {
    const int64_t off = offsets[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanPer;
    for (int k = threadIdx.x; k < kScanPer; k += kBlock)
        if (base + k < n) out[base + k] += off;
}
// scratch for n elements: totals of every level, one after the other
static int64_t scan_scratch_elems(int64_t n) {
    int64_t e = 0;
    while (n > 1) { n = (n + kScanPer - 1) / kScanPer; e += n; if (n == 1) break; }
    return e > 0 ? e : 1;
}
template <typename In>
static int scan_level(In in, int64_t n, int64_t *out, int64_t *scratch, hipStream_t s) {
    const int64_t blocks = (n + kScanPer - 1) / kScanPer;
    hipLaunchKernelGGL((k_scan_block<In>), dim3((unsigned)blocks), dim3(kBlock), 0, s, in, n, out, scratch);
    LSK_LAUNCH_CHECK();
    if (blocks > 1) {
        ScanArrayIn t{scratch};
        if (scan_level<ScanArrayIn>(t, blocks, scratch, scratch + blocks, s) != 0) return -1; // in place: totals -> their exclusive sums
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)blocks), dim3(kBlock), 0, s, n, out, scratch);
        LSK_LAUNCH_CHECK();
    }
    return 0;
}
template <typename In>
static int exclusive_scan(In in, int64_t n, int64_t *out, int64_t *scratch, hipStream_t s) {
    if (n <= 0) return 0;
    if ((n + kScanPer - 1) / kScanPer > 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "scan too large"); return -1; }
    return scan_level<In>(in, n, out, scratch, s);
}
static int exclusive_scan_i64(int64_t n, int64_t const *in, int64_t *out, hipStream_t s) {
    if (n == 0) return 0;
    int64_t *scratch = nullptr;
    LSK_CHECK(hipMalloc((void **)&scratch, 8 * (size_t)scan_scratch_elems(n)));
    ScanArrayIn src{in};
    const int rc = exclusive_scan<ScanArrayIn>(src, n, out, scratch, s);
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (rc != 0) return -1;
    LSK_CHECK(e2);
    return 0;
}
extern "C" int lsk_exclusive_scan_i64(int64_t n, int64_t const *in, int64_t *out, void *stream) {
    return exclusive_scan_i64(n, in, out, (hipStream_t)stream);
}
extern "C" int lsk_rankdir_build(int64_t n, uint64_t const *reps, int sites, int weight, uint64_t const *d_binom, int64_t entries,
                                 lsk_rankdir *dir, int *d_flag, void *stream) {
    if (entries <= 0 || sites < 1 || sites > 64 || weight < 0 || weight >= LSK_BINOM_K) { snprintf(g_err, sizeof(g_err), "lsk_rankdir_build: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    LSK_CHECK(hipMemsetAsync(dir, 0, sizeof(lsk_rankdir) * (size_t)entries, s));
    if (n > 0) {
        hipLaunchKernelGGL(k_rankdir_mark, dim3(grid_for(n)), dim3(kBlock), 0, s, n, reps, sites, weight, d_binom, dir, d_flag);
        LSK_LAUNCH_CHECK();
    }
    int64_t *pre = nullptr, *scratch = nullptr;
    LSK_CHECK(hipMalloc((void **)&pre, 8 * (size_t)entries));
    if (hipMalloc((void **)&scratch, 8 * (size_t)scan_scratch_elems(entries)) != hipSuccess) { (void)hipFree(pre); snprintf(g_err, sizeof(g_err), "lsk_rankdir_build: out of memory"); return -1; }
    ScanDirPopcIn in{dir};
    int rc = exclusive_scan<ScanDirPopcIn>(in, entries, pre, scratch, s);
    if (rc == 0) {
        hipLaunchKernelGGL(k_rankdir_prefix, dim3(grid_for(entries)), dim3(kBlock), 0, s, entries, pre, dir);
        if (hipGetLastError() != hipSuccess) rc = -1;
    }
    if (rc == 0 && n > 0) {
        lsk_index ix;
        memset(&ix, 0, sizeof(ix));
        ix.kind = LSK_INDEX_SEARCH; ix.count = n; ix.reps = reps; ix.binom = d_binom; ix.dir = dir; ix.dir_sites = sites; ix.dir_weight = weight;
        hipLaunchKernelGGL(k_rankdir_check, dim3(grid_for(n)), dim3(kBlock), sizeof(uint64_t) * (size_t)sites * (size_t)(weight + 1), s, ix, n, reps, d_flag);
        if (hipGetLastError() != hipSuccess) rc = -1;
    }
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(pre);
    (void)hipFree(scratch);
    if (rc != 0) { snprintf(g_err, sizeof(g_err), "lsk_rankdir_build: launch failed"); return -1; }
    LSK_CHECK(e2);
    return 0;
}

// ---- all-destinations directory (lsk_gdir): every rank derives it alone -- the owner of a state is a hash of the state ----------
// thread = one word of 64 consecutive global ranks: unrank the first, Gosper-step through the rest, mark each in its owner's entry
__global__ __launch_bounds__(kBlock) void k_gdir_mark(lsk_gdir gd, Owner ow, uint64_t const *__restrict__ g_binom, int64_t words,
                                                      lsk_rankdir *__restrict__ entries) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < words; w += (int64_t)gridDim.x * kBlock) {
        const int64_t g0 = w << 6, g1 = g0 + 64 < gd.n_ranks ? g0 + 64 : gd.n_ranks;
        uint64_t s = unrank_combinadic(g0, gd.weight, s_binom);
        lsk_rankdir *row = entries + w * (int64_t)gd.P;
        for (int64_t g = g0; g < g1; ++g) {
            row[owner_of(s, ow)].bits |= 1ULL << (g - g0); // (this thread owns the whole row)
            s = next_fixed_hamming(s);
        }
    }
}
struct ScanGdirPopcIn {
    lsk_rankdir const *entries;
    int64_t P, d;
    __device__ int64_t operator()(int64_t w) const { return (int64_t)__popcll(entries[w * P + d].bits); }
};
__global__ __launch_bounds__(kBlock) void k_gdir_prefix(int64_t words, int64_t P, int64_t d, int64_t const *__restrict__ pre,
                                                        lsk_rankdir *__restrict__ entries) {
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < words; w += (int64_t)gridDim.x * kBlock) {
        entries[w * P + d].prefix = (uint32_t)pre[w];
        entries[w * P + d].pad = 0;
    }
}
extern "C" int lsk_gdir_build(lsk_gdir gd, lsk_rankdir *entries, uint64_t const *d_binom, void *stream) {
    if (gd.P < 1 || gd.P > LSK_MAX_PARTS || gd.sites < 1 || gd.sites > 64 || gd.weight < 0 || gd.weight >= LSK_BINOM_K - 1 || gd.n_ranks < 1) {
        snprintf(g_err, sizeof(g_err), "lsk_gdir_build: bad arguments"); return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t words = (gd.n_ranks + 63) >> 6;
    LSK_CHECK(hipMemsetAsync(entries, 0, sizeof(lsk_rankdir) * (size_t)words * (size_t)gd.P, s));
    hipLaunchKernelGGL(k_gdir_mark, dim3(grid_for(words)), dim3(kBlock), 0, s, gd, make_owner(gd.P), d_binom, words, entries);
    LSK_LAUNCH_CHECK();
    int64_t *pre = nullptr, *scratch = nullptr;
    LSK_CHECK(hipMalloc((void **)&pre, 8 * (size_t)words));
    if (hipMalloc((void **)&scratch, 8 * (size_t)scan_scratch_elems(words)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(pre); snprintf(g_err, sizeof(g_err), "lsk_gdir_build: out of memory"); return -1; }
    int rc = 0;
    for (int d = 0; d < gd.P && rc == 0; ++d) { // one exclusive scan of the popcounts per destination
        ScanGdirPopcIn in{entries, gd.P, d};
        rc = exclusive_scan<ScanGdirPopcIn>(in, words, pre, scratch, s);
        if (rc == 0) {
            hipLaunchKernelGGL(k_gdir_prefix, dim3(grid_for(words)), dim3(kBlock), 0, s, words, (int64_t)gd.P, (int64_t)d, pre, entries);
            if (hipGetLastError() != hipSuccess) rc = -1;
        }
    }
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(pre);
    (void)hipFree(scratch);
    if (rc != 0) { snprintf(g_err, sizeof(g_err), "lsk_gdir_build: launch failed"); return -1; }
    LSK_CHECK(e2);
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_gdir_check(lsk_gdir gd, Owner ow, int part, int64_t n, uint64_t const *__restrict__ reps,
                                                       uint64_t const *__restrict__ g_binom, int *flag) {
    extern __shared__ uint64_t s_db[];
    gdir_load(gd, g_binom, s_db);
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (owner_of(reps[i], ow) != part || gdir_index(gd, reps[i], part, s_db) != i) atomicExch(flag, 1);
}
extern "C" int lsk_gdir_check(lsk_gdir gd, int part, int64_t n, uint64_t const *reps, uint64_t const *d_binom, int *d_flag, void *stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_gdir_check, dim3(grid_for(n)), dim3(kBlock), sizeof(uint64_t) * (size_t)gd.sites * (size_t)(gd.weight + 1), (hipStream_t)stream,
                       gd, make_owner(gd.P), part, n, reps, d_binom, d_flag);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// enumeration of representatives (enumerateStates, StatesEnumeration.chpl:158-224,516-585)
// Candidate c in [0, n_candidates) is the c-th state in ascending order of the candidate space:
//   fixed Hamming weight -> unrank_combinadic(c) ; otherwise the integer c.
// Each thread owns kEnumChunk consecutive candidates; survivors are flagged in a 64-bit mask, the
// masks' popcounts are scanned, and a second (cheap) kernel writes the survivors in order.
// ---------------------------------------------------------------------------------------------
constexpr int kEnumChunk = 64;

__global__ __launch_bounds__(kBlock) void k_enum_flags(lsk_basis bs, lsk_group_elem const *__restrict__ elems,
                                                       uint64_t const *__restrict__ g_binom, int64_t n_cand,
                                                       int64_t n_threads, uint64_t *__restrict__ flags,
                                                       int64_t *__restrict__ counts) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n_threads; t += (int64_t)gridDim.x * kBlock) {
        int64_t c0 = t * kEnumChunk;
        int64_t c1 = c0 + kEnumChunk < n_cand ? c0 + kEnumChunk : n_cand;
        uint64_t s = bs.hamming_weight >= 0 ? unrank_combinadic(c0, bs.hamming_weight, s_binom) : (uint64_t)c0;
        uint64_t m = 0;
        for (int64_t c = c0; c < c1; ++c) {
            bool keep = true;
            if (bs.proj == LSK_PROJ_FULL) {
                // trivial sector: every orbit has non-zero norm, so "is its own orbit minimum" is the whole test
                if (bs.k4_mode != 0) keep = bs.number_sites <= 32 ? rep_trivial<uint32_t>(bs, elems, (uint32_t)s) == (uint32_t)s
                                                                   : rep_trivial<uint64_t>(bs, elems, s) == s;
                else keep = is_representative(bs, elems, s);
            }
            if (keep) m |= 1ULL << (c - c0);
            if (c + 1 < c1) s = (bs.hamming_weight > 0) ? next_fixed_hamming(s) : s + 1;
        }
        flags[t] = m;
        counts[t] = __popcll(m);
    }
}
__global__ __launch_bounds__(kBlock) void k_enum_write(lsk_basis bs, uint64_t const *__restrict__ g_binom,
                                                       int64_t n_cand, int64_t n_threads,
                                                       uint64_t const *__restrict__ flags,
                                                       int64_t const *__restrict__ offsets,
                                                       uint64_t *__restrict__ out) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n_threads; t += (int64_t)gridDim.x * kBlock) {
        uint64_t m = flags[t];
        if (!m) continue;
        int64_t c0 = t * kEnumChunk;
        int64_t c1 = c0 + kEnumChunk < n_cand ? c0 + kEnumChunk : n_cand;
        uint64_t s = bs.hamming_weight >= 0 ? unrank_combinadic(c0, bs.hamming_weight, s_binom) : (uint64_t)c0;
        int64_t o = offsets[t];
        for (int64_t c = c0; c < c1; ++c) {
            if ((m >> (c - c0)) & 1) out[o++] = s;
            if (c + 1 < c1) s = (bs.hamming_weight > 0) ? next_fixed_hamming(s) : s + 1;
        }
    }
}

extern "C" int lsk_enumerate(lsk_basis bs, uint64_t const *d_binom, int64_t n_cand, uint64_t **d_states,
                             int64_t *count, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    *d_states = nullptr;
    *count = 0;
    if (n_cand <= 0) { LSK_CHECK(hipMalloc((void **)d_states, 8)); return 0; }
    int64_t n_threads = (n_cand + kEnumChunk - 1) / kEnumChunk;
    uint64_t *flags = nullptr;
    int64_t *counts = nullptr, *offsets = nullptr;
    LSK_CHECK(hipMalloc((void **)&flags, 8 * n_threads));
    LSK_CHECK(hipMalloc((void **)&counts, 8 * n_threads));
    LSK_CHECK(hipMalloc((void **)&offsets, 8 * n_threads));
    hipLaunchKernelGGL(k_enum_flags, dim3(grid_for(n_threads)), dim3(kBlock), 0, s, bs, bs.elems, d_binom, n_cand, n_threads, flags, counts);
    LSK_LAUNCH_CHECK();
    if (exclusive_scan_i64(n_threads, counts, offsets, s) != 0) return -1;
    int64_t last_off = 0, last_cnt = 0;
    LSK_CHECK(hipMemcpy(&last_off, offsets + (n_threads - 1), 8, hipMemcpyDeviceToHost));
    LSK_CHECK(hipMemcpy(&last_cnt, counts + (n_threads - 1), 8, hipMemcpyDeviceToHost));
    int64_t total = last_off + last_cnt;
    LSK_CHECK(hipMalloc((void **)d_states, total > 0 ? 8 * total : 8));
    hipLaunchKernelGGL(k_enum_write, dim3(grid_for(n_threads)), dim3(kBlock), 0, s, bs, d_binom, n_cand, n_threads, flags, offsets, *d_states);
    LSK_LAUNCH_CHECK();
    LSK_CHECK(hipStreamSynchronize(s));
    (void)hipFree(flags); (void)hipFree(counts); (void)hipFree(offsets);
    *count = total;
    return 0;
}

// masks[i] = owner of states[i]  (_enumStatesComputeMasksAndCounts, StatesEnumeration.chpl:138-156)
__global__ __launch_bounds__(kBlock) void k_masks(int64_t n, uint64_t const *__restrict__ states, Owner ow,
                                                  uint8_t *__restrict__ masks) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        masks[i] = (uint8_t)owner_of(states[i], ow);
}
extern "C" int lsk_masks(int64_t n, uint64_t const *states, int P, uint8_t *masks, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_masks, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, states, make_owner(P), masks);
    LSK_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_mask_counts(int64_t n, uint8_t const *__restrict__ masks,
                                                        unsigned long long *__restrict__ counts) {
    __shared__ unsigned s_cnt[LSK_MAX_PARTS];
    for (int d = threadIdx.x; d < LSK_MAX_PARTS; d += kBlock) s_cnt[d] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        atomicAdd(&s_cnt[masks[i]], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d < LSK_MAX_PARTS; d += kBlock)
        if (s_cnt[d]) atomicAdd(&counts[d], (unsigned long long)s_cnt[d]);
}
extern "C" int lsk_mask_counts(int64_t n, uint8_t const *masks, int P, int64_t *h_counts, void *stream) {
    unsigned long long *d = nullptr;
    LSK_CHECK(hipMalloc((void **)&d, 8 * LSK_MAX_PARTS));
    LSK_CHECK(hipMemsetAsync(d, 0, 8 * LSK_MAX_PARTS, (hipStream_t)stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_mask_counts, dim3(grid_for(n, kBlock * 16)), dim3(kBlock), 0, (hipStream_t)stream, n, masks, d);
        LSK_LAUNCH_CHECK();
    }
    unsigned long long h[LSK_MAX_PARTS];
    LSK_CHECK(hipStreamSynchronize((hipStream_t)stream));
    LSK_CHECK(hipMemcpy(h, d, 8 * LSK_MAX_PARTS, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    for (int p = 0; p < P; ++p) h_counts[p] = (int64_t)h[p];
    return 0;
}

// Layout converters.  position[i] = rank of element i among the elements with the same mask that
// precede it (exclusive scan of the indicator, one pass per partition); block->hashed then is
// dest[mask[i]][position[i]] = src[i] and hashed->block its inverse.  Stable by construction, which
// is what keeps every hashed part ascending (BlockToHashed.chpl:87-208, HashedToBlock.chpl:67-153).
template <int ELT>
__global__ __launch_bounds__(kBlock) void k_permute(int64_t n, uint8_t const *__restrict__ masks, uint8_t p,
                                                    int64_t const *__restrict__ pos, char const *src, char *dst,
                                                    int to_hashed) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        if (masks[i] != p) continue;
        int64_t j = pos[i];
        if (ELT == 8) {
            if (to_hashed) ((uint64_t *)dst)[j] = ((uint64_t const *)src)[i];
            else ((uint64_t *)dst)[i] = ((uint64_t const *)src)[j];
        } else {
            if (to_hashed) ((ulonglong2 *)dst)[j] = ((ulonglong2 const *)src)[i];
            else ((ulonglong2 *)dst)[i] = ((ulonglong2 const *)src)[j];
        }
    }
}
static int permute_by_masks(int64_t n, uint8_t const *masks, int P, int elt_size, void const *block_const,
                            void *block_mut, void *const *parts, int to_hashed, hipStream_t s) {
    if (n == 0) return 0;
    if (elt_size != 8 && elt_size != 16) { snprintf(g_err, sizeof(g_err), "layout converters support 8/16-byte elements"); return -1; }
    int64_t *pos = nullptr, *tmp = nullptr;
    LSK_CHECK(hipMalloc((void **)&pos, 8 * n));
    if (hipMalloc((void **)&tmp, 8 * (size_t)scan_scratch_elems(n)) != hipSuccess) { (void)hipFree(pos); snprintf(g_err, sizeof(g_err), "layout converter: no memory"); return -1; }
    for (int p = 0; p < P; ++p) {
        ScanMaskIn f{masks, (uint8_t)p};
        if (exclusive_scan<ScanMaskIn>(f, n, pos, tmp, s) != 0) { (void)hipFree(pos); (void)hipFree(tmp); return -1; }
        char const *src = to_hashed ? (char const *)block_const : (char const *)parts[p];
        char *dst = to_hashed ? (char *)parts[p] : (char *)block_mut;
        if (elt_size == 8) hipLaunchKernelGGL(k_permute<8>, dim3(grid_for(n)), dim3(kBlock), 0, s, n, masks, (uint8_t)p, pos, src, dst, to_hashed);
        else hipLaunchKernelGGL(k_permute<16>, dim3(grid_for(n)), dim3(kBlock), 0, s, n, masks, (uint8_t)p, pos, src, dst, to_hashed);
        LSK_LAUNCH_CHECK();
    }
    LSK_CHECK(hipStreamSynchronize(s));
    if (tmp) (void)hipFree(tmp);
    (void)hipFree(pos);
    return 0;
}
extern "C" int lsk_block_to_hashed(int64_t n, uint8_t const *masks, int P, int elt_size, void const *src,
                                   void *const *h_dest, void *stream) {
    return permute_by_masks(n, masks, P, elt_size, src, nullptr, h_dest, 1, (hipStream_t)stream);
}
extern "C" int lsk_hashed_to_block(int64_t n, uint8_t const *masks, int P, int elt_size, void const *const *h_src,
                                   void *dest, void *stream) {
    return permute_by_masks(n, masks, P, elt_size, nullptr, dest, (void *const *)h_src, 0, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// deterministic vectors for tests / bench: the value depends only on (basis state, seed), so every
// partitioning of the same basis sees the same logical vector.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_fill_random(int64_t n, uint64_t const *__restrict__ states, uint64_t seed,
                                                        int cplx, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t h = hash64_01(states[i] ^ (seed * 0x9e3779b97f4a7c15ULL + 0x632be59bd9b4e019ULL));
        double re = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        if (cplx) {
            uint64_t h2 = hash64_01(h ^ 0xd6e8feb86659fd93ULL);
            out[2 * i] = re;
            out[2 * i + 1] = (double)(h2 >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        } else out[i] = re;
    }
}
extern "C" int lsk_fill_random(int64_t n, uint64_t const *states, uint64_t seed, int cplx, void *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_fill_random, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, states, seed, cplx, (double *)out);
    LSK_LAUNCH_CHECK();
    return 0;
}
